"""Multi-GPU plan execution over torch.distributed (one process per GPU): fields shard one per rank.

The reference has no multi-GPU code (its beam loop, src/kernel_wrapper.cu:601, accumulates every beam into one device dose
volume, :92). Beams are independent until that accumulation. What bench.py --gpus N uses is the LAST part of this module,
`BevExchange` + `balanced_slabs`: the ranks all-gather their packed beam's-eye-view slabs (~10 MB each) and every rank writes its
slab of the dose volume with all fields in field order — no dose data crosses xGMI, the volume is bit-identical to the one-GPU
loop and stays sharded by slabs. The first part (`reduce_dose`, `PipelinedBoxReduce`, `PipelinedSlabReduce`: dose boxes summed into
rank 0, round 1's exchange) is kept with its gloo tests, unused by the bench."""


def shard_fields(n_fields, world_size, rank):
    """Indices of the fields rank `rank` computes: round-robin, so 4 fields on 4 ranks = one each."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    return list(range(rank, n_fields, world_size))


def reduce_dose(dose_tensor, dist=None, dst=0):
    """Sum the per-rank dose volumes into rank `dst` (in place on dst). No-op without a process group."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return dose_tensor
    dist.reduce(dose_tensor, dst=dst, op=dist.ReduceOp.SUM)
    return dose_tensor


def run_plan(compute_field, n_fields, dose_tensor, dist=None, dst=0):
    """Compute this rank's shard (compute_field(i) accumulates field i into dose_tensor) and reduce to dst."""
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    for i in shard_fields(n_fields, world, rank):
        compute_field(i)
    return reduce_dose(dose_tensor, dist, dst)


def reduce_dose_bbox(dose_tensor, bbox_min, bbox_max, dist=None, dst=0):
    """Sum only the union of the ranks' dose bounding boxes into rank `dst`.

    Every field writes inside its own bounding box only (kernel_wrapper.cu:1185-1210), so voxels outside the union of
    the boxes are zero on every rank: exchanging the union box (a 6-int all-gather, then one reduce of the packed box)
    moves a fraction of the volume over xGMI instead of all of it. dose_tensor is [Z][Y][X]; bbox_* are (x, y, z)
    inclusive index triples of THIS rank's fields (max < min means "nothing written")."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return dose_tensor
    import torch
    world = dist.get_world_size()
    mine = torch.tensor([int(v) for v in bbox_min] + [int(v) for v in bbox_max], dtype=torch.int64, device=dose_tensor.device)
    boxes = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(boxes, mine)
    boxes = torch.stack(boxes).cpu()
    valid = (boxes[:, 3:] >= boxes[:, :3]).all(dim=1)
    if not bool(valid.any()):
        return dose_tensor
    lo = boxes[valid, :3].min(dim=0).values.tolist()
    hi = boxes[valid, 3:].max(dim=0).values.tolist()
    view = dose_tensor[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1]
    packed = view.contiguous()
    dist.reduce(packed, dst=dst, op=dist.ReduceOp.SUM)
    if dist.get_rank() == dst:
        view.copy_(packed)
    return dose_tensor


class _NullContext:
    def __enter__(self): return self
    def __exit__(self, *a): return False


class _QueuedAdds:
    """An exchange whose work (send, or receive + adds) is already queued on the side stream: retiring it = making the main
    stream wait for its event. Holds the packed / received buffers until then."""

    def __init__(self, event, views, buffers):
        self.record_event_done, self.views, self.buffers = event, views, buffers

    def wait_on_current_stream(self):
        import torch
        torch.cuda.current_stream().wait_event(self.record_event_done)


class PipelinedBoxReduce:
    """Sum of the ranks' dose into rank `dst`, sent as point-to-point boxes and left in flight.

    A field changes only the voxels of its own dose box (rtd_field_info.dose_box_min/max: the image of the BEV rectangle that
    carries dose — 60 MB on the 512^3 bench field, against 170 MB for its reference bounding box and ~300 MB for the union
    box of four fields). So nothing is reduced collectively: every rank packs ITS box and sends it to `dst` over its own xGMI
    link (`isend`), `dst` receives the N-1 boxes concurrently (`irecv`) and adds each into its volume. The transfers of plan
    i stay in flight on the communication stream while the kernels of plan i+1 run, so a sequence of plans costs
    max(compute, transfer) per plan instead of their sum.

    Usage per plan step, with `dose` one of two alternating volumes: `release(dose)` BEFORE the volume is cleared and
    refilled (it completes the exchange that used this volume two steps earlier and, on `dst`, adds the received boxes),
    then the field's kernels and rtd_field_finish, then `submit(dose, box_min, box_max)`; `drain()` before the results
    are read and before the timed region ends.

    The ranks exchange their 6-int boxes with an all_gather, which makes the host wait for the stream; with
    `static_boxes=True` (a plan whose fields keep their geometry, as in bench.py) that happens on the first submit only."""

    def __init__(self, dist, dst=0, static_boxes=False):
        self.dist = dist
        self.dst = dst
        self.static_boxes = static_boxes
        self.boxes = None          # [world][6] ints (x0, y0, z0, x1, y1, z1), x1 < x0 = nothing written
        self.pending = {}          # id(dose tensor) -> list of (work, view or None, buffer)
        self.done = {}             # id(dose tensor) -> views the last completed exchange added into (dst), until release() hands them out
        self.side = None           # (RCCL) stream on which the destination adds the received boxes, concurrently with the next plan's kernels

    def _gather_boxes(self, dose_tensor, box_min, box_max):
        import torch
        dist = self.dist
        if self.static_boxes and self.boxes is not None:
            return self.boxes
        world = dist.get_world_size()
        mine = torch.tensor([int(v) for v in box_min] + [int(v) for v in box_max], dtype=torch.int64, device=dose_tensor.device)
        boxes = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(boxes, mine)
        self.boxes = torch.stack(boxes).cpu().tolist()
        return self.boxes

    @staticmethod
    def _view(dose_tensor, b):
        return dose_tensor[b[2]:b[5] + 1, b[1]:b[4] + 1, b[0]:b[3] + 1]

    def _staged(self, dose_tensor):
        """gloo moves CPU tensors only: device tensors are staged through the host in that (test / rehearsal) configuration."""
        return dose_tensor.is_cuda and self.dist.get_backend() == "gloo"

    def _retire(self, key):
        items = self.pending.pop(key, None)
        if items is None:
            return
        views = []
        waited = set()
        for work, view, buf in items:
            if hasattr(work, "record_event_done"):                    # (dst, RCCL) the adds were queued on the side stream at submit
                work.wait_on_current_stream()
                views.extend(work.views)
                continue
            if id(work) not in waited:
                work.wait()
                waited.add(id(work))
            if view is not None:                                     # dst: add the received box
                view.add_(buf.to(view.device) if buf.device != view.device else buf)
                views.append(view)
        self.done[key] = views

    def release(self, dose_tensor):
        """Completes the exchange that used this volume (if any). On `dst` it returns the list of box views that received
        other ranks' dose since the volume was last handed out (clearing them, plus the rank's own field box, resets the
        volume without touching the rest); elsewhere None."""
        self._retire(id(dose_tensor))
        views = self.done.pop(id(dose_tensor), None)
        return views if self.dist.get_rank() == self.dst else None

    def submit(self, dose_tensor, box_min, box_max, ready=None):
        """Start the exchange of the plan that was computed into dose_tensor. `ready` (RCCL only): a torch.cuda.Event recorded on the
        compute stream right after that plan's kernels were enqueued. With it the exchange is ordered behind THAT plan only; without
        it, it is ordered behind everything already enqueued on the current stream (which, in a pipelined loop, includes the next
        plan: RCCL makes its stream wait for the current stream at enqueue time, and the overlap would be lost)."""
        import torch
        dist = self.dist
        assert id(dose_tensor) not in self.pending, "release() the volume before refilling it"
        boxes = self._gather_boxes(dose_tensor, box_min, box_max)
        rank, world = dist.get_rank(), dist.get_world_size()
        valid = [all(b[3 + a] >= b[a] for a in range(3)) for b in boxes]
        rccl = dose_tensor.is_cuda and dist.get_backend() == "nccl"
        items = []
        if rccl:
            # everything of the exchange (packing, send / receive, the destination's adds) is issued with a side stream current
            if self.side is None:
                self.side = torch.cuda.Stream(device=dose_tensor.device)
            if ready is not None:
                self.side.wait_event(ready)
            else:
                self.side.wait_stream(torch.cuda.current_stream(dose_tensor.device))
        ctx = torch.cuda.stream(self.side) if rccl else _NullContext()
        with ctx:
            ops, meta = [], []
            if rank != self.dst:
                if valid[rank]:
                    packed = self._view(dose_tensor, boxes[rank]).contiguous()
                    if self._staged(dose_tensor):
                        packed = packed.cpu()
                    ops.append(dist.P2POp(dist.isend, packed, self.dst))
                    meta.append((None, packed))
            else:
                for r in range(world):
                    if r == self.dst or not valid[r]:
                        continue
                    view = self._view(dose_tensor, boxes[r])
                    buf = torch.empty(view.shape, dtype=dose_tensor.dtype, device="cpu" if self._staged(dose_tensor) else dose_tensor.device)
                    ops.append(dist.P2POp(dist.irecv, buf, r))
                    meta.append((view, buf))
            if ops:
                # one group: the destination's N-1 receives run concurrently, each over its own link (separate irecv calls would be
                # serialised on the communication stream)
                works = dist.batch_isend_irecv(ops)
                if len(works) == len(ops):
                    items = [(w, v, b) for w, (v, b) in zip(works, meta)]
                else:                                                # coalesced into one work object (RCCL)
                    items = [(works[0], v, b) for (v, b) in meta]
                if rccl:
                    # Work.wait() only makes the CURRENT (= side) stream wait: the destination's adds are queued right now and run
                    # as soon as the boxes have arrived, concurrently with the (compute-bound) kernels on the main stream
                    for w in {id(w): w for w, _, _ in items}.values():
                        w.wait()
                    if rank == self.dst:
                        for _, v, b in items:
                            v.add_(b)
                    done = torch.cuda.Event()
                    done.record(self.side)
                    items = [(_QueuedAdds(done, [v for _, v, _ in items if v is not None], [b for _, _, b in items]), None, None)]
        self.pending[id(dose_tensor)] = items

    def drain(self):
        for key in list(self.pending):
            self._retire(key)


def slab_partition(boxes, world):
    """Owner slabs for PipelinedSlabReduce: the union of the valid boxes is cut into `world` slabs of (nearly) equal thickness
    along one axis; rank r owns slab r. The axis is the one for which the largest piece any rank sends to another (its box cut by
    the other's slab) is smallest — for fields that rotate about an axis that is the rotation axis, along which every field has
    the same extent. Returns (axis, [slab boxes as 6 ints, or None when the slab is empty]); (None, []) if no box is valid."""
    valid = [b for b in boxes if all(b[3 + a] >= b[a] for a in range(3))]
    if not valid:
        return None, []
    lo = [min(b[a] for b in valid) for a in range(3)]
    hi = [max(b[3 + a] for b in valid) for a in range(3)]

    def slabs(axis):
        n = hi[axis] - lo[axis] + 1
        out = []
        for r in range(world):
            a0, a1 = lo[axis] + (r * n) // world, lo[axis] + ((r + 1) * n) // world - 1
            if a1 < a0:
                out.append(None)
                continue
            s = lo + hi
            s[axis], s[3 + axis] = a0, a1
            out.append(s)
        return out

    best = None
    for axis in (2, 1, 0):                                            # ties: z first (whole rows and planes), then y
        sl = slabs(axis)
        cost = 0
        for s, b in enumerate(boxes):
            for r, own in enumerate(sl):
                if r != s:
                    p = box_intersection(b, own)
                    if p is not None:
                        cost = max(cost, (p[3] - p[0] + 1) * (p[4] - p[1] + 1) * (p[5] - p[2] + 1))
        if best is None or cost < best[0]:
            best = (cost, axis, sl)
    return best[1], best[2]


def box_intersection(a, b):
    """Intersection of two inclusive 6-int boxes (x0, y0, z0, x1, y1, z1); None if either is None / empty or they are disjoint."""
    if a is None or b is None:
        return None
    p = [max(a[i], b[i]) for i in range(3)] + [min(a[3 + i], b[3 + i]) for i in range(3)]
    return p if all(p[3 + i] >= p[i] for i in range(3)) else None


def union_rects(boxes, clip):
    """Disjoint inclusive 6-int boxes that cover exactly (union of `boxes`) intersected with `clip`, few and large: the cells of the
    grid spanned by all box faces are merged greedily along x, then y, then z. Deterministic (every rank derives the same list)."""
    cl = [b for b in (box_intersection(b, clip) for b in boxes) if b is not None]
    if not cl:
        return []
    cuts = [sorted({b[a] for b in cl} | {b[3 + a] + 1 for b in cl}) for a in range(3)]
    n = [len(c) - 1 for c in cuts]
    covered = [[[any(b[0] <= cuts[0][i] and cuts[0][i + 1] - 1 <= b[3] and b[1] <= cuts[1][j] and cuts[1][j + 1] - 1 <= b[4] and
                     b[2] <= cuts[2][k] and cuts[2][k + 1] - 1 <= b[5] for b in cl)
                 for i in range(n[0])] for j in range(n[1])] for k in range(n[2])]
    out = []
    for k in range(n[2]):
        for j in range(n[1]):
            for i in range(n[0]):
                if not covered[k][j][i]:
                    continue
                i1 = i
                while i1 + 1 < n[0] and covered[k][j][i1 + 1]:
                    i1 += 1
                j1 = j
                while j1 + 1 < n[1] and all(covered[k][j1 + 1][ii] for ii in range(i, i1 + 1)):
                    j1 += 1
                k1 = k
                while k1 + 1 < n[2] and all(covered[k1 + 1][jj][ii] for jj in range(j, j1 + 1) for ii in range(i, i1 + 1)):
                    k1 += 1
                for kk in range(k, k1 + 1):
                    for jj in range(j, j1 + 1):
                        for ii in range(i, i1 + 1):
                            covered[kk][jj][ii] = False
                out.append([cuts[0][i], cuts[1][j], cuts[2][k], cuts[0][i1 + 1] - 1, cuts[1][j1 + 1] - 1, cuts[2][k1 + 1] - 1])
    return out


class PipelinedSlabReduce(PipelinedBoxReduce):
    """The same sum into rank `dst` as PipelinedBoxReduce, as a point-to-point reduce-scatter + gather for three or more ranks.

    With PipelinedBoxReduce each of the N-1 links into `dst` carries one whole field box (60-83 MB on the 512^3 bench fields:
    1.0-1.4 ms at ~60 GB/s per xGMI link, more than the ~1 ms of compute per plan), while the other (N-1)(N-2) links idle.
    The boxes overlap around the isocentre, so summing first moves less into `dst`:
      phase 1  every rank cuts its box by the owner slabs (slab_partition) and sends each piece straight to its owner over
               their own link (box / N per link); the owner adds the pieces into its volume, where its own field already is;
      phase 2  every owner sends what any field wrote inside its slab (union_rects: a few disjoint boxes, one message) — now
               holding the complete sum — to `dst`, which copies it in.
    Per link into `dst` for 4 fields at 0/90/180/270 degrees: 21 + 31 MB, against 83 MB.
    Sums are formed in rank order (own field, then the pieces of ranks 0, 1, ... as received), so the result is reproducible;
    it differs from the sequential sum by float rounding only. Interface and pipelining as PipelinedBoxReduce, except that
    release() returns views to clear on EVERY rank (owners receive pieces). With two ranks it moves the same bytes as the
    direct form in two steps — use PipelinedBoxReduce there."""

    def __init__(self, dist, dst=0, static_boxes=False):
        super().__init__(dist, dst, static_boxes)
        self.layout = None         # (boxes it was derived from, axis, slabs)
        self.rects = None          # (layout, {owner: disjoint boxes of the union inside its slab})

    def _layout(self, boxes):
        if self.layout is None or self.layout[0] != boxes:
            axis, slabs = slab_partition(boxes, self.dist.get_world_size())
            self.layout = (boxes, axis, slabs)
        return self.layout[2]

    def _slab_rects(self, boxes, slabs, r):
        """Disjoint boxes covering what any field wrote inside owner r's slab (cached with the layout)."""
        if self.rects is None or self.rects[0] is not self.layout:
            self.rects = (self.layout, {})
        cache = self.rects[1]
        if r not in cache:
            cache[r] = union_rects(boxes, slabs[r]) if slabs[r] is not None else []
        return cache[r]

    def release(self, dose_tensor):
        """Completes the exchange that used this volume (if any) and returns the views that received dose from other ranks
        (clearing them, plus the rank's own field box, resets the volume), or None."""
        self._retire(id(dose_tensor))
        return self.done.pop(id(dose_tensor), None)

    def _retire(self, key):
        items = self.pending.pop(key, None)
        if items is None:
            return
        views = []
        for item in items:
            work = item[0]
            if hasattr(work, "record_event_done"):                    # (RCCL) everything was queued on the side stream at submit
                work.wait_on_current_stream()
                views.extend(work.views)
                continue
            if work is not None:
                work.wait()
            if item[1] is not None:                                  # (gloo) phase 2 on dst: unpack the received slab parts
                item[3](item[1], item[2])
                views.extend(item[1])
        self.done[key] = self.done.get(key, []) + views

    def _exchange(self, ops, meta):
        """One grouped batch of point-to-point operations; returns [(work, view, buffer)]."""
        if not ops:
            return []
        works = self.dist.batch_isend_irecv(ops)
        if len(works) == len(ops):
            return [(w, v, b) for w, (v, b) in zip(works, meta)]
        return [(works[0], v, b) for (v, b) in meta]                 # coalesced into one work object (RCCL)

    def submit(self, dose_tensor, box_min, box_max, ready=None):
        import torch
        dist = self.dist
        assert id(dose_tensor) not in self.pending, "release() the volume before refilling it"
        boxes = self._gather_boxes(dose_tensor, box_min, box_max)
        slabs = self._layout(boxes)
        rank, world = dist.get_rank(), dist.get_world_size()
        rccl = dose_tensor.is_cuda and dist.get_backend() == "nccl"
        staged = self._staged(dose_tensor)
        if rccl:
            if self.side is None:
                self.side = torch.cuda.Stream(device=dose_tensor.device)
            if ready is not None:
                self.side.wait_event(ready)
            else:
                self.side.wait_stream(torch.cuda.current_stream(dose_tensor.device))

        def recv_buffer(view):
            return torch.empty(view.shape, dtype=dose_tensor.dtype, device="cpu" if staged else dose_tensor.device)

        def packed(view):
            p = view.contiguous()
            return p.cpu() if staged else p

        ctx = torch.cuda.stream(self.side) if rccl else _NullContext()
        with ctx:
            # ---- phase 1: pieces to their owners ----
            ops, meta = [], []
            if slabs:
                for r in range(world):                                # sends: this rank's box cut by every other owner's slab
                    piece = box_intersection(boxes[rank], slabs[r]) if r != rank else None
                    if piece is not None:
                        p = packed(self._view(dose_tensor, piece))
                        ops.append(dist.P2POp(dist.isend, p, r))
                        meta.append((None, p))
                for s in range(world):                                # receives: every other rank's box cut by this rank's slab
                    piece = box_intersection(boxes[s], slabs[rank]) if s != rank else None
                    if piece is not None:
                        view = self._view(dose_tensor, piece)
                        buf = recv_buffer(view)
                        ops.append(dist.P2POp(dist.irecv, buf, s))
                        meta.append((view, buf))
            items = self._exchange(ops, meta)
            for w in {id(w): w for w, _, _ in items}.values():
                w.wait()                                             # RCCL: the side stream waits; otherwise the host does
            added, held = [], []
            for _, v, b in items:                                    # in rank order: the sum is reproducible
                if v is not None:
                    v.add_(b.to(v.device) if b.device != v.device else b)
                    added.append(v)
                held.append(b)
            # ---- phase 2: complete slabs to dst ----
            # only where some field wrote: the union of the boxes inside the slab, as a few disjoint boxes packed into one message
            ops, meta = [], []
            if slabs:
                if rank != self.dst:
                    rects = self._slab_rects(boxes, slabs, rank)
                    if rects:
                        p = torch.cat([self._view(dose_tensor, q).reshape(-1) for q in rects])
                        p = p.cpu() if staged else p
                        ops.append(dist.P2POp(dist.isend, p, self.dst))
                        meta.append((None, p))
                else:
                    for r in range(world):
                        rects = self._slab_rects(boxes, slabs, r) if r != self.dst else []
                        if rects:
                            views = [self._view(dose_tensor, q) for q in rects]
                            buf = torch.empty(sum(v.numel() for v in views), dtype=dose_tensor.dtype,
                                              device="cpu" if staged else dose_tensor.device)
                            ops.append(dist.P2POp(dist.irecv, buf, r))
                            meta.append((views, buf))
            items2 = self._exchange(ops, meta)

            def unpack(views, buf):
                """dst: the slab parts now hold the complete sum (dst's own partial values there are replaced)."""
                off = 0
                for v in views:
                    chunk = buf[off:off + v.numel()].view(v.shape)
                    v.copy_(chunk.to(v.device) if chunk.device != v.device else chunk)
                    off += v.numel()

            if rccl:
                for w in {id(w): w for w, _, _ in items2}.values():
                    w.wait()
                for _, views, b in items2:
                    if views is not None:
                        unpack(views, b)
                        added.extend(views)
                    held.append(b)
                done = torch.cuda.Event()
                done.record(self.side)
                pending = [(_QueuedAdds(done, added, held), None, None)]
            else:
                self.done[id(dose_tensor)] = added                    # phase 1 is complete; phase 2 is finished by release() / drain()
                pending = [(w, views, b, unpack) for w, views, b in items2] if items2 else [(None, None, None, None)]
        self.pending[id(dose_tensor)] = pending


# ------------------------------------------------------------------------------------------------------------------------
# Exchange in beam's-eye view. What a field hands to the dose grid is its BEV dose cube (the cube the reference copies into a
# 3-D texture before primTransfDiv samples it, kernel_wrapper.cu:1107-1141): on the 512^3 / 0.5 mm bench grid its non-zero
# block is ~10 MB where the dose box it turns into is 60-83 MB. So the ranks all-gather the packed BEV blocks (ONE RCCL
# collective per plan, N x ~10 MB, every xGMI link busy) and every rank runs the transfer of EVERY field — in field order,
# i.e. the `+=` order of the sequential beam loop (kernel_wrapper.cu:601, :92) — restricted to ITS slab of the dose volume.
# The plan's dose volume is left sharded by slabs across the GPUs (each rank copies its slab to the host over its own PCIe
# link); no dose data crosses between GPUs, and the result is bit-identical to the one-GPU loop.

def _water_fill(head, total, per_unit):
    """Shares s_r >= 0 with sum s_r = total and head_r + per_unit * s_r equal for every rank that gets a share (ranks whose head
    already exceeds that level get none)."""
    n = len(head)
    if total <= 0 or per_unit <= 0:
        return [total / n] * n
    order = sorted(range(n), key=lambda r: head[r])
    level = None
    for m in range(1, n + 1):                                        # the m ranks with the smallest heads share the work
        lv = (sum(head[order[j]] for j in range(m)) + per_unit * total) / m
        if m == n or lv <= head[order[m]]:
            level = lv
            break
    return [max(0.0, (level - head[r]) / per_unit) for r in range(n)]


def balanced_slabs(boxes, dims, world, head=None, per_voxel=None):
    """Slabs for BevExchange: the dose grid is cut into `world` slabs along one axis so that the transfer work — for every slab
    the voxels of ALL fields' dose boxes inside it — is as even as it can be. boxes: 6-int inclusive boxes (x0, y0, z0, x1, y1, z1),
    an empty box has max < min. Returns (axis, [(lo, hi) inclusive index range per rank along that axis]); slabs tile [0, dims[axis]).

    A box may carry a 7th entry, the relative cost per voxel of transferring that field (oblique beams have large, mostly empty
    boxes that cost less per voxel); the load of a slab is then the cost-weighted voxel count.
    head[r] (time rank r spends on its own field before it transfers, any unit) and per_voxel (time per unit of load, same
    unit) make the cut uneven on purpose: fields differ in cost (an oblique beam's superposition took 0.55 ms against 0.47 ms,
    profiles/r02_angles.json), so a rank with an expensive field gets a thinner slab and head[r] + per_voxel * load(slab r) — the
    rank's step time — is what is evened out. The dose does not depend on the cut (every voxel receives every field in field order)."""
    valid = [b for b in boxes if all(b[3 + a] >= b[a] for a in range(3))]
    weighted = head is not None and per_voxel is not None and per_voxel > 0
    best = None
    for axis in (2, 1, 0):                                            # ties: z first (contiguous slabs of the [z][y][x] volume)
        n = int(dims[axis])
        prof = [0] * (n + 1)
        for b in valid:
            area = 1
            for a in range(3):
                if a != axis:
                    area *= b[3 + a] - b[a] + 1
            area *= b[6] if len(b) > 6 else 1                         # the field's own cost per voxel, if given
            prof[max(b[axis], 0)] += area
            prof[min(b[3 + axis], n - 1) + 1] -= area
        load, run = [], 0
        for i in range(n):
            run += prof[i]
            load.append(run)
        total = sum(load)
        shares = _water_fill([float(h) for h in head], float(total), float(per_voxel)) if weighted else [total / world] * world
        target, run_t = [], 0.0                                       # cumulative share in front of slab r
        for r in range(world):
            target.append(run_t)
            run_t += shares[r]
        cuts, acc, r = [0], 0, 1
        for i in range(n):
            acc += load[i]
            while r < world and acc >= target[r] and len(cuts) < world:
                cuts.append(i + 1)
                r += 1
        while len(cuts) < world:
            cuts.append(n)
        cuts.append(n)
        ranges = [(cuts[k], cuts[k + 1] - 1) for k in range(world)]
        cost = [(float(head[k]) if weighted else 0.0) + (float(per_voxel) if weighted else 1.0) * (sum(load[a:b + 1]) if b >= a else 0)
                for k, (a, b) in enumerate(ranges)]
        worst = max(cost, default=0)
        if best is None or worst < 0.97 * best[0]:                   # z unless another axis is clearly better (contiguous slabs)
            best = (worst, axis, ranges)
    return best[1], best[2]


class BevExchange:
    """All-gather of the packed BEV slabs + slab-clipped transfers of every field on every rank (see the comment above).

    fields[r] for r != rank are geometry-only ("remote") field objects of the other ranks' beams; fields[rank] alternates between
    the caller's own field objects. The objects need the methods of raytracedicom_amd.engine.Field used here (wait_plan,
    export_bev, attach_bev, transfer, clear_dose_box) — the CPU tests drive this class with numpy stand-ins over gloo.
    All device work is issued on the CURRENT torch stream, which must be the stream the engine launches on."""

    def __init__(self, dist, rank, world, remote_fields, dims, new_bytes, data_ptr=lambda t: t.data_ptr(), n_buffers=2, zero_box=None,
                 transfer_all=None):
        self.dist, self.rank, self.world, self.dims = dist, rank, world, tuple(int(d) for d in dims)
        self.remote = remote_fields            # dict rank -> remote field object
        self.new_bytes, self.data_ptr = new_bytes, data_ptr
        self.n_buffers = n_buffers
        self.zero_box = zero_box               # optional zero_box(b, lo, hi): zero an inclusive index box of dose volume b in ONE launch
        # optional transfer_all(fields in rank order, dose_ptr, lo, hi): WRITE the box with 0 + field 0 + field 1 + ... in one launch
        # (rtd_fields_transfer_init): replaces the per-field transfers AND the clear (the box is rewritten by every plan)
        self.transfer_all = transfer_all
        self.heads_us, self.rates_ps_kvox = None, None
        self.cap = None
        self.send, self.recv, self.work = [], [], []
        self.boxes = None
        self.axis, self.ranges = None, None

    def setup(self, own_field, head_us=None, transfer_ps_per_kvoxel=None):
        """After the first compute_bev of the own field: message capacity (max over ranks), dose boxes, slab partition.
        head_us: measured time of this rank's field up to its BEV dose; transfer_ps_per_kvoxel: measured cost of this rank's
        transfer (+ clear) in picoseconds per 1000 box voxels (both optional, integers after rounding): when every rank supplies them the slabs are cut
        so that the ranks' step times come out even (balanced_slabs), otherwise so that the transfer work does."""
        import torch
        info, nbytes = own_field.wait_plan()
        dev = self.new_bytes(1).device
        have = head_us is not None and transfer_ps_per_kvoxel is not None
        mine = torch.tensor([int(nbytes)] + [int(v) for v in info["dose_box_min"]] + [int(v) for v in info["dose_box_max"]]
                            + [int(round(head_us)) if have else -1, int(round(transfer_ps_per_kvoxel)) if have else -1], dtype=torch.int64, device=dev)
        allv = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(allv, mine)
        rows = torch.stack(allv).cpu().tolist()
        self.cap = (max(int(r[0]) for r in rows) + 255) // 256 * 256
        self.boxes = [[int(v) for v in r[1:7]] for r in rows]
        heads, rates = [int(r[7]) for r in rows], [int(r[8]) for r in rows]
        self.heads_us, self.rates_ps_kvox = heads, rates
        if min(heads) >= 0 and min(rates) > 0:                       # every rank measured: even out the step times
            weighted = [bx + [rates[r]] for r, bx in enumerate(self.boxes)]      # load in ps per 1000 voxels x voxels
            self.axis, self.ranges = balanced_slabs(weighted, self.dims, self.world, head=heads, per_voxel=1e-9)   # -> us
        else:
            self.axis, self.ranges = balanced_slabs(self.boxes, self.dims, self.world)
        self.send = [self.new_bytes(self.cap) for _ in range(self.n_buffers)]
        self.recv = [self.new_bytes(self.cap * self.world) for _ in range(self.n_buffers)]
        self.work = [None] * self.n_buffers
        return self

    def clip(self):
        """(lo, hi) inclusive dose-index box of this rank's slab."""
        lo, hi = [0, 0, 0], [d - 1 for d in self.dims]
        lo[self.axis], hi[self.axis] = self.ranges[self.rank]
        return lo, hi

    def post(self, own_field, b):
        """Pack the own field's BEV slab into send buffer b and start the all-gather (left in flight)."""
        own_field.export_bev(self.data_ptr(self.send[b]), self.cap)
        self.work[b] = self.dist.all_gather_into_tensor(self.recv[b], self.send[b], async_op=True)

    def complete(self, own_field, b, dose_ptr):
        """Wait (in stream order) for all-gather b, then transfer every field, in field order, into this rank's slab of the volume."""
        if self.work[b] is not None:
            self.work[b].wait()
            self.work[b] = None
        lo, hi = self.clip()
        if hi[self.axis] < lo[self.axis]:
            return
        base = self.data_ptr(self.recv[b])
        if self.transfer_all is not None:
            box = self._union_box(lo, hi)
            if box is None:
                return
            fields = []
            for r in range(self.world):
                if r == self.rank:
                    fields.append(own_field)
                else:
                    self.remote[r].attach_bev(base + r * self.cap)
                    fields.append(self.remote[r])
            self.transfer_all(fields, dose_ptr, box[0], box[1])
            return
        for r in range(self.world):
            if r == self.rank:
                own_field.transfer(dose_ptr, lo, hi)
            else:
                f = self.remote[r]
                f.attach_bev(base + r * self.cap)
                f.transfer(dose_ptr, lo, hi)

    def _union_box(self, lo, hi):
        """Bounding box of all fields' dose boxes inside [lo, hi], or None."""
        valid = [bx for bx in self.boxes if all(bx[3 + a] >= bx[a] for a in range(3))]
        if not valid:
            return None
        ulo = [max(lo[a], min(bx[a] for bx in valid)) for a in range(3)]
        uhi = [min(hi[a], max(bx[3 + a] for bx in valid)) for a in range(3)]
        return (ulo, uhi) if all(uhi[a] >= ulo[a] for a in range(3)) else None

    def clear(self, own_field, b, dose_ptr):
        """Zero what complete(own_field, b, dose_ptr) wrote (the fields' dose boxes inside this rank's slab)."""
        lo, hi = self.clip()
        if hi[self.axis] < lo[self.axis]:
            return
        if self.transfer_all is not None:                             # the fused transfer rewrites its whole box: nothing to clear
            return
        if self.zero_box is not None:                                 # one launch: the bounding box of all fields' boxes, inside the slab
            box = self._union_box(lo, hi)
            if box is not None:
                self.zero_box(b, box[0], box[1])
            return
        base = self.data_ptr(self.recv[b])
        for r in range(self.world):
            if r == self.rank:
                own_field.clear_dose_box(dose_ptr, lo, hi)
            else:
                f = self.remote[r]
                f.attach_bev(base + r * self.cap)
                f.clear_dose_box(dose_ptr, lo, hi)
