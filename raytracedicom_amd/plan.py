"""Multi-GPU plan execution over torch.distributed (one process per GPU): fields shard one per rank.

The reference has no multi-GPU code (its beam loop, src/kernel_wrapper.cu:601, accumulates every beam into one device dose
volume, :92). Beams are independent until that accumulation. `BevExchange` + `balanced_slabs` are what bench.py --gpus N uses:
the ranks all-gather their packed beam's-eye-view slabs (~10 MB each) and every rank writes its slab of the dose volume with all
fields in field order — no dose data crosses xGMI, the volume is bit-identical to the one-GPU loop and stays sharded by slabs.
(Round 1's exchange — dose boxes summed into rank 0 — is gone; git history has it.)"""


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

    def check(self, own_field):
        """setup() froze the message capacity and every rank's dose box from the FIRST plan. Call after a plan of the own field has
        finished (wait_plan then returns at once): raises when that plan's packed slab no longer fits the capacity or its dose box
        has grown beyond the box the slab transfers were sized for (new spot weights or CT on the same objects) — the dose outside
        the stale box would otherwise be dropped silently. The remedy is a new setup()."""
        info, nbytes = own_field.wait_plan()
        if int(nbytes) > self.cap:
            raise RuntimeError("BevExchange: the field's packed BEV slab (%d bytes) exceeds the capacity fixed by setup() (%d): call setup() again"
                               % (int(nbytes), self.cap))
        mine = self.boxes[self.rank]
        lo, hi = [int(v) for v in info["dose_box_min"]], [int(v) for v in info["dose_box_max"]]
        if all(hi[a] >= lo[a] for a in range(3)) and not all(lo[a] >= mine[a] and hi[a] <= mine[3 + a] for a in range(3)):
            raise RuntimeError("BevExchange: the field's dose box %s..%s is no longer inside the box setup() saw %s..%s: call setup() again"
                               % (lo, hi, mine[:3], mine[3:]))

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

    def attach_all(self, b):
        """Attach every other rank's message of receive buffer b to its remote field WITHOUT transferring anything — what a rank
        whose slab is empty (complete() returns at once) needs before it can read the senders' headers (rtd_field_finish of a
        remote field reports their device-side error flags)."""
        base = self.data_ptr(self.recv[b])
        for r in range(self.world):
            if r != self.rank:
                self.remote[r].attach_bev(base + r * self.cap)

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
