"""Multi-GPU plan execution: fields shard one-per-rank, per-rank dose volumes are summed into rank 0.

The reference has no multi-GPU code (its beam loop, src/kernel_wrapper.cu:601, accumulates every beam into one
device dose volume, :92). Beams are independent until that accumulation, so a plan shards by field with no
data-path exchange except ONE float32 sum-reduce of the dose volume (RCCL over xGMI when the tensors live on
GPUs; gloo in the CPU tests). Sum order differs from the sequential reference by float rounding only.
"""


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


class PipelinedBoxReduce:
    """reduce_dose_bbox with the collective left in flight: step i's packed box is reduced on the communication stream
    while the kernels of step i+1 run, so a sequence of plans costs max(compute, reduce) per plan instead of their sum.

    Usage per plan step, with `dose` one of two alternating volumes: `release(dose)` BEFORE the volume is zeroed and
    refilled (it completes the reduce that used this volume two steps earlier and, on the destination rank, stores its
    sum), then the field's kernels and rtd_field_finish, then `submit(dose, bbox_min, bbox_max)`; `drain()` before the
    results are read and before the timed region ends."""

    def __init__(self, dist, dst=0):
        self.dist = dist
        self.dst = dst
        self.pending = {}          # id(dose tensor) -> (work, view, packed)
        self.done = {}             # id(dose tensor) -> union-box view of its last completed reduce, until release() hands it out

    def _retire(self, key):
        item = self.pending.pop(key, None)
        if item is None:
            return None
        work, view, packed = item
        work.wait()
        if self.dist.get_rank() == self.dst:
            view.copy_(packed)
        self.done[key] = view

    def release(self, dose_tensor):
        """Completes the reduce that used this volume (if any) and returns the union-box view of the volume it covered
        (None if there was none): on the destination rank that view now holds the plan's sum and is the only part of the
        volume other ranks contributed to, so clearing it (instead of the whole volume) resets the volume."""
        self._retire(id(dose_tensor))
        return self.done.pop(id(dose_tensor), None)

    def submit(self, dose_tensor, bbox_min, bbox_max):
        import torch
        dist = self.dist
        assert id(dose_tensor) not in self.pending, "release() the volume before refilling it"
        world = dist.get_world_size()
        mine = torch.tensor([int(v) for v in bbox_min] + [int(v) for v in bbox_max], dtype=torch.int64, device=dose_tensor.device)
        boxes = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(boxes, mine)
        boxes = torch.stack(boxes).cpu()
        valid = (boxes[:, 3:] >= boxes[:, :3]).all(dim=1)
        if not bool(valid.any()):
            return
        lo = boxes[valid, :3].min(dim=0).values.tolist()
        hi = boxes[valid, 3:].max(dim=0).values.tolist()
        view = dose_tensor[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1]
        packed = view.contiguous()
        work = dist.reduce(packed, dst=self.dst, op=dist.ReduceOp.SUM, async_op=True)
        self.pending[id(dose_tensor)] = (work, view, packed)

    def drain(self):
        for key in list(self.pending):
            self._retire(key)
