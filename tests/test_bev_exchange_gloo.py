"""The multi-GPU exchange of bench.py (plan.BevExchange) on CPU: gloo ranks, world 2 / 3 / 4. Every rank packs a slab of its
field, ONE all-gather moves the slabs, every rank transfers every field — in field order — into its own slab of the volume.
The fields are numpy stand-ins with the method surface of raytracedicom_amd.engine.Field (the BEV -> dose transfer itself is a
GPU kernel, covered on one GPU by tests/test_gpu_multi.py); their per-field dose comes from the oracle. The union of the ranks'
slabs must equal the sequential accumulation of all fields BIT FOR BIT (same `+=` order at every voxel)."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT
from raytracedicom_amd import plan


def test_balanced_slabs():
    """Slabs tile the grid and even out the transfer work (voxels of all dose boxes inside a slab)."""
    boxes = [[143, 158, 46, 368, 352, 464], [12, 152, 152, 498, 358, 358], [141, 154, 46, 370, 356, 464], [12, 156, 156, 498, 354, 354]]
    for world in (1, 2, 3, 4, 8):
        axis, ranges = plan.balanced_slabs(boxes, (512, 512, 512), world)
        assert ranges[0][0] == 0 and ranges[-1][1] == 511 and all(ranges[i][1] + 1 == ranges[i + 1][0] for i in range(world - 1))
        load = []
        for lo, hi in ranges:
            v = 0
            for b in boxes:
                a0, a1 = max(b[axis], lo), min(b[3 + axis], hi)
                if a1 >= a0:
                    v += (a1 - a0 + 1) * np.prod([b[3 + a] - b[a] + 1 for a in range(3) if a != axis])
            load.append(v)
        assert max(load) <= 1.15 * sum(load) / world + 1
    # cost-weighted cut: head[r] = time of rank r's own field, per_voxel = transfer cost; the ranks' step times come out even, the
    # rank with the expensive field gets the thinner slab, a rank slower than the common level gets nothing
    def loads(axis, ranges):
        out = []
        for lo, hi in ranges:
            v = 0
            for b in boxes:
                a0, a1 = max(b[axis], lo), min(b[3 + axis], hi)
                if a1 >= a0:
                    v += (a1 - a0 + 1) * int(np.prod([b[3 + a] - b[a] + 1 for a in range(3) if a != axis]))
            out.append(v)
        return out
    head, c = [744.0, 846.0, 720.0, 745.0], 4.5e-6
    axis, ranges = plan.balanced_slabs(boxes, (512, 512, 512), 4, head=head, per_voxel=c)
    assert ranges[0][0] == 0 and ranges[-1][1] == 511 and all(ranges[i][1] + 1 == ranges[i + 1][0] for i in range(3))
    step = [h + c * v for h, v in zip(head, loads(axis, ranges))]
    axis_e, ranges_e = plan.balanced_slabs(boxes, (512, 512, 512), 4)
    step_e = [h + c * v for h, v in zip(head, loads(axis_e, ranges_e))]
    assert max(step) - min(step) < 12.0 and max(step) < max(step_e) - 40.0     # (a plane of the busiest region costs ~ 5 us)
    assert loads(axis, ranges)[1] == min(loads(axis, ranges))
    axis, ranges = plan.balanced_slabs(boxes, (512, 512, 512), 4, head=[744.0, 5000.0, 720.0, 745.0], per_voxel=c)
    assert ranges[1][1] < ranges[1][0] and ranges[-1][1] == 511                  # empty slab for the rank that is late anyway
    assert plan.balanced_slabs(boxes, (512, 512, 512), 4, head=[7.0] * 4, per_voxel=c) == (axis_e, ranges_e)
    # degenerate inputs: no box at all, more ranks than planes
    axis, ranges = plan.balanced_slabs([[1, 1, 1, 0, 0, 0]], (8, 8, 8), 3)
    assert len(ranges) == 3 and ranges[-1][1] == 7
    axis, ranges = plan.balanced_slabs([[0, 0, 0, 1, 1, 1]], (2, 2, 2), 4)
    assert len(ranges) == 4 and sum(1 for lo, hi in ranges if hi >= lo) <= 2


class _StandInField:
    """engine.Field's exchange surface on numpy: message = [box (6 x int64) | the field's dose inside its box]."""
    HEADER = 64

    def __init__(self, dims, dose=None, box=None):
        self.dims, self.dose, self.box, self.msg = dims, dose, box, None

    def wait_plan(self):
        b = self.box
        n = int(np.prod([b[3 + a] - b[a] + 1 for a in range(3)]))
        return {"dose_box_min": b[:3], "dose_box_max": b[3:]}, self.HEADER + 4 * n

    def _mem(self, ptr, nbytes):
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,))

    def export_bev(self, ptr, cap):
        _, need = self.wait_plan()
        assert need <= cap
        m = self._mem(ptr, need)
        m[:48] = np.array(self.box, dtype=np.int64).view(np.uint8)
        b = self.box
        m[self.HEADER:need] = np.ascontiguousarray(self.dose[b[2]:b[5] + 1, b[1]:b[4] + 1, b[0]:b[3] + 1]).view(np.uint8).ravel()

    def attach_bev(self, ptr):
        self.msg = ptr

    def _block(self):
        if self.msg is None:
            b = self.box
            return b, self.dose[b[2]:b[5] + 1, b[1]:b[4] + 1, b[0]:b[3] + 1]
        b = [int(v) for v in self._mem(self.msg, 48).view(np.int64)]
        shape = (b[5] - b[2] + 1, b[4] - b[1] + 1, b[3] - b[0] + 1)
        return b, self._mem(self.msg + self.HEADER, 4 * int(np.prod(shape))).view(np.float32).reshape(shape)

    def _apply(self, dose_ptr, lo, hi, op):
        nx, ny, nz = self.dims
        vol = np.ctypeslib.as_array(ctypes.cast(dose_ptr, ctypes.POINTER(ctypes.c_float)), shape=(nz, ny, nx))
        b, blk = self._block()
        c = list(lo) + list(hi)
        q = [max(b[i], c[i]) for i in range(3)] + [min(b[3 + i], c[3 + i]) for i in range(3)]
        if any(q[3 + i] < q[i] for i in range(3)):
            return
        dst = vol[q[2]:q[5] + 1, q[1]:q[4] + 1, q[0]:q[3] + 1]
        src = blk[q[2] - b[2]:q[5] - b[2] + 1, q[1] - b[1]:q[4] - b[1] + 1, q[0] - b[0]:q[3] - b[0] + 1]
        op(dst, src)

    def transfer(self, dose_ptr, lo, hi):
        def add(dst, src):
            m = src > 0.0                    # primTransfDiv adds only positive values (kernel_wrapper.cu:91-93)
            dst[m] += src[m]
        self._apply(dose_ptr, lo, hi, add)

    def clear_dose_box(self, dose_ptr, lo, hi):
        self._apply(dose_ptr, lo, hi, lambda dst, src: dst.fill(0.0))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from raytracedicom_amd import luts, scenarios
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    oracle.set_threads(2)
    es = luts.synth_luts()
    n = 64
    ct, _ = scenarios.hetero_phantom(n)
    scn = scenarios.hetero_ct(es, n=n, spots=4, pitch=8.0, n_layers=2, angles=[0.0, 90.0, 200.0, 300.0][:world], ct=ct)
    mine = np.zeros_like(scn.ct)
    f = oracle.run_field(scn, scn.beams[rank], mine, keep_layers=False)
    box = [int(v) for v in f.info["bbox_min"]] + [int(v) for v in f.info["bbox_max"]]
    f.close()
    own = _StandInField(scn.dims, mine, box)
    remote = {r: _StandInField(scn.dims) for r in range(world) if r != rank}
    ex = plan.BevExchange(dist, rank, world, remote, scn.dims, new_bytes=lambda k: torch.empty(int(k), dtype=torch.uint8))
    # measured costs (bench.py supplies them): rank 1 pretends to own an expensive field -> thinner slab, same dose
    # (world 4: so expensive that its slab comes out EMPTY — it transfers nothing and attaches nothing)
    ex.setup(own, head_us=700 + (300 if world < 4 else 50000) * (rank == 1), transfer_ps_per_kvoxel=4000)
    assert ex.cap % 256 == 0 and len(ex.ranges) == world
    if world == 4:
        assert ex.ranges[1][1] < ex.ranges[1][0]
    vols = [torch.zeros((n, n, n), dtype=torch.float32) for _ in range(2)]
    # three pipelined plans on two alternating volumes / buffers (bench.py's loop): post plan i, then complete plan i - 1
    pending = None
    for it in range(3):
        b = it % 2
        if it >= 2:
            ex.clear(own, b, vols[b].data_ptr())
            assert not vols[b].any()
        ex.post(own, b)
        if pending is not None:
            ex.complete(own, pending, vols[pending].data_ptr())
        pending = b
    ex.complete(own, pending, vols[pending].data_ptr())
    assert torch.equal(vols[0], vols[1])
    ex.check(own)                                                     # the plans still fit what setup() froze ...
    grown = _StandInField(scn.dims, mine, [max(box[0] - 1, 0)] + box[1:])
    if grown.box != box:
        with pytest.raises(RuntimeError):                            # ... a field whose dose box has grown does not
            ex.check(grown)
    lo, hi = ex.clip()
    out = vols[0].numpy()
    mask = np.zeros_like(out, dtype=bool)
    mask[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = True
    assert not out[~mask].any()                                       # nothing outside this rank's slab
    # reading the senders' headers (bench.py's last check) needs the messages attached — also on a rank with an empty slab
    if hi[ex.axis] < lo[ex.axis]:
        assert all(fr.msg is None for fr in remote.values())
    ex.attach_all(pending)
    assert all(fr.msg is not None for fr in remote.values())
    np.save(os.path.join(out_dir, "slab%d.npy" % rank), out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_bev_exchange_equals_sequential_bitwise(orc, synth, tmp_path, world):
    import torch.multiprocessing as mp
    from raytracedicom_amd import scenarios
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    slabs = [np.load(str(tmp_path / ("slab%d.npy" % r))) for r in range(world)]
    got = np.zeros_like(slabs[0])
    for s_ in slabs:
        assert not (got != 0)[s_ != 0].any()                         # the slabs are disjoint
        got += s_
    ct, _ = scenarios.hetero_phantom(64)
    scn = scenarios.hetero_ct(synth, n=64, spots=4, pitch=8.0, n_layers=2, angles=[0.0, 90.0, 200.0, 300.0][:world], ct=ct)
    ref = orc.compute(scn)                                            # sequential accumulation, field 0, 1, ...
    assert ref.max() > 0
    np.testing.assert_array_equal(got, ref)
