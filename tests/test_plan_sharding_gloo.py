"""N>1 path on CPU: two gloo ranks each compute their shard of a 2-field plan (the oracle stands in for the per-rank
engine), one sum-reduce to rank 0, result == sequential accumulation of both fields (the reference's beam loop)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT
from raytracedicom_amd import plan


def test_shard_fields():
    assert plan.shard_fields(4, 4, 2) == [2]
    assert plan.shard_fields(8, 4, 1) == [1, 5]
    assert plan.shard_fields(1, 2, 1) == []
    assert sorted(sum((plan.shard_fields(7, 3, r) for r in range(3)), [])) == list(range(7))
    with pytest.raises(ValueError):
        plan.shard_fields(4, 2, 2)


def test_slab_partition():
    """Owner slabs of the reduce-scatter form: equal cuts of the union box along the axis that balances the pieces."""
    # four fields rotating about y (bench N=4): every box has the same y extent -> cut along y, four slabs tiling the union
    boxes = [[143, 158, 46, 368, 352, 464], [12, 152, 152, 498, 358, 358], [141, 154, 46, 370, 356, 464], [12, 156, 156, 498, 354, 354]]
    axis, slabs = plan.slab_partition(boxes, 4)
    assert axis == 1 and [s[1] for s in slabs] == [152, 203, 255, 307] and [s[4] for s in slabs] == [202, 254, 306, 358]
    assert all(s[0] == 12 and s[3] == 498 and s[2] == 46 and s[5] == 464 for s in slabs)
    # an empty box takes no part; more ranks than planes along the cut axis leave empty slabs
    axis, slabs = plan.slab_partition([[0, 0, 0, 1, 1, 1], [5, 5, 5, 4, 4, 4]], 4)
    assert slabs.count(None) == 2 and [s for s in slabs if s is not None] == [[0, 0, 0, 1, 1, 0], [0, 0, 1, 1, 1, 1]]
    assert plan.slab_partition([[1, 1, 1, 0, 0, 0]] * 2, 2) == (None, [])
    assert plan.box_intersection([0, 0, 0, 4, 4, 4], [3, 2, 1, 9, 9, 9]) == [3, 2, 1, 4, 4, 4]
    assert plan.box_intersection([0, 0, 0, 4, 4, 4], [5, 0, 0, 9, 9, 9]) is None and plan.box_intersection(None, [0] * 6) is None


def test_union_rects_cover_the_union_exactly():
    """Phase 2 of the slab reduce sends the union of the field boxes inside a slab as disjoint boxes: exact cover, no overlap."""
    rng = np.random.default_rng(7)
    clip = [1, 0, 2, 9, 8, 7]
    for _ in range(150):
        boxes = []
        for _ in range(int(rng.integers(1, 5))):
            lo = rng.integers(0, 8, 3)
            hi = lo + rng.integers(0, 6, 3)
            boxes.append([int(v) for v in lo] + [int(v) for v in hi])
        want = np.zeros((14, 14, 14), dtype=int)
        for b in boxes:
            q = plan.box_intersection(b, clip)
            if q:
                want[q[2]:q[5] + 1, q[1]:q[4] + 1, q[0]:q[3] + 1] = 1
        got = np.zeros_like(want)
        for q in plan.union_rects(boxes, clip):
            got[q[2]:q[5] + 1, q[1]:q[4] + 1, q[0]:q[3] + 1] += 1
        assert (got == want).all()
    # the four bench fields (cross of two beam directions) inside one slab: a centre block and two arms instead of the full slab
    boxes = [[143, 158, 46, 368, 352, 464], [12, 152, 152, 498, 358, 358], [141, 154, 46, 370, 356, 464], [12, 156, 156, 498, 354, 354]]
    _, slabs = plan.slab_partition(boxes, 4)
    rects = plan.union_rects(boxes, slabs[1])
    vol = lambda q: (q[3] - q[0] + 1) * (q[4] - q[1] + 1) * (q[5] - q[2] + 1)
    assert len(rects) == 3 and sum(vol(q) for q in rects) < 0.75 * vol(slabs[1])
    assert plan.union_rects(boxes, [600, 0, 0, 700, 9, 9]) == []


def _worker(rank, world, port, out_path, use_bbox=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from raytracedicom_amd import luts, scenarios
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    oracle.set_threads(2)
    es = luts.synth_luts()
    ct, _ = scenarios.hetero_phantom(64)
    scn = scenarios.hetero_ct(es, n=64, spots=4, pitch=8.0, n_layers=2, angles=[0.0, 90.0, 200.0, 300.0][:max(2, world)], ct=ct)
    dose = np.zeros_like(scn.ct)
    t = torch.from_numpy(dose)

    boxes = []

    def compute_field(i):
        f = oracle.run_field(scn, scn.beams[i], dose, keep_layers=False)
        assert f.status == 0
        boxes.append((f.info["bbox_min"], f.info["bbox_max"]))
        f.close()

    if use_bbox in ("pipe", "slab"):
        # four plan iterations on two alternating volumes, reduces left in flight (bench.py's N>1 path). A reused volume is
        # not zeroed as a whole: every rank clears the box its own fields wrote, the destination rank also the boxes it received
        # from the others; a drain() in the middle (bench.py's barrier) must not lose that bookkeeping.
        # "slab": the reduce-scatter + gather form (pieces to owner slabs, complete slabs to rank 0): every rank receives dose.
        red = plan.PipelinedBoxReduce(dist, dst=0) if use_bbox == "pipe" else plan.PipelinedSlabReduce(dist, dst=0)
        vols = [np.zeros_like(scn.ct), np.zeros_like(scn.ct)]
        tens = [torch.from_numpy(v) for v in vols]
        own = [None, None]
        for it in range(4):
            v = vols[it % 2]
            views = red.release(tens[it % 2])
            if it >= 2:
                lo, hi = own[it % 2]
                v[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = 0.0     # what this rank's own fields wrote
                for w in views or ():                                        # (destination) what the other ranks' boxes added
                    w.zero_()
                assert not v.any()
            bl, bh = [10 ** 9] * 3, [-1] * 3
            for i in plan.shard_fields(len(scn.beams), world, rank):
                f = oracle.run_field(scn, scn.beams[i], v, keep_layers=False)
                bl = [min(a, b) for a, b in zip(bl, f.info["bbox_min"])]
                bh = [max(a, b) for a, b in zip(bh, f.info["bbox_max"])]
                f.close()
            own[it % 2] = (bl, bh)
            red.submit(tens[it % 2], bl, bh)
            if it == 1:
                red.drain()
        red.drain()
        dose[:] = vols[0]            # iteration 2 used volume 0
        assert np.array_equal(vols[0], vols[1]) or rank != 0   # iteration 3 (volume 1) gives the same sum on rank 0
    elif use_bbox:
        for i in plan.shard_fields(len(scn.beams), world, rank):
            compute_field(i)
        lo = [min(b[0][a] for b in boxes) for a in range(3)] if boxes else [0, 0, 0]
        hi = [max(b[1][a] for b in boxes) for a in range(3)] if boxes else [-1, -1, -1]
        plan.reduce_dose_bbox(t, lo, hi, dist, dst=0)
    else:
        plan.run_plan(compute_field, len(scn.beams), t, dist, dst=0)
    if rank == 0:
        np.save(out_path, dose)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("use_bbox,world", [(False, 2), (True, 2), ("pipe", 2), ("pipe", 3), ("slab", 2), ("slab", 3), ("slab", 4)])
def test_sharded_plan_equals_sequential(orc, synth, tmp_path, use_bbox, world):
    import torch.multiprocessing as mp
    from raytracedicom_amd import scenarios
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "dose.npy")
    mp.spawn(_worker, args=(world, port, out, use_bbox), nprocs=world, join=True)
    got = np.load(out)
    ct, _ = scenarios.hetero_phantom(64)
    scn = scenarios.hetero_ct(synth, n=64, spots=4, pitch=8.0, n_layers=2, angles=[0.0, 90.0, 200.0, 300.0][:max(2, world)], ct=ct)
    ref = orc.compute(scn)
    assert ref.max() > 0
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-12 * float(ref.max()))
