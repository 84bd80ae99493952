"""GPU tests of size-independent properties and edge cases (through the C ABI):
bitwise reproducibility (the engine uses no float atomics), exact linearity in the spot weights, the reference water
cube C2 at full size, beams that never enter the patient, ragged sizes (steps / layers / spot maps that are not
multiples of any tile or batch size)."""
import math

import numpy as np
import pytest

from raytracedicom_amd import abi, scenarios

pytestmark = pytest.mark.gpu


def _dose(engine, scn, beams=None, options=None):
    dose = np.zeros_like(scn.ct)
    with engine.Engine(0) as eng:
        if options is not None:
            eng.set_options(options)
        eng.set_luts(scn.luts)
        eng.set_ct(scn.ct)
        eng.compute(beams if beams is not None else scn.beams, dose)
    return dose


def test_bitwise_reproducible(engine, synth):
    """Two runs of the same plan give identical bits (the reference's atomicAdd flush does not)."""
    ct, _ = scenarios.hetero_phantom(128)
    scn = scenarios.hetero_ct(synth, n=128, spots=6, pitch=7.0, n_layers=5, angles=[30.0], ct=ct)
    a = _dose(engine, scn)
    b = _dose(engine, scn)
    assert a.max() > 0
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_exact_linearity_in_spot_weights(engine, synth):
    """Scaling every spot weight by 2 (exact in binary) scales every stage, hence the dose, by exactly 2 when the
    ray-weight cut-off cannot change the live set (RAY_WEIGHT_CUTOFF = 0)."""
    ct, _ = scenarios.hetero_phantom(96)
    scn = scenarios.hetero_ct(synth, n=96, spots=5, pitch=8.0, n_layers=3, angles=[0.0], ct=ct)
    opt = abi.default_options()
    opt.ray_weight_cutoff = 0.0
    d1 = _dose(engine, scn, options=opt)
    b = scn.beams[0]
    b2 = scenarios.BeamSettings(b.spotWeights * np.float32(2.0), b.beamEnergies, b.spotSigmas, b.raySpacing, b.tracerSteps, b.sourceDist,
                                b.spotIdxToGantry, b.gantryToImIdx, b.gantryToDoseIdx)
    d2 = _dose(engine, scn, beams=[b2], options=opt)
    assert d1.max() > 0
    np.testing.assert_array_equal(d2, d1 * np.float32(2.0))


def test_c2_reference_water_cube_full_size(orc, engine, synth):
    """BASELINE.json configs[1]: the reference's WATER_CUBE_TEST, 256^3, 33x33 spots x 20 layers, 512 steps (a water field: the engine
    takes it through k_superpose_uniform, the separable superposition on the matrix cores — tests/test_gpu_parity.py checks the path)."""
    scn = scenarios.water_cube(synth, n=256, n_layers=20)
    ref = orc.compute(scn)
    dose = _dose(engine, scn)
    mx = float(ref.max())
    thr = ref > 0.1 * mx
    rel = np.abs(dose - ref)[thr] / ref[thr]
    assert rel.max() < 1e-4, rel.max()
    assert np.abs(dose - ref).max() < 1e-5 * mx
    rate, n_eval, gmax = orc.gamma_pass_rate(ref, dose, scn.spacing)
    assert rate == 1.0 and n_eval > 100000
    # spread-out Bragg peak: 20 layers -> a plateau between the first and the last peak depth on the central axis
    prof = ref[:, 128, 128]
    z = np.nonzero(prof > 0.5 * prof.max())[0]
    assert (z.max() - z.min()) > 60


def test_beam_that_never_enters_the_patient(orc, engine, synth):
    """All-air CT: nothing is inside, the field is skipped (the reference would launch with negative sizes)."""
    scn = scenarios.water_cube(synth, n=48, n_layers=2, spots=5, pitch=6.0)
    scn.ct[:] = 0.0
    dose = _dose(engine, scn)
    assert dose.max() == 0.0 and dose.min() == 0.0
    ref = orc.compute(scn)
    assert ref.max() == 0.0


@pytest.mark.parametrize("steps,n_layers,spots", [(300, 23, 3), (77, 1, 1), (513, 11, 7)])
def test_ragged_sizes(orc, engine, synth, steps, n_layers, spots):
    """Steps, layers and spot maps that are not multiples of the 32-step trace segments, the 16/8-step batches or the
    10 layer groups; a single spot; more layers than groups."""
    ct, _ = scenarios.hetero_phantom(96)
    scn = scenarios.hetero_ct(synth, n=96, spots=spots, pitch=9.0, n_layers=n_layers, angles=[10.0], steps=steps, ct=ct)
    ref = orc.compute(scn)
    dose = _dose(engine, scn)
    mx = float(ref.max())
    if mx > 0:
        assert np.abs(dose - ref).max() <= 2e-5 * mx
        thr = ref > 0.1 * mx
        assert (np.abs(dose - ref)[thr] / ref[thr]).max() < 1e-4
    else:
        assert dose.max() == 0.0


def test_fetch_unknown_name_and_double_finish(engine, synth):
    scn = scenarios.water_cube(synth, n=48, n_layers=1, spots=5, pitch=6.0)
    eng = engine.Engine(0)
    eng.set_luts(scn.luts)
    eng.set_ct(scn.ct)
    d = eng.device_alloc(4 * scn.n_voxels)
    eng.device_zero(d, 4 * scn.n_voxels)
    fld = eng.create_field(scn.beams[0], scn.dims)
    with pytest.raises(engine.RtdError):
        fld.finish()                                  # not computed yet
    fld.compute(d)
    t1, i1 = fld.finish()
    t2, i2 = fld.finish()
    assert i1 == i2 and t1["total_ms"] > 0
    with pytest.raises(engine.RtdError) as e:
        fld.fetch("no_such_buffer")
    assert "unknown name" in str(e.value)
    fld.destroy()
    eng.device_free(d)
    eng.close()


def test_clear_dose_resets_exactly_the_written_box(engine, synth):
    """rtd_field_clear_dose zeroes every voxel the field's compute changed (so a zero volume is zero again) and nothing
    outside the device-side dose box (a volume pre-filled with ones keeps its ones there)."""
    ct, _ = scenarios.hetero_phantom(96)
    scn = scenarios.hetero_ct(synth, n=96, spots=4, pitch=7.0, n_layers=3, angles=[20.0], ct=ct)
    n = scn.n_voxels
    eng = engine.Engine(0)
    eng.set_luts(scn.luts)
    eng.set_ct(scn.ct)
    d = eng.device_alloc(4 * n)
    fld = eng.create_field(scn.beams[0], scn.dims)
    with pytest.raises(engine.RtdError):
        fld.clear_dose(d)                             # nothing computed yet
    eng.device_zero(d, 4 * n)
    fld.compute(d)
    fld.finish()
    out = np.empty(n, dtype=np.float32)
    eng.to_host(out, d)
    assert out.max() > 0
    fld.clear_dose(d)
    eng.sync()
    eng.to_host(out, d)
    assert not out.any()
    ones = np.ones(n, dtype=np.float32)
    eng.to_device(d, ones)
    fld.compute(d)
    _, info = fld.finish()
    eng.to_host(out, d)
    changed = out != 1.0
    assert changed.any()
    fld.clear_dose(d)
    eng.sync()
    eng.to_host(out, d)
    assert (out[changed] == 0.0).all()                # everything the compute touched is cleared
    vol = out.reshape(scn.ct.shape)
    lo, hi = info["bbox_min"], info["bbox_max"]
    outside = np.ones(vol.shape, dtype=bool)
    outside[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = False
    assert (vol[outside] == 1.0).all()                # the clear stays inside the reported bounding box
    fld.destroy()
    eng.device_free(d)
    eng.close()


def test_pipelined_exchange_queues_destination_adds_on_a_side_stream():
    """plan.PipelinedBoxReduce under RCCL (destination rank): the adds of the received boxes are queued on a side stream at submit
    and retired by an event wait. Exercised on one GPU with a stand-in process group whose receives deliver known boxes."""
    import torch
    from raytracedicom_amd import plan

    class Work:
        def wait(self):
            return True

    class FakeDist:
        P2POp = staticmethod(lambda op, tensor, peer: (op, tensor, peer))
        isend, irecv = "isend", "irecv"

        def get_rank(self): return 0
        def get_world_size(self): return 3
        def get_backend(self): return "nccl"

        def all_gather(self, out, mine):
            boxes = [[2, 3, 4, 9, 8, 7], [0, 0, 0, 5, 5, 5], [6, 6, 6, 5, 5, 5]]     # rank 2 wrote nothing (max < min)
            for o, b in zip(out, boxes):
                o.copy_(torch.tensor(b, dtype=o.dtype))

        def batch_isend_irecv(self, ops):
            for op, tensor, peer in ops:
                assert op == "irecv" and peer == 1
                tensor.fill_(float(peer) + 0.5)
            return [Work()]                                           # coalesced, as RCCL returns it

    dev = torch.device("cuda:0")
    red = plan.PipelinedBoxReduce(FakeDist(), dst=0, static_boxes=True)
    vols = [torch.zeros((12, 12, 12), device=dev) for _ in range(2)]
    for it in range(4):
        v = vols[it % 2]
        views = red.release(v)
        if it >= 2:
            assert views is not None and len(views) == 1
            torch.cuda.synchronize()
            ref = torch.zeros_like(v)
            ref[4:8, 3:9, 2:10] = 1.0                                  # own field
            ref[0:6, 0:6, 0:6] += 1.5                                  # + the box received from rank 1
            assert torch.equal(v, ref)
            v[4:8, 3:9, 2:10] = 0.0
            for w in views:
                w.zero_()
            assert not bool(v.any())
        v[4:8, 3:9, 2:10] += 1.0                                       # "compute" this rank's field into its box
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        other = vols[(it + 1) % 2]
        other.mul_(1.0)                                                # later work on the main stream must not delay the exchange
        red.submit(v, (2, 3, 4), (9, 8, 7), ready=ready if it % 2 else None)
    red.drain()
    torch.cuda.synchronize()
    assert float(vols[1][0, 0, 0]) == 1.5 and float(vols[1][5, 5, 5]) == 2.5 and red.side is not None


def test_slab_reduce_between_three_ranks_on_one_gpu():
    """plan.PipelinedSlabReduce, RCCL branch (everything queued on a side stream at submit, retired by an event wait), between three
    ranks played by three threads on one GPU: a stand-in process group moves the packed tensors through mailboxes in stream
    order. Rank 0 must end with the sum of the three fields; on every rank, clearing the own box plus the returned views must
    restore the zero volume (bench.py's dirty-box clearing)."""
    import queue
    import threading
    import torch
    from raytracedicom_amd import plan

    world, n = 3, 20
    boxes = [[2, 3, 1, 13, 12, 17], [0, 5, 6, 18, 10, 11], [6, 0, 4, 9, 19, 15]]       # (x0, y0, z0, x1, y1, z1), overlapping
    mail = {(s, r): queue.Queue() for s in range(world) for r in range(world)}
    dev = torch.device("cuda:0")

    class Work:
        def __init__(self, recvs): self.recvs = recvs
        def wait(self):                                              # stream-ordered, like RCCL: the current stream waits for the data
            for tensor, peer, me in self.recvs:
                src, ev = mail[(peer, me)].get(timeout=60)
                torch.cuda.current_stream().wait_event(ev)
                tensor.copy_(src)
            self.recvs = []
            return True

    class FakeDist:
        isend, irecv = "isend", "irecv"
        def __init__(self, rank): self.rank = rank
        def P2POp(self, op, tensor, peer): return (op, tensor, peer)
        def get_rank(self): return self.rank
        def get_world_size(self): return world
        def get_backend(self): return "nccl"
        def all_gather(self, out, mine):
            for o, b in zip(out, boxes):
                o.copy_(torch.tensor(b, dtype=o.dtype))
        def batch_isend_irecv(self, ops):
            recvs = []
            for op, tensor, peer in ops:
                if op == "isend":
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream())
                    mail[(self.rank, peer)].put((tensor, ev))
                else:
                    recvs.append((tensor, peer, self.rank))
            return [Work(recvs)]                                      # coalesced, as RCCL returns it

    expect = torch.zeros((n, n, n), device=dev)
    for r, b in enumerate(boxes):
        expect[b[2]:b[5] + 1, b[1]:b[4] + 1, b[0]:b[3] + 1] += float(r + 1)
    errors, results = [], {}

    def run(rank):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                red = plan.PipelinedSlabReduce(FakeDist(rank), dst=0, static_boxes=True)
                b = boxes[rank]
                vols = [torch.zeros((n, n, n), device=dev) for _ in range(2)]
                for it in range(4):
                    v = vols[it % 2]
                    views = red.release(v)
                    if it >= 2:
                        if rank == 0:
                            torch.cuda.current_stream().synchronize()
                            assert torch.equal(v, expect), "rank 0 does not hold the sum of the three fields"
                        v[b[2]:b[5] + 1, b[1]:b[4] + 1, b[0]:b[3] + 1] = 0.0
                        for w in views or ():
                            w.zero_()
                        assert not bool(v.any()), "own box + returned views do not cover what the exchange wrote"
                    v[b[2]:b[5] + 1, b[1]:b[4] + 1, b[0]:b[3] + 1] += float(rank + 1)     # "compute" this rank's field
                    ready = torch.cuda.Event()
                    ready.record(torch.cuda.current_stream())
                    red.submit(v, b[:3], b[3:], ready=ready if it % 2 else None)
                red.drain()
                torch.cuda.current_stream().synchronize()
                results[rank] = (vols[0].clone(), red.side is not None)
        except Exception as e:                                        # noqa: BLE001 - reported by the main thread
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads)
    assert torch.equal(results[0][0], expect) and all(results[r][1] for r in range(world))
    axis, slabs = plan.slab_partition(boxes, world)
    for r in (1, 2):                                                  # an owner holds the complete sum inside its slab
        sl = slabs[r]
        assert torch.equal(results[r][0][sl[2]:sl[5] + 1, sl[1]:sl[4] + 1, sl[0]:sl[3] + 1],
                           expect[sl[2]:sl[5] + 1, sl[1]:sl[4] + 1, sl[0]:sl[3] + 1])


def test_field_launch_sequence_is_hipgraph_capturable(engine, synth):
    """rtd_field_clear_dose + rtd_field_compute make no host round trip, allocation or synchronisation, so the whole plan
    iteration can be captured into a hipGraph on the caller's stream; replays reproduce the directly launched dose bit for bit."""
    import torch
    n = 96
    ct, _ = scenarios.hetero_phantom(n)
    scn = scenarios.hetero_ct(synth, n=n, spots=5, pitch=7.0, n_layers=4, angles=[25.0], ct=ct)
    dev = torch.device("cuda:0")
    eng = engine.Engine(0)
    eng.set_luts(scn.luts)
    ct_dev = torch.from_numpy(scn.ct).to(dev)
    eng.set_ct_device(ct_dev.data_ptr(), scn.dims)
    fld = eng.create_field(scn.beams[0], scn.dims)
    s = torch.cuda.Stream()
    eng.set_stream(s.cuda_stream)
    ref = torch.zeros((n, n, n), dtype=torch.float32, device=dev)
    dose = torch.zeros_like(ref)
    with torch.cuda.stream(s):
        fld.compute(ref.data_ptr())
        fld.finish()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fld.clear_dose(dose.data_ptr())
            fld.compute(dose.data_ptr())
        for _ in range(3):
            g.replay()
    torch.cuda.synchronize()
    assert float(ref.max()) > 0 and torch.equal(dose, ref)
    eng.set_stream(None)
    fld.destroy()
    eng.close()


def test_reduction_tree_hand_offs_are_race_free_over_many_launches(engine, synth):
    """The superposition adds the layer groups' accumulators in a tree whose hand-offs cross XCDs (agent-scope stores / loads, node
    counters that the second arriver resets). 40 launches of the bench-sized field (14 groups, 15 output tiles, ~230 live slices:
    ~40 k hand-offs per launch) must give the same BEV dose bit for bit, and the counters must be back at zero every time (a stale
    slot or a lost arrival would change bits or leave a tile unwritten)."""
    import zlib
    ct, _ = scenarios.hetero_phantom(256)
    scn = scenarios.hetero_ct(synth, n=256, angles=[0.0], ct=ct)      # same ray grid, layers and steps as the 512^3 bench field
    with engine.Engine(0) as eng:
        eng.set_luts(scn.luts)
        eng.set_ct(scn.ct)
        f = eng.create_field(scn.beams[0], scn.dims)
        first = None
        for it in range(40):
            f.compute_bev()
            f.finish()
            bev = f.fetch("bev")
            h = zlib.crc32(bev.tobytes())
            if first is None:
                first, ref = h, bev.copy()
                assert ref.max() > 0
            elif h != first:
                bad = np.nonzero(bev.view(np.uint32) != ref.view(np.uint32))[0]
                raise AssertionError("launch %d differs from launch 0 in %d BEV values (first at flat index %d)" % (it, bad.size, int(bad[0])))
        f.destroy()
