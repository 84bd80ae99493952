#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X dose engine (BASELINE.json metric).

metric   Mvoxels/s of dose deposited: dose-grid voxels x fields / wall time of the plan, inputs (CT, LUTs, spot
         weights, workspace) already resident in HBM; the dose volume is restored to zero inside the timed step.
workload N=1: C3 = 512^3 synthetic heterogeneous CT, one field, 10x10 spots x 20 energy layers, 512 tracer steps
         (BASELINE.json configs[2] — the configuration the metric is quoted on).
         N>1: one field per GPU (gantry angles 360/N apart, same CT replicated): weak scaling, per-GPU work fixed.
step     N=1: all kernels of the field, its transfer WRITING the field's dose box into the volume that is zero elsewhere
         (= a fresh zero volume + the field, without clearing anything) + rtd_field_finish (pipelined by one plan).
         N>1: every rank computes its field up to the beam's-eye-view (BEV) dose, ONE RCCL all-gather moves the packed BEV
         slabs (~10 MB each; the dose boxes they turn into are 60-83 MB), and every rank runs the fan -> dose transfer of EVERY
         field, in field order, into its own slab of the dose volume (plan.BevExchange). The plan's volume is left sharded by
         slabs across the GPUs; it is bit-identical to the sequential one-GPU accumulation of the N fields.

One JSON line on rank 0; `roofline` describes the dominant kernel (kernel superposition) from the kernels' own dispatch
timestamps taken by the engine on the launch stream during the timed steps; `cpu_baseline` is the CPU oracle timed on this host.
Also reported: `ms_plan_latency` (ONE plan, nothing to pipeline against) and, at N=1, `ms_plan_end_to_end` (the reference's
"total global execution time": CT + LUT upload, dose up, all kernels, dose down, from page-locked host buffers).

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--size 512] [--no-cpu]
       N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
            or plain `python bench.py --gpus N ...`, which starts exactly that launcher as a child process and relays its output
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md, Chip-level parameters)
F32_MATRIX_PEAK_TF = 157.3   # v_mfma_f32_16x16x4_f32 peak = f32 vector peak (MI355X_MICROARCH.md, Matrix cores)


def algorithmic_bytes(info, dims, ct_footprint):
    """Algorithmic HBM bytes of one field, per stage (SURVEY.md 8(d) formula; fp32).
    R rays, S steps, P padded BEV slice, SA = sum over layers of live steps, V_bb = dose voxels in the bounding box, N_fp = distinct
    CT voxels the tracer reads (counted by the oracle; without the oracle leg the bound min(N, 8*R*S))."""
    W, H, L = info["ray_dims"]
    R, P = W * H, (W + 64) * (H + 64)
    S = info["_steps"]
    SA = info["live_steps"]
    Z = max(0, info["beam_first_calculated_passive"] - info["beam_first_inside"])
    bb = [max(0, info["bbox_max"][i] - info["bbox_min"][i] + 1) for i in range(3)]
    vbb = bb[0] * bb[1] * bb[2]
    n = dims[0] * dims[1] * dims[2]
    nfp = ct_footprint if ct_footprint is not None else min(n, 8 * R * S)
    out = {
        "tracer": 4 * nfp + 8 * R * S + 8 * R + 4 * R * S,   # CT + density,WEPL + 2 int maps + min-WEPL read
        "fill": 16 * R * SA + 4 * R * L + 4 * R * SA,        # read rho,WEPL; write idd,1/sigma; weights; tile radius
        "superposition": 8 * R * SA + 8 * P * SA,            # read idd,1/sigma; RMW padded BEV
        "bev_zero_read": 4 * P * S + 4 * P * Z,
        "transfer": 8 * vbb,
    }
    out["total"] = sum(out.values())
    return out


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=512, help="CT edge in voxels (512 = C3/C4, 768 = C5)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / parity / end-to-end legs")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--streams", type=int, default=1,
                    help="N=1 only, not the default: launch consecutive plans round-robin on this many HIP streams, so the kernels of "
                         "independent plans overlap (throughput of a multi-field plan on one GPU; per-kernel durations in stage_ms / "
                         "roofline then include the sharing of the GPU)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary leg (throughput_streams: the same steps once more with consecutive plans on 4 HIP streams). "
                         "profiles/collect.sh passes this: the leg's overlapped kernels would enter the per-kernel averages of a "
                         "`rocprofv3 --stats -- python bench.py` beside the timed single-stream loop, whose kernel durations are what `roofline` reports")
    ap.add_argument("--secondary", action="store_true", help=argparse.SUPPRESS)      # (accepted for older command lines: the default now)
    ap.add_argument("--secondary-streams", type=int, default=4, help="HIP streams of the secondary leg")
    ap.add_argument("--exchange-selftest", action="store_true",
                    help="N=1 only: run the N>1 code path (process group, all-gather, fused slab transfer) with a world of one rank — "
                         "exercises the RCCL calls on a one-GPU box; not a benchmark mode")
    ap.add_argument("--backend", default="nccl", help="process-group backend; 'gloo' only to rehearse N>1 on a one-GPU box")
    ap.add_argument("--debug-head-us", default="", help=argparse.SUPPRESS)   # tests only: per-rank field times for the slab cut, e.g. "600,90000" (rank 1 gets an empty slab)
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a CHILD process (this process has made no
    # GPU call yet and never will: it only relays the child's output — rank 0's JSON line — and exit code).
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    from raytracedicom_amd import abi, engine, luts, plan, scenarios

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d started with WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the dose engine has no CPU fallback")
    xchg = world > 1 or args.exchange_selftest          # the N>1 path (a world of one rank only with --exchange-selftest)
    if args.exchange_selftest and world == 1:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
    dev_index = local_rank % torch.cuda.device_count()     # one GPU per rank on a real node; shared only in a gloo rehearsal
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if xchg:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)
    # Every torch op, every engine launch and every collective of this process is issued on ONE explicit stream: torch's default
    # stream has handle 0, which the C ABI reads as "the handle's own stream" — the engine would then run unordered against torch.
    main_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(main_stream)

    # ---- synthetic inputs (same CT on every rank; field r at gantry angle r*360/world) ----
    n = args.size
    es = luts.synth_luts()
    ct_np, voxel = scenarios.hetero_phantom(n)
    angles = [r * 360.0 / world for r in range(world)]
    scn = scenarios.hetero_ct(es, n=n, angles=angles, ct=ct_np)
    beam = scn.beams[rank]
    n_vox = scn.n_voxels

    eng = engine.Engine(dev_index)
    opt = abi.default_options()
    opt.fine_grained_timing = 1
    eng.set_options(opt)
    eng.set_stream(main_stream.cuda_stream)
    assert int(eng.stream() or 0) == int(main_stream.cuda_stream) != 0
    eng.set_luts(es)
    ct_dev = torch.from_numpy(ct_np).to(dev)
    eng.set_ct_device(ct_dev.data_ptr(), scn.dims)
    n_streams = max(1, args.streams)
    assert n_streams == 1 or world == 1, "--streams is a one-GPU mode"
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams > 1 else None
    n_vol = 2 if xchg else (n_streams + 1 if n_streams > 1 else 1)
    doses = [torch.zeros((n, n, n), dtype=torch.float32, device=dev) for _ in range(n_vol)]
    dose = doses[0]
    # Field objects of the same beam alternate, so plan i+1 is launched before plan i is finished (rtd_field_finish waits for
    # its own field's last kernel only): the device does not idle while the host reads back timing and geometry.
    flds = [eng.create_field(beam, scn.dims) for _ in range(3 if xchg else n_streams + 1)]
    fld = flds[0]
    ex = None
    if xchg:
        remote = {r: eng.create_field(scn.beams[r], scn.dims, remote=True) for r in range(world) if r != rank}
        ex = plan.BevExchange(dist, rank, world, remote, scn.dims, new_bytes=lambda k: torch.empty(int(k), dtype=torch.uint8, device=dev),
                               zero_box=lambda b, lo, hi: doses[b][lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1].zero_(),
                               transfer_all=lambda fs, d, lo, hi: eng.transfer_fields_init(fs, d, lo, hi))
        # This rank's field measured alone (best of 8 runs after 3 warm-ups): time up to its BEV dose and transfer cost per box
        # voxel. The slabs are then cut so that the ranks' step times come out even — fields differ in cost
        # (profiles/r02_angles.json) and a rank with an expensive field gets a thinner slab; the dose does not depend on the cut.
        head_us, ps_kvox = [], []
        for j in range(11):
            fld.compute_bev()
            fld.transfer(doses[0].data_ptr())
            t_j, _ = fld.finish()
            if j >= 3 and t_j.get("transfer_voxels", 0) > 0:
                head_us.append(1000.0 * (t_j["total_ms"] - t_j["transforming_ms"]))
                ps_kvox.append(1.25 * t_j["transforming_ms"] * 1e9 / t_j["transfer_voxels"] * 1000.0)   # + 25 %: the clear of the box
        doses[0].zero_()
        fld.compute_bev()
        if args.debug_head_us:
            head_us = [float(args.debug_head_us.split(",")[rank])]
        ex.setup(fld, head_us=min(head_us) if head_us else None, transfer_ps_per_kvoxel=min(ps_kvox) if ps_kvox else None)   # message capacity, dose boxes, slab partition: the fields keep their geometry
        fld.finish()
    main_stream.synchronize()
    step_no = [0]
    in_flight = []                      # (step index, field) launched, not yet finished

    def launch():
        """Launch one plan iteration (asynchronous). N=1: fresh dose volume + all kernels of the field. N>1: this rank's field up to
        its BEV dose, slab packed, all-gather started."""
        i = step_no[0]
        step_no[0] += 1
        f = flds[i % len(flds)]
        if ex is None:
            d = doses[i % len(doses)]
            if streams is not None:
                eng.set_stream(streams[i % n_streams].cuda_stream)
            # fresh dose volume: the volume is zero outside the box the previous plan (same field geometry) wrote, so the field's
            # transfer WRITES its dose box (rtd_field_transfer_init) — no clear of 512^3, no clear of the box, no read-modify-write
            f.compute_bev()
            f.transfer_init(d.data_ptr())
        else:
            b = i % 2
            if i >= 2:
                ex.clear(f, b, doses[b].data_ptr())      # what plan i-2 wrote into this volume (all fields, this rank's slab)
            f.compute_bev()
            ex.post(f, b)
        in_flight.append((i, f))

    completed = set()                   # N>1: plans whose exchange-side work (all-gather wait + fused slab transfer) has been issued

    def complete_pending():
        for i, f in in_flight:
            if i not in completed:
                ex.complete(f, i % 2, doses[i % 2].data_ptr())
                completed.add(i)

    def retire():
        """Finish the oldest launched plan (N>1: after issuing its slab transfer if that has not happened yet)."""
        i, f = in_flight.pop(0)
        if ex is not None and i not in completed:
            ex.complete(f, i % 2, doses[i % 2].data_ptr())
        completed.discard(i)
        r = f.finish()
        if ex is not None:
            ex.check(f)                 # the plan still fits what setup() froze (message capacity, dose box)
        return r

    def step():
        """One plan iteration in steady state. N=1: launch plan i, then finish plan i-1 — its last kernel sits in front of plan i's in
        the stream, so the host never waits with an empty queue. N>1: launch plan i (up to the all-gather), issue the slab transfer of
        plan i-1 (it waits, in stream order, for that plan's all-gather, which had a whole plan's kernels to finish behind), then
        finish plan i-2 for the same reason — its transfer was queued in front of plan i's kernels (three field objects alternate)."""
        launch()
        if ex is not None:
            complete_pending()
            if len(in_flight) > 2:
                return retire()
            return None
        if len(in_flight) > n_streams:
            return retire()
        return None

    def barrier():
        torch.cuda.synchronize()
        if xchg:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    while in_flight:
        retire()
    buckets = {}
    n_timed = 0
    info = None
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = step()
        if r is not None:
            for k, v in r[0].items():
                if isinstance(v, float):
                    buckets[k] = buckets.get(k, 0.0) + float(v)
            info = r[1]
            n_timed += 1
    while in_flight:                    # the last plan is finished inside the timed region
        t, info = retire()
        for k, v in t.items():
            if isinstance(v, float):
                buckets[k] = buckets.get(k, 0.0) + float(v)
        n_timed += 1
    barrier()
    elapsed = time.perf_counter() - t0
    assert n_timed == args.steps
    if xchg:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = 1000.0 * elapsed / args.steps

    # ---- latency of ONE plan: nothing to pipeline against (launch, finish, drained stream), mean of 5 ----
    lat = []
    for _ in range(5):
        barrier()
        l0 = time.perf_counter()
        launch()
        retire()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - l0)
    ms_latency = 1000.0 * sum(lat) / len(lat)
    if xchg:
        tl = torch.tensor([ms_latency], dtype=torch.float64, device=dev)
        dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        ms_latency = float(tl.item())

    # ---- secondary figure (N=1): throughput with consecutive plans on 4 HIP streams — independent plans (a plan's fields, or a queue
    #      of patients) overlap: the latency-bound launches of one (scan, plans, the tails of fill / superposition / transfer) run beside
    #      the issue-bound kernels of another. NOT the headline: `value`, stage_ms and the roofline come from the single-stream loop
    #      above, where a kernel's duration is that of the kernel alone and a step is one plan's latency chain. ----
    multi = None
    if not xchg and n_streams == 1 and not args.no_secondary:
        ns = max(2, args.secondary_streams)
        ss = [torch.cuda.Stream(device=dev) for _ in range(ns)]
        fl = [eng.create_field(beam, scn.dims) for _ in range(ns)]
        vols = [torch.zeros((n, n, n), dtype=torch.float32, device=dev) for _ in range(ns)]
        torch.cuda.synchronize()
        busy = [False] * ns

        def go(j):
            eng.set_stream(ss[j].cuda_stream)
            if busy[j]:
                fl[j].finish()
            fl[j].compute_bev()
            fl[j].transfer_init(vols[j].data_ptr())
            busy[j] = True

        for j in range(ns):
            go(j)
        torch.cuda.synchronize()
        m0 = time.perf_counter()
        for i in range(args.steps):
            go(i % ns)
        for j in range(ns):
            fl[j].finish()
        torch.cuda.synchronize()
        m_el = time.perf_counter() - m0
        same = all(torch.equal(v, vols[0]) for v in vols[1:])
        eng.set_stream(main_stream.cuda_stream)
        multi = {"streams": ns, "ms_per_step": round(1000.0 * m_el / args.steps, 4), "mvoxels_s": round(n_vox * args.steps / m_el / 1e6, 1),
                 "volumes_identical": bool(same)}       # (+ path_frac_of_hbm_peak below, once the algorithmic bytes are known)
        for f in fl:
            f.destroy()
        del vols

    # ---- self-checks (untimed) ----
    # (a) a volume restored by the dirty-box clears of launch() is bit-identical to the same plan computed into a fully zeroed volume
    launch()
    i_chk, f_chk = in_flight[0]
    retire()
    last = doses[i_chk % len(doses)]
    ref = torch.zeros_like(last)
    if ex is None:
        fld2 = flds[(i_chk + 1) % len(flds)]
        fld2.compute(ref.data_ptr())
        fld2.finish()
    else:
        ex.complete(f_chk, i_chk % 2, ref.data_ptr())      # same gathered slabs, same field order, fresh volume
        f_chk.finish()
    torch.cuda.synchronize()
    clear_ok = torch.tensor([1 if torch.equal(last, ref) else 0], dtype=torch.int64, device=dev)
    if xchg:
        dist.all_reduce(clear_ok, op=dist.ReduceOp.MIN)
    clear_check = bool(int(clear_ok.item()))
    # (b) N>1: the slabs of all ranks together hold the sum of all fields (each rank also transfers its own field unclipped)
    reduce_check = None
    if xchg:
        slab_sum = last.sum(dtype=torch.float64).reshape(1)
        ref.zero_()
        f_chk.transfer(ref.data_ptr())
        f_chk.finish()
        torch.cuda.synchronize()
        field_sum = ref.sum(dtype=torch.float64).reshape(1)
        dist.all_reduce(slab_sum, op=dist.ReduceOp.SUM)
        dist.all_reduce(field_sum, op=dist.ReduceOp.SUM)
        reduce_check = abs(float(slab_sum.item()) - float(field_sum.item())) / max(float(field_sum.item()), 1e-300)
    # (c) N>1: the plan's volume ASSEMBLED on rank 0 — every rank's volume is zero outside its slab, so a sum over the ranks
    #     (x + 0 = x exactly) is the gather of the slabs — against the reference's sequential beam loop on one GPU
    #     (kernel_wrapper.cu:601, `+=` at :92): all fields accumulated into one zeroed volume in field order. Bit for bit.
    assembled_check = None
    if xchg:
        if args.backend == "nccl":
            dist.reduce(last, dst=0, op=dist.ReduceOp.SUM)
            assembled = last
        else:
            host_part = last.cpu()
            dist.reduce(host_part, dst=0, op=dist.ReduceOp.SUM)
            assembled = host_part.to(dev) if rank == 0 else None
        if rank == 0:
            ref.zero_()
            for r in range(world):
                fr = flds[0] if r == rank else eng.create_field(scn.beams[r], scn.dims)
                fr.compute(ref.data_ptr())
                fr.finish()
                if r != rank:
                    fr.destroy()
            torch.cuda.synchronize()
            assembled_check = bool(torch.equal(assembled, ref))
        # the headers of the messages this rank received carry the senders' device-side error flags: none may be set
        # (attached here: a rank whose slab came out empty — a late rank gets none — has transferred nothing and attached nothing)
        ex.attach_all(i_chk % 2)
        for f_r in ex.remote.values():
            f_r.finish()
        last.zero_()
    del ref

    if rank == 0:
        info["_steps"] = beam.tracerSteps
        stage_ms = {k: buckets[k] / args.steps for k in buckets if k.endswith("_ms")}
        ct_fp = None
        oracle = None
        if not args.no_cpu and not xchg:
            from oracle import oracle
            ncpu = args.cpu_threads or min(16, os.cpu_count() or 1)
            oracle.set_threads(ncpu)
            ct_fp = oracle.ct_footprint(scn, beam)
        alg = algorithmic_bytes(info, scn.dims, ct_fp)
        ks_ms = stage_ms["superp_kernel_ms"]
        ks_gbs = alg["superposition"] / (ks_ms * 1e-3) / 1e9 if ks_ms > 0 else 0.0
        traffic, pmc = None, {}
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                pmc = json.load(open(tfile)).get("k_superpose", {})
                traffic = pmc.get("hbm_bytes_per_launch")
            except Exception:
                traffic, pmc = None, {}
        mfma_per_launch = pmc.get("mfma_insts_per_launch")
        valu_per_launch = pmc.get("valu_insts_per_launch")
        # the general superposition of the field: k_superpose_sweep when every batch radius is within its reach (<= 16), else k_superpose_mfma
        ks_name = pmc.get("kernel") or ("rtd::k_superpose_sweep" if info["max_radius"] <= 16 and not os.environ.get("RTD_NO_SWEEP") else "rtd::k_superpose_mfma")
        if os.environ.get("RTD_NO_SWEEP"):
            ks_name, traffic, pmc = "rtd::k_superpose_mfma", None, {}      # (profiles/traffic.json describes the default path)
        roof = {"kernel": ks_name, "bound": "valu+mfma issue", "achieved": round(ks_gbs, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(ks_gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
                "avg_launch_ms": round(ks_ms, 4), "algorithmic_bytes_per_launch": alg["superposition"],
                "note": "contract form: algorithmic bytes = SURVEY.md 8(d) superposition term 8*(R+P)*sum(A_l), against the 8 TB/s HBM peak. "
                        "The kernel is NOT HBM-bound: PMC counters (profiles/, per launch) show its SIMDs busy issuing vector ALU + f32-MFMA "
                        "instructions (issue_cycle_frac); traffic = (2*FETCH_SIZE+WRITE_SIZE)*1024 from profiles/traffic.json"}
        if mfma_per_launch and ks_ms > 0:
            roof["mfma_tflops"] = round(mfma_per_launch * 2048 / (ks_ms * 1e-3) / 1e12, 2)      # 16x16x4 MFMA = 1024 MACs
            roof["mfma_peak_tflops"] = F32_MATRIX_PEAK_TF
            if valu_per_launch and pmc.get("simd_cycles_per_launch"):
                roof["issue_cycle_frac"] = round((4.0 * valu_per_launch + 32.0 * mfma_per_launch) / pmc["simd_cycles_per_launch"], 3)
        result = {
            "metric": "Mvoxels/s dose deposited (%d^3 CT, 1 field per GPU), inputs resident in HBM" % n,
            "value": round(world * n_vox / elapsed * args.steps / 1e6, 3),
            "unit": "Mvoxels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d^3 synthetic heterogeneous CT (HU->density/SP LUTs), %d field(s) one per GPU, "
                                   "10x10 spots x 20 layers = 2000 spots, 512 tracer steps, 1 mm rays"
                                   % ("C5" if n == 768 else ("C3" if world == 1 else "C4" if world == 4 else "C3 fields at %d angles" % world) if n == 512 else "size %d" % n, n, world),
                       "plans_in_flight_on_streams": n_streams, "ray_grid": info["ray_dims"], "live_steps": info["live_steps"], "max_radius": info["max_radius"],
                       "bbox_voxels": int(np.prod([info["bbox_max"][i] - info["bbox_min"][i] + 1 for i in range(3)])),
                       "ct_footprint_voxels": ct_fp,
                       "exchange": ("none" if not xchg else
                                    "one RCCL all-gather of the packed BEV slabs per plan (%d x %.1f MB), overlapped with the next plan's kernels; every "
                                    "rank writes its slab of the dose volume (axis %d, ranges %s) with all fields, in field order, in one fused launch: the volume stays "
                                    "sharded by slabs, no dose data crosses xGMI; slabs cut for even step times from the ranks' measured field times %s us and "
                                    "transfer rates %s ps per 1000 voxels" % (world, ex.cap / 1e6, ex.axis, ex.ranges, ex.heads_us, ex.rates_ps_kvox))},
            "ms_plan": round(ms_per_step, 4),
            "ms_plan_latency": round(ms_latency, 4),
            "throughput_streams": multi,
            "reduce_check_rel_err": reduce_check, "clear_check": clear_check, "assembled_check": assembled_check,
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "algorithmic_bytes": alg,
            "path_gbs": round(alg["total"] / (stage_ms["total_ms"] * 1e-3) / 1e9, 2),
            "path_frac_of_hbm_peak": round(alg["total"] / (stage_ms["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "roofline": roof,
        }
        if multi is not None:       # the pipelined leg against the same algorithmic bytes (wall time of its steps, not a stage sum)
            multi["path_frac_of_hbm_peak"] = round(alg["total"] / (multi["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            multi["what"] = ("the same %d plans once more, consecutive plans round-robin on %d HIP streams (one field object and one dose volume "
                             "each): independent plans overlap on the GPU; not the headline — a plan's own kernels are a latency chain, which "
                             "`value` and `ms_plan_latency` report" % (args.steps, multi["streams"]))
        if oracle is not None:
            # ---- CPU baseline: the oracle (kind "port") on this host. Headline sample: 4 fields of the bench workload on ncpu threads. ----
            cpu_dose = np.zeros_like(scn.ct)
            c0 = time.perf_counter()
            of = oracle.run_field(scn, beam, cpu_dose, keep_layers=False)
            cpu_first = time.perf_counter() - c0
            extra = scenarios.hetero_ct(es, n=n, angles=[90.0, 180.0, 270.0], ct=ct_np)
            scratch = np.zeros_like(scn.ct)
            for b2 in extra.beams:
                oracle.run_field(extra, b2, scratch, keep_layers=False).close()
            cpu_s = time.perf_counter() - c0
            n_cpu_fields = 1 + len(extra.beams)
            # one thread: the bench field (C3) once, and the mandatory C1 (water cube 128^3, one layer) on 1 and ncpu threads
            oracle.set_threads(1)
            scratch[:] = 0.0
            c1 = time.perf_counter()
            oracle.run_field(scn, beam, scratch, keep_layers=False).close()
            cpu_1t = time.perf_counter() - c1
            del scratch
            c1scn = scenarios.water_cube(es, n=128, n_layers=1)
            c1d = np.zeros_like(c1scn.ct)
            c1 = time.perf_counter()
            oracle.run_field(c1scn, c1scn.beams[0], c1d, keep_layers=False).close()
            c1_1t = time.perf_counter() - c1
            oracle.set_threads(ncpu)
            c1d[:] = 0.0
            c1 = time.perf_counter()
            oracle.run_field(c1scn, c1scn.beams[0], c1d, keep_layers=False).close()
            c1_nt = time.perf_counter() - c1
            # the reference's OWN CPU code on C1: in water the superposition of a slice is xConvCpuScat + yConvCpu of that slice
            # (src/cpu_convolution_1d.cpp, compiled where it lies into oracle/_ref/libref.so); timed on one core, checked against
            # the oracle's superposition stage and against the HIP engine's BEV dose of the same field
            from oracle import ref_cpu_path
            ref_leg = None
            ofc1 = oracle.run_field(c1scn, c1scn.beams[0], np.zeros_like(c1scn.ct), keep_layers=True)
            kind = "reference" if ref_cpu_path.ref_lib() is not None else "port"
            sep = ref_cpu_path.separable_bev(ofc1, c1scn.beams[0], which=kind, repeat=20)
            if sep is not None:
                sep_bev, sep_s, sep_slices, sep_rad = sep
                cw, ch, _ = ofc1.info["ray_dims"]
                obev = ofc1.get("bev").reshape(-1, ch + 64, cw + 64)
                big = obev > 1e-3 * obev.max()
                with engine.Engine(dev_index) as e1:
                    e1.set_luts(es)
                    e1.set_ct(c1scn.ct)
                    f1 = e1.create_field(c1scn.beams[0], c1scn.dims)
                    f1.compute_bev()
                    f1.finish()
                    gbev = f1.fetch("bev").reshape(-1, ch + 64, cw + 64)
                    f1.destroy()
                sb = sep_bev[:obev.shape[0]].astype(np.float64)
                ref_leg = {"kind": kind, "cores": 1,
                           "what": "C1's %d BEV slices (%dx%d rays, radius <= %d) through the reference's xConvCpuScat + yConvCpu%s"
                                   % (sep_slices, cw, ch, sep_rad, "" if kind == "reference" else " (oracle's restatement: oracle/_ref not built)"),
                           "seconds": round(sep_s, 4), "bev_mvoxels_s": round(sep_slices * (cw + 2 * sep_rad) * (ch + 2 * sep_rad) / sep_s / 1e6, 1),
                           "max_rel_diff_oracle_superposition": float((np.abs(sb - obev)[big] / obev[big]).max()),
                           "max_rel_diff_hip_engine_bev": float((np.abs(sb - gbev)[big] / obev[big]).max())}
            ofc1.close()
            host = dose.cpu().numpy() if not xchg else None
            result["cpu_baseline"] = {
                "value": round(n_cpu_fields * n_vox / cpu_s / 1e6, 3), "unit": "Mvoxels/s", "cores": ncpu, "kind": "port",
                "cpu_model": cpu_model(),
                "cores_note": "threads used = %d; this process may use %d of the host's logical CPUs (the GPU box's container share, not the whole socket)"
                              % (ncpu, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)),
                "sample": "%d fields of the same workload (the N=1 field + gantry 90/180/270), CPU oracle (oracle/rtd_oracle.c) "
                          "with %d OpenMP threads: %.1f s wall = %.0f CPU-s; first field alone %.2f s"
                          % (n_cpu_fields, ncpu, cpu_s, cpu_s * ncpu, cpu_first),
                "one_thread": {"value": round(n_vox / cpu_1t / 1e6, 3), "unit": "Mvoxels/s", "seconds_per_field": round(cpu_1t, 3), "workload": "C3 field, 1 thread"},
                "c1": {"workload": "C1: water cube 128^3, one G000 field, one energy layer (BASELINE.json configs[0])",
                       "one_thread_s": round(c1_1t, 4), "all_threads_s": round(c1_nt, 4),
                       "one_thread_mvox_s": round(c1scn.n_voxels / c1_1t / 1e6, 3), "all_threads_mvox_s": round(c1scn.n_voxels / c1_nt / 1e6, 3)},
                "reference_cpu_convolution_1d": ref_leg}
            # the timed loop's last volume was cleared by the self-check: recompute the field for the parity leg
            dose.zero_()
            fld.compute(dose.data_ptr())
            fld.finish()
            torch.cuda.synchronize()
            host = dose.cpu().numpy()
            rate, n_eval, gmax = oracle.gamma_pass_rate(cpu_dose, host, scn.spacing)
            thr = cpu_dose > 0.1 * cpu_dose.max()
            max_rel = float((np.abs(host - cpu_dose)[thr] / cpu_dose[thr]).max())
            result["parity"] = {"gamma_1pct_1mm_pass": rate, "gamma_voxels": n_eval, "gamma_max": round(gmax, 4),
                                "max_rel_diff_above_10pct": max_rel}
            of.close()
            # ---- end to end, as the reference times it (kernel_wrapper.cu:410-414 -> 1356-1360): context exists; LUT + CT upload, dose up,
            #      all kernels, dose down, from page-locked host buffers (HostPinnedImage3D, host_image_3d.cuh:23-32) ----
            e2e = {}
            ct_host = np.ascontiguousarray(ct_np)
            dose_host = np.zeros_like(ct_host)
            engine.host_register(ct_host)
            engine.host_register(dose_host)
            try:
                with engine.Plan([dev_index]) as pl:
                    for name in ("first_call", "second_call"):
                        dose_host[:] = 0.0
                        e0 = time.perf_counter()
                        pl.set_luts(es)
                        pl.set_ct(ct_host, deferred=True)     # as the C++ shim does: the CT outlives the call
                        _, pt = pl.compute([beam], dose_host)
                        e2e[name] = {"ms": round(1000.0 * (time.perf_counter() - e0), 3),
                                     "compute_call": {k: round(v, 3) for k, v in pt.items() if k.endswith("_ms")}}
                    e2e["matches_resident_path"] = bool(np.array_equal(dose_host, host))
            finally:
                engine.host_unregister(ct_host)
                engine.host_unregister(dose_host)
            e2e["what"] = ("the reference's timed span (kernel_wrapper.cu:410-414 -> 1356-1360) through rtd_plan_* as the C++ shim calls it, pinned host "
                           "buffers: LUTs up, the box of the %d MB CT that the beam's rays cross up (rtd_plan_set_ct_deferred), the block of the %d MB "
                           "dose volume that the field can change up, all kernels, that block down; first_call includes the workspace allocations "
                           "(as the reference's per-beam cudaMallocs do), second_call reuses them. Round-2 start, whole volumes both ways: 30 ms"
                           % (ct_host.nbytes // 10 ** 6, dose_host.nbytes // 10 ** 6))
            result["ms_plan_end_to_end"] = e2e
            # the metric on the REFERENCE'S timed span (uploads + kernels + download); `value` above is the resident-data rate
            result["value_reference_span_mvoxels_s"] = round(n_vox / (e2e["second_call"]["ms"] * 1e-3) / 1e6, 1)
        print(json.dumps(result))
    for f in flds:
        f.destroy()
    if ex is not None:
        for f in ex.remote.values():
            f.destroy()
    eng.close()
    if xchg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
