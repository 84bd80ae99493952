#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X dose engine (BASELINE.json metric).

metric   Mvoxels/s of dose deposited: dose-grid voxels x fields / wall time of the plan, inputs (CT, LUTs, spot
         weights, workspace) already resident in HBM; the dose volume is zeroed inside the timed step.
workload N=1: C3 = 512^3 synthetic heterogeneous CT, one field, 10x10 spots x 20 energy layers, 512 tracer steps
         (BASELINE.json configs[2] — the configuration the metric is quoted on).
         N>1: one field per GPU (gantry angles 360/N apart, same CT replicated), per-rank dose volumes summed into
         rank 0 with one RCCL reduce inside the timed step (weak scaling: per-GPU work fixed).
step     = zero the dose volume + rtd_field_compute (all kernels of the field) + rtd_field_finish (sync)
         [+ N>1: reduce of the union of the fields' bounding boxes into rank 0].

One JSON line on rank 0; `roofline` describes the dominant kernel (kernel superposition) from HIP events recorded
by the engine on the launch stream during the timed steps; `cpu_baseline` is the CPU oracle timed on this host.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--size 512] [--no-cpu]
       N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
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

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, Chip-level parameters)


def algorithmic_bytes(info, dims):
    """Algorithmic HBM bytes of one field, per stage (SURVEY.md §8d formula; fp32).
    R rays, S steps, P padded BEV slice, SA = sum over layers of live steps, V_bb = dose voxels in the bounding box.
    The tracer's CT term uses the ray-grid footprint bound min(N, 8*R*S) (distinct voxels touched <= samples*8)."""
    W, H, L = info["ray_dims"]
    R, P = W * H, (W + 64) * (H + 64)
    S = info["_steps"]
    SA = info["live_steps"]
    Z = max(0, info["beam_first_calculated_passive"] - info["beam_first_inside"])
    bb = [max(0, info["bbox_max"][i] - info["bbox_min"][i] + 1) for i in range(3)]
    vbb = bb[0] * bb[1] * bb[2]
    n = dims[0] * dims[1] * dims[2]
    out = {
        "tracer": 4 * min(n, info.get("_ct_footprint", 8 * R * S)) + 8 * R * S + 8 * R + 4 * R * S,   # CT + density,WEPL + 2 int maps + min-WEPL read
        "fill": 16 * R * SA + 4 * R * L + 4 * R * SA,                                       # read rho,WEPL; write idd,1/sigma; weights; tile radius
        "superposition": 8 * R * SA + 8 * P * SA,                                           # read idd,1/sigma; RMW padded BEV
        "bev_zero_read": 4 * P * S + 4 * P * Z,
        "transfer": 8 * vbb,
    }
    out["total"] = sum(out.values())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=512, help="CT edge in voxels (512 = C3/C4, 768 = C5)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--streams", type=int, default=1,
                    help="N=1 only, not the default: launch consecutive plans round-robin on this many HIP streams, so the kernels of "
                         "independent plans overlap (throughput of a multi-field plan on one GPU; per-kernel durations in stage_ms / "
                         "roofline then include the sharing of the GPU)")
    ap.add_argument("--backend", default="nccl", help="process-group backend; 'gloo' only to rehearse N>1 on a one-GPU box")
    args = ap.parse_args()

    import torch
    from raytracedicom_amd import abi, engine, luts, plan, scenarios

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the dose engine has no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()     # one GPU per rank on a real node; shared only in a gloo rehearsal
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)

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
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_luts(es)
    ct_dev = torch.from_numpy(ct_np).to(dev)
    eng.set_ct_device(ct_dev.data_ptr(), scn.dims)
    # N>1: two alternating dose volumes so that the reduce of plan i (communication stream) overlaps the kernels of plan i+1
    n_streams = max(1, args.streams)
    assert n_streams == 1 or world == 1, "--streams is a one-GPU mode"
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams > 1 else None
    doses = [torch.zeros((n, n, n), dtype=torch.float32, device=dev) for _ in range(2 if world > 1 else (n_streams + 1 if n_streams > 1 else 1))]
    dose = doses[0]
    # Two field objects of the same beam alternate, so plan i+1 is launched before plan i is finished (rtd_field_finish waits
    # for its own field's last kernel only): the device does not idle while the host reads back timing and geometry.
    flds = [eng.create_field(beam, scn.dims) for _ in range(n_streams + 1)]
    fld = flds[0]
    # N = 2: rank 1 sends its dose box straight to rank 0. N >= 3: point-to-point reduce-scatter + gather (pieces to owner slabs,
    # complete slabs to rank 0): the links into rank 0 carry (box + union box) / N instead of a whole box each (plan.py).
    # RTD_BENCH_REDUCE=direct|slab overrides. The fields keep their geometry: the 6-int boxes are exchanged once.
    reduce_mode = os.environ.get("RTD_BENCH_REDUCE", "slab" if world >= 3 else "direct")
    reducer = None
    if world > 1:
        reducer = (plan.PipelinedSlabReduce if reduce_mode == "slab" else plan.PipelinedBoxReduce)(dist, static_boxes=True)
    torch.cuda.synchronize()
    step_no = [0]
    in_flight = []                      # (field, dose volume) launched, not yet finished

    def launch():
        """Launch one plan iteration: fresh dose volume + all kernels of this rank's field (asynchronous)."""
        i = step_no[0]
        d, f = doses[i % len(doses)], flds[i % len(flds)]
        if streams is not None:
            eng.set_stream(streams[i % n_streams].cuda_stream)
        first_use = i < len(doses)              # the volume is still the all-zero allocation
        step_no[0] += 1
        views = reducer.release(d) if reducer is not None else None  # the exchange that used this volume two plans ago has completed
        # fresh dose volume: only the voxels the previous plan wrote are cleared (rtd_field_clear_dose: the field's device-side
        # dose box; on the destination rank also the boxes that received the other ranks' dose), not all 512^3
        if not first_use:
            if f.computed:
                f.clear_dose(d.data_ptr())
            else:
                d.zero_()
            for v in views or ():
                v.zero_()
        f.compute(d.data_ptr())
        ready = torch.cuda.Event() if reducer is not None else None
        if ready is not None:
            ready.record(torch.cuda.current_stream())     # the exchange of this plan is ordered behind this plan only
        in_flight.append((f, d, ready))

    def retire():
        """Finish the oldest launched plan: wait for its last kernel, per-stage hipEvent times + bounding box, [N>1: start the
        reduce of the union bounding box into rank 0, left in flight while later plans run]."""
        f, d, ready = in_flight.pop(0)
        t, info = f.finish()
        if reducer is not None:
            reducer.submit(d, info["dose_box_min"], info["dose_box_max"], ready=ready)
        return t, info

    def step():
        """One plan iteration in steady state: launch plan i, then finish plan i-1 (pipelined by one)."""
        launch()
        if len(in_flight) > n_streams:
            return retire()
        return None

    def barrier():
        if reducer is not None:
            reducer.drain()             # every plan's reduce has completed and rank 0 holds the sums
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    while in_flight:
        retire()
    if reducer is not None:
        reducer.drain()
    buckets = {}
    n_timed = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = step()
        if r is not None:
            for k, v in r[0].items():
                buckets[k] = buckets.get(k, 0.0) + float(v)
            n_timed += 1
    while in_flight:                    # the last plan is finished inside the timed region
        t, info = retire()
        for k, v in t.items():
            buckets[k] = buckets.get(k, 0.0) + float(v)
        n_timed += 1
    barrier()
    elapsed = time.perf_counter() - t0
    assert n_timed == args.steps
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = 1000.0 * elapsed / args.steps

    # self-check (untimed): a volume restored by the dirty-box clear of launch() must be bit-identical to the same field
    # computed into a fully zeroed volume (compared BEFORE this plan's exchange is submitted: under RCCL the destination's
    # adds are queued at submit time and may already have run)
    launch()
    f_chk, last, _ = in_flight.pop(0)
    _, chk_info0 = f_chk.finish()
    ref = torch.zeros_like(last)
    fld.compute(ref.data_ptr())
    fld.finish()
    clear_ok = torch.tensor([1 if torch.equal(last, ref) else 0], dtype=torch.int64, device=dev)
    del ref
    if world > 1:
        dist.all_reduce(clear_ok, op=dist.ReduceOp.MIN)
    clear_check = bool(int(clear_ok.item()))
    if reducer is not None:
        reducer.submit(last, chk_info0["dose_box_min"], chk_info0["dose_box_max"])
        reducer.drain()

    # N>1 self-check (untimed): the reduced volume on rank 0 must hold the sum of all ranks' fields
    reduce_check = None
    if world > 1:
        d = doses[0]
        reducer.release(d)
        d.zero_()
        fld.compute(d.data_ptr())
        _, chk_info = fld.finish()
        local_sum = d.sum(dtype=torch.float64).reshape(1)
        reducer.submit(d, chk_info["dose_box_min"], chk_info["dose_box_max"])
        reducer.drain()
        dist.all_reduce(local_sum, op=dist.ReduceOp.SUM)
        if rank == 0:
            total = float(d.sum(dtype=torch.float64).item())
            reduce_check = abs(total - float(local_sum.item())) / max(float(local_sum.item()), 1e-300)

    if rank == 0:
        info["_steps"] = beam.tracerSteps
        stage_ms = {k: buckets[k] / args.steps for k in buckets if k.endswith("_ms")}
        alg = algorithmic_bytes(info, scn.dims)
        ks_ms = stage_ms["superp_kernel_ms"]
        ks_gbs = alg["superposition"] / (ks_ms * 1e-3) / 1e9 if ks_ms > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get("k_superpose", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "Mvoxels/s dose deposited (%d^3 CT, 1 field per GPU)" % n,
            "value": round(world * n_vox / elapsed * args.steps / 1e6, 3),
            "unit": "Mvoxels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C3: %d^3 synthetic heterogeneous CT (HU->density/SP LUTs), %d field(s) one per GPU, "
                                   "10x10 spots x 20 layers = 2000 spots, 512 tracer steps, 1 mm rays" % (n, world),
                       "plans_in_flight_on_streams": n_streams, "ray_grid": info["ray_dims"], "live_steps": info["live_steps"], "max_radius": info["max_radius"],
                       "bbox_voxels": int(np.prod([info["bbox_max"][i] - info["bbox_min"][i] + 1 for i in range(3)])),
                       "reduce": ("none" if world == 1 else
                                  "point-to-point reduce-scatter + gather over xGMI (rccl send/recv): every rank sends the pieces of its dose box "
                                  "(rtd_field_info.dose_box) to the ranks that own those slabs, owners add, then send their complete slab "
                                  "to rank 0; overlapped with the next plan" if reduce_mode == "slab" else
                                  "each rank sends its packed dose box (rtd_field_info.dose_box) to rank 0 (rccl send/recv over its own xGMI "
                                  "link), rank 0 adds the N-1 boxes; overlapped with the next plan")},
            "ms_plan": round(ms_per_step, 4),
            "reduce_check_rel_err": reduce_check, "clear_check": clear_check,
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "algorithmic_bytes": alg,
            "path_gbs": round(alg["total"] / (stage_ms["total_ms"] * 1e-3) / 1e9, 2),
            "roofline": {"kernel": "rtd::k_superpose_mfma", "bound": "hbm", "achieved": round(ks_gbs, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ks_gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "avg_launch_ms": round(ks_ms, 4), "algorithmic_bytes_per_launch": alg["superposition"],
                         "note": "algorithmic bytes = SURVEY.md 8(d) superposition term 8*(R+P)*sum(A_l); the kernel is bound by vector + f32-matrix "
                                 "issue cycles (91 % of its SIMD cycles, DESIGN.md section 4), not by HBM; traffic = (2*FETCH_SIZE+WRITE_SIZE)*1024 from profiles/traffic.json"},
        }
        if not args.no_cpu and world == 1:
            from oracle import oracle
            ncpu = args.cpu_threads or min(16, os.cpu_count() or 1)
            oracle.set_threads(ncpu)
            cpu_dose = np.zeros_like(scn.ct)
            c0 = time.perf_counter()
            of = oracle.run_field(scn, beam, cpu_dose, keep_layers=False)
            cpu_first = time.perf_counter() - c0
            # bounded sample of ~10-30 s of CPU work: the N=1 field plus the three other C4 gantry angles
            extra = scenarios.hetero_ct(es, n=n, angles=[90.0, 180.0, 270.0], ct=ct_np)
            scratch = np.zeros_like(scn.ct)
            for b2 in extra.beams:
                oracle.run_field(extra, b2, scratch, keep_layers=False).close()
            cpu_s = time.perf_counter() - c0
            n_cpu_fields = 1 + len(extra.beams)
            del scratch
            host = dose.cpu().numpy()
            rate, n_eval, gmax = oracle.gamma_pass_rate(cpu_dose, host, scn.spacing)
            thr = cpu_dose > 0.1 * cpu_dose.max()
            max_rel = float((np.abs(host - cpu_dose)[thr] / cpu_dose[thr]).max())
            result["cpu_baseline"] = {"value": round(n_cpu_fields * n_vox / cpu_s / 1e6, 3), "unit": "Mvoxels/s", "cores": ncpu, "kind": "port",
                                      "sample": "%d fields of the same workload (the N=1 field + gantry 90/180/270), CPU oracle (oracle/rtd_oracle.c) "
                                                "with %d OpenMP threads: %.1f s wall = %.0f CPU-s; first field alone %.2f s"
                                                % (n_cpu_fields, ncpu, cpu_s, cpu_s * ncpu, cpu_first)}
            result["parity"] = {"gamma_1pct_1mm_pass": rate, "gamma_voxels": n_eval, "gamma_max": round(gmax, 4),
                                "max_rel_diff_above_10pct": max_rel}
            of.close()
        print(json.dumps(result))
    for f in flds:
        f.destroy()
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
