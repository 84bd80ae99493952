#!/usr/bin/env python3
"""One-GPU emulation of what ONE rank of `bench.py --gpus N` does per plan step, for every rank in turn: its own field up to the
BEV dose, pack, the all-gather replaced by device copies of the other ranks' (pre-exported) messages into the receive buffer, then
every field transferred into the rank's slab, pipelined by one plan like bench.py. Everything but the xGMI traffic and RCCL's own
kernels. Predicted N-GPU value = N * voxels / max over ranks of the step time. Usage: multi_emulation.py N [steps] [balanced 0|1] [fused transfer 0|1] [CT edge 512|768]
-> one JSON line (profiles/r02_multi_emulation_*.json)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from raytracedicom_amd import abi, engine, luts, plan, scenarios


class _Work:
    def wait(self):
        return True


class _FakeDist:
    """torch.distributed's surface as BevExchange uses it, for one emulated rank: the other ranks' rows / messages are precomputed."""

    def __init__(self, rank, world, rows, messages):
        self.rank, self.world, self.rows, self.messages = rank, world, rows, messages

    def all_gather(self, out, mine):
        for r in range(self.world):
            out[r].copy_(mine if r == self.rank else torch.tensor(self.rows[r], dtype=mine.dtype, device=mine.device))

    def all_gather_into_tensor(self, recv, send, async_op=True):
        cap = send.numel()
        for r in range(self.world):
            src = send if r == self.rank else self.messages[r]
            recv[r * cap:r * cap + src.numel()].copy_(src, non_blocking=True)
        return _Work()


def main():
    world = int(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    balanced = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0
    fused = (int(sys.argv[4]) if len(sys.argv) > 4 else 1) != 0
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 512
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    es = luts.synth_luts()
    ct_np, _ = scenarios.hetero_phantom(n)
    angles = [r * 360.0 / world for r in range(world)]
    scn = scenarios.hetero_ct(es, n=n, angles=angles, ct=ct_np)
    eng = engine.Engine(0)
    opt = abi.default_options()
    opt.fine_grained_timing = 1
    eng.set_options(opt)
    eng.set_stream(stream.cuda_stream)
    eng.set_luts(es)
    ct_dev = torch.from_numpy(ct_np).to(dev)
    eng.set_ct_device(ct_dev.data_ptr(), scn.dims)
    doses = [torch.zeros((n, n, n), dtype=torch.float32, device=dev) for _ in range(2)]
    # ---- every rank's setup row and exported message, computed once ----
    rows, messages = [], []
    for r in range(world):
        f = eng.create_field(scn.beams[r], scn.dims)
        head, rate = [], []
        for j in range(11):
            f.compute_bev(); f.transfer(doses[0].data_ptr()); t, _ = f.finish()
            if j >= 3:
                head.append(1000.0 * (t["total_ms"] - t["transforming_ms"]))
                rate.append(1.25 * t["transforming_ms"] * 1e9 / t["transfer_voxels"] * 1000.0)
        f.compute_bev()
        info, nbytes = f.wait_plan()
        msg = torch.empty((nbytes + 255) // 256 * 256, dtype=torch.uint8, device=dev)
        f.export_bev(msg.data_ptr(), msg.numel())
        f.finish()
        rows.append([int(nbytes)] + [int(v) for v in info["dose_box_min"]] + [int(v) for v in info["dose_box_max"]]
                    + ([int(round(min(head))), int(round(min(rate)))] if balanced else [-1, -1]))
        messages.append(msg)
        f.destroy()
    doses[0].zero_()
    out = {"n_gpus_emulated": world, "ct_edge": n, "balanced": balanced, "fused_transfer": fused, "steps": steps, "ranks": []}
    for rank in range(world):
        flds = [eng.create_field(scn.beams[rank], scn.dims) for _ in range(3)]
        remote = {r: eng.create_field(scn.beams[r], scn.dims, remote=True) for r in range(world) if r != rank}
        ex = plan.BevExchange(_FakeDist(rank, world, rows, messages), rank, world, remote, scn.dims,
                              new_bytes=lambda k: torch.empty(int(k), dtype=torch.uint8, device=dev),
                              zero_box=lambda b, lo, hi: doses[b][lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1].zero_(),
                              transfer_all=(lambda fs, d, lo, hi: eng.transfer_fields_init(fs, d, lo, hi)) if fused else None)
        flds[0].compute_bev()
        ex.setup(flds[0], head_us=rows[rank][7] if balanced else None, transfer_ps_per_kvoxel=rows[rank][8] if balanced else None)
        flds[0].finish()
        for d in doses:
            d.zero_()
        in_flight, step_no, completed = [], [0], set()

        def launch():
            i = step_no[0]; step_no[0] += 1
            f = flds[i % 3]; b = i % 2
            if i >= 2:
                ex.clear(f, b, doses[b].data_ptr())
            f.compute_bev()
            ex.post(f, b)
            in_flight.append((i, f))

        def retire():
            i, f = in_flight.pop(0)
            if i not in completed:
                ex.complete(f, i % 2, doses[i % 2].data_ptr())
            completed.discard(i)
            return f.finish()

        def step():                                                   # bench.py's N>1 step
            launch()
            for i, f in in_flight:
                if i not in completed:
                    ex.complete(f, i % 2, doses[i % 2].data_ptr())
                    completed.add(i)
            if len(in_flight) > 2:
                retire()

        for _ in range(3):
            step()
        while in_flight:
            retire()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        while in_flight:
            retire()
        torch.cuda.synchronize()
        ms = 1000.0 * (time.perf_counter() - t0) / steps
        lo, hi = ex.clip()
        out["ranks"].append({"rank": rank, "angle": angles[rank], "ms_per_step": round(ms, 4), "slab": [lo[ex.axis], hi[ex.axis]], "axis": ex.axis,
                             "head_us": rows[rank][7], "ps_per_kvoxel": rows[rank][8]})
        print(out["ranks"][-1], flush=True)
        for f in flds + list(remote.values()):
            f.destroy()
    worst = max(r["ms_per_step"] for r in out["ranks"])
    out["ms_per_step_max_over_ranks"] = worst
    out["predicted_mvoxels_s"] = round(world * n ** 3 / (worst * 1e-3) / 1e6, 1)
    print(json.dumps(out))


main()
