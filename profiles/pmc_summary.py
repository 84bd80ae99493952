#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch for each kernel.
  pmc_summary.py [kernel-substring] a.csv b.csv ...      text table
  pmc_summary.py --traffic-json fetch.csv write.csv      profiles/traffic.json (HBM bytes per launch, gfx950-corrected)"""
import csv, sys, collections, re, json

NOTE = ("HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, profiles/collect.sh: bench.py "
        "--steps 20 --warmup 3, C3 workload). Units KiB. gfx950 correction: FETCH_SIZE reports half of coalesced streamed "
        "reads (MI355X_MICROARCH.md, HBM); calibrated on this code's own kernels: k_slice_min (known 17.30 MB dword-per-lane "
        "read) and k_superpose_reduce (known float4 reads; WRITE_SIZE matches the bytes written). "
        "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.")

def collect(paths, filt=None):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        for row in csv.DictReader(open(p)):
            k = re.sub(r"\(.*", "", row["Kernel_Name"])
            if filt and filt not in k: continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc

def main(paths, filt=None):
    for k, d in collect(paths, filt).items():
        print(k)
        for c, v in sorted(d.items()):
            print("   %-28s mean %.4g  (n=%d)" % (c, sum(v)/len(v), len(v)))

def traffic(paths):
    out = {"_note": NOTE}
    for k, d in collect(paths).items():
        m = re.search(r"rtd::(k_\w+)", k)
        if not m or "FETCH_SIZE" not in d or "WRITE_SIZE" not in d: continue
        f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        out[m.group(1)] = {"fetch_size_kib": round(f, 3), "write_size_kib": round(w, 3), "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    # the dominant kernel of the bench field: k_superpose_sweep when it took the field (every batch radius <= 16), else k_superpose_mfma
    # (the other of the two returns at once or is not launched: its traffic is a few KiB)
    cand = [k for k in ("k_superpose_sweep", "k_superpose_mfma") if k in out]
    if cand:
        dom = max(cand, key=lambda k: out[k]["hbm_bytes_per_launch"])
        out["k_superpose"] = dict(out[dom], kernel="rtd::" + dom)
        # issue counters of the dominant kernel, when the SQ passes are given too (bench.py: roofline.issue_cycle_frac, mfma_tflops)
        for k, d in collect(paths).items():
            m = re.search(r"rtd::(k_\w+)", k)
            if not m or m.group(1) != dom: continue     # (exactly: k_superpose_sweep_big is another kernel)
            mean = lambda c: sum(d[c]) / len(d[c]) if c in d else None
            if mean("SQ_INSTS_MFMA"): out["k_superpose"]["mfma_insts_per_launch"] = int(mean("SQ_INSTS_MFMA"))
            if mean("SQ_INSTS_VALU"): out["k_superpose"]["valu_insts_per_launch"] = int(mean("SQ_INSTS_VALU"))
            if mean("SQ_INSTS_SALU"): out["k_superpose"]["salu_insts_per_launch"] = int(mean("SQ_INSTS_SALU"))
            if mean("SQ_INSTS_LDS"): out["k_superpose"]["lds_insts_per_launch"] = int(mean("SQ_INSTS_LDS"))
            # GRBM_GUI_ACTIVE sums the 8 XCDs: cycles of the launch = /8; SIMD-cycles = x 256 CUs x 4 SIMDs
            if mean("GRBM_GUI_ACTIVE"): out["k_superpose"]["simd_cycles_per_launch"] = int(mean("GRBM_GUI_ACTIVE") / 8 * 1024)
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--traffic-json":
        traffic(args[1:])
    else:
        filt = None
        if args and not args[0].endswith(".csv"): filt, args = args[0], args[1:]
        main(args, filt)
