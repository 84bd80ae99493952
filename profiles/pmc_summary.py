#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch for each kernel."""
import csv, sys, collections, re
def main(paths, filt=None):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        for row in csv.DictReader(open(p)):
            k = re.sub(r"\(.*", "", row["Kernel_Name"])
            if filt and filt not in k: continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        print(k)
        for c, v in sorted(d.items()):
            print("   %-28s mean %.4g  (n=%d)" % (c, sum(v)/len(v), len(v)))
if __name__ == "__main__":
    args = sys.argv[1:]
    filt = None
    if args and not args[0].endswith(".csv"): filt, args = args[0], args[1:]
    main(args, filt)
