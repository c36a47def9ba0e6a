"""Summarises rocprofv3 --pmc counter_collection CSVs (one directory per pass) into mean-per-dispatch rows per kernel.
Usage: python tools/pmc_summary.py <dir> [<dir> ...] > profiles/rNN_pmc.csv"""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                key = (row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = row["Kernel_Name"]
        for (disp, ctr), v in per_dispatch.items():
            a = acc[(names[disp], ctr)]
            a[0] += v; a[1] += 1
print("kernel,counter,mean_per_dispatch,dispatches")
for (k, c), (s, n) in sorted(acc.items()):
    if "td_kernel" in k or "reduce_kernel" in k or "fit_kernel" in k:
        print(f'"{k}",{c},{s / n:.1f},{n}')
