"""Summarises rocprofv3 --pmc counter_collection CSVs (one directory per pass) into mean-per-dispatch rows per kernel.
Usage: python tools/pmc_summary.py <dir> [<dir> ...] > profiles/rNN_pmc.csv
       python tools/pmc_summary.py --traffic profiles/rNN_pmc.csv > profiles/rNN_traffic.json   (HBM bytes per launch of the
       step kernel and of the reduce launch, corrected as MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE doubled on
       gfx950, WRITE_SIZE exact; the workload is the default bench)"""
import csv, glob, json, os, sys
from collections import defaultdict
if len(sys.argv) > 2 and sys.argv[1] == "--traffic":
    rows = {(r["kernel"], r["counter"]): float(r["mean_per_dispatch"]) for r in csv.DictReader(l for l in open(sys.argv[2]) if not l.startswith("#"))}
    def kb(kern, ctr):
        return next(v for (k, c), v in rows.items() if kern in k and c == ctr)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out = {"_comment": "HBM traffic per launch, default bench workload. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes "
                       "(the pmc.csv next to this file); counter units are KB. Correction per MI355X_MICROARCH.md (HBM / rocprofv3 section): on gfx950 "
                       "FETCH_SIZE tallies 128-B read requests at 64 B, so it is DOUBLED; WRITE_SIZE is exact. traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024. "
                       "The doubling is calibrated for wide coalesced reads; part of the step kernel's reads are 4-byte gathers through the env order, so its "
                       "corrected figure is an upper bound (raw values kept).",
           "workload": {"envs_per_gpu": bench.ENVS_PER_GPU, "n_options": bench.N_OPTIONS, "map": bench.MAP},
           "kernel": "td_kernel<MODE_FUSED>", "fetch_correction": 2.0}
    f, w = kb("td_kernel<0>", "FETCH_SIZE"), kb("td_kernel<0>", "WRITE_SIZE")
    out.update(fetch_size_kb=f, write_size_kb=w, traffic_bytes_per_launch=int((2 * f + w) * 1024))
    f2, w2 = kb("reduce_kernel", "FETCH_SIZE"), kb("reduce_kernel", "WRITE_SIZE")
    out["reduce_kernel"] = {"fetch_size_kb": f2, "write_size_kb": w2, "traffic_bytes_per_launch": int((2 * f2 + w2) * 1024)}
    alg = bench.ENVS_PER_GPU * bench.BYTES_PER_ENV_STEP
    out["algorithmic_bytes_per_launch"] = alg
    out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / alg                      # the step kernel alone (roofline.traffic)
    out["step_traffic_over_algorithmic"] = (out["traffic_bytes_per_launch"] + out["reduce_kernel"]["traffic_bytes_per_launch"]) / alg   # both launches of a step-batch
    print(json.dumps(out, indent=1))
    sys.exit(0)
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
