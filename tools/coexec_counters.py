"""rocprofv3 counter_collection.csv -> one line per dispatch: the counters side by side (dispatch order = launch order)."""
import csv, sys, collections
rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    rows.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"][:40]})[r["Counter_Name"]] = float(r["Counter_Value"])
names = ["SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA"]
print("dispatch  " + "  ".join(f"{n[3:]:>24s}" for n in names) + "  coexec/mfma_busy")
for d, v in rows.items():
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    print(f"{d:8d}  " + "  ".join(f"{v.get(n, float('nan')):24.0f}" for n in names) + f"  {v.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0.0) / busy if busy else float('nan'):.3f}")
