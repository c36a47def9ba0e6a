set -e -o pipefail
root=$(pwd); out=$root/gpurun_out/r05/icache; mkdir -p "$out"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d "$out/a" -o run -- python3 $root/bench.py --steps 10 --warmup 5 --no-extras --no-cpu-baseline --ramp 20 > "$out/a.out" 2> "$out/a.err"
rocprofv3 --pmc SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d "$out/b" -o run -- python3 $root/bench.py --steps 10 --warmup 5 --no-extras --no-cpu-baseline --ramp 20 > "$out/b.out" 2> "$out/b.err"
cd $root
python3 - <<'PY'
import csv, glob, collections
for d in "ab":
    f = glob.glob(f"gpurun_out/r05/icache/{d}/**/run_counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for r in csv.DictReader(open(f)):
        pass
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)): disp[r["Kernel_Name"][:40]].add(r["Dispatch_Id"])
    for k, v in acc.items():
        print(k, len(disp[k]), {c: round(x / len(disp[k])) for c, x in v.items()})
PY
