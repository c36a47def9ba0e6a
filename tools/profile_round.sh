#!/bin/bash
# The per-round evidence set behind DESIGN §5 / profiles/rNN_*: run on the GPU box from the repo root,
#   bash tools/profile_round.sh gpurun_out/<dir>
# then copy <dir>/{bench_default.json,bench_20_steps.json,kernel_stats.csv,pmc.csv,stamps.txt,traffic.json} into profiles/.
# Counter passes are separate rocprofv3 runs with --kernel-trace only (never combined with other trace domains).
set -e -o pipefail
root=$(pwd); out=$root/$1; mkdir -p "$out"
python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
python bench.py --steps 20 --warmup 5 > "$out/bench_20_steps.json" 2> "$out/bench_20_steps.err"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- python3 "$root/bench.py" --no-cpu-baseline --no-extras > "$out/stats.json" 2> "$out/stats.err"
cp "$out"/stats/*/run_kernel_stats.csv "$out/kernel_stats.csv" 2>/dev/null || cp "$out"/stats/run_kernel_stats.csv "$out/kernel_stats.csv"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pmc$i" -o run -- python3 "$root/bench.py" --steps 30 --warmup 5 --ramp 20 --no-cpu-baseline --no-extras > "$out/pmc$i.json" 2> "$out/pmc$i.err" || echo "counter set $i failed: $set" >> "$out/pmc_failures.txt"
    i=$((i + 1))
done
cd "$root"
python tools/pmc_summary.py "$out"/pmc[0-9] > "$out/pmc.csv"
python tools/pmc_summary.py --traffic "$out/pmc.csv" > "$out/traffic.json"
python tools/stamp_report.py > "$out/stamps.txt" 2>&1
rm -rf "$out"/pmc[0-9] "$out/stats"
echo "profile set written to $1"
