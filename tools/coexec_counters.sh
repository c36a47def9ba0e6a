#!/bin/bash
# does SQ_VALU_MFMA_COEXEC_CYCLES count f32 MFMAs on gfx950 at all? Counter pass over the probes whose
# co-execution is known from their own timing. Run on the GPU box from the repo root: bash tools/coexec_counters.sh <outdir>
set -e -o pipefail
root=$(pwd); out=$root/$1; mkdir -p "$out"
cd /tmp; export TMPDIR=/tmp
for p in mfma_valu_coexec build_beside_mfma coexec_probe; do
    rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d "$out/$p" -o run -- "$root/tools/$p" > "$out/$p.stdout" 2> "$out/$p.err"
    f=$(find "$out/$p" -name 'run_counter_collection.csv' | head -1)
    python3 "$root/tools/coexec_counters.py" "$f" > "$out/${p}_counters.txt"
done
