#!/bin/bash
# Round-end evidence job for one gpurun call (run from the repo root on the GPU box):
#   gpurun --timeout 1200 -- bash tools/job_round_end.sh r04
# GPU tests, the randomised parity sweep, the profile set of tools/profile_round.sh, the batch sweep and the behavioural evidence.
set -e -o pipefail
tag=${1:-rXX}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $out/gputests.txt 2>&1 || { tail -30 $out/gputests.txt; exit 1; }
tail -1 $out/gputests.txt
timeout -k 10 300 python tests/fuzz_parity.py 200 20000 > $out/fuzz.txt 2>&1 || { tail -20 $out/fuzz.txt; exit 1; }
tail -1 $out/fuzz.txt
timeout -k 10 500 bash tools/profile_round.sh $out/prof
timeout -k 10 200 bash tools/batch_sweep.sh > $out/batch_sweep.txt 2>&1 || true
cat $out/batch_sweep.txt
timeout -k 10 400 python tools/chain_evidence.py --seeds 1 2 3 --json $out/chain_evidence.jsonl > $out/chain_evidence.txt 2>&1 || true
tail -8 $out/chain_evidence.txt
