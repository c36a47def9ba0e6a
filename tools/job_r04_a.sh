set -e
mkdir -p gpurun_out/r04
python bench.py --steps 20 --warmup 5 > gpurun_out/r04/base_bench_20.json 2> gpurun_out/r04/base_bench_20.err
python bench.py --no-cpu-baseline > gpurun_out/r04/base_bench_default.json 2>> gpurun_out/r04/base_bench_20.err
for s in 1 2 3; do python tools/learning_curve.py --seed $s --iters 16; done > gpurun_out/r04/learning_curve.txt 2>&1
python tools/chain_evidence.py --json gpurun_out/r04/chain_evidence.jsonl > gpurun_out/r04/chain_evidence.txt 2>&1
tail -5 gpurun_out/r04/chain_evidence.txt
