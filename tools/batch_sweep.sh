#!/bin/bash
# env-steps/s of the shipped build against the number of envs on ONE MI355X (5 options, pinball_simple): python bench.py --envs-per-gpu N
echo "# python bench.py --envs-per-gpu N --steps 400 --warmup 50 --no-cpu-baseline --no-extras, one MI355X, shipped build (5 options, pinball_simple)"
for n in 4096 16384 32768 49152 65536 98304 131072 262144; do
  python bench.py --envs-per-gpu $n --steps 400 --warmup 50 --no-cpu-baseline --no-extras | python -c "
import json,sys; d=json.loads(sys.stdin.read()); n=$n
print(f'envs {n:7d}  blocks {n//256:5d}  {d[\"value\"]/1e6:7.1f} M env-steps/s  step {d[\"ms_per_step\"]*1e3:7.2f} us  td {d[\"roofline\"][\"kernel_ms\"]*1e3:7.2f} us')"
done
python bench.py --no-learn --steps 400 --warmup 50 --no-cpu-baseline --no-extras | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(f'# acting-only (--no-learn, 65536 envs): {d[\"value\"]/1e6:.1f} M env-steps/s, {d[\"ms_per_step\"]*1e3:.1f} us/step')"
python bench.py --options 1 --envs-per-gpu 4096 --steps 400 --warmup 50 --no-cpu-baseline --no-extras | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(f'# BASELINE configs[1] (4096 envs, root + 1 option): {d[\"value\"]/1e6:.1f} M env-steps/s, {d[\"ms_per_step\"]*1e3:.2f} us/step (td {d[\"roofline\"][\"kernel_ms\"]*1e3:.2f})')"
