#!/bin/bash
# VERDICT r3 item 5 (weak item 8): the step kernel built for 64- and 128-env blocks against the shipped 256 at small env counts.
# The block size is SPEC §5's B (order of the partial sums), so each build is checked against an oracle built with the same B.
#   (cd skill-chaining-with-graphs_amd/csrc && for b in 64 128; do hipcc <Makefile FLAGS> -DSCG_BLOCK_ENVS=$b -o libscg_hip_v_b$b.so scg_kernels.hip; done)
#   (cd oracle && for b in 64 128; do gcc <Makefile CFLAGS> -shared -DSCO_BLOCK_ENVS=$b -o libsc_oracle_b$b.so sc_oracle.c -lm; done)
C=skill-chaining-with-graphs_amd/csrc
for b in 64 128 256; do
  if [ $b = 256 ]; then lib=$PWD/$C/libscg_hip.so; orc=$PWD/oracle/libsc_oracle.so; else lib=$PWD/$C/libscg_hip_v_b$b.so; orc=$PWD/oracle/libsc_oracle_b$b.so; fi
  echo "== B = $b"
  SCG_LIB=$lib SCO_LIB=$orc timeout -k 10 200 python tests/fuzz_parity.py 40 31000 2>&1 | tail -1
  for cfg in "4096 1" "4096 5" "16384 5" "32768 5" "65536 5"; do
    set -- $cfg
    SCG_LIB=$lib timeout -k 10 100 python bench.py --envs-per-gpu $1 --options $2 --steps 400 --warmup 50 --no-cpu-baseline --no-extras | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(f'  envs $1 options $2: {d[\"value\"]/1e6:7.1f} M env-steps/s  step {d[\"ms_per_step\"]*1e3:7.2f} us  td {d[\"roofline\"][\"kernel_ms\"]*1e3:7.2f} us')"
  done
done
