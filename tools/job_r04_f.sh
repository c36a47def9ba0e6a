mkdir -p gpurun_out/r04
C=$PWD/skill-chaining-with-graphs_amd/csrc
for v in libscg_hip.so libscg_hip_v_oldu2.so libscg_hip_v_p0bar.so; do
  echo "== $v"
  SCG_LIB=$C/$v timeout -k 10 300 python -m pytest tests/test_golden.py tests/test_gpu_parity.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED" | head -20
done
