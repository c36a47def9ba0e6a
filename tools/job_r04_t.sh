set -e
mkdir -p gpurun_out/r04
C=skill-chaining-with-graphs_amd/csrc
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r04/t_gputests.txt 2>&1 || { tail -30 gpurun_out/r04/t_gputests.txt; exit 1; }
tail -1 gpurun_out/r04/t_gputests.txt
for v in u2db edyn edyn_u2db; do SCG_LIB=$PWD/$C/libscg_hip_v_$v.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_scale.py tests/test_gpu_stress.py -x -q -m gpu 2>&1 | tail -1; done
python tools/ab_bench.py --rounds 3 $C/libscg_hip.so $C/libscg_hip_v_u2db.so $C/libscg_hip_v_edyn.so $C/libscg_hip_v_edyn_u2db.so > gpurun_out/r04/t_ab.txt 2>&1 || true
grep median gpurun_out/r04/t_ab.txt
