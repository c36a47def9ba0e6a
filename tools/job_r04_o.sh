set -e
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r04/o_gputests.txt 2>&1 || { tail -30 gpurun_out/r04/o_gputests.txt; exit 1; }
tail -2 gpurun_out/r04/o_gputests.txt
C=skill-chaining-with-graphs_amd/csrc
for v in h6 h7; do SCG_LIB=$PWD/$C/libscg_hip_v_$v.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -1; done
python tools/ab_bench.py --rounds 2 $C/libscg_hip.so $C/libscg_hip_v_h6.so $C/libscg_hip_v_h7.so > gpurun_out/r04/o_ab.txt 2>&1 || true
grep median gpurun_out/r04/o_ab.txt
python tools/stamp_report.py > gpurun_out/r04/o_stamps.txt 2>&1 || true
head -38 gpurun_out/r04/o_stamps.txt
