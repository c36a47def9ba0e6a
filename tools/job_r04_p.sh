set -e
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r04/p_gputests.txt 2>&1 || { tail -30 gpurun_out/r04/p_gputests.txt; exit 1; }
tail -2 gpurun_out/r04/p_gputests.txt
C=skill-chaining-with-graphs_amd/csrc
python tools/ab_bench.py --rounds 2 $C/libscg_hip.so $C/libscg_hip_v_h8.so $C/libscg_hip_v_etg6.so $C/libscg_hip_v_l0.so > gpurun_out/r04/p_ab.txt 2>&1 || true
grep median gpurun_out/r04/p_ab.txt
SCG_LIB_ABI=1 python tools/ab_bench.py --rounds 1 $C/libscg_hip_r03.so 2>&1 | grep median
python tools/stamp_report.py > gpurun_out/r04/p_stamps.txt 2>&1 || true
head -38 gpurun_out/r04/p_stamps.txt
