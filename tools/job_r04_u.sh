set -e
mkdir -p gpurun_out/r04
C=skill-chaining-with-graphs_amd/csrc
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r04/u_gputests.txt 2>&1 || { tail -30 gpurun_out/r04/u_gputests.txt; exit 1; }
tail -1 gpurun_out/r04/u_gputests.txt
python tools/ab_bench.py --rounds 2 $C/libscg_hip.so $C/libscg_hip_v_ey1.so $C/libscg_hip_v_ey2.so $C/libscg_hip_v_h8.so $C/libscg_hip_v_h6.so $C/libscg_hip_v_etg6.so $C/libscg_hip_v_etg3.so > gpurun_out/r04/u_ab.txt 2>&1 || true
grep median gpurun_out/r04/u_ab.txt
python tools/stamp_report.py > gpurun_out/r04/u_stamps.txt 2>&1 || true
head -40 gpurun_out/r04/u_stamps.txt
