set -e
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r04/n_gputests.txt 2>&1 || { tail -30 gpurun_out/r04/n_gputests.txt; exit 1; }
tail -2 gpurun_out/r04/n_gputests.txt
C=skill-chaining-with-graphs_amd/csrc
python tools/ab_bench.py --rounds 2 $C/libscg_hip.so $C/libscg_hip_v_tg6.so $C/libscg_hip_v_tg4.so $C/libscg_hip_v_tg2.so $C/libscg_hip_v_l0.so > gpurun_out/r04/n_ab.txt 2>&1 || true
grep median gpurun_out/r04/n_ab.txt
python tools/stamp_report.py > gpurun_out/r04/n_stamps.txt 2>&1 || true
head -36 gpurun_out/r04/n_stamps.txt
