set -e
mkdir -p gpurun_out/r04
timeout -k 10 120 python tools/debug_u2.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r04/e_gputests.txt 2>&1 || { tail -30 gpurun_out/r04/e_gputests.txt; exit 1; }
tail -2 gpurun_out/r04/e_gputests.txt
C=skill-chaining-with-graphs_amd/csrc
python tools/ab_bench.py --rounds 2 $C/libscg_hip.so $C/libscg_hip_v_oldu2.so $C/libscg_hip_v_p0.so $C/libscg_hip_v_l0.so $C/libscg_hip_v_h1.so $C/libscg_hip_v_p0l0.so > gpurun_out/r04/e_ab.txt 2>&1 || true
grep median gpurun_out/r04/e_ab.txt
python tools/stamp_report.py > gpurun_out/r04/e_stamps.txt 2>&1 || true
head -36 gpurun_out/r04/e_stamps.txt
