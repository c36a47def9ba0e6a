set -e
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r04/k_gputests.txt 2>&1 || { tail -30 gpurun_out/r04/k_gputests.txt; exit 1; }
tail -2 gpurun_out/r04/k_gputests.txt
timeout -k 10 600 python tests/fuzz_parity.py 60 6000 > gpurun_out/r04/k_fuzz.txt 2>&1 || { tail -20 gpurun_out/r04/k_fuzz.txt; exit 1; }
tail -1 gpurun_out/r04/k_fuzz.txt
C=skill-chaining-with-graphs_amd/csrc
python tools/ab_bench.py --rounds 2 $C/libscg_hip.so $C/libscg_hip_v_eomfma.so > gpurun_out/r04/k_ab.txt 2>&1 || true
grep median gpurun_out/r04/k_ab.txt
python tools/stamp_report.py > gpurun_out/r04/k_stamps.txt 2>&1 || true
head -36 gpurun_out/r04/k_stamps.txt
