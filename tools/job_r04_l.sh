set -e
mkdir -p gpurun_out/r04
C=skill-chaining-with-graphs_amd/csrc
python tools/ab_bench.py --rounds 2 $C/libscg_hip.so $C/libscg_hip_v_eomfma.so > gpurun_out/r04/l_ab.txt 2>&1 || true
grep median gpurun_out/r04/l_ab.txt
python tools/stamp_report.py > gpurun_out/r04/l_stamps_eomfma.txt 2>&1 || true
head -36 gpurun_out/r04/l_stamps_eomfma.txt
