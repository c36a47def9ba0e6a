set -e
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r04/c_gputests.txt 2>&1 || { tail -60 gpurun_out/r04/c_gputests.txt; exit 1; }
tail -3 gpurun_out/r04/c_gputests.txt
python tools/ab_bench.py --rounds 2 skill-chaining-with-graphs_amd/csrc/libscg_hip.so > gpurun_out/r04/c_ab_new.txt 2>&1 || true
tail -4 gpurun_out/r04/c_ab_new.txt
SCG_LIB_ABI=1 python tools/ab_bench.py --rounds 2 skill-chaining-with-graphs_amd/csrc/libscg_hip_r03.so > gpurun_out/r04/c_ab_old.txt 2>&1 || true
tail -3 gpurun_out/r04/c_ab_old.txt
