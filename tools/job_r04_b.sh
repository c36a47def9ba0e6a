set -e
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04/b_parity.txt 2>&1 || { tail -40 gpurun_out/r04/b_parity.txt; exit 1; }
tail -3 gpurun_out/r04/b_parity.txt
export SCG_LIB=$PWD/skill-chaining-with-graphs_amd/csrc/libscg_hip_r03.so SCG_LIB_ABI=1
for rs in 1000 10000; do
python tools/chain_evidence.py --seeds 1 2 --r-succ $rs > gpurun_out/r04/b_evidence_rsucc$rs.txt 2>&1
grep -E "verdict|after" gpurun_out/r04/b_evidence_rsucc$rs.txt
done
