set -e -o pipefail
mkdir -p gpurun_out/g1
python -m pytest tests -m gpu -x -q > gpurun_out/g1/pytest.log 2>&1 || { tail -30 gpurun_out/g1/pytest.log; exit 1; }
tail -3 gpurun_out/g1/pytest.log
python bench.py --no-cpu-baseline > gpurun_out/g1/bench_default.json 2> gpurun_out/g1/bench_default.err
cat gpurun_out/g1/bench_default.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3)"
python tools/time_discovery.py > gpurun_out/g1/discovery.json 2> gpurun_out/g1/discovery.err
cat gpurun_out/g1/discovery.json
python tools/stamp_report.py > gpurun_out/g1/stamps.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/g1/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 300 > $GRAFT_REPO_ROOT/gpurun_out/g1/stats.json 2> $GRAFT_REPO_ROOT/gpurun_out/g1/stats.err
cd $GRAFT_REPO_ROOT
find gpurun_out/g1/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/g1/kernel_stats.csv \;
rm -rf gpurun_out/g1/stats
head -8 gpurun_out/g1/kernel_stats.csv
