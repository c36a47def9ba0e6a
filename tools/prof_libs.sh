# usage (on the GPU box, from the repo root): bash tools/prof_libs.sh NAME1 NAME2 ... — rocprofv3 kernel stats of the bench for csrc/libscg_hip_NAME.so builds (A/B of kernel times)
cd /tmp; export TMPDIR=/tmp
for l in "$@"; do
  rm -rf /tmp/st_$l
  SCG_LIB=$GRAFT_REPO_ROOT/skill-chaining-with-graphs_amd/csrc/libscg_hip_$l.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$l -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 300 > /tmp/st_$l.json 2>/dev/null
  echo "== $l  $(python3 -c "import json;d=json.load(open('/tmp/st_$l.json'));print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step']*1e3,2),'us/step')")"
  find /tmp/st_$l -name "*kernel_stats.csv" -exec head -4 {} \; | cut -d, -f1-4
done
