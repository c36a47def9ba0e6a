#!/bin/bash
# rocprofv3 kernel stats of the reduce launch for variant libraries (timing experiments): bash tools/reduce_variants.sh out lib1.so lib2.so ...
set -e -o pipefail
root=$(pwd); out=$root/$1; shift; mkdir -p "$out"
cd /tmp; export TMPDIR=/tmp
for l in "$@"; do
    n=$(basename $l .so)
    export SCG_LIB=$root/$l
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$n" -o run -- python3 $root/bench.py --steps 300 --warmup 50 --ramp 100 --no-extras --no-cpu-baseline > "$out/$n.out" 2> "$out/$n.err"
    f=$(find "$out/$n" -name 'run_kernel_stats.csv' | head -1)
    echo "== $n: $(python3 -c "import json,sys; d=json.loads(open('$out/$n.out').read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), 'M/s')")"
    grep "td_kernel<0>\|reduce_kernel" "$f" | cut -d, -f1-5
done
