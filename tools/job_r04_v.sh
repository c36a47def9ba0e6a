set -e
mkdir -p gpurun_out/r04
C=skill-chaining-with-graphs_amd/csrc
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r04/v_gputests.txt 2>&1 || { tail -30 gpurun_out/r04/v_gputests.txt; exit 1; }
tail -1 gpurun_out/r04/v_gputests.txt
python tools/ab_bench.py --rounds 3 $C/libscg_hip.so $C/libscg_hip_v_u1s.so $C/libscg_hip_v_h8.so $C/libscg_hip_v_scheddef.so $C/libscg_hip_v_schedmem.so $C/libscg_hip_v_schedb50.so > gpurun_out/r04/v_ab.txt 2>&1 || true
grep median gpurun_out/r04/v_ab.txt
for r in 200 1000 4000; do python bench.py --steps 20 --warmup 5 --ramp $r --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ramp', $r, round(d['value']/1e6,1), 'M/s', round(d['ms_per_step']*1e3,2), 'us td', round(d['roofline']['kernel_ms']*1e3,2))"; done
python tools/stamp_report.py > gpurun_out/r04/v_stamps.txt 2>&1 || true
sed -n 2,12p gpurun_out/r04/v_stamps.txt; grep "E phase per wave\|helper wave" gpurun_out/r04/v_stamps.txt
