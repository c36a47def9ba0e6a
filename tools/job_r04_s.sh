set -e
bash tools/profile_round.sh gpurun_out/r04/prof_a > gpurun_out/r04_prof_a.log 2>&1 || { tail -20 gpurun_out/r04_prof_a.log; exit 1; }
cat gpurun_out/r04/prof_a/pmc.csv | grep td_kernel
head -4 gpurun_out/r04/prof_a/kernel_stats.csv | cut -c1-150
python -c "
import json
for f in ('bench_default','bench_20_steps'):
    d=json.load(open('gpurun_out/r04/prof_a/'+f+'.json')); print(f, round(d['value']/1e6,1), round(d['ms_per_step']*1e3,2), round(d['roofline']['kernel_ms']*1e3,2), d.get('extras'))
"
