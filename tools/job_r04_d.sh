set -e
mkdir -p gpurun_out/r04
python tools/stamp_report.py > gpurun_out/r04/d_stamps.txt 2>&1 || true
cat gpurun_out/r04/d_stamps.txt | head -40
python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-learn > gpurun_out/r04/d_nolearn.json 2>/dev/null || true
python bench.py --steps 400 --warmup 50 --no-cpu-baseline --diag-no-td > gpurun_out/r04/d_notd.json 2>/dev/null || true
python - <<'PY'
import json
for f in ("d_nolearn","d_notd"):
    try:
        d=json.load(open(f"gpurun_out/r04/{f}.json")); print(f, round(d["value"]/1e6,1), "M/s", round(d["ms_per_step"]*1e3,2), "us/step td", round(d["roofline"]["kernel_ms"]*1e3,2))
    except Exception as e: print(f, "failed", e)
PY
