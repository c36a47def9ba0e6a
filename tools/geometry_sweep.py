"""Is the automatic block geometry (auto_block_envs) the fastest of the three builds at every env count? One MI355X;
per env count: bench.py (root + 5 options, pinball_simple) with --block-envs 64 / 128 / 256, and which one the context picks by itself.
    python tools/geometry_sweep.py [--steps 300] [N ...]"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from skill_chaining_with_graphs_amd import auto_block_envs
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300); ap.add_argument("--options", type=int, default=5)
ap.add_argument("envs", type=int, nargs="*", default=[2048, 4096, 8192, 16384, 20480, 24576, 32768, 40960, 49152, 65536, 98304, 131072])
a = ap.parse_args()
print(f"# python tools/geometry_sweep.py: M env-steps/s (us per step-batch), root + {a.options} options, pinball_simple, {a.steps} steps; * = what the context picks", flush=True)
worst = 1.0
for n in a.envs:
    row, best = {}, None
    for b in (64, 128, 256):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--envs-per-gpu", str(n), "--options", str(a.options), "--block-envs", str(b),
                              "--steps", str(a.steps), "--warmup", "50", "--ramp", "100", "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            row[b] = (d["value"] / 1e6, d["ms_per_step"] * 1e3)
        except Exception:
            row[b] = (0.0, 0.0)
    pick = auto_block_envs(n)
    best = max(row, key=lambda b: row[b][0])
    ratio = row[pick][0] / max(row[best][0], 1e-9)
    worst = min(worst, ratio)
    print(f"envs {n:7d}: " + "  ".join(f"B={b:3d}{'*' if b == pick else ' '} {row[b][0]:7.1f} ({row[b][1]:6.1f})" for b in (64, 128, 256))
          + f"   pick / best = {ratio:.3f}", flush=True)
print(f"# worst pick / best over the sweep: {worst:.3f}")
