"""A/B timing of library builds on one box: python tools/ab_bench.py [--rounds R] [--steps K] lib1.so lib2.so ...
Runs bench.py once per library per round (interleaved, so clock drift hits every arm alike) with SCG_LIB pointing at
the build, and prints M env-steps/s, us per step and the td_kernel's event time per run, then the per-arm medians."""
import argparse, json, os, statistics, subprocess, sys
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {l: [] for l in a.libs}
for r in range(a.rounds):
    for l in a.libs:
        env = dict(os.environ, SCG_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", str(a.steps), "--warmup", "50",
                              "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        row = (d["value"] / 1e6, d["ms_per_step"] * 1e3, d["roofline"]["kernel_ms"] * 1e3)
        res[l].append(row)
        print(f"{os.path.basename(l):28s} {row[0]:7.1f} M/s  {row[1]:7.2f} us/step  td {row[2]:6.2f} us", flush=True)
for l in a.libs:
    print(f"median {os.path.basename(l):28s} {statistics.median(x[0] for x in res[l]):7.1f} M/s  "
          f"td {statistics.median(x[2] for x in res[l]):6.2f} us")
