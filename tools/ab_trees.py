"""A/B timing across source TREES and library builds on one box (round 5: the ABI moved, so a round-4 library no longer loads into
this tree's binding): python tools/ab_trees.py [--rounds R] [--steps K] name=tree_root[:lib.so] ...
Runs `python <tree_root>/bench.py` once per arm per round, interleaved (clock drift hits every arm alike), with SCG_LIB pointing at
the arm's library when one is given; prints M env-steps/s, us per step and the fused kernel's event time per run, then the medians."""
import argparse, json, os, statistics, subprocess, sys
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3); ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--extra", default="", help="extra bench.py flags for every arm")
ap.add_argument("arms", nargs="+")
a = ap.parse_args()
arms = []
for spec in a.arms:
    name, rest = spec.split("=", 1)
    root, _, lib = rest.partition(":")
    arms.append((name, os.path.abspath(root), os.path.abspath(lib) if lib else None))
res = {n: [] for n, _, _ in arms}
for r in range(a.rounds):
    for name, root, lib in arms:
        env = dict(os.environ)
        env.pop("SCG_LIB", None)
        if lib:
            env["SCG_LIB"] = lib
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", str(a.steps), "--warmup", "50", "--no-cpu-baseline", "--no-extras"]
                           + a.extra.split(), env=env, capture_output=True, text=True, cwd=root)
        try:
            d = json.loads(p.stdout.strip().splitlines()[-1])
        except Exception:
            print(f"{name}: bench failed: {p.stderr[-400:]}", flush=True)
            continue
        row = (d["value"] / 1e6, d["ms_per_step"] * 1e3, d["roofline"]["kernel_ms"] * 1e3)
        res[name].append(row)
        print(f"{name:24s} {row[0]:7.1f} M/s  {row[1]:7.2f} us/step  td {row[2]:6.2f} us", flush=True)
for name, _, _ in arms:
    if res[name]:
        print(f"median {name:24s} {statistics.median(x[0] for x in res[name]):7.1f} M/s  {statistics.median(x[1] for x in res[name]):7.2f} us/step  "
              f"td {statistics.median(x[2] for x in res[name]):6.2f} us")
