"""Does the cached baseline of SPEC §5.4 (Q(s, a) from the previous step's evaluation: one update stale) learn like the exact form?
Runs the CPU ORACLE (the checker — hence under tests/) on Pinball in both forms from the same seeds and prints goal arrivals per
1000 env-steps per chunk.   python tests/oracle_learning_curve.py [--options K] [--seeds 1 2 3] [--envs N] [--iters I] [--chunk C]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [p for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")) if p not in sys.path]
import numpy as np
import sc_oracle
from util import SCALE, chain_classifiers
import skill_chaining_with_graphs_amd as scg

ap = argparse.ArgumentParser()
ap.add_argument("--map", default="pinball_simple"); ap.add_argument("--envs", type=int, default=2048)
ap.add_argument("--options", type=int, default=0); ap.add_argument("--alpha", type=float, default=0.02)
ap.add_argument("--iters", type=int, default=24); ap.add_argument("--chunk", type=int, default=500)
ap.add_argument("--seeds", type=int, nargs="+", default=[1, 2, 3]); ap.add_argument("--threads", type=int, default=8)
a = ap.parse_args()
m = scg.load_map(a.map)
mask = sum(1 << k for k in range(1, a.options + 1))
print(f"# oracle_learning_curve map {a.map} envs {a.envs} options {a.options} (synthetic nested-disc initiation sets) alpha {a.alpha} eps 0.05 gamma 0.99 "
      f"r_option_success 10000; goal arrivals per 1000 env-steps per chunk of {a.chunk} step-batches", flush=True)
for seed in a.seeds:
    for cached in (False, True):
        orc = sc_oracle.Oracle(m, SCALE, n_envs=a.envs, n_options=a.options, seed=seed, enabled_mask=mask, n_threads=a.threads,
                               gamma=0.99, alpha=a.alpha, epsilon=0.05, r_option_success=10000.0, max_episode_steps=2000, max_option_steps=200)
        if cached:
            orc.set_cached_baseline(True)
        st = sc_oracle.new_state(a.envs, m)
        W = np.zeros((a.options + 1, 5, 1296), np.float32)
        clf = chain_classifiers(m, a.options)
        rates, t = [], 0
        for it in range(a.iters):
            goals = 0
            for _ in range(a.chunk):
                G, n_k = orc.step(st, W, clf, t)
                orc.apply(W, G, n_k)
                goals += int((st["done"] == 1).sum()); t += 1
            rates.append(1000.0 * goals / (a.chunk * a.envs))
        print(f"seed {seed} {'cached baseline' if cached else 'exact          '}: " + " ".join(f"{r:5.2f}" for r in rates)
              + f"   |W|max {np.abs(W).max():7.1f}  in options at end {int((st['option_id'] > 0).sum())}", flush=True)
