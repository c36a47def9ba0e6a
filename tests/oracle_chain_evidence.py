"""Does the discovered chain help the learner, and under which EXIT RULE of SPEC §4.2?

Runs the CPU ORACLE (the checker — hence under tests/) through the same protocol as tools/chain_evidence.py runs the HIP
library: per seed a root-only warm-up, then either `chain_skills` (mirrored here on the oracle's own collect / fit entry
points, SkillChainingAgent.chain_skills's defaults) followed by `after` step-batches with the discovered options, or the flat
learner for the same number of step-batches. Printed: goal arrivals per 1000 env-steps over the last `after` step-batches and
chain / flat per (seed, rule, r_option_success). The warm-up (identical for every arm of a seed) is run once per seed.
    python tests/oracle_chain_evidence.py [--rules 0 1 2] [--r-succ 50 1000 10000] [--seeds 1 2 3] [--envs 4096]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [p for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")) if p not in sys.path]
import numpy as np
import sc_oracle
from util import SCALE
import skill_chaining_with_graphs_amd as scg

ap = argparse.ArgumentParser()
ap.add_argument("--map", default="pinball_simple"); ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--options", type=int, default=3); ap.add_argument("--alpha", type=float, default=0.02)
ap.add_argument("--warm", type=int, default=3000); ap.add_argument("--after", type=int, default=3000)
ap.add_argument("--seeds", type=int, nargs="+", default=[1, 2, 3]); ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--rules", type=int, nargs="+", default=[0, 1, 2]); ap.add_argument("--r-succ", type=float, nargs="+", default=[50.0, 1000.0, 10000.0])
ap.add_argument("--max-option-steps", type=int, default=200)
ap.add_argument("--arms", nargs="+", default=None, help="explicit arms rule:r_succ (e.g. 0:10000 2:50) instead of the rules x r-succ grid")
ap.add_argument("--reoffer", type=int, default=1, help="SPEC 4.2: an env staying out of an option is offered it again every this many steps")
ap.add_argument("--nk-floor", type=int, default=0, help="SPEC 5 apply divisor max(n_k, floor)")
ap.add_argument("--chunk", type=int, default=0, help="also print the goal rate per chunk of this many step-batches of the `after` window")
a = ap.parse_args()
m = scg.load_map(a.map)
N, K = a.envs, a.options
print(f"# oracle_chain_evidence map {a.map} envs {N} options {K} alpha {a.alpha} eps 0.05 gamma 0.99 max_episode_steps 2000 "
      f"max_option_steps {a.max_option_steps} warm {a.warm} after {a.after}; goal arrivals per 1000 env-steps", flush=True)


def make(seed, r_succ, rule):
    orc = sc_oracle.Oracle(m, SCALE, n_envs=N, n_options=K, seed=seed, enabled_mask=0, n_threads=a.threads, gamma=0.99, alpha=a.alpha,
                           epsilon=0.05, r_option_success=r_succ, max_episode_steps=2000, max_option_steps=a.max_option_steps)
    orc.p.exit_rule = rule % 10
    orc.p.nk_floor = a.nk_floor
    orc.p.reoffer_period = a.reoffer
    orc.p.select_rule = rule // 10          # (rules >= 10: value-gated entry on top of exit rule rule % 10)
    orc.set_trace(64)
    return orc


def run(orc, st, W, clf, t, steps, mask, series=None):
    goals = cg = 0
    for i in range(steps):
        G, n_k = orc.step(st, W, clf, t, enabled_mask=mask)
        orc.apply(W, G, n_k)
        g = int((st["done"] == 1).sum()); goals += g; cg += g; t += 1
        if series is not None and a.chunk and (i + 1) % a.chunk == 0:
            series.append(1000.0 * cg / (a.chunk * N)); cg = 0
    return t, 1000.0 * goals / max(steps * N, 1)


def chain_skills(orc, st, W, clf, t, steps_per_option=400, min_examples=3000, max_examples=40000, l_pos=24, l_neg=24,
                 start_coverage=0.9, poll_every=8, cap=65536):
    """SkillChainingAgent.chain_skills on the oracle (no gestation)."""
    mask, report = 0, []
    sx, sy = m.starts[:, 0].astype(np.float32).copy(), m.starts[:, 1].astype(np.float32).copy()
    for k in range(1, K + 1):
        parent = k - 1
        ex_xy, ex_lab, cnt = np.zeros((cap, 2), np.float32), np.zeros(cap, np.uint8), np.zeros(1, np.int32)
        prev = np.zeros(N, np.uint8) if parent else None
        steps = 0
        while steps < steps_per_option and cnt[0] < min(max_examples, cap):
            for _ in range(min(poll_every, steps_per_option - steps)):
                G, n_k = orc.step(st, W, clf, t, enabled_mask=mask)
                orc.apply(W, G, n_k); t += 1
                orc.collect_examples(1 if parent == 0 else (1 << parent), prev, l_pos, l_neg, ex_xy, ex_lab, cnt)
                steps += 1
        got = int(cnt[0])
        if got < min_examples:
            break
        w = np.zeros((1, 8), np.float32)
        orc.fit_initiation(ex_xy[:got], ex_lab[:got], np.array([0, got], np.int32), w, 400, 3.0, 1e-4)
        clf[k] = w[0]
        W[k] = W[0]
        mask |= 1 << k
        pred = orc.classifier_predict(ex_xy[:got, 0].copy(), ex_xy[:got, 1].copy(), clf[k])
        acc = float((pred == ex_lab[:got]).mean())
        cov = float(orc.classifier_predict(sx, sy, clf[k]).mean())
        report.append(dict(option=k, steps=steps, examples=got, accuracy=round(acc, 3), start_coverage=round(cov, 3)))
        if cov >= start_coverage:
            break
    return t, mask, report


for seed in a.seeds:
    t0w = time.time()
    orc = make(seed, a.r_succ[0], 0)                       # (no options yet: r_succ and the rule do not matter)
    st = sc_oracle.new_state(N, m)
    W = np.zeros((K + 1, 5, 1296), np.float32)
    clf = np.zeros((K + 1, 8), np.float32)
    t, warm_rate = run(orc, st, W, clf, 0, a.warm, 0)
    snap = ({k: v.copy() for k, v in st.items()}, W.copy(), orc.ring_x.copy(), orc.ring_y.copy(), orc.events.copy(), orc.ev_len.copy(), t)
    print(f"seed {seed} warm-up   : {warm_rate:6.3f} ({time.time() - t0w:.0f} s)", flush=True)
    flat, flat_series = {}, {}
    arms = [(float(x.split(":")[1]), int(x.split(":")[0])) for x in a.arms] if a.arms else [(rs, ru) for rs in a.r_succ for ru in a.rules]
    for r_succ, rule in arms:
        if True:
            t0 = time.time()
            orc = make(seed, r_succ, rule)
            st = {k: v.copy() for k, v in snap[0].items()}; W = snap[1].copy(); clf = np.zeros((K + 1, 8), np.float32)
            orc.ring_x[:], orc.ring_y[:], orc.events[:], orc.ev_len[:] = snap[2], snap[3], snap[4], snap[5]
            t, mask, report = chain_skills(orc, st, W, clf, snap[6])
            disc = t - snap[6]
            ser = []
            t, rate = run(orc, st, W, clf, t, a.after, mask, ser)
            inside = [int((st["option_id"] == k).sum()) for k in range(K + 1)]
            if disc not in flat:                           # control: the flat learner over the same env-steps
                o2 = make(seed, r_succ, 0)
                st2 = {k: v.copy() for k, v in snap[0].items()}; W2 = snap[1].copy(); c2 = np.zeros((K + 1, 8), np.float32)
                t2, _ = run(o2, st2, W2, c2, snap[6], disc, 0)
                flat_series[disc] = []
                _, flat[disc] = run(o2, st2, W2, c2, t2, a.after, 0, flat_series[disc])
            print(f"seed {seed} rule {rule} r_succ {r_succ:7.0f}: chain {rate:6.3f}  flat {flat[disc]:6.3f}  chain/flat {rate / max(flat[disc], 1e-9):5.2f}  "
                  f"({len(report)} options, {disc} discovery step-batches, envs per option {inside}, |W|max {np.abs(W).max():.0f}, "
                  f"acc {[r['accuracy'] for r in report]}, {time.time() - t0:.0f} s)", flush=True)
            if a.chunk:
                print("        chain per chunk: " + " ".join(f"{v:6.2f}" for v in ser) + "\n        flat  per chunk: " + " ".join(f"{v:6.2f}" for v in flat_series[disc]), flush=True)
