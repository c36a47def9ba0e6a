"""Behavioural evidence that the learner learns and that the chain helps.

For each seed, three runs on the same map and env count, all with the same number of env-steps:
  root-only : SkillChainingAgent with 0 options, W warm-up + A "after" step-batches
  chain     : W warm-up step-batches, chain_skills() (its step-batches are counted and matched in the control),
              then A step-batches with the discovered options
  chain+gest: the same with a gestation period (SPEC 4.4)
Prints goal arrivals per 1000 env-steps over the LAST `after` step-batches of every run (the control's window starts
at the same env-step count as the chain run's window), the per-option reports and a one-line verdict per seed.
With no upstream code to pin against (reference = README.md:1-2, the paper's title), "chaining is not worse than the
flat learner at equal env-steps" is the only external anchor this repo has; it is a loose, statistical one."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from skill_chaining_with_graphs_amd import SkillChainingAgent

ap = argparse.ArgumentParser()
ap.add_argument("--map", default="pinball_simple"); ap.add_argument("--envs", type=int, default=8192)
ap.add_argument("--options", type=int, default=3); ap.add_argument("--alpha", type=float, default=0.02)
ap.add_argument("--warm", type=int, default=3000); ap.add_argument("--after", type=int, default=3000)
ap.add_argument("--seeds", type=int, nargs="+", default=[1, 2, 3])
ap.add_argument("--gestation", type=int, default=200)
ap.add_argument("--r-succ", type=float, default=0.0, help="option completion reward (SPEC 4.2 r_option_success)")
ap.add_argument("--reoffer", type=int, default=4, help="SPEC 4.2 reoffer_period")
ap.add_argument("--floor-div", type=int, default=16, help="update_count_floor = envs / this (0: no floor)")
ap.add_argument("--max-option-steps", type=int, default=200)
ap.add_argument("--json", default=None, help="also write the rows as JSON lines to this file")
a = ap.parse_args()
HP = dict(alpha=a.alpha, epsilon=0.05, gamma=0.99, max_episode_steps=2000, max_option_steps=a.max_option_steps, r_option_success=a.r_succ,
          update_count_floor=(a.envs // a.floor_div if a.floor_div else 0), reoffer_period=a.reoffer)
print(f"# chain_evidence map {a.map} envs {a.envs} options {a.options} warm {a.warm} after {a.after} gestation {a.gestation} hparams {HP}", flush=True)


def run(ag, steps):
    goals = torch.zeros((), device="cuda")
    for _ in range(steps):
        ag.step_batch()
        goals += (ag.state.done == 1).sum()
    return 1000.0 * float(goals) / max(steps * a.envs, 1)


rows = []
for seed in a.seeds:
    res = {}
    for mode in ("chain", "chain+gest"):
        ag = SkillChainingAgent(a.map, a.envs, a.options, seed=seed, **HP)
        ag.enable_tracing(64)
        warm_rate = run(ag, a.warm)
        t0 = ag.t
        report = ag.chain_skills(steps_per_option=400, min_examples=3000, max_examples=40000, start_coverage=0.9,
                                 gestation=a.gestation if mode == "chain+gest" else 0)
        discovery_steps = ag.t - t0
        after_rate = run(ag, a.after)
        inside = [int((ag.state.option_id == k).sum()) for k in range(a.options + 1)]
        res[mode] = dict(seed=seed, mode=mode, warm_rate=warm_rate, discovery_steps=discovery_steps, after_rate=after_rate,
                         options=[{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()} for r in report],
                         envs_per_option=inside, edges=sorted(ag.skill_graph().edges()))
        print(f"seed {seed} {mode:10s}: warm {warm_rate:6.3f} -> after {after_rate:6.3f} goals/1k env-steps "
              f"({len(report)} options, {discovery_steps} discovery step-batches, envs per option {inside})", flush=True)
        for r in res[mode]["options"]:
            print("    created", r, flush=True)
        del ag
    # control: the flat learner for as many step-batches as the plain chain run, same window
    ag = SkillChainingAgent(a.map, a.envs, 0, seed=seed, **HP)
    warm_rate = run(ag, a.warm)
    run(ag, res["chain"]["discovery_steps"])
    after_rate = run(ag, a.after)
    res["root-only"] = dict(seed=seed, mode="root-only", warm_rate=warm_rate, after_rate=after_rate)
    print(f"seed {seed} root-only : warm {warm_rate:6.3f} -> after {after_rate:6.3f} goals/1k env-steps "
          f"(same env-steps as the chain run)", flush=True)
    del ag
    c, g, r0 = res["chain"]["after_rate"], res["chain+gest"]["after_rate"], after_rate
    print(f"seed {seed} verdict   : chain / root-only = {c / max(r0, 1e-9):.2f}, chain+gest / root-only = {g / max(r0, 1e-9):.2f}", flush=True)
    rows += list(res.values())
if a.json:
    with open(a.json, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
