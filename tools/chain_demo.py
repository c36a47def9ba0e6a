"""Demo: skill chaining end to end on the GPU. The root policy learns Pinball for a while, then
SkillChainingAgent.chain_skills() creates options backwards from the goal (trajectory ring -> harvest -> GPU
logistic regression -> enable), and learning continues with the chain. Prints goal arrivals per 1000 env-steps
before and after, the per-option report, and the skill graph's edges."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from skill_chaining_with_graphs_amd import SkillChainingAgent

ap = argparse.ArgumentParser()
ap.add_argument("--map", default="pinball_simple"); ap.add_argument("--envs", type=int, default=8192)
ap.add_argument("--options", type=int, default=3); ap.add_argument("--alpha", type=float, default=0.02)
ap.add_argument("--warm", type=int, default=3000); ap.add_argument("--after", type=int, default=3000)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--gestation", type=int, default=0, help="successes a new option must see before it is enabled (SPEC 4.4)")
a = ap.parse_args()
ag = SkillChainingAgent(a.map, a.envs, a.options, seed=a.seed, alpha=a.alpha, epsilon=0.05, gamma=0.99,
                        max_episode_steps=2000, max_option_steps=200, update_count_floor=a.envs // 16)
ag.enable_tracing(64)


def run(steps, tag):
    goals = torch.zeros((), device="cuda")
    for _ in range(steps):
        ag.step_batch()
        goals += (ag.state.done == 1).sum()
    rate = 1000.0 * float(goals) / (steps * a.envs)
    print(f"{tag}: {steps} step-batches, goals per 1000 env-steps = {rate:.3f}", flush=True)
    return rate


run(a.warm // 2, "root policy, first half ")
run(a.warm // 2, "root policy, second half")
report = ag.chain_skills(steps_per_option=400, min_examples=3000, max_examples=40000, start_coverage=0.9, gestation=a.gestation)
for r in report:
    print("created", r, flush=True)
inside = [(int((ag.state.option_id == k).sum())) for k in range(a.options + 1)]
print("envs per running option (0 = none):", inside)
run(a.after, "with the chain           ")
print("skill graph edges (option -> target):", sorted(ag.skill_graph().edges()))
