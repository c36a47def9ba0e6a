"""Diagnostic: does the batched root Q-learning actually learn Pinball? Prints goal arrivals per 1000 env-steps."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from skill_chaining_with_graphs_amd import SkillChainingAgent
ap = argparse.ArgumentParser()
ap.add_argument("--map", default="pinball_simple"); ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--alpha", type=float, default=0.02); ap.add_argument("--eps", type=float, default=0.05)
ap.add_argument("--gamma", type=float, default=0.99); ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--chunk", type=int, default=500); ap.add_argument("--maxep", type=int, default=2000)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
ag = SkillChainingAgent(a.map, a.envs, 0, seed=a.seed, alpha=a.alpha, epsilon=a.eps, gamma=a.gamma, max_episode_steps=a.maxep)
print(f"# learning_curve map {a.map} envs {a.envs} alpha {a.alpha} eps {a.eps} gamma {a.gamma} seed {a.seed}", flush=True)
for it in range(a.iters):
    goals = torch.zeros((), device="cuda"); touts = torch.zeros((), device="cuda")
    for _ in range(a.chunk):
        ag.step_batch()
        goals += (ag.state.done == 1).sum(); touts += (ag.state.done == 2).sum()
    g, t = int(goals), int(touts)
    print(f"iter {it:3d}  goals/1k env-steps {1000*g/(a.chunk*a.envs):7.3f}  timeouts {t:6d}  |W|max {float(ag.W.abs().max()):9.3f}", flush=True)
