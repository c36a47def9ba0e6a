"""What the outer skill-discovery loop costs per step-batch: env-steps/s of
    plain      step_batch()                                   — the bench's hot path
    traced     step_batch() with the trajectory ring + events attached (SPEC §7)
    discovery  traced + collect_examples() after every step-batch — exactly what chain_skills issues
on the bench workload (65 536 envs, map pinball_simple), for two stages of a chain: creating option 1 (no option
enabled, trigger = goal bit) and creating option 5 (options 1-4 enabled, trigger = entering option 4's initiation set).
Prints one JSON object; run it under rocprofv3 --kernel-trace --stats for the per-kernel view (collect_* kernels).
Usage: python tools/time_discovery.py [--steps K] [--envs N]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--ring", type=int, default=256)
a = ap.parse_args()
import torch
import bench
from skill_chaining_with_graphs_amd import SkillChainingAgent


def run(stage_k: int) -> dict:
    agent = SkillChainingAgent(bench.MAP, a.envs, 5, seed=0, **bench.HP)
    agent.clf.copy_(torch.as_tensor(bench.chain_discs(agent.map, 5)))
    for k in range(1, stage_k):
        agent.enable_option(k)
    agent.init_weights(std=1e-3); agent.domain.reset_random(seed=1000, v_max=1.0)

    def timed(fn, steps):
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"env_steps_per_s": a.envs * steps / dt, "us_per_step_batch": dt / steps * 1e6}

    out = {"creating_option": stage_k, "options_enabled": stage_k - 1}
    out["plain"] = timed(agent.step_batch, a.steps)
    agent.enable_tracing(ring_len=a.ring, max_examples=1 << 20)
    out["traced"] = timed(agent.step_batch, a.steps)

    def disc():
        agent.step_batch()
        agent.collect_examples(stage_k, 24, 24)
    out["discovery"] = timed(disc, a.steps)
    out["examples_collected"] = agent.examples_held(stage_k)
    for k in ("traced", "discovery"):
        out[k]["vs_plain"] = out[k]["env_steps_per_s"] / out["plain"]["env_steps_per_s"]
    out["collect_us_per_step_batch"] = out["discovery"]["us_per_step_batch"] - out["traced"]["us_per_step_batch"]
    return out


res = {"workload": {"envs": a.envs, "map": bench.MAP, "ring_len": a.ring, "l_pos": 24, "l_neg": 24, "steps": a.steps},
       "stages": [run(1), run(5)]}
print(json.dumps(res))
