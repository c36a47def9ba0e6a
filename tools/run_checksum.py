"""Cross-process reproducibility: run K learning step-batches from fixed seeds and print a SHA-256 over W and the env states.
Two processes must print the same line; --poison first fills (and frees) GPU memory with 0xFF bytes, so a kernel that reads a
buffer nobody initialised would show up as a different checksum.   python tools/run_checksum.py [--poison] [--steps K] [--envs N] [--options K]"""
import argparse, hashlib, os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch
import bench
from skill_chaining_with_graphs_amd import SkillChainingAgent

ap = argparse.ArgumentParser()
ap.add_argument("--poison", action="store_true"); ap.add_argument("--steps", type=int, default=600)
ap.add_argument("--envs", type=int, default=8192); ap.add_argument("--options", type=int, default=0)
ap.add_argument("--trace", action="store_true")
a = ap.parse_args()
if a.poison:
    junk = [torch.full((1 << 28,), 0xFF, dtype=torch.uint8, device="cuda") for _ in range(24)]      # 6 GiB of 0xFF
    torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
hp = dict(alpha=0.02, epsilon=0.05, gamma=0.99, max_episode_steps=2000, max_option_steps=200, r_option_success=10000.0)
ag = SkillChainingAgent(bench.MAP, a.envs, a.options, seed=1, **hp)
if a.options:
    ag.clf.copy_(torch.as_tensor(bench.chain_discs(ag.map, a.options)))
    for k in range(1, a.options + 1):
        ag.enable_option(k)
if a.trace:
    ag.enable_tracing(64)
h = hashlib.sha256()
goals = 0
for t in range(a.steps):
    ag.step_batch()
    goals += int((ag.state.done == 1).sum()) if t % 50 == 0 else 0
for tns in (ag.W, ag.state.x, ag.state.y, ag.state.vx, ag.state.vy, ag.state.qcache, ag.state.option_id, ag.state.ep_steps):
    h.update(tns.cpu().numpy().tobytes())
print(f"envs {a.envs} options {a.options} steps {a.steps} trace {a.trace} poison {a.poison}: sha256 {h.hexdigest()[:24]} |W|max {float(ag.W.abs().max()):.4f} status {ag.ctx.async_status(True)}")
