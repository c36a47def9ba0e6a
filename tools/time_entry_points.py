"""Times the un-fused entry points on one GPU (diagnostic; not part of the bench contract)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
import skill_chaining_with_graphs_amd as scg
from skill_chaining_with_graphs_amd.core import ScgContext

def t_ms(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

N = 65536
for name in ("pinball_empty", "pinball_simple", "pinball_maze"):
    m = scg.load_map(name)
    ctx = ScgContext(N, 0, m)
    rng = np.random.default_rng(0)
    pos = m.sample_free(N, rng)
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda:0")
    x, y = d(pos[:, 0]), d(pos[:, 1])
    v = rng.uniform(-1, 1, (2, N)).astype(np.float32)
    vx, vy = d(v[0]), d(v[1])
    act = d(rng.integers(0, 5, N).astype(np.uint8))
    x0, y0, vx0, vy0 = x.clone(), y.clone(), vx.clone(), vy.clone()
    def step():
        x.copy_(x0); y.copy_(y0); vx.copy_(vx0); vy.copy_(vy0)
        ctx.pinball_step((x, y, vx, vy), act)
    def copies():
        x.copy_(x0); y.copy_(y0); vx.copy_(vx0); vy.copy_(vy0)
    print(name, m.n_edges, "edges: pinball_step", round(t_ms(step) - t_ms(copies), 4), "ms / 65536 envs")
W = torch.randn(5 * 1296, device="cuda:0") * 0.01
print("q_values", round(t_ms(lambda: ctx.q_values((x, y, vx, vy), W)), 4), "ms / 65536 envs")
# SkillChainingAgent.q_update on explicit transitions (td_kernel<MODE_TRANS> + reduce) and its pieces' context: the fused step's own
# kernel on the same 65 536 envs, root only (no option work), for scale
xn, yn, vxn, vyn = x0.clone(), y0.clone(), vx0.clone(), vy0.clone()
ctx.pinball_step((xn, yn, vxn, vyn), act)
r = torch.full((N,), -1.0, device="cuda:0"); cont = torch.full((N,), 0.99, device="cuda:0")
Wall = torch.randn(1 * 5 * 1296, device="cuda:0") * 0.01
print("q_update (65536 transitions, one value function, apply)", round(1e3 * t_ms(lambda: ctx.q_update(0, (x0, y0, vx0, vy0), act, r, cont, (xn, yn, vxn, vyn), Wall)), 1), "us")
print("q_update without apply", round(1e3 * t_ms(lambda: ctx.q_update(0, (x0, y0, vx0, vy0), act, r, cont, (xn, yn, vxn, vyn), Wall, apply=False)), 1), "us")
from skill_chaining_with_graphs_amd import SkillChainingAgent
ag = SkillChainingAgent("pinball_simple", N, 0, seed=0)
ag.domain.reset_random(seed=1)
print("fused step-batch, root only (act + physics + TD + apply)", round(1e3 * t_ms(lambda: ag.step_batch(), 200), 1), "us")
