import os, sys, numpy as np, torch
sys.path[:0] = ["/root/repo"]
import bench
from skill_chaining_with_graphs_amd import SkillChainingAgent
n = 65536
ag = SkillChainingAgent(bench.MAP, n, 5, seed=0, **bench.HP)
ag.clf.copy_(torch.as_tensor(bench.chain_discs(ag.map, 5)))
for k in range(1, 6): ag.enable_option(k)
ag.init_weights(std=1e-3, seed=0); ag.domain.reset_random(seed=1000, v_max=1.0)
for t in (300, 1000, 3000):
    while ag.t < t: ag.step_batch()
    a = ag.state.action.cpu().numpy()
    print("t", t, "action shares", np.round(np.bincount(a, minlength=5) / n, 3).tolist())
# per block (in the env order the kernel uses: approximate by env id blocks of 256)
a = ag.state.action.cpu().numpy().reshape(-1, 256)
cnt = np.stack([(a == k).sum(1) for k in range(5)], 1)
g = (cnt + 3) // 4
S = np.stack([g[:, 0] + g[:, 1] + g[:, 2] + g[:, 4], g[:, 0] + g[:, 1] + g[:, 3] + g[:, 4], g[:, 0] + g[:, 2] + g[:, 3] + g[:, 4], g[:, 1] + g[:, 2] + g[:, 3]], 1)
print("groups per action per block: mean", g.mean(0).round(1).tolist(), "max-SIMD load / mean-SIMD load per block: mean %.3f" % (S.max(1) / (g.sum(1) * 15 / 16 / 4 * 1.0)).mean(), " (SIMD loads mean", S.mean(0).round(1).tolist(), ")")
