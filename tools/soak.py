"""Soak: 30 000 learning step-batches of the bench workload on one GPU; prints the asynchronous status word, whether W stayed finite,
|W|max and how many envs sit in an option at the end (rounds 2 and 3 print the same numbers: the arithmetic did not change)."""
import sys, time, torch
import os; sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import bench
from skill_chaining_with_graphs_amd import SkillChainingAgent
ag = SkillChainingAgent(bench.MAP, 65536, 5, seed=0, **bench.HP)
ag.clf.copy_(torch.as_tensor(bench.chain_discs(ag.map, 5)))
for k in range(1,6): ag.enable_option(k)
ag.init_weights(std=1e-3); ag.domain.reset_random(seed=1000, v_max=1.0)
t0=time.time()
for i in range(30000): ag.step_batch()
torch.cuda.synchronize()
print("30000 learning steps in %.1f s; status word %d; W finite %s |W|max %.1f; envs in an option %d; goals this step %d" % (
    time.time()-t0, ag.ctx.async_status(True), bool(torch.isfinite(ag.W).all()), float(ag.W.abs().max()), int((ag.state.option_id>0).sum()), int((ag.state.done==1).sum())))
