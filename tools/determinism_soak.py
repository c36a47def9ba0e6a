"""Long determinism check: two agents with the same seeds stepped side by side must agree bit for bit (W, states, qcache) however
long the run is — short parity rollouts would not see a rare race. Prints the first step-batch at which they differ, if any.
   python tools/determinism_soak.py [steps]"""
import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch
import bench
from skill_chaining_with_graphs_amd import SkillChainingAgent

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
cases = [("8192 envs, root only, alpha 0.02 (the warm-up of tools/chain_evidence.py)", 8192, 0, dict(alpha=0.02, epsilon=0.05, gamma=0.99, max_episode_steps=2000, max_option_steps=200, r_option_success=10000.0)),
         ("65536 envs, root + 5 options (the bench workload)", 65536, 5, dict(bench.HP)),
         ("5000 envs, root + 3 options, ragged last block", 5000, 3, dict(bench.HP, alpha=0.01))]
bad = 0
for name, n, nopt, hp in cases:
    ags = []
    for _ in range(2):
        ag = SkillChainingAgent(bench.MAP, n, nopt, seed=1, **hp)
        if nopt:
            ag.clf.copy_(torch.as_tensor(bench.chain_discs(ag.map, nopt)))
            for k in range(1, nopt + 1):
                ag.enable_option(k)
        ag.init_weights(std=1e-3, seed=1)
        ag.domain.reset_random(seed=1001, v_max=1.0)
        ags.append(ag)
    first = None
    for t in range(steps):
        for ag in ags:
            ag.step_batch()
        if (t + 1) % 100 == 0 or t == steps - 1:
            same = torch.equal(ags[0].W, ags[1].W) and all(torch.equal(getattr(ags[0].state, k), getattr(ags[1].state, k))
                                                            for k in ("x", "y", "vx", "vy", "option_id", "opt_steps", "ep_steps", "qcache", "action"))
            if not same:
                first = t + 1
                break
    goals = int((ags[0].state.done == 1).sum())
    print(f"{name}: {steps} step-batches x 2 agents: " + ("bit-identical throughout" if first is None else f"DIFFER by step-batch {first}")
          + f"; async status {ags[0].ctx.async_status(True)}; goals in the last step-batch {goals}", flush=True)
    bad += first is not None
    del ags
    torch.cuda.empty_cache()
sys.exit(1 if bad else 0)
