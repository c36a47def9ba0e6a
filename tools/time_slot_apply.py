"""What the order-pinned multi-rank apply launch costs on the critical path of a shared-weights step (the price
of NOT folding the slot sum into the next step's weight staging). One GPU; the all-gather itself is not here (no second GPU): the
slots are this rank's own packed operand copied n_ranks times, which exercises the launch exactly as a node would after its gather.
    python tools/time_slot_apply.py [--ranks 8] [--steps 400]
Prints us per step-batch of (a) step with its own fused apply (independent shards: the headline), (b) step without apply + copy
into the slots + scg_apply_update_slots, (c) step without apply alone; (b) - (c) is what a fold into stage_w could remove at most."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import skill_chaining_with_graphs_amd as scg
from bench import HP, MAP, chain_discs     # the headline workload (synthetic nested-disc initiation sets)

ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, default=8); ap.add_argument("--steps", type=int, default=400); ap.add_argument("--envs", type=int, default=65536)
a = ap.parse_args()


def run(mode):
    ag = scg.SkillChainingAgent(MAP, a.envs, 5, seed=0, **HP)
    ag.clf.copy_(torch.as_tensor(chain_discs(ag.map, 5)))
    for k in range(1, 6):
        ag.enable_option(k)
    ag.init_weights(std=1e-3, seed=0)
    ag.domain.reset_random(seed=1000, v_max=1.0)
    ctx = ag.ctx
    gp = ctx.grad_packed()
    slots = torch.zeros((a.ranks, gp.numel()), dtype=torch.float32, device=gp.device)
    def one():
        if mode == "fused":
            ctx.step(ag.state, ag.W, ag.clf, ag.enabled_mask, ag.t, learn=True, apply=True)
        else:
            ctx.step(ag.state, ag.W, ag.clf, ag.enabled_mask, ag.t, learn=True, apply=False)
            if mode == "slots":
                slots[0].copy_(gp)                       # (stands in for the all-gather's landing: the other slots stay zero operands)
                ctx.apply_update_slots(ag.W, slots)
        ag.t += 1
    for _ in range(200): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps): one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.steps * 1e6


res = {m: run(m) for m in ("fused", "slots", "noapply")}
print(f"envs {a.envs} ranks {a.ranks} steps {a.steps}: step with fused apply {res['fused']:.2f} us; step + slot copy + scg_apply_update_slots "
      f"{res['slots']:.2f} us; step without any apply {res['noapply']:.2f} us; the slot apply on the critical path = {res['slots'] - res['noapply']:.2f} us "
      f"(of which a fold into the weight staging could remove at most that)")
