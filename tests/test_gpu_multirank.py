"""The shared-weights multi-rank product path (BASELINE configs[4]) on ONE GPU:

 * two ScgContexts in one process stand in for two ranks (env shards [0, n) and [n, 2n) by global id): each
   scg_step(LEARN) leaves its rank-local gradient in its packed operand, the two operands are summed (what the
   all-reduce does — with two ranks the sum is a single addition, hence order-independent and exact),
   scg_apply_update_packed runs on both; compared bit for bit with the oracle's two shards;
 * two child processes on cuda:0 (gloo stages the collectives through the host) run
   SkillChainingAgent(group=WORLD).step_batch — the code bench.py --shared-weights runs under RCCL — and must end
   with identical weights, equal to the in-process two-context result; the sharded outer loop (chain_skills) must
   terminate with identical classifier tables on both ranks (rank-local loop decisions would deadlock)."""
import os
import sys

import numpy as np
import pytest
import torch

import sc_oracle
from gpu_util import assert_state_equal, dev, make_pair, state_to_device
from util import chain_classifiers, random_states, random_weights

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NS, NOPT, MASK, STEPS, SEED = 640, 2, 0b110, 4, 17          # 640 envs per rank: 5 blocks


def _shard_inputs(m):
    x, y, vx, vy = random_states(m, 2 * NS, 31, vmax=1.0)
    return x, y, vx, vy


def test_two_contexts_packed_operand_equals_the_oracles_two_shards():
    ranks = []
    for r in range(2):
        ctx, orc, m = make_pair("pinball_simple", NS, n_options=NOPT, seed=SEED, env_id_base=r * NS, enabled_mask=MASK)
        ranks.append((ctx, orc, m))
    m = ranks[0][2]
    x, y, vx, vy = _shard_inputs(m)
    clf = chain_classifiers(m, NOPT)
    W_o = random_weights(NOPT + 1, 4, std=0.05)
    st_o, st_d, W_d, gp = [], [], [], []
    for r, (ctx, orc, _) in enumerate(ranks):
        st = sc_oracle.new_state(NS, m)
        sl = slice(r * NS, (r + 1) * NS)
        st["x"][:], st["y"][:], st["vx"][:], st["vy"][:] = x[sl], y[sl], vx[sl], vy[sl]
        st_o.append(st); st_d.append(state_to_device(st, ctx)); W_d.append(dev(W_o.copy())); gp.append(ctx.grad_packed())
    clf_d = dev(clf)
    nw = (NOPT + 1) * 5 * 1296
    for t in range(STEPS):
        G, n = [], []
        for r, (ctx, orc, _) in enumerate(ranks):
            g_r, n_r = orc.step(st_o[r], W_o, clf, t)
            G.append(g_r); n.append(n_r)
            ctx.step(st_d[r], W_d[r].view(-1), clf_d.view(-1), MASK, t, learn=True, apply=False)
        total = gp[0] + gp[1]                                       # the all-reduce
        for r, (ctx, orc, _) in enumerate(ranks):
            assert np.array_equal(gp[r][:nw].cpu().numpy().reshape(NOPT + 1, 5, 1296), G[r]), f"rank {r} gradient, step {t}"
            gp[r].copy_(total)
            ctx.apply_update_packed(W_d[r].view(-1), gp[r])
        ranks[0][1].apply(W_o, G[0] + G[1], n[0] + n[1])
        torch.cuda.synchronize()
        assert np.array_equal(total[nw:].cpu().numpy(), (n[0] + n[1]).astype(np.float32))      # counts: exact
        for r in range(2):
            assert_state_equal(st_d[r], st_o[r], msg=f"rank {r} step {t}")
        assert torch.equal(W_d[0], W_d[1])                                                     # same weights on both ranks
        assert np.array_equal(W_d[0].cpu().numpy(), W_o), f"weights differ from the oracle's two-shard result at step {t}"


def _rank_main(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    import skill_chaining_with_graphs_amd as scg
    from skill_chaining_with_graphs_amd import SkillChainingAgent
    from util import HP
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = scg.load_map("pinball_simple")
    x, y, vx, vy = _shard_inputs(m)
    ag = SkillChainingAgent(m, NS, NOPT, device=0, seed=SEED, env_id_base=rank * NS, group=dist.group.WORLD, **HP)
    sl = slice(rank * NS, (rank + 1) * NS)
    for name, v in (("x", x), ("y", y), ("vx", vx), ("vy", vy)):
        getattr(ag.state, name).copy_(torch.as_tensor(v[sl].copy(), device="cuda:0"))
    ag.clf.copy_(torch.as_tensor(chain_classifiers(m, NOPT), device="cuda:0"))
    ag.enabled_mask = MASK
    ag.W.copy_(torch.as_tensor(random_weights(NOPT + 1, 4, std=0.05), device="cuda:0"))
    ag.time_allreduce(2)                           # bench.py's hook: event pairs round every second packed all-reduce
    for _ in range(STEPS):
        ag.step_batch()
    torch.cuda.synchronize()
    ar = ag.time_allreduce(0)
    assert ar is not None and ar["samples"] == (STEPS + 1) // 2 and 0 < ar["mean_us"] <= ar["max_us"] and ag.allreduce_timing is None
    res = {"W": ag.W.cpu().numpy(), "x": ag.state.x.cpu().numpy(), "option_id": ag.state.option_id.cpu().numpy()}
    # the sharded outer loop: decisions on all-reduced counts, fit on the examples of both ranks
    ag2 = SkillChainingAgent("pinball_empty", 4096, 2, device=0, seed=5, env_id_base=rank * 4096, group=dist.group.WORLD,
                             epsilon=1.0, alpha=1e-4, max_episode_steps=400)
    ag2.enable_tracing(64)
    ag2.domain.reset_random(seed=11 + rank, v_max=0.5)
    report = ag2.chain_skills(steps_per_option=200, min_examples=2000, max_examples=20000, start_coverage=2.0)
    res["clf"] = ag2.clf.cpu().numpy()
    res["n_created"] = np.array([len(report)])
    res["W2"] = ag2.W.cpu().numpy()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_processes_share_weights_through_the_agent(tmp_path):
    import torch.multiprocessing as mp
    port = 29700 + os.getpid() % 200
    mp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["W"], r1["W"])                          # one all-reduce per step keeps the ranks in lock step
    # the same two shards as two contexts in this process
    ranks = [make_pair("pinball_simple", NS, n_options=NOPT, seed=SEED, env_id_base=r * NS, enabled_mask=MASK) for r in range(2)]
    m = ranks[0][2]
    x, y, vx, vy = _shard_inputs(m)
    clf_d = dev(chain_classifiers(m, NOPT))
    st_d, W_d, gp = [], [], []
    for r, (ctx, _, _) in enumerate(ranks):
        st = sc_oracle.new_state(NS, m)
        sl = slice(r * NS, (r + 1) * NS)
        st["x"][:], st["y"][:], st["vx"][:], st["vy"][:] = x[sl], y[sl], vx[sl], vy[sl]
        st_d.append(state_to_device(st, ctx)); W_d.append(dev(random_weights(NOPT + 1, 4, std=0.05))); gp.append(ctx.grad_packed())
    for t in range(STEPS):
        for r, (ctx, _, _) in enumerate(ranks):
            ctx.step(st_d[r], W_d[r].view(-1), clf_d.view(-1), MASK, t, learn=True, apply=False)
        total = gp[0] + gp[1]
        for r, (ctx, _, _) in enumerate(ranks):
            gp[r].copy_(total)
            ctx.apply_update_packed(W_d[r].view(-1), gp[r])
    torch.cuda.synchronize()
    assert np.array_equal(r0["W"], W_d[0].cpu().numpy())
    assert np.array_equal(r0["x"], st_d[0].x.cpu().numpy()) and np.array_equal(r1["x"], st_d[1].x.cpu().numpy())
    assert np.array_equal(r1["option_id"], st_d[1].option_id.cpu().numpy())
    # sharded skill discovery: same number of options, same classifier table, same (shared) weights on both ranks
    assert int(r0["n_created"][0]) == int(r1["n_created"][0]) >= 1
    assert np.array_equal(r0["clf"], r1["clf"]) and np.array_equal(r0["W2"], r1["W2"])


def test_five_shards_with_the_order_pinned_sum_equal_the_oracles_five_shards():
    """Beyond two ranks an all-reduce's order of additions is the library's; the order-pinned form (all-gather of the packed
    operands + scg_apply_update_slots: the sum ((G_0 + G_1) + G_2) + ... element by element) is reproduced by the oracle
    exactly: five contexts stand in for five ranks, the float32 sums are re-done in numpy in the same order."""
    R, n = 5, 300                                                    # 300 envs per rank: a ragged second block
    ranks = [make_pair("pinball_simple", n, n_options=NOPT, seed=SEED, env_id_base=r * n, enabled_mask=MASK) for r in range(R)]
    m = ranks[0][2]
    x, y, vx, vy = random_states(m, R * n, 77, vmax=1.0)
    clf = chain_classifiers(m, NOPT)
    W_o = random_weights(NOPT + 1, 4, std=0.05)
    st_o, st_d, W_d, gp = [], [], [], []
    for r, (ctx, orc, _) in enumerate(ranks):
        st = sc_oracle.new_state(n, m)
        sl = slice(r * n, (r + 1) * n)
        st["x"][:], st["y"][:], st["vx"][:], st["vy"][:] = x[sl], y[sl], vx[sl], vy[sl]
        st_o.append(st); st_d.append(state_to_device(st, ctx)); W_d.append(dev(W_o.copy())); gp.append(ctx.grad_packed())
    clf_d = dev(clf)
    differs_from_other_orders = False
    for t in range(6):
        G, cnt = [], []
        for r, (ctx, orc, _) in enumerate(ranks):
            g_r, n_r = orc.step(st_o[r], W_o, clf, t)
            G.append(g_r); cnt.append(n_r)
            ctx.step(st_d[r], W_d[r].view(-1), clf_d.view(-1), MASK, t, learn=True, apply=False)
        slots = torch.stack(gp).contiguous()                         # what all_gather_into_tensor leaves on every rank
        for r, (ctx, _, _) in enumerate(ranks):
            ctx.apply_update_slots(W_d[r].view(-1), slots)
        g_sum, n_sum = G[0].copy(), cnt[0].copy()
        for r in range(1, R):
            g_sum = (g_sum + G[r]).astype(np.float32)                # float32 additions in rank order
            n_sum = n_sum + cnt[r]
        g_rev = G[R - 1].copy()
        for r in range(R - 2, -1, -1):
            g_rev = (g_rev + G[r]).astype(np.float32)
        differs_from_other_orders |= not np.array_equal(g_rev, g_sum)
        ranks[0][1].apply(W_o, g_sum, n_sum)
        torch.cuda.synchronize()
        for r in range(R):
            assert_state_equal(st_d[r], st_o[r], msg=f"rank {r} step {t}")
            assert torch.equal(W_d[r], W_d[0])
        assert np.array_equal(W_d[0].cpu().numpy(), W_o), f"weights differ from the oracle's five-shard result at step {t}"
    assert differs_from_other_orders                                 # (the order does matter at five ranks: the test can tell)


def _rank_main_ordered(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    import skill_chaining_with_graphs_amd as scg
    from skill_chaining_with_graphs_amd import SkillChainingAgent
    from util import HP
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = scg.load_map("pinball_simple")
    n = 300
    x, y, vx, vy = random_states(m, world * n, 77, vmax=1.0)
    ag = SkillChainingAgent(m, n, NOPT, device=0, seed=SEED, env_id_base=rank * n, group=dist.group.WORLD, ordered_sum=True, **HP)
    sl = slice(rank * n, (rank + 1) * n)
    for name, v in (("x", x), ("y", y), ("vx", vx), ("vy", vy)):
        getattr(ag.state, name).copy_(torch.as_tensor(v[sl].copy(), device="cuda:0"))
    ag.clf.copy_(torch.as_tensor(chain_classifiers(m, NOPT), device="cuda:0"))
    ag.enabled_mask = MASK
    ag.W.copy_(torch.as_tensor(random_weights(NOPT + 1, 4, std=0.05), device="cuda:0"))
    for _ in range(6):
        ag.step_batch()
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"ordered{rank}.npz"), W=ag.W.cpu().numpy(), x=ag.state.x.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_processes_with_the_order_pinned_sum_equal_the_oracle(tmp_path):
    """SkillChainingAgent(group=WORLD, ordered_sum=True) in three rank processes on cuda:0 (gloo through the host): identical
    weights on all ranks, equal bit for bit to the oracle's three shards summed in rank order."""
    import torch.multiprocessing as mp
    R, n = 3, 300
    port = 29900 + os.getpid() % 90
    mp.spawn(_rank_main_ordered, args=(R, port, str(tmp_path)), nprocs=R, join=True)
    res = [np.load(tmp_path / f"ordered{r}.npz") for r in range(R)]
    assert all(np.array_equal(res[0]["W"], res[r]["W"]) for r in range(R))
    from util import HP, SCALE
    import skill_chaining_with_graphs_amd as scg
    m = scg.load_map("pinball_simple")
    x, y, vx, vy = random_states(m, R * n, 77, vmax=1.0)
    clf = chain_classifiers(m, NOPT)
    W_o = random_weights(NOPT + 1, 4, std=0.05)
    orcs, sts = [], []
    for r in range(R):
        orcs.append(sc_oracle.Oracle(m, SCALE, n_envs=n, n_options=NOPT, seed=SEED, env_id_base=r * n, enabled_mask=MASK, n_threads=4, **HP))
        st = sc_oracle.new_state(n, m)
        sl = slice(r * n, (r + 1) * n)
        st["x"][:], st["y"][:], st["vx"][:], st["vy"][:] = x[sl], y[sl], vx[sl], vy[sl]
        sts.append(st)
    for t in range(6):
        out = [orcs[r].step(sts[r], W_o, clf, t) for r in range(R)]
        g_sum, n_sum = out[0][0].copy(), out[0][1].copy()
        for r in range(1, R):
            g_sum = (g_sum + out[r][0]).astype(np.float32)
            n_sum = n_sum + out[r][1]
        orcs[0].apply(W_o, g_sum, n_sum)
    assert np.array_equal(res[0]["W"], W_o)
    for r in range(R):
        assert np.array_equal(res[r]["x"], sts[r]["x"])
