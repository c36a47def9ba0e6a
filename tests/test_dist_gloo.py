"""N>1 path on CPU (world_size 2, gloo). The kernels need a GPU, so the CPU oracle stands in for scg_step and produces
each rank's (G, n_k); everything else is the product's own multi-rank code: `dist.shard_range` for the env shards and
`dist.allreduce_packed` on the ONE packed operand (G, then the counts as floats) that `SkillChainingAgent.step_batch`
all-reduces, followed by the packed apply rule (counts recovered from the float tail, as apply_kernel does).
Checked against the single-process run: exact on counts and on every env trajectory, to tolerance on the weights
(the all-reduce sum is not order-pinned — SPEC §5). The outer-loop collectives (`allreduce_sum_int`,
`allgather_rows`) are exercised with ragged per-rank inputs. The same path on the GPU: tests/test_gpu_multirank.py."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, NOPT, MASK, STEPS = 512, 2, 0b110, 4


def _init_state(m, lo, hi):
    import sc_oracle
    from util import random_states
    x, y, vx, vy = random_states(m, N, 21, vmax=1.0)
    st = sc_oracle.new_state(hi - lo, m)
    st["x"][:], st["y"][:], st["vx"][:], st["vy"][:] = x[lo:hi], y[lo:hi], vx[lo:hi], vy[lo:hi]
    return st


def _worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from skill_chaining_with_graphs_amd.dist import allgather_packed, allgather_rows, allreduce_packed, allreduce_sum_int, shard_range
    from util import chain_classifiers, make_oracle, random_weights
    lo, hi = shard_range(N, rank, world)
    orc, m = make_oracle("pinball_simple", n_envs=hi - lo, n_options=NOPT, seed=9, env_id_base=lo,
                         enabled_mask=MASK, n_threads=1)
    st, W, clf = _init_state(m, lo, hi), random_weights(NOPT + 1, 5, std=0.05), chain_classifiers(m, NOPT)
    n_vf, nw = NOPT + 1, (NOPT + 1) * 5 * 1296
    for t in range(STEPS):
        G, n_k = orc.step(st, W, clf, t)
        gp = torch.cat([torch.from_numpy(G).view(-1), torch.from_numpy(n_k.astype(np.float32))])   # the packed operand
        allreduce_packed(gp, dist.group.WORLD)
        counts = (gp[nw:].numpy() + 0.5).astype(np.int32)                   # apply_kernel: (int)(nk_f + 0.5)
        orc.apply(W, gp[:nw].numpy().reshape(n_vf, 5, 1296), counts)
    # the order-pinned form: every rank's operand in rank order (what scg_apply_update_slots sums)
    mine = torch.arange(6, dtype=torch.float32) + 100.0 * rank
    slots = torch.zeros((world, 6))
    allgather_packed(mine, slots, dist.group.WORLD)
    assert torch.equal(slots, torch.stack([torch.arange(6, dtype=torch.float32) + 100.0 * r for r in range(world)]))
    total = allreduce_sum_int(rank + 3, dist.group.WORLD, "cpu")
    rows = allgather_rows(torch.full((rank + 1, 2), float(rank)), dist.group.WORLD)
    out[rank] = (W, st["x"], st["option_id"], st["done"], counts, total, rows.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_gloo_ranks_match_one_process(oracle_mod):
    from util import chain_classifiers, make_oracle, random_weights
    world, port = 2, 29500 + os.getpid() % 1000
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    orc, m = make_oracle("pinball_simple", n_envs=N, n_options=NOPT, seed=9, enabled_mask=MASK, n_threads=2)
    st, W, clf = _init_state(m, 0, N), random_weights(NOPT + 1, 5, std=0.05), chain_classifiers(m, NOPT)
    for t in range(STEPS):
        G, n_k = orc.step(st, W, clf, t)
        orc.apply(W, G, n_k)
    W0, x0, o0, d0, n0, tot0, rows0 = out[0]
    W1, x1, o1, d1, n1, tot1, rows1 = out[1]
    assert np.array_equal(W0, W1)                                   # both ranks hold the same weights
    assert np.array_equal(n0, n_k) and np.array_equal(n1, n_k)      # counts are exact
    assert np.allclose(W0, W, rtol=1e-4, atol=1e-6)                 # sums differ only by association
    # trajectories: the acting policy reads qcache, whose values depend on W only to ~1e-7 relative — the greedy
    # action could flip on an exact near-tie; none does on this workload, so the shards reproduce the batch exactly
    assert np.array_equal(np.concatenate([o0, o1]), st["option_id"])
    assert np.array_equal(np.concatenate([d0, d1]), st["done"])
    assert np.array_equal(np.concatenate([x0, x1]), st["x"])
    assert tot0 == tot1 == 7
    want = np.array([[0, 0], [1, 1], [1, 1]], np.float32)
    assert np.array_equal(rows0, want) and np.array_equal(rows1, want)
