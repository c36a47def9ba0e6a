"""Full-size (BASELINE configs[2]: 65 536 envs, 5 options) checks through size-independent properties:
determinism, shard equivalence (SPEC §2), legality of every env state, conservation of update counts,
plus an oracle spot-check of a few blocks of the big batch."""
import numpy as np
import pytest
import torch

import sc_oracle
from gpu_util import dev, make_pair, state_to_device
from util import chain_classifiers, random_weights

pytestmark = pytest.mark.gpu
N, NOPT, MASK = 65536, 5, 0b111110


def _run(n, base, x, y, vx, vy, steps, learn, seed=5):
    ctx, orc, m = make_pair("pinball_simple", n, n_options=NOPT, seed=seed, env_id_base=base, enabled_mask=MASK,
                            max_episode_steps=40)
    st = sc_oracle.new_state(n, m)
    st["x"][:], st["y"][:], st["vx"][:], st["vy"][:] = x, y, vx, vy
    st_d = state_to_device(st, ctx)
    W = dev(random_weights(NOPT + 1, 3, std=0.02))
    clf = dev(chain_classifiers(m, NOPT))
    G, n_k = ctx.grad_buffers()
    counts = []
    for t in range(steps):
        ctx.step(st_d, W.view(-1), clf.view(-1), MASK, t, learn=learn)
        if learn:
            counts.append((n_k.cpu().numpy().copy(), (st_d.option_id.cpu().numpy())))
    torch.cuda.synchronize()
    return st_d, W, counts, m, ctx


@pytest.fixture(scope="module")
def init():
    import skill_chaining_with_graphs_amd as scg
    m = scg.load_map("pinball_simple")
    rng = np.random.default_rng(0)
    pos = m.sample_free(N, rng)
    v = rng.uniform(-1, 1, (2, N)).astype(np.float32)
    return pos[:, 0].copy(), pos[:, 1].copy(), v[0].copy(), v[1].copy()


def test_full_size_properties_and_determinism(init):
    x, y, vx, vy = init
    st1, W1, c1, m, _ = _run(N, 0, x, y, vx, vy, 45, True)
    st2, W2, c2, _, _ = _run(N, 0, x, y, vx, vy, 45, True)
    for k in ("x", "y", "vx", "vy", "option_id", "qcache", "done"):
        assert torch.equal(getattr(st1, k), getattr(st2, k)), k
    assert torch.equal(W1, W2) and bool(torch.isfinite(W1).all())
    xs, ys = st1.x.cpu().numpy(), st1.y.cpu().numpy()
    assert np.all((xs >= 0) & (xs <= 1) & (ys >= 0) & (ys <= 1))
    assert m.free_mask(np.stack([xs, ys], 1), margin=0.0).all()          # no ball centre inside an obstacle
    sp = st1.vx.cpu().numpy().astype(np.float64) ** 2 + st1.vy.cpu().numpy().astype(np.float64) ** 2
    assert sp.max() <= 8.0 + 1e-4
    prev_opt = None
    for n_k, opt in c1:
        assert n_k[0] == N and np.all(n_k[1:] >= 0) and n_k[1:].sum() <= N
        if prev_opt is not None:                                          # n_k[k] = envs that were in option k
            assert np.array_equal(n_k[1:], np.bincount(np.maximum(prev_opt, 0), minlength=NOPT + 1)[1:])
        prev_opt = opt
    done = st1.done.cpu().numpy()
    assert set(np.unique(done)) <= {0, 1, 2}
    starts = {tuple(s) for s in m.starts.tolist()}
    for e in np.nonzero(done)[0][:200]:
        assert (float(xs[e]), float(ys[e])) in starts


def test_two_shards_equal_one_batch(init):
    x, y, vx, vy = init
    full, _, _, _, _ = _run(N, 0, x, y, vx, vy, 6, False)
    h = N // 2
    a, _, _, _, _ = _run(h, 0, x[:h], y[:h], vx[:h], vy[:h], 6, False)
    b, _, _, _, _ = _run(h, h, x[h:], y[h:], vx[h:], vy[h:], 6, False)
    for k in ("x", "y", "vx", "vy", "option_id", "action", "done", "reward"):
        assert torch.equal(getattr(full, k), torch.cat([getattr(a, k), getattr(b, k)])), k
    assert torch.equal(full.qcache, torch.cat([a.qcache, b.qcache], 1))


def test_oracle_spot_check_of_the_big_batch(init):
    """Blocks are independent given W, so the first 512 envs of the 65 536-env step must equal a 512-env
    oracle step with the same global ids (act + physics + options + qcache; learn off)."""
    x, y, vx, vy = init
    n = 512
    full, W, _, m, _ = _run(N, 0, x, y, vx, vy, 3, False)
    orc = sc_oracle.Oracle(m, __import__("util").SCALE, n_envs=n, n_options=NOPT, seed=5, enabled_mask=MASK,
                           n_threads=8, **dict(__import__("util").HP, max_episode_steps=40))
    st = sc_oracle.new_state(n, m)
    st["x"][:], st["y"][:], st["vx"][:], st["vy"][:] = x[:n], y[:n], vx[:n], vy[:n]
    Wn = W.cpu().numpy()
    clf = chain_classifiers(m, NOPT)
    for t in range(3):
        orc.step(st, Wn, clf, t)
    for k in ("x", "y", "vx", "vy", "option_id", "action", "done", "reward"):
        assert np.array_equal(getattr(full, k).cpu().numpy()[:n], st[k]), k
    assert np.array_equal(full.qcache.cpu().numpy()[:, :n], st["qcache"])


def test_quarter_million_envs_single_gpu():
    """4x the bench size on one GPU (1024 workgroups, 64 first-level reduction segments): invariants +
    an oracle spot check of the sorted-order bookkeeping (learn off => every env is independent)."""
    import skill_chaining_with_graphs_amd as scg
    n = 262144
    m = scg.load_map("pinball_maze")
    rng = np.random.default_rng(7)
    pos = m.sample_free(n, rng)
    v = rng.uniform(-1, 1, (2, n)).astype(np.float32)
    ctx, orc, m = make_pair("pinball_maze", n, n_options=NOPT, seed=8, enabled_mask=MASK, max_episode_steps=30)
    st = sc_oracle.new_state(n, m)
    st["x"][:], st["y"][:], st["vx"][:], st["vy"][:] = pos[:, 0], pos[:, 1], v[0], v[1]
    st_d = state_to_device(st, ctx)
    W = dev(random_weights(NOPT + 1, 3, std=0.02))
    clf = dev(chain_classifiers(m, NOPT))
    G, n_k = ctx.grad_buffers()
    for t in range(4):
        ctx.step(st_d, W.view(-1), clf.view(-1), MASK, t)
        nk = n_k.cpu().numpy()
        assert nk[0] == n and nk[1:].sum() <= n
    assert bool(torch.isfinite(W).all()) and bool(torch.isfinite(st_d.qcache).all())
    xs, ys = st_d.x.cpu().numpy(), st_d.y.cpu().numpy()
    assert m.free_mask(np.stack([xs, ys], 1)[::64], margin=0.0).all()
    opt = st_d.option_id.cpu().numpy()
    assert opt.min() >= -NOPT and opt.max() <= NOPT and len(np.unique(opt)) >= 3      # (-k: inside I_k, staying out of option k)


def test_the_headline_batch_learn_on_bit_exact():
    """The exact batch of the bench line — 65 536 envs, root + 5 options, learn on, one workgroup per CU — compared with the oracle directly
    (it was covered only through the 70 000-env case and through properties), eight step-batches."""
    test_more_than_256_workgroups_bit_exact(n=65536, steps=8, seed=11)


def test_more_than_256_workgroups_bit_exact(n=70000, steps=3, seed=8):
    """70 000 envs = 274 workgroups = 18 first-level segments: the reduce launch needs a second round of
    segments and the row placement a second stride over the count table — bit-exact against the oracle."""
    import skill_chaining_with_graphs_amd as scg
    m0 = scg.load_map("pinball_simple")
    rng = np.random.default_rng(21)
    pos = m0.sample_free(n, rng)
    v = rng.uniform(-1, 1, (2, n)).astype(np.float32)
    ctx, orc, m = make_pair("pinball_simple", n, n_options=NOPT, seed=seed, enabled_mask=MASK, max_episode_steps=40)
    st_o = sc_oracle.new_state(n, m)
    st_o["x"][:], st_o["y"][:], st_o["vx"][:], st_o["vy"][:] = pos[:, 0], pos[:, 1], v[0], v[1]
    st_o["ep_steps"][:] = rng.integers(0, 39, n)
    W_o = random_weights(NOPT + 1, 4, std=0.02)
    clf = chain_classifiers(m, NOPT)
    st_d = state_to_device(st_o, ctx)
    W_d, clf_d = dev(W_o.copy()), dev(clf)
    G_d, n_d = ctx.grad_buffers()
    for t in range(steps):
        G, n_k = orc.step(st_o, W_o, clf, t)
        orc.apply(W_o, G, n_k)
        ctx.step(st_d, W_d.view(-1), clf_d.view(-1), MASK, t)
        for k in ("x", "y", "vx", "vy", "option_id", "opt_steps", "ep_steps", "action", "done", "reward", "qcache"):
            assert np.array_equal(getattr(st_d, k).cpu().numpy().reshape(-1), st_o[k].reshape(-1)), (k, t)
        assert np.array_equal(n_d.cpu().numpy(), n_k) and np.array_equal(G_d.cpu().numpy(), G), t
        assert np.array_equal(W_d.cpu().numpy(), W_o), t
    assert n_k[1:].sum() > 0
