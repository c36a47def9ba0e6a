"""Parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same seeded
inputs — BIT-EXACT on every output, floats included (SPEC.md preamble). "Parity" here means parity with
this repo's oracle; the upstream reference has no code to compare with."""
import numpy as np
import pytest
import torch

import sc_oracle
import skill_chaining_with_graphs_amd as scg
from gpu_util import assert_state_equal, dev, make_pair, state_to_device
from util import chain_classifiers, random_states, random_weights

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("map_name", ["pinball_empty", "pinball_simple", "pinball_maze"])
def test_pinball_step_bit_exact(map_name):
    n = 20000
    ctx, orc, m = make_pair(map_name, n)
    x, y, vx, vy = random_states(m, n, 1, vmax=2.8)          # incl. speeds above the clip
    act = np.random.default_rng(2).integers(0, 5, n).astype(np.uint8)
    # a few envs next to the target so that the goal branch is exercised
    tx, ty, tr = m.target
    x[:64] = tx - tr - 0.003; y[:64] = ty; vx[:64] = 1.0; vy[:64] = 0.0
    d = [dev(a) for a in (x, y, vx, vy)]
    for it in range(4):
        r_o, g_o = orc.pinball_step(x, y, vx, vy, act)
        r_d, g_d = ctx.pinball_step(d, dev(act))
        for name, a, b in zip("x y vx vy".split(), d, (x, y, vx, vy)):
            assert np.array_equal(a.cpu().numpy(), b), (it, name)
        assert np.array_equal(r_d.cpu().numpy(), r_o) and np.array_equal(g_d.cpu().numpy(), g_o)
        if it == 0:
            assert g_o[:64].all() and 0 < g_o.sum() < n


def test_pinball_step_dense_map_four_mask_words_bit_exact():
    """160 edges (> 128: four 64-bit candidate words) and tight corridors (lanes with > 3 candidates)."""
    import sc_oracle
    from skill_chaining_with_graphs_amd.core import ScgContext
    from util import HP, SCALE, dense_map
    m = dense_map()
    assert m.n_edges > 128
    n = 30000
    ctx = ScgContext(n, 0, m, **HP)
    orc = sc_oracle.Oracle(m, SCALE, n_envs=n, **HP)
    x, y, vx, vy = random_states(m, n, 21, vmax=2.8)
    act = np.random.default_rng(22).integers(0, 5, n).astype(np.uint8)
    d = [dev(a) for a in (x, y, vx, vy)]
    for it in range(5):
        r_o, g_o = orc.pinball_step(x, y, vx, vy, act)
        r_d, g_d = ctx.pinball_step(d, dev(act))
        for name, a, b in zip("x y vx vy".split(), d, (x, y, vx, vy)):
            assert np.array_equal(a.cpu().numpy(), b), (it, name)
        assert np.array_equal(r_d.cpu().numpy(), r_o) and np.array_equal(g_d.cpu().numpy(), g_o)


def test_features_and_q_values_bit_exact():
    n = 3000                                                  # 12 blocks, last one ragged
    ctx, orc, m = make_pair("pinball_simple", n)
    x, y, vx, vy = random_states(m, n, 3, vmax=2.8)
    d = [dev(a) for a in (x, y, vx, vy)]
    assert np.array_equal(ctx.features(d).cpu().numpy(), orc.features(x, y, vx, vy))
    W = random_weights(1, 4, std=1.0)[0]
    q = ctx.q_values(d, dev(W).view(-1))
    assert np.array_equal(q.cpu().numpy(), orc.q_values(x, y, vx, vy, W))


def test_classifier_predict_and_fit_bit_exact():
    ctx, orc, m = make_pair("pinball_simple", 256)
    rng = np.random.default_rng(5)
    xy = rng.random((5000, 2)).astype(np.float32)
    lab = (((xy[:, 0] - 0.6) ** 2 + (xy[:, 1] - 0.4) ** 2) < 0.3 ** 2).astype(np.uint8)
    off = np.array([0, 3000, 3000, 5000], np.int32)          # middle problem is empty
    w_o = np.zeros((3, 8), np.float32)
    w_o[2, :3] = [0.1, -0.2, 0.3]
    w_d = dev(w_o.copy())
    orc.fit_initiation(xy, lab, off, w_o, iters=150, lr=3.0, l2=1e-4)
    ctx.fit_initiation(dev(xy).view(-1), dev(lab), dev(off), w_d.view(-1), iters=150, lr=3.0, l2=1e-4)
    assert np.array_equal(w_d.cpu().numpy(), w_o)
    x, y = xy[:, 0].copy(), xy[:, 1].copy()
    pred = ctx.classifier_predict(dev(x), dev(y), w_d[0].contiguous())
    assert np.array_equal(pred.cpu().numpy(), orc.classifier_predict(x, y, w_o[0]))
    assert (pred.cpu().numpy()[:3000] == lab[:3000]).mean() > 0.93


def test_fit_many_examples_and_many_problems_bit_exact():
    """SPEC §6 geometry: 8 workgroups x 1024 chains per option. Covers several examples per chain (20 000), the part
    beyond the 8 register-resident examples per chain (70 000 > 65 536), one example only, and more problems than one
    launch batch holds (10 > 8)."""
    ctx, orc, m = make_pair("pinball_simple", 256)
    rng = np.random.default_rng(15)
    sizes = [20000, 70000, 1, 300, 0, 9000, 64, 8192, 8193, 777]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    xy = rng.random((off[-1], 2)).astype(np.float32)
    lab = (((xy[:, 0] - 0.45) ** 2 + (xy[:, 1] - 0.55) ** 2) < 0.28 ** 2).astype(np.uint8)
    w_o = (rng.standard_normal((len(sizes), 8)) * 0.05).astype(np.float32)
    w_d = dev(w_o.copy())
    orc.fit_initiation(xy, lab, off, w_o, iters=12, lr=2.0, l2=1e-3)
    ctx.fit_initiation(dev(xy).view(-1), dev(lab), dev(off), w_d.view(-1), iters=12, lr=2.0, l2=1e-3)
    assert np.array_equal(w_d.cpu().numpy(), w_o)


@pytest.mark.parametrize("n,k", [(1, 0), (700, 0), (700, 2)])
def test_q_update_bit_exact(n, k):
    ctx, orc, m = make_pair("pinball_simple", 700, n_options=2)
    x, y, vx, vy = random_states(m, n, 6)
    xn, yn, vxn, vyn = random_states(m, n, 7)
    rng = np.random.default_rng(8)
    act = rng.integers(0, 5, n).astype(np.uint8)
    r = rng.choice([-1.0, -5.0, 10000.0], n).astype(np.float32)
    cont = np.where(rng.random(n) < 0.2, 0.0, 0.99).astype(np.float32)
    W = random_weights(3, 9, std=0.5)
    G_o, cnt = orc.q_update_grad((x, y, vx, vy), act, r, cont, (xn, yn, vxn, vyn), W[k])
    W_o = W.copy()
    n_k = np.zeros(3, np.int32); n_k[k] = cnt
    G_all = np.zeros((3, 5, 1296), np.float32); G_all[k] = G_o
    orc.apply(W_o, G_all, n_k)
    W_d = dev(W.copy())
    G_d, n_d = ctx.grad_buffers()
    ctx.q_update(k, [dev(a) for a in (x, y, vx, vy)], dev(act), dev(r), dev(cont),
                 [dev(a) for a in (xn, yn, vxn, vyn)], W_d.view(-1))
    assert n_d.cpu().numpy().tolist() == n_k.tolist()
    assert np.array_equal(G_d[k].cpu().numpy(), G_o)
    assert np.array_equal(W_d.cpu().numpy(), W_o)


@pytest.mark.parametrize("map_name,n,n_options,steps", [
    ("pinball_simple", 1, 0, 25),            # BASELINE config 1 shape on the GPU path
    ("pinball_simple", 4096, 1, 10),         # BASELINE config 2: root + 1 chained option
    ("pinball_maze", 1000, 5, 12),           # full chain, ragged last block, 74-edge map
])
def test_fused_step_rollout_bit_exact(map_name, n, n_options, steps):
    mask = sum(1 << k for k in range(1, n_options + 1))
    ctx, orc, m = make_pair(map_name, n, n_options=n_options, seed=42, enabled_mask=mask)
    clf = chain_classifiers(m, n_options)
    st_o = sc_oracle.new_state(n, m)
    if n > 1:
        x, y, vx, vy = random_states(m, n, 10, vmax=1.0)
        st_o["x"][:], st_o["y"][:], st_o["vx"][:], st_o["vy"][:] = x, y, vx, vy
        st_o["ep_steps"][:] = np.random.default_rng(11).integers(0, 59, n)
    W_o = random_weights(n_options + 1, 12, std=0.05)
    st_d = state_to_device(st_o, ctx)
    W_d, clf_d = dev(W_o.copy()), dev(clf)
    G_d, n_d = ctx.grad_buffers()
    seen_done = set()
    for t in range(steps):
        G, n_k = orc.step(st_o, W_o, clf, t)
        orc.apply(W_o, G, n_k)
        ctx.step(st_d, W_d.view(-1), clf_d.view(-1), mask, t)
        assert_state_equal(st_d, st_o, msg=f"t={t}")
        assert np.array_equal(n_d.cpu().numpy(), n_k), t
        assert np.array_equal(G_d.cpu().numpy(), G), t
        assert np.array_equal(W_d.cpu().numpy(), W_o), t
        seen_done |= set(np.unique(st_o["done"]).tolist())
    if n >= 1000:
        assert {0, 2} <= seen_done                     # time-limit resets happened
        assert n_k[1:].sum() > 0                       # option VFs were updated


def test_fused_step_act_only_and_split_apply():
    """learn=False leaves W and G alone; LEARN without APPLY + scg_apply_update == LEARN|APPLY."""
    n, n_options, mask = 777, 2, 0b110
    ctx, orc, m = make_pair("pinball_simple", n, n_options=n_options, seed=1, enabled_mask=mask)
    clf = chain_classifiers(m, n_options)
    st_o = sc_oracle.new_state(n, m)
    W_o = random_weights(3, 13, std=0.05)
    st_d = state_to_device(st_o, ctx)
    W_d, clf_d = dev(W_o.copy()), dev(clf)
    G_d, n_d = ctx.grad_buffers()
    ctx.step(st_d, W_d.view(-1), clf_d.view(-1), mask, 0, learn=False)
    orc.step(st_o, W_o, clf, 0)
    assert_state_equal(st_d, st_o)
    assert np.array_equal(W_d.cpu().numpy(), W_o)
    ctx.step(st_d, W_d.view(-1), clf_d.view(-1), mask, 1, learn=True, apply=False)
    G, n_k = orc.step(st_o, W_o, clf, 1)
    assert np.array_equal(W_d.cpu().numpy(), W_o) and np.array_equal(G_d.cpu().numpy(), G)
    ctx.apply_update(W_d.view(-1), G_d, n_d)
    orc.apply(W_o, G, n_k)
    assert np.array_equal(W_d.cpu().numpy(), W_o)


def test_env_order_prepared_by_the_previous_step_and_invalidated_on_outside_writes():
    """A learning step leaves the next step's env order (SPEC §5) behind; acting-only steps and option ids
    written by the caller (followed by scg_invalidate_order) fall back to a fresh sort. All bit-exact."""
    import torch
    n, n_options, mask = 3000, 3, 0b1110
    ctx, orc, m = make_pair("pinball_simple", n, n_options=n_options, seed=9, enabled_mask=mask)
    clf = chain_classifiers(m, n_options)
    st_o = sc_oracle.new_state(n, m)
    x, y, vx, vy = random_states(m, n, 3, vmax=1.0)
    st_o["x"][:], st_o["y"][:], st_o["vx"][:], st_o["vy"][:] = x, y, vx, vy
    W_o = random_weights(n_options + 1, 14, std=0.05)
    st_d = state_to_device(st_o, ctx)
    W_d, clf_d = dev(W_o.copy()), dev(clf)
    G_d, n_d = ctx.grad_buffers()
    rng = np.random.default_rng(5)
    for t in range(14):
        learn = t not in (4, 5, 9)                       # acting-only steps leave no prepared order
        if t in (3, 7, 10):                              # the caller rewrites some option ids in place
            idx = rng.choice(n, 500, replace=False)
            st_o["option_id"][idx] = 0
            st_o["opt_steps"][idx] = 0
            st_d.option_id.copy_(torch.as_tensor(st_o["option_id"]))
            st_d.opt_steps.copy_(torch.as_tensor(st_o["opt_steps"]))
            ctx.invalidate_order()
        G, n_k = orc.step(st_o, W_o, clf, t)
        ctx.step(st_d, W_d.view(-1), clf_d.view(-1), mask, t, learn=learn)
        assert_state_equal(st_d, st_o, msg=f"t={t}")
        if learn:
            orc.apply(W_o, G, n_k)
            assert np.array_equal(G_d.cpu().numpy(), G), t
        assert np.array_equal(W_d.cpu().numpy(), W_o), t
    assert n_k[1:].sum() > 0


def test_packed_gradient_operand_matches_the_split_pair():
    """scg_set_grad_buffer_packed: G and the counts (as floats) in one buffer — the single all-reduce operand of
    the shared-weights path — and scg_apply_update_packed give the same W as LEARN|APPLY."""
    n, n_options, mask = 1300, 2, 0b110
    ctx, orc, m = make_pair("pinball_simple", n, n_options=n_options, seed=3, enabled_mask=mask)
    clf = chain_classifiers(m, n_options)
    st_o = sc_oracle.new_state(n, m)
    x, y, vx, vy = random_states(m, n, 17, vmax=1.0)
    st_o["x"][:], st_o["y"][:], st_o["vx"][:], st_o["vy"][:] = x, y, vx, vy
    W_o = random_weights(3, 15, std=0.05)
    st_d = state_to_device(st_o, ctx)
    W_d, clf_d = dev(W_o.copy()), dev(clf)
    gp = ctx.grad_packed()
    for t in range(4):
        G, n_k = orc.step(st_o, W_o, clf, t)
        orc.apply(W_o, G, n_k)
        ctx.step(st_d, W_d.view(-1), clf_d.view(-1), mask, t, learn=True, apply=False)
        flat = gp.cpu().numpy()
        assert np.array_equal(flat[:G.size].reshape(G.shape), G)
        assert np.array_equal(flat[G.size:], n_k.astype(np.float32))
        ctx.apply_update_packed(W_d.view(-1), gp)
        assert np.array_equal(W_d.cpu().numpy(), W_o), t
    assert n_k[1:].sum() > 0


@pytest.mark.parametrize("n_slots", [1, 3, 8])
def test_order_pinned_slot_sum_is_the_sequential_float_sum(n_slots):
    """scg_apply_update_slots: G and the counts are the slots' sums IN SLOT ORDER (SPEC §5 multi-rank form), then the usual
    update — against float32 additions in numpy in the same order and the oracle's apply; host checks included."""
    nopt = 2
    ctx, orc, m = make_pair("pinball_simple", 64, n_options=nopt, seed=1)
    per = (nopt + 1) * 5 * 1296 + nopt + 1
    rng = np.random.default_rng(50 + n_slots)
    slots = (rng.standard_normal((n_slots, per)) * rng.choice([1e-3, 1.0, 1e3], size=(n_slots, 1))).astype(np.float32)
    counts = rng.integers(0, 5000, size=(n_slots, nopt + 1))
    counts[:, 1] = 0                                                  # a value function nobody updated stays untouched
    slots[:, per - (nopt + 1):] = counts.astype(np.float32)
    W_o = random_weights(nopt + 1, 9, std=0.05)
    W_d = dev(W_o.copy())
    ctx.apply_update_slots(W_d.view(-1), dev(slots))
    g = slots[0, :per - (nopt + 1)].copy()
    for r in range(1, n_slots):
        g = (g + slots[r, :per - (nopt + 1)]).astype(np.float32)
    orc.apply(W_o, g.reshape(nopt + 1, 5, 1296), counts.sum(axis=0).astype(np.int32))
    assert np.array_equal(W_d.cpu().numpy(), W_o)
    with pytest.raises(scg.ScgError):
        ctx.apply_update_slots(W_d.view(-1), dev(slots)[:, :-1].contiguous())


@pytest.mark.parametrize("n,dist", [
    (127, "uniform"), (129, "uniform"), (1000, "all3"), (1000, "none"), (1000, "heavy"), (4100, "uniform"),
    (4100, "one_each"), (4100, "heavy"), (2048, "no_root"),
])
def test_env_order_layouts_bit_exact(n, dist):
    """SPEC §5's env order in all its regimes — chunked and padded layouts, ragged last block, empty runs, no
    filler envs at all — through both sort paths (stand-alone kernels on step 0, the reduce launch's row
    workgroups afterwards), bit-exact against the oracle's procedural construction."""
    import torch
    nopt, mask = 5, 0b111110
    ctx, orc, m = make_pair("pinball_simple", n, n_options=nopt, seed=6, enabled_mask=mask)
    clf = chain_classifiers(m, nopt)
    rng = np.random.default_rng(n + len(dist))
    st_o = sc_oracle.new_state(n, m)
    x, y, vx, vy = random_states(m, n, 23, vmax=1.0)
    st_o["x"][:], st_o["y"][:], st_o["vx"][:], st_o["vy"][:] = x, y, vx, vy
    if dist == "uniform":
        opt = rng.integers(0, nopt + 1, n)
    elif dist == "all3":
        opt = np.full(n, 3)
    elif dist == "none":
        opt = np.zeros(n, int)
    elif dist == "heavy":
        opt = np.where(rng.random(n) < 0.95, rng.integers(1, nopt + 1, n), 0)
    elif dist == "one_each":
        opt = np.zeros(n, int); opt[[5, 700, 1300, 2500, 4000]] = [1, 2, 3, 4, 5]
    else:                                                # no_root: nobody can serve as filler
        opt = rng.integers(1, nopt + 1, n)
    st_o["option_id"][:] = opt
    st_o["opt_steps"][:] = rng.integers(0, 5, n) * (opt > 0)
    W_o = random_weights(nopt + 1, 16, std=0.05)
    st_d = state_to_device(st_o, ctx)
    W_d, clf_d = dev(W_o.copy()), dev(clf)
    G_d, n_d = ctx.grad_buffers()
    for t in range(3):
        G, n_k = orc.step(st_o, W_o, clf, t)
        orc.apply(W_o, G, n_k)
        ctx.step(st_d, W_d.view(-1), clf_d.view(-1), mask, t)
        assert_state_equal(st_d, st_o, msg=f"t={t}")
        assert np.array_equal(n_d.cpu().numpy(), n_k), t
        assert np.array_equal(G_d.cpu().numpy(), G), t
        assert np.array_equal(W_d.cpu().numpy(), W_o), t


def test_host_checks_fail_before_any_launch():
    from skill_chaining_with_graphs_amd import ScgError
    ctx, orc, m = make_pair("pinball_simple", 64)
    x = torch.zeros(64, device="cuda:0")
    with pytest.raises(ScgError):
        ctx.pinball_step((x, x, x, x.double()), torch.zeros(64, dtype=torch.uint8, device="cuda:0"))
    with pytest.raises(ScgError):
        ctx.pinball_step((x, x, x, x), torch.full((64,), 7, dtype=torch.uint8, device="cuda:0"))
    with pytest.raises(ScgError):
        ctx.q_values((x, x, x, x), torch.zeros(10, device="cuda:0"))
    with pytest.raises(ScgError):
        ctx.features((x.cpu(), x, x, x))


def test_skill_tree_rollout_bit_exact_and_graph_export():
    """SPEC §4.2 option graph: a tree (1 -> goal, 2 -> goal, 3 -> 1, 4 -> 2, 5 -> 4) instead of the default chain."""
    from util import disc_weights
    from skill_chaining_with_graphs_amd import ScgError
    n, n_options, mask, parents = 2000, 5, 0b111110, [0, 0, 0, 1, 2, 4]
    ctx, orc, m = make_pair("pinball_simple", n, n_options=n_options, seed=17, enabled_mask=mask)
    ctx.set_option_parents(parents)
    orc.set_parents(parents)
    tx, ty, _ = m.target
    clf = np.zeros((6, 8), np.float32)
    for k, (cx, cy, r) in enumerate([(tx, ty, 0.15), (tx - 0.25, ty + 0.1, 0.15), (tx, ty + 0.3, 0.2),
                                     (tx - 0.45, ty + 0.3, 0.2), (tx - 0.6, ty + 0.55, 0.25)], start=1):
        clf[k] = disc_weights(cx, cy, r)
    st_o = sc_oracle.new_state(n, m)
    x, y, vx, vy = random_states(m, n, 18, vmax=1.0)
    st_o["x"][:], st_o["y"][:], st_o["vx"][:], st_o["vy"][:] = x, y, vx, vy
    W_o = random_weights(6, 19, std=0.05)
    st_d, W_d, clf_d = state_to_device(st_o, ctx), dev(W_o.copy()), dev(clf)
    G_d, n_d = ctx.grad_buffers()
    seen = set()
    for t in range(10):
        G, n_k = orc.step(st_o, W_o, clf, t)
        orc.apply(W_o, G, n_k)
        ctx.step(st_d, W_d.view(-1), clf_d.view(-1), mask, t)
        assert_state_equal(st_d, st_o, msg=f"t={t}")
        assert np.array_equal(W_d.cpu().numpy(), W_o), t
        seen |= set(np.unique(st_o["option_id"]).tolist())
    assert {1, 2, 3, 4, 5} <= seen
    for bad in ([0, 2, 1, 0, 0, 0], [0, 0, 0, 3, 0, 0], [0, 0, 0, 0, 0, 9]):      # 1<->2 cycle, self loop, out of range
        with pytest.raises(ScgError):
            ctx.set_option_parents(bad)


def test_agent_skill_graph_export():
    from skill_chaining_with_graphs_amd import SkillChainingAgent
    ag = SkillChainingAgent("pinball_simple", 256, 3)
    ag.set_option_parents([0, 0, 1, 1])
    ag.enable_option(1)
    g = ag.skill_graph()
    assert sorted(g.edges()) == [(1, 0), (2, 1), (3, 1)] and g.nodes[1]["enabled"] and not g.nodes[2]["enabled"]
