"""Oracle-level tests of the batched agent semantics (SPEC §4–§6) — CPU only."""
import numpy as np
import pytest

import sc_oracle
from util import (HP, SCALE, chain_classifiers, disc_weights, fourier_reference, make_oracle, random_states,
                  random_weights)


def test_config1_single_env_root_only_is_online_q_learning():
    """BASELINE config 1: 1 env, order 5, root Q only. One step-batch == the textbook online update."""
    orc, m = make_oracle("pinball_simple", n_envs=1, n_options=0, seed=7, epsilon=0.0)
    st = sc_oracle.new_state(1, m)
    W = random_weights(1, 0)
    clf = np.zeros((1, 8), np.float32)
    st["qcache"][:, 0] = orc.q_values(st["x"], st["y"], st["vx"], st["vy"], W[0])[:, 0]
    s = [st[k].copy() for k in ("x", "y", "vx", "vy")]
    a_expect = int(np.argmax(st["qcache"][:, 0]))
    G, n_k = orc.step(st, W, clf, t=0)
    assert n_k.tolist() == [1] and st["action"][0] == a_expect and st["done"][0] == 0
    # float64 restatement of delta * phi(s)
    phi_s = fourier_reference(*s)[0]
    phi_n = fourier_reference(st["x"], st["y"], st["vx"], st["vy"])[0]
    W64 = W[0].astype(np.float64)
    delta = st["reward"][0] + HP["gamma"] * np.max(W64 @ phi_n) - W64[a_expect] @ phi_s
    assert np.allclose(G[0, a_expect], delta * phi_s, rtol=1e-4, atol=1e-4)
    others = [a for a in range(5) if a != a_expect]
    assert np.all(G[0, others] == 0)
    Wn = W.copy()
    orc.apply(Wn, G, n_k)
    assert np.allclose(Wn[0, a_expect], W[0, a_expect] + HP["alpha"] * SCALE * delta * phi_s, rtol=1e-4, atol=1e-6)
    # qcache now holds Q(s', .) under the pre-update weights
    assert np.allclose(st["qcache"][:, 0], W64 @ phi_n, atol=2e-4)


def test_rollout_invariants_and_determinism():
    n = 600                                  # 3 blocks (one ragged)
    orc, m = make_oracle("pinball_simple", n_envs=n, n_options=2, seed=3, enabled_mask=0b110)
    clf = chain_classifiers(m, 2)

    def run():
        st = sc_oracle.new_state(n, m)
        x, y, vx, vy = random_states(m, n, 5, vmax=1.0)
        st["x"][:], st["y"][:], st["vx"][:], st["vy"][:] = x, y, vx, vy
        W = random_weights(3, 1)
        hist = []
        for t in range(12):
            G, n_k = orc.step(st, W, clf, t)
            orc.apply(W, G, n_k)
            assert n_k[0] == n and 0 <= n_k[1] <= n and 0 <= n_k[2] <= n
            assert np.all((st["option_id"] >= -2) & (st["option_id"] <= 2))      # (-k: inside I_k, staying out of option k — SPEC §4.2)
            assert np.all(st["action"] < 5) and np.all(st["done"] <= 2)
            assert np.all(np.isfinite(W)) and np.all(np.isfinite(st["qcache"]))
            hist.append((st["x"].copy(), st["option_id"].copy(), st["done"].copy(), n_k.copy()))
        return st, W, hist

    st1, W1, h1 = run()
    st2, W2, h2 = run()
    assert np.array_equal(W1, W2) and all(np.array_equal(a[0], b[0]) for a, b in zip(h1, h2))
    assert any(h[3][1] > 0 for h in h1) and any(h[3][2] > 0 for h in h1)      # options really ran


def test_option_termination_and_selection_rules():
    """One env walked through SPEC §4.2 by construction."""
    orc, m = make_oracle("pinball_empty", n_envs=1, n_options=2, seed=0, epsilon=0.0, enabled_mask=0b110)
    clf = np.zeros((3, 8), np.float32)
    clf[1] = disc_weights(0.8, 0.8, 0.15)            # I_1: disc round the goal
    clf[2] = disc_weights(0.8, 0.8, 0.40)            # I_2 contains I_1
    W = np.zeros((3, 5, 1296), np.float32)
    st = sc_oracle.new_state(1, m)
    # (a) outside both sets -> root
    st["x"][0], st["y"][0] = 0.2, 0.2
    orc.step(st, W, clf, 0)
    assert st["option_id"][0] == 0
    # (b) inside I_2 only -> option 2 selected
    st["x"][0], st["y"][0] = 0.55, 0.8
    orc.step(st, W, clf, 1)
    assert st["option_id"][0] == 2 and st["opt_steps"][0] == 0
    # (c) still inside I_2, not in I_1 -> keeps running, counter advances, VF 2 updated
    G, n_k = orc.step(st, W, clf, 2)
    assert st["option_id"][0] == 2 and st["opt_steps"][0] == 1 and n_k.tolist() == [1, 0, 1]
    # (d) teleport into I_1: option 2 succeeds (reward bonus), option 1 takes over
    st["x"][0], st["y"][0] = 0.7, 0.8
    G, n_k = orc.step(st, W, clf, 3)
    assert st["option_id"][0] == 1 and st["opt_steps"][0] == 0 and n_k.tolist() == [1, 0, 1]
    # W = 0 => Q = 0 => delta = r: root sees -1 (NONE is the greedy action 0? no: argmax of zeros = 0 = ACC_X)
    a = st["action"][0]
    assert a == 0 and st["reward"][0] == -5.0
    assert G[2, a, 0] == pytest.approx(-5.0 + HP["r_option_success"])      # phi_0 = 1, terminal: target = r_o
    assert G[0, a, 0] == pytest.approx(-5.0)
    # (e) leave I_1 without reaching the goal -> fail, falls back to option 2
    st["x"][0], st["y"][0] = 0.55, 0.8
    G, n_k = orc.step(st, W, clf, 4)
    assert st["option_id"][0] == 2 and n_k.tolist() == [1, 1, 0]
    assert G[1, st["action"][0], 0] == pytest.approx(-5.0)                 # failure: no bonus, no bootstrap
    # (f) option time-out
    orc.p.max_option_steps = 2
    orc.step(st, W, clf, 5)
    assert st["opt_steps"][0] == 1
    orc.step(st, W, clf, 6)
    assert st["opt_steps"][0] == 0 and st["option_id"][0] == 2             # timed out, re-selected at once
    # (g) disabled options are never selected
    orc.p.enabled_mask = 0
    orc.step(st, W, clf, 7)
    assert st["option_id"][0] == 0


def test_episode_reset_and_timeout():
    orc, m = make_oracle("pinball_empty", n_envs=4, n_options=0, seed=5, epsilon=1.0, max_episode_steps=3)
    st = sc_oracle.new_state(4, m)
    W = np.zeros((1, 5, 1296), np.float32)
    clf = np.zeros((1, 8), np.float32)
    starts = {tuple(s) for s in m.starts.tolist()}
    for t in range(3):
        orc.step(st, W, clf, t)
    assert np.all(st["done"] == 2) and np.all(st["ep_steps"] == 0)
    assert all((float(a), float(b)) in starts for a, b in zip(st["x"], st["y"]))
    assert np.all(st["vx"] == 0) and np.all(st["vy"] == 0)
    # goal: put an env next to the target moving in
    tx, ty, tr = m.target
    st["x"][0], st["y"][0], st["vx"][0], st["vy"][0] = tx - tr - 0.004, ty, 1.0, 0.0
    G, n_k = orc.step(st, W, clf, 3)
    assert st["done"][0] == 1 and st["reward"][0] == 10000.0
    assert G[0, st["action"][0], 0] == pytest.approx(10000.0)              # terminal: no bootstrap


def test_sharded_envs_reproduce_the_single_process_streams():
    """SPEC §2: keyed by global env id, 2 x 300 envs == 1 x 600 envs (acting only; W frozen)."""
    n = 600
    full, m = make_oracle("pinball_maze", n_envs=n, n_options=1, seed=11, enabled_mask=0b10)
    clf = chain_classifiers(m, 1)
    W = random_weights(2, 2)
    x, y, vx, vy = random_states(m, n, 9, vmax=1.0)

    def init(lo, hi):
        st = sc_oracle.new_state(hi - lo, m)
        st["x"][:], st["y"][:], st["vx"][:], st["vy"][:] = x[lo:hi], y[lo:hi], vx[lo:hi], vy[lo:hi]
        return st

    sf = init(0, n)
    shards = []
    for lo, hi in ((0, 300), (300, 600)):
        o, _ = make_oracle("pinball_maze", n_envs=hi - lo, n_options=1, seed=11, env_id_base=lo, enabled_mask=0b10)
        shards.append((o, init(lo, hi)))
    for t in range(6):
        full.step(sf, W, clf, t)
        for o, s in shards:
            o.step(s, W, clf, t)
    for key in ("x", "y", "vx", "vy", "option_id", "action", "done", "reward"):
        assert np.array_equal(sf[key], np.concatenate([s[key] for _, s in shards])), key
    assert np.array_equal(sf["qcache"], np.concatenate([s["qcache"] for _, s in shards], 1))


def test_fit_initiation_learns_a_disc():
    orc, _ = make_oracle("pinball_empty")
    rng = np.random.default_rng(0)
    xy = rng.random((3000, 2)).astype(np.float32)
    lab = (((xy[:, 0] - 0.65) ** 2 + (xy[:, 1] - 0.35) ** 2) < 0.25 ** 2).astype(np.uint8)
    w = np.zeros((2, 8), np.float32)
    off = np.array([0, 3000, 3000 + 1000], np.int32)
    xy2 = np.concatenate([xy, xy[:1000]])
    lab2 = np.concatenate([lab, lab[:1000]])
    orc.fit_initiation(xy2, lab2, off, w, iters=400, lr=4.0, l2=1e-5)
    for q, (lo, hi) in enumerate(((0, 3000), (3000, 4000))):
        pred = orc.classifier_predict(xy2[lo:hi, 0].copy(), xy2[lo:hi, 1].copy(), w[q])
        assert (pred == lab2[lo:hi]).mean() > 0.95
    # float64 gradient descent lands on (nearly) the same weights
    u, v = 2 * xy[:, 0].astype(np.float64) - 1, 2 * xy[:, 1].astype(np.float64) - 1
    psi = np.stack([np.ones_like(u), u, v, u * u, u * v, v * v], 1)
    w64 = np.zeros(6)
    for _ in range(400):
        p = 1 / (1 + np.exp(-(psi @ w64)))
        g = psi.T @ (p - lab) / 3000
        reg = 1e-5 * w64
        reg[0] = 0
        w64 -= 4.0 * (g + reg)
    assert np.allclose(w[0, :6], w64, rtol=2e-3, atol=2e-3)


def test_skill_graph_parents_generalise_the_chain():
    """SPEC §4.2 with a tree: options 1 and 2 both target the goal, option 3 targets option 2."""
    orc, m = make_oracle("pinball_empty", n_envs=1, n_options=3, seed=0, epsilon=0.0, enabled_mask=0b1110)
    orc.set_parents([0, 0, 0, 2])
    clf = np.zeros((4, 8), np.float32)
    clf[1] = disc_weights(0.8, 0.8, 0.12)            # I_1: small disc round the goal
    clf[2] = disc_weights(0.6, 0.8, 0.12)            # I_2: a different region, also leads to the goal
    clf[3] = disc_weights(0.35, 0.8, 0.15)           # I_3: leads into I_2
    W = np.zeros((4, 5, 1296), np.float32)
    st = sc_oracle.new_state(1, m)
    st["x"][0], st["y"][0] = 0.35, 0.8
    orc.step(st, W, clf, 0)
    assert st["option_id"][0] == 3
    st["x"][0], st["y"][0] = 0.58, 0.8               # inside I_2 = the target of option 3 -> success, 2 takes over
    G, n_k = orc.step(st, W, clf, 1)
    assert st["option_id"][0] == 2 and n_k.tolist() == [1, 0, 0, 1]
    assert G[3, st["action"][0], 0] == pytest.approx(-5.0 + HP["r_option_success"])
    st["x"][0], st["y"][0] = 0.8, 0.72               # inside I_1 only: not option 2's target (the goal is) -> 2 fails, 1 selected
    G, n_k = orc.step(st, W, clf, 2)
    assert st["option_id"][0] == 1 and G[2, st["action"][0], 0] == pytest.approx(-5.0)


def test_value_gated_entry_and_the_exit_rule_on_one_env():
    """SPEC §4.2 / §5 (round 5) by construction, one env (global id 0) on the empty map, epsilon = 0, reoffer_period = 4:
    (a) a candidate option whose value function promises LESS than the root's at s_next is declined: option_id = -k, qcache = the
        root's Q(s_next, .); while the env stays inside I_k it is offered k again only when (t + env id) % 4 == 0 — raising the
        option's weights above the root's at t = 1 changes nothing at t = 1, 2, 3 and makes it enter at t = 4;
    (b) an option that ends while the episode goes on (here: it leaves its initiation set) bootstraps from the ROOT's value of
        s_next: target = r + gamma * max_a Q_0(s_next, a), checked against float64 values."""
    orc, m = make_oracle("pinball_empty", n_envs=1, n_options=1, seed=0, epsilon=0.0, enabled_mask=0b10, gamma=0.9, r_option_success=0.0)
    assert orc.p.reoffer_period == 4 and orc.p.select_rule == 1 and orc.p.exit_rule == 2
    clf = np.zeros((2, 8), np.float32)
    clf[1] = disc_weights(0.3, 0.3, 0.12)            # I_1: a disc far from the goal (0.9, 0.9): leaving it is "fail", never "success"
    W = np.zeros((2, 5, 1296), np.float32)
    W[0, :, 0] = 2.0                                 # the root promises 2 everywhere (feature 0 = 1), the option 1
    W[1, :, 0] = 1.0
    st = sc_oracle.new_state(1, m)

    def park():                                      # back to the disc's centre, at rest (the no-op action keeps it there)
        st["x"][0], st["y"][0], st["vx"][0], st["vy"][0] = 0.3, 0.3, 0.0, 0.0

    park()
    orc.step(st, W, clf, 0)
    assert st["option_id"][0] == -1 and np.allclose(st["qcache"][:, 0], 2.0)          # (a) declined, the root's values cached
    W[1, :, 0] = 3.0                                 # the option now promises more ...
    for t in (1, 2, 3):
        park()
        orc.step(st, W, clf, t)
        assert st["option_id"][0] == -1, t                                            # ... but is not offered before t = 4
    park()
    orc.step(st, W, clf, 4)
    assert st["option_id"][0] == 1 and np.allclose(st["qcache"][:, 0], 3.0)           # offered again and entered
    # (b) throw the env out of I_1 within one step: the option fails, the episode goes on
    st["x"][0], st["y"][0], st["vx"][0], st["vy"][0] = 0.41, 0.3, 2.0, 0.0
    s = [st[k].copy() for k in ("x", "y", "vx", "vy")]
    G, n_k = orc.step(st, W, clf, 5)
    assert st["option_id"][0] == 0 and n_k.tolist() == [1, 1]
    a, r = int(st["action"][0]), float(st["reward"][0])
    phi_s = fourier_reference(*s)[0]
    sn = [st[k] for k in ("x", "y", "vx", "vy")]
    v0_next = float(np.max(W[0].astype(np.float64) @ fourier_reference(*sn)[0]))
    delta = r + 0.9 * v0_next - float(W[1, a].astype(np.float64) @ phi_s)            # bootstrapped from the ROOT (2), not 0 and not the option (3)
    assert np.allclose(G[1, a], delta * phi_s, rtol=1e-4, atol=1e-4)
    assert abs(v0_next - 2.0) < 1e-5


def test_update_count_floor_divides_small_batches_by_the_floor():
    """SPEC §5 apply: step = alpha / max(n_k, update_count_floor)."""
    orc, m = make_oracle("pinball_empty", n_envs=1, n_options=0, alpha=0.5, update_count_floor=4)
    W = np.zeros((1, 5, 1296), np.float32)
    G = np.zeros((1, 5, 1296), np.float32); G[0, 2, 0] = 8.0
    orc.apply(W, G, np.array([1], np.int32))
    assert W[0, 2, 0] == np.float32(0.5 / 4 * 8.0)                                     # scale_0 = 1; divided by the floor, not by n_k = 1
    orc.apply(W, G, np.array([16], np.int32))
    assert W[0, 2, 0] == np.float32(1.0 + 0.5 / 16 * 8.0)
