"""SPEC §7 (outer-loop support): trajectory ring, events, harvest — oracle semantics on CPU, bit-exact parity and
an end-to-end option discovery on the GPU."""
import numpy as np
import pytest

import sc_oracle
from util import chain_classifiers, make_oracle, random_states, random_weights


def _run_oracle(n, steps, ring_len, n_options=2, mask=0b110, seed=4):
    orc, m = make_oracle("pinball_simple", n_envs=n, n_options=n_options, seed=seed, enabled_mask=mask,
                         max_episode_steps=20)
    orc.set_trace(ring_len)
    st = sc_oracle.new_state(n, m)
    x, y, vx, vy = random_states(m, n, 31, vmax=1.0)
    st["x"][:], st["y"][:], st["vx"][:], st["vy"][:] = x, y, vx, vy
    W = random_weights(n_options + 1, 32, std=0.05)
    clf = chain_classifiers(m, n_options)
    hist = []
    for t in range(steps):
        pre = (st["x"].copy(), st["y"].copy(), st["ep_steps"].copy())
        G, n_k = orc.step(st, W, clf, t)
        orc.apply(W, G, n_k)
        hist.append(pre)
    return orc, m, st, W, clf, hist


def test_ring_and_events_record_what_happened():
    n, steps, H = 300, 26, 8
    orc, m, st, W, clf, hist = _run_oracle(n, steps, H)
    x0, y0, ep0 = hist[-1]
    rows = ep0 & (H - 1)
    assert np.array_equal(orc.ring_x[rows, np.arange(n)], x0) and np.array_equal(orc.ring_y[rows, np.arange(n)], y0)
    assert np.array_equal(orc.ev_len, ep0 + 1)
    assert np.array_equal((orc.events & 1).astype(bool), st["done"] == 1)
    # in-set bits agree with the classifier at the post-physics position of live envs
    live = st["done"] == 0
    for k in (1, 2):
        want = orc.classifier_predict(st["x"][live].copy(), st["y"][live].copy(), clf[k]).astype(bool)
        assert np.array_equal(((orc.events[live] >> k) & 1).astype(bool), want)


def test_harvest_walks_back_through_the_episode():
    n, steps, H = 64, 13, 8
    orc, m, st, W, clf, hist = _run_oracle(n, steps, H)
    sel = np.arange(0, n, 3, dtype=np.int32)
    xy, lab = orc.harvest(sel, 3, 4)
    for si, e in enumerate(sel):
        for j in range(7):
            idx = orc.ev_len[e] - 1 - j
            if idx < 0 or j >= H:
                assert lab[si, j] == 255
                continue
            assert lab[si, j] == (1 if j < 3 else 0)
            # the state recorded (steps-1-j) batches ago, provided the env has not been reset since
            xs, ys, eps = hist[steps - 1 - j]
            assert eps[e] == idx and xy[si, j, 0] == xs[e] and xy[si, j, 1] == ys[e]


@pytest.mark.gpu
def test_trace_and_harvest_bit_exact_on_gpu():
    import torch
    from gpu_util import dev, make_pair, state_to_device
    n, steps, H, nopt, mask = 1500, 24, 16, 2, 0b110
    ctx, orc, m = make_pair("pinball_simple", n, n_options=nopt, seed=4, enabled_mask=mask, max_episode_steps=20)
    orc.set_trace(H)
    ring_x, ring_y, events, ev_len = ctx.set_trace_buffers(H)
    st_o = sc_oracle.new_state(n, m)
    x, y, vx, vy = random_states(m, n, 31, vmax=1.0)
    st_o["x"][:], st_o["y"][:], st_o["vx"][:], st_o["vy"][:] = x, y, vx, vy
    W_o = random_weights(nopt + 1, 32, std=0.05)
    clf = chain_classifiers(m, nopt)
    st_d, W_d, clf_d = state_to_device(st_o, ctx), dev(W_o.copy()), dev(clf)
    for t in range(steps):
        G, n_k = orc.step(st_o, W_o, clf, t)
        orc.apply(W_o, G, n_k)
        ctx.step(st_d, W_d.view(-1), clf_d.view(-1), mask, t)
        assert np.array_equal(events.cpu().numpy(), orc.events) and np.array_equal(ev_len.cpu().numpy(), orc.ev_len)
    assert np.array_equal(ring_x.cpu().numpy(), orc.ring_x) and np.array_equal(ring_y.cpu().numpy(), orc.ring_y)
    sel = np.union1d(np.nonzero(orc.events & 6)[0], np.arange(0, n, 7)).astype(np.int32)   # in-set envs + a spread
    xy_o, lab_o = orc.harvest(sel, 5, 6)
    xy_d, lab_d = ctx.harvest(dev(sel), 5, 6)
    assert np.array_equal(xy_d.cpu().numpy(), xy_o) and np.array_equal(lab_d.cpu().numpy(), lab_o)
    assert np.array_equal(W_d.cpu().numpy(), W_o)


@pytest.mark.gpu
def test_discover_first_option_end_to_end():
    """Skill chaining's first link, on the GPU only: run the root policy until envs reach the goal, harvest
    their trajectories from the ring, fit initiation set 1, enable option 1, and see envs execute it."""
    import torch
    from skill_chaining_with_graphs_amd import SkillChainingAgent
    agent = SkillChainingAgent("pinball_empty", 8192, 1, seed=3, epsilon=1.0, alpha=1e-4, max_episode_steps=400)
    agent.enable_tracing(64)
    agent.domain.reset_random(seed=9, v_max=0.5)
    got = 0
    for _ in range(150):
        agent.step_batch()
        got = agent.collect_examples(1, l_pos=24, l_neg=24)
        if got > 20000:
            break
    assert got > 2000, "random exploration never reached the goal"
    acc = agent.create_option(1)
    assert acc > 0.7      # random-walk positives and negatives overlap in space; the fit still separates them
    xy, lab = agent._examples[1]
    tx, ty, _ = agent.map.target
    d = torch.hypot(xy[:, 0] - tx, xy[:, 1] - ty)
    assert float(d[lab == 1].mean()) < float(d[lab == 0].mean())       # positives sit nearer the goal
    for _ in range(5):
        agent.step_batch()
    assert int((agent.state.option_id == 1).sum()) > 0


@pytest.mark.gpu
def test_chain_skills_builds_a_chain():
    """The whole outer loop (GPU only): options are created one after another, each chaining to the one before,
    until the start state is covered or the option slots run out; created options get executed."""
    import torch
    from skill_chaining_with_graphs_amd import SkillChainingAgent
    agent = SkillChainingAgent("pinball_empty", 8192, 3, seed=5, epsilon=1.0, alpha=1e-4, max_episode_steps=400)
    agent.enable_tracing(64)
    agent.domain.reset_random(seed=11, v_max=0.5)
    report = agent.chain_skills(steps_per_option=250, min_examples=2000, max_examples=20000, start_coverage=2.0)
    assert len(report) >= 2, report
    assert [r["option"] for r in report] == list(range(1, len(report) + 1))
    assert all(r["parent"] == r["option"] - 1 and r["accuracy"] > 0.55 for r in report), report
    for _ in range(5):
        agent.step_batch()
    ids = agent.state.option_id
    assert int((ids == 1).sum()) > 0 and int((ids == 2).sum()) > 0
    g = agent.skill_graph()
    assert all(g.nodes[r["option"]]["enabled"] for r in report)


@pytest.mark.gpu
def test_batched_q_learning_learns_pinball():
    """The hot path is a learner, not just a throughput kernel: with the root value function only, goal
    arrivals per env-step must rise several-fold within 2500 step-batches (GPU only; no oracle involved)."""
    import torch
    from skill_chaining_with_graphs_amd import SkillChainingAgent
    ag = SkillChainingAgent("pinball_simple", 4096, 0, seed=1, alpha=0.02, epsilon=0.05, gamma=0.99,
                            max_episode_steps=2000)
    rates = []
    for _ in range(5):
        goals = torch.zeros((), device="cuda")
        for _ in range(500):
            ag.step_batch()
            goals += (ag.state.done == 1).sum()
        rates.append(float(goals) / (500 * 4096))
    assert bool(torch.isfinite(ag.W).all())
    assert rates[-1] > 3 * rates[0] and rates[-1] > 0.0015, rates     # chaotic in the rounding: loose on purpose


@pytest.mark.gpu
def test_checkpoint_resume_continues_bit_identically(tmp_path):
    """SkillChainingAgent.save / load: a restored agent continues exactly like the uninterrupted one —
    env state, weights, option bookkeeping and the trace buffers (RNG streams are keyed by (seed, env id, t))."""
    import torch
    from skill_chaining_with_graphs_amd import SkillChainingAgent
    from util import HP, chain_classifiers, random_weights

    def make():
        ag = SkillChainingAgent("pinball_simple", 3000, 3, seed=21, **HP)
        ag.clf.copy_(torch.as_tensor(chain_classifiers(ag.map, 3), device="cuda:0"))
        ag.enabled_mask = 0b1110
        ag.set_option_parents([0, 0, 1, 1])                       # a small tree: 2 and 3 both chain to 1
        ag.W.copy_(torch.as_tensor(random_weights(4, 2, std=0.05), device="cuda:0"))
        ag.enable_tracing(32)
        return ag

    a = make()
    a.domain.reset_random(seed=4, v_max=1.0)
    for _ in range(9):
        a.step_batch()
        a.collect_examples(2, l_pos=8, l_neg=8)
    path = str(tmp_path / "agent.pt")
    a.save(path)
    for _ in range(6):
        a.step_batch()
        a.collect_examples(2, l_pos=8, l_neg=8)
    b = make()
    b.load(path)
    assert b.t == 9 and b.enabled_mask == 0b1110 and b.ctx.parents.tolist() == [0, 0, 1, 1]
    for _ in range(6):
        b.step_batch()
        b.collect_examples(2, l_pos=8, l_neg=8)
    torch.cuda.synchronize()
    for f in SkillChainingAgent._STATE_FIELDS:
        assert torch.equal(getattr(a.state, f), getattr(b.state, f)), f
    assert torch.equal(a.W, b.W) and a.t == b.t
    for ta, tb in zip(a.trace, b.trace):
        assert torch.equal(ta, tb)
    assert torch.equal(a._examples[2][0], b._examples[2][0]) and torch.equal(a._examples[2][1], b._examples[2][1])
