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


def test_collect_examples_matches_select_then_harvest():
    """The device-side trigger (one pass: select, compact, gather) against its two-step definition on the oracle:
    entering envs in env order, each contributing its valid harvest rows (a prefix of the ages)."""
    n, steps, H = 400, 30, 8
    orc, m = make_oracle("pinball_simple", n_envs=n, n_options=2, seed=4, enabled_mask=0b110, max_episode_steps=20)
    orc.set_trace(H)
    st = sc_oracle.new_state(n, m)
    x, y, vx, vy = random_states(m, n, 31, vmax=1.0)
    st["x"][:], st["y"][:], st["vx"][:], st["vy"][:] = x, y, vx, vy
    W, clf = random_weights(3, 32, std=0.05), chain_classifiers(m, 2)
    cap = 120
    ex_xy, ex_lab, cnt = np.zeros((cap, 2), np.float32), np.zeros(cap, np.uint8), np.zeros(1, np.int32)
    prev = np.zeros(n, np.uint8)
    want_xy, want_lab, was_in = [], [], np.zeros(n, bool)
    for t in range(steps):
        G, n_k = orc.step(st, W, clf, t)
        orc.apply(W, G, n_k)
        orc.collect_examples(0b10, prev, 3, 4, ex_xy, ex_lab, cnt)        # envs ENTERING initiation set 1
        now_in = (orc.events & 2) != 0
        sel = np.nonzero(now_in & ~was_in)[0].astype(np.int32)
        was_in = now_in
        if len(sel):
            xy, lab = orc.harvest(sel, 3, 4)
            keep = lab.reshape(-1) != 255
            want_xy.append(xy.reshape(-1, 2)[keep]); want_lab.append(lab.reshape(-1)[keep])
    want_xy, want_lab = np.concatenate(want_xy), np.concatenate(want_lab)
    assert len(want_lab) > cap                                            # the buffer overflows: the tail is dropped
    assert cnt[0] == cap and np.array_equal(ex_xy, want_xy[:cap]) and np.array_equal(ex_lab, want_lab[:cap])
    assert np.array_equal(prev.astype(bool), was_in)


def test_gestating_option_learns_off_policy_and_is_never_selected():
    """SPEC §4.4 on the oracle: option 2 gestates — nobody runs it, yet VF 2 gets update items (the envs inside its
    initiation set), its successes are counted, and its classifier still serves as a membership test (events bit 2)."""
    n = 600
    orc, m = make_oracle("pinball_simple", n_envs=n, n_options=2, seed=6, enabled_mask=0b010)
    orc.set_gestation(0b100)
    orc.set_trace(8)
    st = sc_oracle.new_state(n, m)
    x, y, vx, vy = random_states(m, n, 33, vmax=1.0)
    st["x"][:], st["y"][:], st["vx"][:], st["vy"][:] = x, y, vx, vy
    W, clf = random_weights(3, 34, std=0.05), chain_classifiers(m, 2)
    W2_before = W[2].copy()
    n2 = 0
    for t in range(25):
        inside = orc.classifier_predict(st["x"].copy(), st["y"].copy(), clf[2]).astype(bool)
        G, n_k = orc.step(st, W, clf, t)
        orc.apply(W, G, n_k)
        assert not np.any(st["option_id"] == 2)                            # never selected
        assert n_k[2] == inside.sum()                                      # one off-policy item per env inside I_2
        n2 += int(n_k[2])
    assert n2 > 0 and not np.array_equal(W[2], W2_before)
    assert orc.gest_succ[2] > 0 and orc.gest_succ[0] == 0 and orc.gest_succ[1] == 0
    assert np.any(orc.events & 4)


@pytest.mark.gpu
def test_gestation_and_device_side_collect_bit_exact_on_gpu():
    import torch
    from gpu_util import assert_state_equal, dev, make_pair, state_to_device
    n, steps, H, nopt = 1500, 30, 16, 3
    enabled, gest = 0b0010, 0b1100                                        # option 1 runs, 2 and 3 gestate
    ctx, orc, m = make_pair("pinball_simple", n, n_options=nopt, seed=8, enabled_mask=enabled, max_episode_steps=30)
    orc.set_gestation(gest); orc.set_trace(H)
    ctx.set_trace_buffers(H)
    succ_d = ctx.set_gestation(gest)
    st_o = sc_oracle.new_state(n, m)
    x, y, vx, vy = random_states(m, n, 35, vmax=1.0)
    st_o["x"][:], st_o["y"][:], st_o["vx"][:], st_o["vy"][:] = x, y, vx, vy
    W_o, clf = random_weights(nopt + 1, 36, std=0.05), chain_classifiers(m, nopt)
    st_d, W_d, clf_d = state_to_device(st_o, ctx), dev(W_o.copy()), dev(clf)
    cap = 5000
    ex_xy_o, ex_lab_o, cnt_o, prev_o = np.zeros((cap, 2), np.float32), np.zeros(cap, np.uint8), np.zeros(1, np.int32), np.zeros(n, np.uint8)
    ex_xy_d, ex_lab_d = torch.zeros((cap, 2), device="cuda:0"), torch.zeros(cap, dtype=torch.uint8, device="cuda:0")
    cnt_d, prev_d = torch.zeros(1, dtype=torch.int32, device="cuda:0"), torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    gp = ctx.grad_buffers()
    for t in range(steps):
        G, n_k = orc.step(st_o, W_o, clf, t)
        orc.apply(W_o, G, n_k)
        ctx.step(st_d, W_d.view(-1), clf_d.view(-1), enabled, t)
        orc.collect_examples(0b100, prev_o, 6, 5, ex_xy_o, ex_lab_o, cnt_o)          # envs entering I_2 (gestating, known)
        ctx.collect_examples(0b100, prev_d, 6, 5, ex_xy_d.view(-1), ex_lab_d, cnt_d)
        assert np.array_equal(gp[1].cpu().numpy(), n_k), t
    torch.cuda.synchronize()
    assert_state_equal(st_d, st_o, msg="gestation rollout")
    assert np.array_equal(W_d.cpu().numpy(), W_o)
    assert np.array_equal(succ_d.cpu().numpy(), orc.gest_succ) and orc.gest_succ[2] + orc.gest_succ[3] > 0
    assert int(cnt_d.item()) == int(cnt_o[0]) > 0
    k = int(cnt_o[0])
    assert np.array_equal(ex_xy_d.cpu().numpy()[:k], ex_xy_o[:k]) and np.array_equal(ex_lab_d.cpu().numpy()[:k], ex_lab_o[:k])
    assert np.array_equal(prev_d.cpu().numpy(), prev_o)
    # the first collect with a goal trigger (no prev_in) on top of a non-empty buffer
    orc.collect_examples(0b1, None, 4, 4, ex_xy_o, ex_lab_o, cnt_o)
    ctx.collect_examples(0b1, None, 4, 4, ex_xy_d.view(-1), ex_lab_d, cnt_d)
    k = int(cnt_o[0])
    assert int(cnt_d.item()) == k and np.array_equal(ex_xy_d.cpu().numpy()[:k], ex_xy_o[:k])


@pytest.mark.gpu
def test_chain_skills_with_gestation():
    """The outer loop with a gestation period (SPEC §4.4): created options first learn off-policy, then get enabled by
    the success counters (or by the step limit), and are executed afterwards."""
    from skill_chaining_with_graphs_amd import SkillChainingAgent
    agent = SkillChainingAgent("pinball_empty", 8192, 2, seed=5, epsilon=1.0, alpha=1e-4, max_episode_steps=400)
    agent.enable_tracing(64)
    agent.domain.reset_random(seed=11, v_max=0.5)
    report = agent.chain_skills(steps_per_option=250, min_examples=2000, max_examples=20000, start_coverage=2.0,
                                gestation=200, gestation_steps=120)
    assert len(report) >= 1 and agent.gest_mask == 0, report
    assert all(0 < r["gestation_steps"] <= 120 for r in report), report
    assert all((agent.enabled_mask >> r["option"]) & 1 for r in report)
    for _ in range(5):
        agent.step_batch()
    assert int((agent.state.option_id == 1).sum()) > 0


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
    for i in range(150):
        agent.step_batch()
        agent.collect_examples(1, l_pos=24, l_neg=24)          # device-side trigger: no host round trip
        if i % 10 == 9:
            got = agent.examples_held(1)
            if got > 20000:
                break
    assert got > 2000, "random exploration never reached the goal"
    acc = agent.create_option(1)
    assert acc > 0.7      # random-walk positives and negatives overlap in space; the fit still separates them
    xy, lab = agent.examples(1)
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
def test_the_chain_is_not_worse_than_the_flat_learner_at_equal_env_steps():
    """The only external anchor this repo has (README.md:2 names the skill-chaining paper, whose claim is that chaining helps). With
    SPEC §4.2's value-gated entry and exit rule (round 5) goal arrivals with the discovered chain are 0.98-1.45 of the flat
    learner's over the same env-steps on ten seeds at 8192 envs (mean 1.08; profiles/r05_chain_evidence_gpu_a.txt) — rounds 1-4,
    with forced entry, ranged from 0.15 to 1.15 and collapsed for good whenever the options were cut while the root was still weak.
    Chaotic in the rounding, hence two seeds, a bound below the measured range, GPU only, no oracle."""
    import torch
    from skill_chaining_with_graphs_amd import SkillChainingAgent
    n, warm, after = 4096, 2500, 2000
    hp = dict(alpha=0.02, epsilon=0.05, gamma=0.99, max_episode_steps=2000, max_option_steps=200, update_count_floor=n // 16)

    def run(ag, steps):
        goals = torch.zeros((), device="cuda")
        for _ in range(steps):
            ag.step_batch()
            goals += (ag.state.done == 1).sum()
        return float(goals) / (steps * n)

    for seed in (2, 5):
        ag = SkillChainingAgent("pinball_simple", n, 3, seed=seed, **hp)
        ag.enable_tracing(64)
        run(ag, warm)
        t0 = ag.t
        report = ag.chain_skills(steps_per_option=400, min_examples=3000, max_examples=40000, start_coverage=0.9)
        disc = ag.t - t0
        chain = run(ag, after)
        flat_ag = SkillChainingAgent("pinball_simple", n, 0, seed=seed, **hp)
        run(flat_ag, warm + disc)
        flat = run(flat_ag, after)
        assert len(report) >= 1 and bool(torch.isfinite(ag.W).all())
        assert chain > 0.85 * flat and chain > 0.0003, (seed, chain, flat, report)
        assert int((ag.state.option_id < 0).sum()) > 0           # some envs stay out of an option they are inside of (SPEC §4.2)

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
    assert a.examples_held(2) == b.examples_held(2) > 0
    assert torch.equal(a.examples(2)[0], b.examples(2)[0]) and torch.equal(a.examples(2)[1], b.examples(2)[1])


@pytest.mark.gpu
def test_announced_trigger_gives_the_same_examples_in_one_launch():
    """scg_arm_collect: the step's commit rows leave the per-row example totals, the following collect skips its count
    launch. Two agents on the same seed, one announcing (the default of ctx.collect_examples) and one not, must hold
    identical example buffers, counts and prev_in after every step-batch — including a step whose collect is skipped
    (stale announcement: the next collect must fall back to counting itself)."""
    import torch
    from skill_chaining_with_graphs_amd import SkillChainingAgent
    from util import HP
    ags = []
    for _ in range(2):
        ag = SkillChainingAgent("pinball_simple", 3000, 2, seed=9, **dict(HP, max_episode_steps=60))
        ag.clf.copy_(torch.as_tensor(chain_classifiers(ag.map, 2), device="cuda:0"))
        ag.enable_option(1)
        ag.init_weights(std=1e-2)
        ag.domain.reset_random(seed=4, v_max=1.5)
        ag.enable_tracing(ring_len=32, max_examples=1 << 16)
        ags.append(ag)
    a, b = ags
    for t in range(40):
        for ag in ags:
            ag.step_batch()
        if t % 7 == 3:
            continue                                      # no collect after this step: a's announcement goes stale
        xy, lab, cnt, prev = a._ex_buffers(2)
        a.ctx.collect_examples(1 << 1, prev, 8, 8, xy.view(-1), lab, cnt, rearm=True)
        xy, lab, cnt, prev = b._ex_buffers(2)
        b.ctx.collect_examples(1 << 1, prev, 8, 8, xy.view(-1), lab, cnt, rearm=False)
    torch.cuda.synchronize()
    na, nb = a.examples_held(2), b.examples_held(2)
    assert na == nb and na > 100
    for ta, tb in zip(a._ex_buffers(2), b._ex_buffers(2)):
        assert torch.equal(ta, tb)
