"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the in-repo
oracle): the oracle must keep reproducing them (CPU), and the HIP path must hit them bit for bit (GPU)."""
import os

import numpy as np
import pytest

import sc_oracle
from util import make_oracle

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(G, name + ".npz"), allow_pickle=False))


def test_oracle_reproduces_golden_physics():
    d = load("physics")
    orc, m = make_oracle("pinball_maze")
    x, y, vx, vy = (d[k].copy() for k in ("x0", "y0", "vx0", "vy0"))
    for t in range(3):
        r, g = orc.pinball_step(x, y, vx, vy, d["action"])
        assert np.array_equal(r, d["reward"][t]) and np.array_equal(g, d["goal"][t])
    for k, a in zip(("x", "y", "vx", "vy"), (x, y, vx, vy)):
        assert np.array_equal(a, d[k]), k
    assert d["goal"][0][:8].all()


def test_oracle_reproduces_golden_values_step_fit():
    d = load("values")
    orc, _ = make_oracle("pinball_simple")
    assert np.array_equal(orc.features(d["x"], d["y"], d["vx"], d["vy"]), d["phi"])
    assert np.array_equal(orc.q_values(d["x"], d["y"], d["vx"], d["vy"], d["W"]), d["q"])
    d = load("step")
    n, nopt = int(d["n"]), int(d["n_options"])
    orc, m = make_oracle("pinball_simple", n_envs=n, n_options=nopt, seed=int(d["seed"]), enabled_mask=int(d["mask"]))
    st = {k: d[k + "0"].copy() for k in sc_oracle.new_state(1, m)}
    W = d["W0"].copy()
    for t in range(int(d["steps"])):
        Gr, n_k = orc.step(st, W, d["clf"], t)
        orc.apply(W, Gr, n_k)
        assert np.array_equal(n_k, d["n_k"][t])
    assert np.array_equal(W, d["W"])
    for k in st:
        assert np.array_equal(st[k], d[k]), k
    d = load("fit")
    w = np.zeros((2, 8), np.float32)
    orc.fit_initiation(d["xy"], d["label"], d["offsets"], w, int(d["iters"]), float(d["lr"]), float(d["l2"]))
    assert np.array_equal(w, d["w"])


@pytest.mark.gpu
def test_hip_reproduces_golden():
    import torch
    from gpu_util import dev, make_pair, state_to_device
    d = load("physics")
    ctx, _, _ = make_pair("pinball_maze", 96)
    s = [dev(d[k].copy()) for k in ("x0", "y0", "vx0", "vy0")]
    for t in range(3):
        r, g = ctx.pinball_step(s, dev(d["action"]))
        assert np.array_equal(r.cpu().numpy(), d["reward"][t]) and np.array_equal(g.cpu().numpy(), d["goal"][t])
    for k, a in zip(("x", "y", "vx", "vy"), s):
        assert np.array_equal(a.cpu().numpy(), d[k]), k
    d = load("values")
    ctx, _, _ = make_pair("pinball_simple", 12)
    s = [dev(d[k]) for k in ("x", "y", "vx", "vy")]
    assert np.array_equal(ctx.features(s).cpu().numpy(), d["phi"])
    assert np.array_equal(ctx.q_values(s, dev(d["W"]).view(-1)).cpu().numpy(), d["q"])
    d = load("step")
    n, nopt, mask = int(d["n"]), int(d["n_options"]), int(d["mask"])
    ctx, _, m = make_pair("pinball_simple", n, n_options=nopt, seed=int(d["seed"]), enabled_mask=mask)
    if ctx.lib.scg_block_envs() == 256:      # the step fixture pins SPEC §5's geometry
        st = state_to_device({k: d[k + "0"] for k in sc_oracle.new_state(1, m)}, ctx)
        W, clf = dev(d["W0"].copy()), dev(d["clf"])
        _, n_d = ctx.grad_buffers()
        for t in range(int(d["steps"])):
            ctx.step(st, W.view(-1), clf.view(-1), mask, t)
            assert np.array_equal(n_d.cpu().numpy(), d["n_k"][t])
        assert np.array_equal(W.cpu().numpy(), d["W"])
        for k in sc_oracle.new_state(1, m):
            assert np.array_equal(getattr(st, k).cpu().numpy(), d[k]), k
    d = load("fit")
    ctx, _, _ = make_pair("pinball_empty", 8)
    w = torch.zeros((2, 8), device="cuda:0")
    ctx.fit_initiation(dev(d["xy"]).view(-1), dev(d["label"]), dev(d["offsets"]), w.view(-1), int(d["iters"]),
                       float(d["lr"]), float(d["l2"]))
    assert np.array_equal(w.cpu().numpy(), d["w"])
