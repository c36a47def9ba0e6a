"""Generates tests/golden/*.npz — inputs and expected outputs produced by the in-repo CPU oracle
(oracle/sc_oracle.c, SPEC.md). The upstream reference holds no code or fixtures, so these
vectors pin the oracle against regressions and give the HIP path a committed, box-independent target;
they are NOT reference outputs. Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import sc_oracle  # noqa: E402
from util import chain_classifiers, make_oracle, random_states, random_weights  # noqa: E402


def physics_case():
    orc, m = make_oracle("pinball_maze")
    x, y, vx, vy = random_states(m, 96, 101, vmax=2.5)
    tx, ty, tr = m.target
    x[:8] = tx - tr - 0.003; y[:8] = ty; vx[:8] = 1.0; vy[:8] = 0.0
    act = np.random.default_rng(102).integers(0, 5, 96).astype(np.uint8)
    inp = dict(x0=x.copy(), y0=y.copy(), vx0=vx.copy(), vy0=vy.copy(), action=act)
    rs, gs = [], []
    for _ in range(3):
        r, g = orc.pinball_step(x, y, vx, vy, act)
        rs.append(r); gs.append(g)
    return dict(inp, x=x, y=y, vx=vx, vy=vy, reward=np.stack(rs), goal=np.stack(gs))


def value_case():
    orc, m = make_oracle("pinball_simple")
    x, y, vx, vy = random_states(m, 12, 103, vmax=2.0)
    W = random_weights(1, 104, std=1.0)[0]
    return dict(x=x, y=y, vx=vx, vy=vy, W=W, phi=orc.features(x, y, vx, vy), q=orc.q_values(x, y, vx, vy, W))


def step_case():
    n, n_opt, mask, steps = 700, 2, 0b110, 4
    orc, m = make_oracle("pinball_simple", n_envs=n, n_options=n_opt, seed=77, enabled_mask=mask)
    clf = chain_classifiers(m, n_opt)
    st = sc_oracle.new_state(n, m)
    x, y, vx, vy = random_states(m, n, 105, vmax=1.0)
    st["x"][:], st["y"][:], st["vx"][:], st["vy"][:] = x, y, vx, vy
    st["ep_steps"][:] = np.random.default_rng(106).integers(0, 59, n)
    W = random_weights(n_opt + 1, 107, std=0.05)
    out = dict(n=n, n_options=n_opt, mask=mask, steps=steps, seed=77, clf=clf, W0=W.copy(),
               **{k + "0": v.copy() for k, v in st.items()})
    nks = []
    for t in range(steps):
        G, n_k = orc.step(st, W, clf, t)
        orc.apply(W, G, n_k)
        nks.append(n_k.copy())
    out.update(W=W, n_k=np.stack(nks), **st)
    return out


def fit_case():
    orc, _ = make_oracle("pinball_empty")
    rng = np.random.default_rng(108)
    xy = rng.random((700, 2)).astype(np.float32)
    lab = (((xy[:, 0] - 0.4) ** 2 + (xy[:, 1] - 0.6) ** 2) < 0.3 ** 2).astype(np.uint8)
    off = np.array([0, 500, 700], np.int32)
    w = np.zeros((2, 8), np.float32)
    orc.fit_initiation(xy, lab, off, w, iters=120, lr=3.0, l2=1e-4)
    return dict(xy=xy, label=lab, offsets=off, iters=120, lr=3.0, l2=1e-4, w=w)


if __name__ == "__main__":
    for name, fn in (("physics", physics_case), ("values", value_case), ("step", step_case), ("fit", fit_case)):
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **fn())
        print("wrote", name)
