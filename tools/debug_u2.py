"""Debug: where does G of scg_q_update differ from the oracle? (run from the repo root on a GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from gpu_util import dev, make_pair
from util import random_states, random_weights
for n in (1, 5, 40):
    k = 0
    ctx, orc, m = make_pair("pinball_simple", 700, n_options=2)
    x, y, vx, vy = random_states(m, n, 6); xn, yn, vxn, vyn = random_states(m, n, 7)
    rng = np.random.default_rng(8)
    act = rng.integers(0, 5, n).astype(np.uint8)
    r = rng.choice([-1.0, -5.0, 10000.0], n).astype(np.float32)
    cont = np.where(rng.random(n) < 0.2, 0.0, 0.99).astype(np.float32)
    W = random_weights(3, 9, std=0.5)
    G_o, cnt = orc.q_update_grad((x, y, vx, vy), act, r, cont, (xn, yn, vxn, vyn), W[k])
    W_d = dev(W.copy()); G_d, n_d = ctx.grad_buffers()
    ctx.q_update(k, [dev(a) for a in (x, y, vx, vy)], dev(act), dev(r), dev(cont), [dev(a) for a in (xn, yn, vxn, vyn)], W_d.view(-1), apply=False)
    torch.cuda.synchronize()
    G = G_d[k].cpu().numpy().reshape(5, 36, 36); Go = G_o.reshape(5, 36, 36)
    bad = G != Go
    print(f"n={n} actions {act.tolist()[:10]} n_k {n_d.cpu().numpy().tolist()} differing entries {int(bad.sum())} of {bad.size}")
    for a in range(5):
        if bad[a].any():
            rows = np.where(bad[a].any(1))[0]; cols = np.where(bad[a].any(0))[0]
            print(f"  action {a}: rows(c12) {rows.min()}..{rows.max()} ({len(rows)}) cols(c34) {cols.min()}..{cols.max()} ({len(cols)})  sample gpu {G[a][bad[a]][:3]} oracle {Go[a][bad[a]][:3]}")
            tiles = [(mi, ni) for mi in range(3) for ni in range(3) if bad[a][16*mi:16*mi+16, 16*ni:16*ni+16].any()]
            print("   tiles (mi, ni) with differences:", tiles)
