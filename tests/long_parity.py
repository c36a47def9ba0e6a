"""Long parity run against the CPU oracle (the checker): a learning rollout of many hundred step-batches, W / G / states compared
bit for bit — the fixed parity cases run 10-30 step-batches, which a rare race could survive.
   python tests/long_parity.py [steps] [envs] [options]        (tests/test_gpu_stress.py runs a 400-step-batch case of it)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [p for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")) if p not in sys.path]
import numpy as np


def run(steps=1500, n=2048, nopt=3, check_every=10):
    import sc_oracle
    from gpu_util import assert_state_equal, dev, make_pair, state_to_device
    from util import chain_classifiers, random_states, random_weights
    mask = sum(1 << k for k in range(1, nopt + 1))
    ctx, orc, m = make_pair("pinball_simple", n, n_options=nopt, seed=5, enabled_mask=mask, alpha=0.02, max_episode_steps=300)
    clf = chain_classifiers(m, nopt)
    st_o = sc_oracle.new_state(n, m)
    x, y, vx, vy = random_states(m, n, 3, vmax=1.0)
    st_o["x"][:], st_o["y"][:], st_o["vx"][:], st_o["vy"][:] = x, y, vx, vy
    W_o = random_weights(nopt + 1, 4, std=0.01)
    st_d, W_d, clf_d = state_to_device(st_o, ctx), dev(W_o.copy()), dev(clf)
    G_d, n_d = ctx.grad_buffers()
    goals = 0
    for t in range(steps):
        G, n_k = orc.step(st_o, W_o, clf, t)
        orc.apply(W_o, G, n_k)
        ctx.step(st_d, W_d.view(-1), clf_d.view(-1), mask, t)
        goals += int((st_o["done"] == 1).sum())
        if t % check_every == 0 or t == steps - 1:
            assert_state_equal(st_d, st_o, msg=f"t={t}")
            assert np.array_equal(G_d.cpu().numpy(), G), t
            assert np.array_equal(W_d.cpu().numpy(), W_o), t
    return goals, float(np.abs(W_o).max())


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    nopt = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    goals, wmax = run(steps, n, nopt)
    print(f"{steps} learning step-batches x {n} envs, root + {nopt} options: bit-identical to the oracle throughout "
          f"({goals} goals reached, |W|max {wmax:.2f})")
