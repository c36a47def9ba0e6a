"""Oracle pinned to known answers: published Philox4x32-10 vectors, float64 math, hand-computed physics.
(The upstream reference has no tests or fixtures — SURVEY.md §4 — so these KATs are the pin.)"""
import numpy as np
import pytest

from util import SCALE, fourier_reference, make_oracle, random_states


@pytest.fixture(scope="module")
def orc():
    return make_oracle("pinball_empty")[0]


# Random123 kat_vectors, philox4x32 10 rounds (Salmon et al., SC'11)
PHILOX_KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.mark.parametrize("ctr,key,want", PHILOX_KAT)
def test_philox_published_vectors(orc, ctr, key, want):
    assert tuple(orc.philox(ctr, key)) == want


def test_sincospi_accuracy_and_exact_points(orc):
    ts = np.concatenate([np.linspace(-3, 3, 4001), np.linspace(0, 1.25, 3001)]).astype(np.float32)
    err = 0.0
    for t in ts:
        c, s = orc.sincospi(float(t))
        err = max(err, abs(c - np.cos(np.pi * float(t))), abs(s - np.sin(np.pi * float(t))))
    assert err < 3e-7
    assert orc.sincospi(0.0) == (1.0, 0.0)
    assert orc.sincospi(0.5) == (0.0, 1.0)
    assert orc.sincospi(1.0) == (-1.0, 0.0)
    assert orc.sincospi(1.5) == (0.0, -1.0)


def test_sigmoid_accuracy(orc):
    zs = np.linspace(-30, 30, 2001)
    got = np.array([orc.sigmoid(float(z)) for z in zs])
    assert np.max(np.abs(got - 1 / (1 + np.exp(-zs)))) < 2e-7
    assert orc.sigmoid(0.0) == 0.5
    assert orc.sigmoid(200.0) == pytest.approx(1.0) and orc.sigmoid(-200.0) < 1e-37


def _q_numpy_f32(x, y, vx, vy, W):
    """SPEC §3.1 restated with numpy float32 scalars (float64 fma emulation: exact product, one rounding)."""
    f32 = np.float32

    def fma(a, b, c):                     # binary32 fma via float64: the product of two floats is exact in double,
        return f32(np.float64(a) * np.float64(b) + np.float64(c))   # the sum rounds once more -> may double-round

    orc = make_oracle("pinball_empty")[0]
    sh = [x, y, fma(vx, f32(0.25), f32(0.5)), fma(vy, f32(0.25), f32(0.5))]
    Z = []
    for d in range(4):
        c, s = orc.sincospi(float(sh[d]))
        z = [(f32(1), f32(0)), (f32(c), f32(s))]
        for _ in range(4):
            a, b = z[-1], z[1]
            z.append((fma(-a[1], b[1], f32(a[0] * b[0])), fma(a[0], b[1], f32(a[1] * b[0]))))
        Z.append(z)

    def cm(a, b):
        return (fma(-a[1], b[1], f32(a[0] * b[0])), fma(a[0], b[1], f32(a[1] * b[0])))

    def table(zrow, zcol):                # SPEC §3: row 0 = powers of zcol, row i = row i - 1 times zrow^1
        t = [[None] * 6 for _ in range(6)]
        for j in range(6):
            t[0][j] = zcol[j]
            for i in range(1, 6):
                t[i][j] = cm(t[i - 1][j], zrow[1])
        return [t[i][j] for i in range(6) for j in range(6)]

    AB, CD = table(Z[0], Z[1]), table(Z[2], Z[3])
    out = []
    for a in range(5):
        q = [[f32(0), f32(0)] for _ in range(4)]
        for c12 in range(36):
            tre = tim = f32(0)
            for kb in range(9):
                for g in range(4):
                    c34 = 9 * g + kb
                    w = W[a, c12 * 36 + c34]
                    tre = fma(w, CD[c34][0], tre)
                    tim = fma(w, CD[c34][1], tim)
            grp = ((36 * a + c12) % 16) // 4
            q[grp][0] = fma(tre, AB[c12][0], q[grp][0])
            q[grp][1] = fma(tim, -AB[c12][1], q[grp][1])
        u = [f32(q[g][0] + q[g][1]) for g in range(4)]
        out.append(f32(f32(u[0] + u[1]) + f32(u[2] + u[3])))
    return np.array(out, np.float32)


def test_q_value_order_matches_an_independent_restatement(orc):
    """SPEC §3.1 (two contractions, c34 = 9 g + kb order, four row-group chains, fixed tree) restated in numpy:
    the C oracle must agree to the last bit except where the float64-emulated fma double-rounds (rare)."""
    m = make_oracle("pinball_simple")[1]
    x, y, vx, vy = random_states(m, 6, 11)
    W = (np.random.default_rng(5).standard_normal((5, 1296)) * 0.1).astype(np.float32)
    q = orc.q_values(x, y, vx, vy, W)
    exact = total = 0
    for e in range(6):
        ref = _q_numpy_f32(x[e], y[e], vx[e], vy[e], W)
        total += 5
        exact += int(np.sum(ref == q[:, e]))
        assert np.max(np.abs(ref - q[:, e])) < 1e-6
    assert exact >= total - 2


def test_features_match_float64_cosines(orc):
    m = make_oracle("pinball_simple")[1]
    x, y, vx, vy = random_states(m, 200, 1, vmax=2.8)
    phi = orc.features(x, y, vx, vy)
    ref = fourier_reference(x, y, vx, vy)
    assert phi.shape == (200, 1296)
    assert np.max(np.abs(phi - ref)) < 2e-6
    assert np.all(phi[:, 0] == 1.0)                 # c = 0 term
    assert np.max(np.abs(phi)) <= 1.0 + 1e-6


def test_q_values_match_float64_dot(orc):
    m = make_oracle("pinball_simple")[1]
    x, y, vx, vy = random_states(m, 64, 2)
    W = (np.random.default_rng(0).standard_normal((5, 1296))).astype(np.float32)
    q = orc.q_values(x, y, vx, vy, W)
    ref = (fourier_reference(x, y, vx, vy) @ W.astype(np.float64).T).T
    assert np.max(np.abs(q - ref)) < 2e-4           # 1296-term float32 sums of O(1) terms
    assert SCALE[0] == 1.0 and SCALE[1] == 1.0 and SCALE[7] == pytest.approx(1 / np.sqrt(2))


def test_classifier_predict_is_sign_of_quadratic(orc):
    from util import disc_weights
    rng = np.random.default_rng(3)
    x, y = rng.random(5000).astype(np.float32), rng.random(5000).astype(np.float32)
    w = disc_weights(0.6, 0.4, 0.25)
    got = orc.classifier_predict(x, y, w)
    d2 = (x.astype(np.float64) - 0.6) ** 2 + (y.astype(np.float64) - 0.4) ** 2
    far = np.abs(d2 - 0.25 ** 2) > 1e-5
    assert np.array_equal(got[far], (d2 < 0.25 ** 2)[far].astype(np.uint8))
