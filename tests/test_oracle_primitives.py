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


def test_wave_order_is_a_bijection(orc):
    seen = [orc.feature_index(l, j) for l in range(64) for j in range(21)]
    valid = [f for f in seen if f >= 0]
    assert sorted(valid) == list(range(1296))
    assert seen.count(-1) == 64 * 21 - 1296
    assert orc.feature_index(0, 0) == 0 and orc.feature_index(33, 0) == 4 * 36 + 1      # lane 33: upper half, col 1
    assert orc.feature_index(2, 5) == (8 + 1) * 36 + 2 and orc.feature_index(40, 17) == 35 * 36 + 8
    assert orc.feature_index(5, 18) == 1 * 36 + 32 + 1 and orc.feature_index(16, 20) == -1


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
