"""Hand-computed known answers for SPEC §1.3 + property tests (hypothesis)."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from util import make_oracle, random_states

f32 = np.float32


def arr(v, dt=np.float32):
    return np.array([v], dt)


@pytest.fixture(scope="module")
def empty():
    return make_oracle("pinball_empty")


@pytest.fixture(scope="module")
def simple():
    return make_oracle("pinball_simple")


def step1(orc, x, y, vx, vy, a):
    X, Y, VX, VY = arr(x), arr(y), arr(vx), arr(vy)
    r, g = orc.pinball_step(X, Y, VX, VY, arr(a, np.uint8))
    return float(X[0]), float(Y[0]), float(VX[0]), float(VY[0]), float(r[0]), int(g[0])


def test_rest_stays_at_rest(empty):
    orc, _ = empty
    assert step1(orc, 0.5, 0.5, 0.0, 0.0, 4) == (0.5, 0.5, 0.0, 0.0, -1.0, 0)


def test_free_flight_acc_x_by_hand(empty):
    orc, m = empty
    h = f32(m.scalars[1])
    vx = f32(0.0) + f32(0.2)                       # impulse 1/5
    x = f32(0.5)
    for _ in range(20):                            # x = fma(vx, h, x): emulate in float64 then round once
        x = f32(np.float64(vx) * np.float64(h) + np.float64(x))
    want_vx = f32(vx * f32(0.995))
    got = step1(orc, 0.5, 0.5, 0.0, 0.0, 0)
    assert got == (float(x), 0.5, float(want_vx), 0.0, -5.0, 0)
    # other actions: signs and axes
    assert step1(orc, 0.5, 0.5, 0.0, 0.0, 2)[2] == -float(want_vx)
    assert step1(orc, 0.5, 0.5, 0.0, 0.0, 1)[3] == float(want_vx)
    assert step1(orc, 0.5, 0.5, 0.0, 0.0, 3)[3] == -float(want_vx)


def test_velocity_clip_at_two(empty):
    orc, _ = empty
    x, y, vx, vy, r, g = step1(orc, 0.5, 0.5, 1.95, 0.0, 0)
    assert vx == float(f32(2.0) * f32(0.995))


def test_bounce_off_right_wall_mirrors_vx(empty):
    orc, m = empty
    # right wall inner face at x = 0.99; ball radius 0.02 -> contact when x >= 0.97
    x, y, vx, vy, r, g = step1(orc, 0.9695, 0.5, 1.0, 0.25, 4)
    assert vx == pytest.approx(-0.995, abs=1e-6)       # mirrored about the vertical edge, then drag
    assert vy == pytest.approx(0.25 * 0.995, abs=1e-6)
    assert x < 0.9705 and r == -1.0 and g == 0


def test_corner_double_hit_reverses_velocity(empty):
    orc, _ = empty
    # moving diagonally into the top-right corner: two edges intercept in the same sub-step
    x, y, vx, vy, r, g = step1(orc, 0.9695, 0.9695, 1.0, 1.0, 4)
    assert vx == pytest.approx(-0.995, abs=1e-6) and vy == pytest.approx(-0.995, abs=1e-6)


def test_moving_away_is_not_a_collision(empty):
    orc, _ = empty
    # overlapping the right wall but already moving away: no reflection
    x, y, vx, vy, r, g = step1(orc, 0.9705, 0.5, -1.0, 0.0, 4)
    assert vx == pytest.approx(-0.995, abs=1e-6)


def test_goal_terminates_without_drag(empty):
    orc, m = empty
    tx, ty, tr = m.target
    x, y, vx, vy, r, g = step1(orc, tx - tr - 0.004, ty, 1.0, 0.0, 4)
    assert g == 1 and r == 10000.0
    assert vx == 1.0                                   # terminal sub-step: no drag applied
    assert (x - tx) ** 2 + (y - ty) ** 2 < tr * tr


def test_clamp_to_unit_square():
    orc, _ = make_oracle("pinball_empty")
    orc.p.n_edges = 0                                  # no walls: only the clamp keeps the ball inside
    x, y, vx, vy, r, g = step1(orc, 0.999, 0.001, 2.0, -2.0, 4)
    assert x == 1.0 and y == 0.0


@settings(max_examples=60, deadline=None)
@given(seed=st.integers(0, 10_000), a=st.integers(0, 4))
def test_properties_ball_stays_legal(seed, a):
    orc, m = make_oracle("pinball_simple")
    n = 32
    x, y, vx, vy = random_states(m, n, seed, vmax=2.0)
    act = np.full(n, a, np.uint8)
    for _ in range(5):
        r, g = orc.pinball_step(x, y, vx, vy, act)
        assert np.all((x >= 0) & (x <= 1) & (y >= 0) & (y <= 1))
        assert np.all(vx.astype(np.float64) ** 2 + vy.astype(np.float64) ** 2 <= 8.0 + 1e-4)
        assert set(np.unique(r)) <= {-1.0, -5.0, 10000.0}
        live = g == 0
        for xi, yi in zip(x[live], y[live]):
            assert not m.inside_obstacle(float(xi), float(yi))
        if not live.any():
            break
        x, y, vx, vy, act = x[live], y[live], vx[live], vy[live], act[live]


def test_speed_never_grows_by_reflection(simple):
    orc, m = simple
    x, y, vx, vy = random_states(m, 512, 11, vmax=1.5)
    s0 = np.hypot(vx, vy)
    orc.pinball_step(x, y, vx, vy, np.full(512, 4, np.uint8))
    assert np.all(np.hypot(vx, vy) <= s0 * (1 + 1e-5))
