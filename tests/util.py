"""Shared helpers for the tests (host side only)."""
import numpy as np

import sc_oracle
import skill_chaining_with_graphs_amd as scg
from skill_chaining_with_graphs_amd.core import fourier_scale_table

SCALE = fourier_scale_table()
HP = dict(gamma=0.99, alpha=1e-3, epsilon=0.1, r_option_success=100.0, max_episode_steps=60,
          max_option_steps=25)


def make_oracle(map_name, n_envs=1, n_options=0, seed=0, env_id_base=0, enabled_mask=0, n_threads=4, **hp):
    m = scg.load_map(map_name)
    kw = dict(HP)
    kw.update(hp)
    return sc_oracle.Oracle(m, SCALE, n_envs=n_envs, n_options=n_options, seed=seed, env_id_base=env_id_base,
                            enabled_mask=enabled_mask, n_threads=n_threads, **kw), m


def random_states(m, n, seed, vmax=2.0, near_walls=True):
    """Collision-free positions (some hugging obstacles) + velocities in [-vmax, vmax]."""
    rng = np.random.default_rng(seed)
    pos = m.sample_free(n, rng, margin=1.05 if near_walls else 2.0)
    v = rng.uniform(-vmax, vmax, (n, 2)).astype(np.float32)
    return pos[:, 0].copy(), pos[:, 1].copy(), v[:, 0].copy(), v[:, 1].copy()


def disc_weights(cx, cy, radius):
    uc, vc, r = 2 * cx - 1, 2 * cy - 1, 2 * radius
    w = np.zeros(8, np.float32)
    w[:6] = [r * r - uc * uc - vc * vc, 2 * uc, 2 * vc, -1.0, 0.0, -1.0]
    return w


def chain_classifiers(m, n_options):
    """A synthetic skill chain: option 1's initiation set is a disc round the goal, option k's a
    larger disc (so that I_k contains I_(k-1)) — stands in for fitted classifiers."""
    clf = np.zeros((n_options + 1, 8), np.float32)
    tx, ty, _ = m.target
    for k in range(1, n_options + 1):
        clf[k] = disc_weights(tx, ty, 0.18 + 0.17 * (k - 1))
    return clf


def random_weights(n_vf, seed, std=1e-2):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((n_vf, 5, 1296)) * std).astype(np.float32)


def fourier_reference(x, y, vx, vy):
    """float64 cos(pi c.s_hat) in canonical feature order."""
    s = np.stack([x, y, vx * 0.25 + 0.5, vy * 0.25 + 0.5], 1).astype(np.float64)
    idx = np.arange(1296)
    c = np.stack([(idx // 6 ** (3 - d)) % 6 for d in range(4)], 1).astype(np.float64)
    return np.cos(np.pi * s @ c.T)


def dense_map(n_side=6, size=0.035):
    """A synthetic map with many small square obstacles (16 + 4 n_side^2 edges): exercises the 4-word candidate
    masks (> 128 edges) and lanes with more than three candidate edges (overflow path of the HIP physics)."""
    lines = ["ball 0.015", "target 0.93 0.07 0.03", "start 0.07 0.93",
             "polygon 0.0 0.0 0.0 0.01 1.0 0.01 1.0 0.0", "polygon 0.0 0.0 0.01 0.0 0.01 1.0 0.0 1.0",
             "polygon 0.0 1.0 0.0 0.99 1.0 0.99 1.0 1.0", "polygon 1.0 1.0 0.99 1.0 0.99 0.0 1.0 0.0"]
    for i in range(n_side):
        for j in range(n_side):
            cx, cy = 0.16 + 0.136 * i, 0.16 + 0.136 * j
            lines.append(f"polygon {cx - size} {cy - size} {cx + size} {cy - size} {cx + size} {cy + size} {cx - size} {cy + size}")
    return scg.parse_map("\n".join(lines), "dense_synthetic")


def hub_map(n_spokes=12, r_in=0.035, r_out=0.30, half_width=0.006):
    """Thin spokes radiating from the centre (16 + 4 n_spokes edges): a ball near the hub has a dozen or more edges
    within reach at once — more than the (env, edge) pair form of the HIP physics takes per env (> 8: its per-lane
    loop) — and the balls around it have 3..8 (pair runs of every length, several groups of 64 pairs per wave)."""
    lines = ["ball 0.02", "target 0.9 0.1 0.04", "start 0.5 0.5 0.1 0.9 0.5 0.5",
             "polygon 0.0 0.0 0.0 0.01 1.0 0.01 1.0 0.0", "polygon 0.0 0.0 0.01 0.0 0.01 1.0 0.0 1.0",
             "polygon 0.0 1.0 0.0 0.99 1.0 0.99 1.0 1.0", "polygon 1.0 1.0 0.99 1.0 0.99 0.0 1.0 0.0"]
    for k in range(n_spokes):
        t = 2 * np.pi * (k + 0.5) / n_spokes
        c, s_ = np.cos(t), np.sin(t)
        pts = [(0.5 + r_in * c - half_width * s_, 0.5 + r_in * s_ + half_width * c),
               (0.5 + r_out * c - half_width * s_, 0.5 + r_out * s_ + half_width * c),
               (0.5 + r_out * c + half_width * s_, 0.5 + r_out * s_ - half_width * c),
               (0.5 + r_in * c + half_width * s_, 0.5 + r_in * s_ - half_width * c)]
        lines.append("polygon " + " ".join(f"{px:.6f} {py:.6f}" for px, py in pts))
    return scg.parse_map("\n".join(lines), f"hub_{n_spokes}")
