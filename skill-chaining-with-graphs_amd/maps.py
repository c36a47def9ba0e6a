"""Pinball obstacle maps (SPEC.md §1.1): text format, loader, validator, edge table.

The upstream reference ships no maps (it ships no files but README.md); the maps under `maps/` are
authored by this build. Host-side data preparation only — nothing here is on the per-step path."""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np

MAPS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "maps")
MAX_EDGES = 256


class MapError(ValueError):
    pass


def _f32(v) -> np.ndarray:
    return np.asarray(v, dtype=np.float32)


@dataclass
class PinballMap:
    radius: float
    target: Sequence[float]                 # (tx, ty, tr)
    starts: np.ndarray                      # [n_starts, 2] float32
    polygons: List[np.ndarray] = field(default_factory=list)   # each [k, 2] float32
    name: str = ""

    # ---------------------------------------------------------------- derived tables (SPEC §1.1)
    @property
    def edges(self) -> np.ndarray:
        """[n_edges, 8] float32: x0, y0, ex, ey, inv_len2, ux, uy, 0 — float64 math on the binary32
        vertices, each field rounded to binary32 once."""
        rows = []
        for poly in self.polygons:
            p = poly.astype(np.float64)
            for i in range(len(p)):
                p0, p1 = p[i], p[(i + 1) % len(p)]
                e = _f32(p1 - p0).astype(np.float64)      # ex, ey are themselves binary32 fields
                l2 = e[0] * e[0] + e[1] * e[1]
                if l2 <= 0.0:
                    raise MapError(f"degenerate edge in polygon of map {self.name!r}")
                ln = np.sqrt(l2)
                rows.append([p0[0], p0[1], e[0], e[1], 1.0 / l2, e[0] / ln, e[1] / ln, 0.0])
        out = _f32(rows).reshape(-1, 8)
        if len(out) > MAX_EDGES:
            raise MapError(f"map {self.name!r} has {len(out)} edges (max {MAX_EDGES})")
        return np.ascontiguousarray(out)

    @property
    def scalars(self) -> np.ndarray:
        """[R, hstep, R2, TX, TY, TR2] float32."""
        R = float(np.float32(self.radius))
        tx, ty, tr = (float(np.float32(v)) for v in self.target)
        return _f32([R, R / 20.0, R * R, tx, ty, tr * tr])

    @property
    def n_edges(self) -> int:
        return int(sum(len(p) for p in self.polygons))

    # ---------------------------------------------------------------- geometry helpers (host, float64)
    def distance_to_obstacles(self, x: float, y: float) -> float:
        """Distance from a point to the nearest edge (inf for an empty map)."""
        E = self.edges.astype(np.float64)
        if len(E) == 0:
            return float("inf")
        d = np.array([x, y]) - E[:, 0:2]
        t = np.clip((d * E[:, 2:4]).sum(1) * E[:, 4], 0.0, 1.0)
        c = E[:, 0:2] + E[:, 2:4] * t[:, None]
        return float(np.sqrt(((c - [x, y]) ** 2).sum(1)).min())

    def inside_obstacle(self, x: float, y: float) -> bool:
        for poly in self.polygons:
            p = poly.astype(np.float64)
            inside = False
            j = len(p) - 1
            for i in range(len(p)):
                if (p[i, 1] > y) != (p[j, 1] > y):
                    xi = (p[j, 0] - p[i, 0]) * (y - p[i, 1]) / (p[j, 1] - p[i, 1]) + p[i, 0]
                    if x < xi:
                        inside = not inside
                j = i
            if inside:
                return True
        return False

    def is_free(self, x: float, y: float, margin: float = 1.0) -> bool:
        """True if a ball centred at (x, y) overlaps no obstacle (with `margin` radii of clearance)."""
        if not (0.0 <= x <= 1.0 and 0.0 <= y <= 1.0):
            return False
        if self.inside_obstacle(x, y):
            return False
        return self.distance_to_obstacles(x, y) > margin * float(self.radius)

    def validate(self) -> None:
        if not (0.0 < self.radius < 0.25):
            raise MapError("ball radius out of range (0, 0.25)")
        tx, ty, tr = self.target
        if tr <= 0:
            raise MapError("target radius must be positive")
        if len(self.starts) < 1:
            raise MapError("need at least one start position")
        for poly in self.polygons:
            if len(poly) < 3:
                raise MapError("polygon with fewer than 3 vertices")
        _ = self.edges
        if self.inside_obstacle(tx, ty):
            raise MapError("target centre lies inside an obstacle")
        for sx, sy in self.starts:
            if not self.is_free(float(sx), float(sy)):
                raise MapError(f"start ({sx}, {sy}) overlaps an obstacle")
            if (sx - tx) ** 2 + (sy - ty) ** 2 < tr * tr:
                raise MapError(f"start ({sx}, {sy}) lies inside the target")

    def free_mask(self, pts: np.ndarray, margin: float = 1.0) -> np.ndarray:
        """Vectorised is_free for [n,2] points (host, float64)."""
        p = np.asarray(pts, np.float64)
        ok = (p[:, 0] >= 0) & (p[:, 0] <= 1) & (p[:, 1] >= 0) & (p[:, 1] <= 1)
        E = self.edges.astype(np.float64)
        if len(E):
            d = p[:, None, :] - E[None, :, 0:2]
            t = np.clip((d * E[None, :, 2:4]).sum(2) * E[None, :, 4], 0.0, 1.0)
            c = E[None, :, 0:2] + E[None, :, 2:4] * t[..., None]
            dist = np.sqrt(((c - p[:, None, :]) ** 2).sum(2)).min(1)
            ok &= dist > margin * float(self.radius)
        for poly in self.polygons:
            q = poly.astype(np.float64)
            inside = np.zeros(len(p), bool)
            j = len(q) - 1
            for i in range(len(q)):
                cond = (q[i, 1] > p[:, 1]) != (q[j, 1] > p[:, 1])
                with np.errstate(divide="ignore", invalid="ignore"):
                    xi = (q[j, 0] - q[i, 0]) * (p[:, 1] - q[i, 1]) / (q[j, 1] - q[i, 1]) + q[i, 0]
                inside ^= cond & (p[:, 0] < xi)
                j = i
            ok &= ~inside
        return ok

    def sample_free(self, n: int, rng: np.random.Generator, margin: float = 1.5) -> np.ndarray:
        """n collision-free positions outside the target, [n, 2] float32 (rejection sampling, host)."""
        tx, ty, tr = self.target
        out = np.empty((0, 2), np.float32)
        while len(out) < n:
            c = rng.random((max(256, 2 * (n - len(out))), 2)).astype(np.float32)
            keep = self.free_mask(c, margin)
            keep &= (c[:, 0] - tx) ** 2 + (c[:, 1] - ty) ** 2 >= (tr + self.radius) ** 2
            out = np.concatenate([out, c[keep]])
        return np.ascontiguousarray(out[:n])


def parse_map(text: str, name: str = "") -> PinballMap:
    radius = None
    target = None
    starts: List[List[float]] = []
    polys: List[np.ndarray] = []
    for ln_no, raw in enumerate(text.splitlines(), 1):
        line = raw.split("#", 1)[0].strip()
        if not line:
            continue
        tok = line.split()
        key, vals = tok[0].lower(), tok[1:]
        try:
            nums = [float(v) for v in vals]
        except ValueError as e:
            raise MapError(f"{name}:{ln_no}: not a number: {e}") from None
        if key == "ball":
            if len(nums) != 1:
                raise MapError(f"{name}:{ln_no}: ball takes one number")
            radius = nums[0]
        elif key == "target":
            if len(nums) != 3:
                raise MapError(f"{name}:{ln_no}: target takes three numbers")
            target = nums
        elif key == "start":
            if len(nums) < 2 or len(nums) % 2:
                raise MapError(f"{name}:{ln_no}: start takes an even number of coordinates")
            starts += [nums[i:i + 2] for i in range(0, len(nums), 2)]
        elif key == "polygon":
            if len(nums) < 6 or len(nums) % 2:
                raise MapError(f"{name}:{ln_no}: polygon needs >= 3 vertices")
            polys.append(_f32(nums).reshape(-1, 2))
        else:
            raise MapError(f"{name}:{ln_no}: unknown record {key!r}")
    if radius is None or target is None or not starts:
        raise MapError(f"{name}: ball, target and start records are required")
    m = PinballMap(radius=float(np.float32(radius)), target=[float(np.float32(v)) for v in target],
                   starts=_f32(starts).reshape(-1, 2), polygons=polys, name=name)
    m.validate()
    return m


def load_map(name_or_path: str) -> PinballMap:
    path = name_or_path
    if not os.path.exists(path):
        path = os.path.join(MAPS_DIR, name_or_path if name_or_path.endswith(".cfg") else name_or_path + ".cfg")
    if not os.path.exists(path):
        raise MapError(f"no such map: {name_or_path}")
    with open(path) as f:
        return parse_map(f.read(), os.path.basename(path))


def available_maps() -> List[str]:
    return sorted(f[:-4] for f in os.listdir(MAPS_DIR) if f.endswith(".cfg"))
