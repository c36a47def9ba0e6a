import numpy as np
import pytest


def test_shipped_maps_load_and_validate(pkg):
    for name in pkg.available_maps():
        m = pkg.load_map(name)
        E = m.edges
        assert E.dtype == np.float32 and E.shape == (m.n_edges, 8) and m.n_edges <= 256
        assert np.allclose(E[:, 5] ** 2 + E[:, 6] ** 2, 1.0, atol=1e-6)
        assert np.allclose(E[:, 4] * (E[:, 2] ** 2 + E[:, 3] ** 2), 1.0, atol=1e-6)
        for sx, sy in m.starts:
            assert m.is_free(float(sx), float(sy))
    assert pkg.load_map("pinball_maze").n_edges > 64      # exercises the second candidate-mask word


@pytest.mark.parametrize("text,msg", [
    ("target 0.5 0.5 0.1\nstart 0.1 0.1", "required"),
    ("ball 0.02\ntarget 0.5 0.5 0.1\nstart 0.1", "even number"),
    ("ball 0.02\ntarget 0.5 0.5 0.1\nstart 0.1 0.1\npolygon 0 0 1 1", ">= 3"),
    ("ball 0.02\ntarget 0.5 0.5 0.1\nstart 0.1 0.1\nwall 0 0 1 1", "unknown record"),
    ("ball 0.02\ntarget 0.5 0.5 0.1\nstart 0.5 0.5", "inside the target"),
    ("ball 0.02\ntarget 0.9 0.9 0.05\nstart 0.3 0.3\npolygon 0.2 0.2 0.4 0.2 0.4 0.4 0.2 0.4", "overlaps"),
    ("ball x", "not a number"),
])
def test_parser_rejects_bad_maps(pkg, text, msg):
    with pytest.raises(pkg.MapError, match=msg):
        pkg.parse_map(text, "t")


def test_shard_range_partitions(pkg):
    for n, w in ((65536, 8), (10, 3), (5, 8), (524288, 8)):
        r = [pkg.shard_range(n, i, w) for i in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert max(hi - lo for lo, hi in r) - min(hi - lo for lo, hi in r) <= 1
    with pytest.raises(ValueError):
        pkg.shard_range(4, 4, 4)
