"""The C-ABI library loads here (no GPU) and exports every symbol include/scg_abi.h declares;
error behaviour that needs no device is checked too. No compute call is made."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "scg_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(scg_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree(pkg):
    assert declared_symbols() == sorted(pkg.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.scg_abi_version() == 5
    assert lib.scg_strerror(0) == b"ok" and b"invalid" in lib.scg_strerror(-1)


def test_create_rejects_bad_config_without_touching_a_device(pkg):
    from skill_chaining_with_graphs_amd._lib import ScgConfig
    lib = pkg.load_library()
    ctx = C.c_void_p()
    bad = [dict(n_envs=0), dict(n_options=6), dict(n_options=-1), dict(fourier_order=3)]
    for kw in bad:
        cfg = dict(n_envs=4, n_options=0, fourier_order=5, device=0)
        cfg.update(kw)
        assert lib.scg_create(C.byref(ctx), C.byref(ScgConfig(**cfg))) == -1
        assert not ctx.value and lib.scg_last_error(None)
    assert lib.scg_create(None, None) == -1
    assert lib.scg_destroy(None) == 0
    assert lib.scg_step(None, *([None] * 13), 0, 0, 0, None) == -1


def test_missing_library_fails_loudly(pkg, monkeypatch):
    from skill_chaining_with_graphs_amd import _lib
    monkeypatch.setattr(_lib, "_libs", {})
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libscg_hip.so")
    with pytest.raises(_lib.ScgError, match="no CPU fallback"):
        _lib.load()
    with pytest.raises(_lib.ScgError, match="block_envs must be one of"):
        _lib.load(96)


def test_every_block_geometry_build_exports_the_abi_and_reports_its_block(pkg):
    """SPEC §5's block size is a build parameter: three libraries from one source, chosen per context (block_envs=...);
    the checker is built for the same three."""
    from skill_chaining_with_graphs_amd import _lib
    from oracle import sc_oracle as O
    assert _lib.block_envs() == 256
    try:
        for b in _lib.BLOCK_ENVS_BUILDS:
            lib = _lib.load(b)
            for name in declared_symbols():
                assert hasattr(lib, name), (b, name)
            assert lib.scg_block_envs() == b and lib.scg_abi_version() == 5
            O.use_block_envs(b)
            assert O.lib().sco_block_envs() == b
    finally:
        O.use_block_envs(256)
    assert _lib.load(256) is _lib.load()


def test_product_never_imports_the_oracle():
    pk = os.path.join(ROOT, "skill-chaining-with-graphs_amd")
    for dp, _, fs in os.walk(pk):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "sc_oracle" not in src and "oracle/" not in src, os.path.join(dp, f)


def test_every_ctx_entry_point_rejects_a_null_ctx(pkg):
    """Host-side argument checks come before any HIP call: with a null ctx every entry point returns
    SCG_ERR_INVALID and leaves a message behind (runs without a GPU)."""
    from skill_chaining_with_graphs_amd import _lib
    lib = pkg.load_library()
    skip = {"scg_abi_version", "scg_block_envs", "scg_strerror", "scg_last_error", "scg_create", "scg_destroy",
            "scg_decode_async_word"}
    for name, (res, args) in _lib._SIGS.items():
        if name in skip:
            continue
        assert args and args[0] is C.c_void_p, name
        zeros = [None if (a is C.c_void_p or hasattr(a, "contents")) else a(0) for a in args]
        assert getattr(lib, name)(*zeros) == -1, name
        assert lib.scg_last_error(None), name


def test_async_status_word_decoding(pkg):
    """The mapping from the device-written status word to a status and a message is pure host code (the error path of a
    fit that gave up on the device, include/scg_abi.h "asynchronous failures"): it runs here without a GPU."""
    lib = pkg.load_library()
    buf = C.create_string_buffer(256)
    assert lib.scg_decode_async_word(0, buf, 256) == 0 and buf.value == b""
    rc = lib.scg_decode_async_word(0x1 | (0x100 << 3), buf, 256)
    assert rc == -5 and b"scg_fit_initiation" in buf.value and b"0x8" in buf.value and b"unchanged" in buf.value
    assert lib.scg_decode_async_word(0x80000000, buf, 256) == -5 and b"unknown" in buf.value
    rc = lib.scg_decode_async_word(0x2, buf, 256)                      # SCG_ASYNC_STEP_HANDOFF (round 4): a blown hand-off poll in scg_step
    assert rc == -5 and b"scg_step" in buf.value and b"hand-off poll" in buf.value
    assert lib.scg_decode_async_word(0x2 | 0x1, buf, 256) == -5 and b"scg_step" in buf.value    # the step's failure is named first
    assert lib.scg_decode_async_word(0x1, None, 0) == -5              # no buffer: status only
    small = C.create_string_buffer(8)
    assert lib.scg_decode_async_word(0x1, small, 8) == -5 and len(small.value) == 7      # truncated, terminated
    assert lib.scg_strerror(-5).startswith(b"an earlier launch")


def test_automatic_block_geometry_table(pkg):
    """SPEC §5 / DESIGN §3.6: the block size a context picks from its env count (no GPU needed for the rule itself)."""
    f = pkg.auto_block_envs
    assert [f(n) for n in (1, 4096, 64 * 256, 64 * 256 + 1, 128 * 256, 128 * 256 + 1, 65536, 524288)] == [64, 64, 64, 128, 128, 256, 256, 256]
    assert all(f(n) in pkg.BLOCK_ENVS_BUILDS for n in range(1, 70000, 997))


def test_library_and_oracle_agree_on_their_default_hyper_parameters(pkg, oracle_mod):
    """a library / oracle pair built on defaults must be the same learner (round 4 moved one default and not the other)."""
    import inspect
    from skill_chaining_with_graphs_amd.core import ScgContext
    ctx_d = {k: v.default for k, v in inspect.signature(ScgContext.__init__).parameters.items() if v.default is not inspect.Parameter.empty}
    orc_d = {k: v.default for k, v in inspect.signature(oracle_mod.Oracle.__init__).parameters.items() if v.default is not inspect.Parameter.empty}
    for k in ("gamma", "alpha", "epsilon", "r_option_success", "max_episode_steps", "max_option_steps", "update_count_floor", "reoffer_period"):
        assert ctx_d[k] == orc_d[k], (k, ctx_d[k], orc_d[k])
