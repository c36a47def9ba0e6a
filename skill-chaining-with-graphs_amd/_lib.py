"""ctypes binding of libscg_hip.so (include/scg_abi.h). The product path has NO fallback: if the HIP
library is missing or a call fails, this module raises — it never routes to a CPU implementation."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCG_LIB") or os.path.join(_HERE, "csrc", "libscg_hip.so")   # SCG_LIB: A/B builds

NUM_ACTIONS = 5
FOURIER_ORDER = 5
NUM_FEATURES = 1296
MAX_OPTIONS = 5
MAX_EDGES = 256
CLF_STRIDE = 8

STEP_LEARN = 1
STEP_APPLY = 2
ABI_VERSION = 5            # include/scg_abi.h SCG_ABI_VERSION
ASYNC_FIT_TIMEOUT = 0x1
ASYNC_STEP_HANDOFF = 0x2


class ScgError(RuntimeError):
    pass


class ScgConfig(C.Structure):
    _fields_ = [
        ("n_envs", C.c_int32),
        ("n_options", C.c_int32),
        ("fourier_order", C.c_int32),
        ("device", C.c_int32),
        ("env_id_base", C.c_int64),
        ("seed", C.c_uint64),
        ("gamma", C.c_float),
        ("alpha", C.c_float),
        ("epsilon", C.c_float),
        ("r_option_success", C.c_float),
        ("max_episode_steps", C.c_int32),
        ("max_option_steps", C.c_int32),
        ("update_count_floor", C.c_int32),
        ("reoffer_period", C.c_int32),
    ]


_P = C.c_void_p
_SIGS = {
    "scg_abi_version": (C.c_int, []),
    "scg_block_envs": (C.c_int, []),
    "scg_strerror": (C.c_char_p, [C.c_int]),
    "scg_last_error": (C.c_char_p, [_P]),
    "scg_create": (C.c_int, [C.POINTER(_P), C.POINTER(ScgConfig)]),
    "scg_destroy": (C.c_int, [_P]),
    "scg_set_hparams": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "scg_set_map": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P, _P]),
    "scg_step": (C.c_int, [_P] + [_P] * 13 + [C.c_uint32, C.c_uint64, C.c_uint32, _P]),
    "scg_grad_buffers": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P)]),
    "scg_set_grad_buffers": (C.c_int, [_P, _P, _P]),
    "scg_apply_update": (C.c_int, [_P, _P, _P, _P, _P]),
    "scg_set_grad_buffer_packed": (C.c_int, [_P, _P]),
    "scg_apply_update_packed": (C.c_int, [_P, _P, _P, _P]),
    "scg_apply_update_slots": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, _P]),
    "scg_pinball_step": (C.c_int, [_P, C.c_int32] + [_P] * 7 + [_P]),
    "scg_fourier_features": (C.c_int, [_P, C.c_int32] + [_P] * 5 + [_P]),
    "scg_q_values": (C.c_int, [_P, C.c_int32] + [_P] * 6 + [_P]),
    "scg_q_update": (C.c_int, [_P, C.c_int32, C.c_int32] + [_P] * 12 + [C.c_uint32, _P]),
    "scg_classifier_predict": (C.c_int, [_P, C.c_int32] + [_P] * 4 + [_P]),
    "scg_set_option_parents": (C.c_int, [_P, _P]),
    "scg_invalidate_order": (C.c_int, [_P]),
    "scg_set_trace_buffers": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P]),
    "scg_harvest": (C.c_int, [_P, C.c_int32, _P, _P, _P, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "scg_collect_examples": (C.c_int, [_P, C.c_uint32, _P, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32, _P]),
    "scg_arm_collect": (C.c_int, [_P, C.c_uint32, _P, C.c_int32, C.c_int32, _P]),
    "scg_set_gestation": (C.c_int, [_P, C.c_uint32, _P]),
    "scg_profile_reset": (C.c_int, [_P, C.c_int32]),
    "scg_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "scg_fit_initiation": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, C.c_int32, C.c_float, C.c_float, _P]),
    "scg_async_status": (C.c_int, [_P, _P, C.c_int32, C.POINTER(C.c_uint32)]),
    "scg_clear_async_error": (C.c_int, [_P]),
    "scg_set_fit_timeout": (C.c_int, [_P, C.c_double]),
    "scg_decode_async_word": (C.c_int, [C.c_uint32, C.c_char_p, C.c_int32]),
    "scg_debug_raise_async": (C.c_int, [_P, C.c_uint32]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)

BLOCK_ENVS_DEFAULT = 256
BLOCK_ENVS_BUILDS = (64, 128, 256)     # SPEC §5 geometry is a build parameter: csrc/Makefile builds one library per value

CHIP_CUS = 256            # MI355X: 8 XCDs x 32 CUs; one 16-wavefront workgroup owns a CU


def auto_block_envs(n_envs: int) -> int:
    """The SPEC §5 block size a context of `n_envs` envs picks when none is given: the SMALLEST build whose workgroups — one per
    block, one per CU — still fit the chip in one round (a workgroup is an ~80 us latency chain whatever its size, so below 65 536
    envs smaller blocks on more CUs win; DESIGN §3.6: 4096 envs 127 / 104 / 69 M env-steps/s at 64 / 128 / 256). Deterministic in
    n_envs alone (the CU count is the chip's constant, not queried): a run's geometry is part of its identity like its seed."""
    for b in BLOCK_ENVS_BUILDS:
        if -(-int(n_envs) // b) <= CHIP_CUS:
            return b
    return BLOCK_ENVS_DEFAULT


_libs = {}


def lib_path(block_envs: int | None = None) -> str:
    """The library of a block geometry: libscg_hip.so (256 envs per block = per workgroup, the throughput build) or
    libscg_hip_b64.so / _b128.so (the small-batch builds). SCG_LIB overrides the default build only."""
    if block_envs in (None, BLOCK_ENVS_DEFAULT):
        return LIB_PATH
    if block_envs not in BLOCK_ENVS_BUILDS:
        raise ScgError(f"block_envs must be one of {BLOCK_ENVS_BUILDS}")
    return os.path.join(_HERE, "csrc", f"libscg_hip_b{block_envs}.so")


def load(block_envs: int | None = None, path: str | None = None) -> C.CDLL:
    """Load the HIP library of a block geometry (once each). Raises ScgError loudly when it has not been built.
    `path`: a variant build of the same ABI (timing / fault-injection builds of csrc/Makefile) instead of the geometry's library."""
    key = (BLOCK_ENVS_DEFAULT if block_envs is None else int(block_envs)) if path is None else os.path.abspath(path)
    if key in _libs:
        return _libs[key]
    if path is None:
        path = lib_path(key)
    if not os.path.exists(path):
        raise ScgError(
            f"{path} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C "
            f"{os.path.dirname(path)}`). There is no CPU fallback."
        )
    lib = C.CDLL(path)
    try:
        ver = int(lib.scg_abi_version())
    except AttributeError:
        raise ScgError(f"{path} does not export scg_abi_version: not a libscg_hip.so") from None
    if ver != ABI_VERSION:
        raise ScgError(f"{path} implements ABI version {ver}, this binding expects {ABI_VERSION} "
                       "(include/scg_abi.h): rebuild the library (a stale .so?)")
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise ScgError(f"{path} does not export {name} although it reports ABI version {ver}: rebuild it") from None
        fn.restype = res
        fn.argtypes = args
    if block_envs is not None and isinstance(key, int) and int(lib.scg_block_envs()) != key:
        raise ScgError(f"{path} is built for {int(lib.scg_block_envs())}-env blocks, not {key}: rebuild it")
    _libs[key] = lib
    return lib


def block_envs(block_envs: int | None = None) -> int:
    """SPEC §5 block size of a loaded library (envs whose update items share one accumulation chain)."""
    return int(load(block_envs).scg_block_envs())


def check(status: int, ctx=None, what: str = "") -> None:
    if status == 0:
        return
    lib = load()
    msg = lib.scg_last_error(ctx).decode() if True else ""
    raise ScgError(f"{what or 'scg call'} failed ({lib.scg_strerror(status).decode()}): {msg}")
