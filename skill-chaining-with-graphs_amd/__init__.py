"""skill-chaining-with-graphs_amd — MI355X-native vectorized skill-chaining inner loop (Pinball env step,
Fourier features, option policy/termination/initiation tests, batched intra-option Q-learning) behind
the C-ABI of include/scg_abi.h. Import as `skill_chaining_with_graphs_amd` (the hyphenated directory
name is not a Python identifier; the shim package of that name points here)."""
from ._lib import (BLOCK_ENVS_BUILDS, auto_block_envs, CLF_STRIDE, EXPORTED_SYMBOLS, LIB_PATH, MAX_EDGES, MAX_OPTIONS, NUM_ACTIONS, NUM_FEATURES, ScgError,
                   block_envs, load as load_library)
from .maps import MapError, PinballMap, available_maps, load_map, parse_map
from .dist import shard_range

__all__ = ["ScgError", "MapError", "PinballMap", "load_map", "parse_map", "available_maps", "shard_range",
           "load_library", "EXPORTED_SYMBOLS", "LIB_PATH", "NUM_ACTIONS", "NUM_FEATURES", "MAX_OPTIONS",
           "MAX_EDGES", "CLF_STRIDE", "block_envs", "BLOCK_ENVS_BUILDS", "auto_block_envs"]


def __getattr__(name):
    # torch-dependent classes are imported lazily so that map/ABI utilities work without touching torch
    if name in ("ScgContext", "EnvState", "fourier_scale_table"):
        from . import core
        return getattr(core, name)
    if name == "FourierBasis":
        from .fourier import FourierBasis
        return FourierBasis
    if name == "PinballDomain":
        from .pinball import PinballDomain
        return PinballDomain
    if name in ("Option", "InitiationClassifier"):
        from . import option
        return getattr(option, name)
    if name == "SkillChainingAgent":
        from .agent import SkillChainingAgent
        return SkillChainingAgent
    raise AttributeError(name)
