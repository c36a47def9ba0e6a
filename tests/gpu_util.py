"""GPU-side helpers: build a product context + oracle pair on the same map/hyper-parameters."""
import numpy as np
import torch

import sc_oracle
import skill_chaining_with_graphs_amd as scg
from skill_chaining_with_graphs_amd.core import EnvState, ScgContext
from util import HP, SCALE


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda:0")


_BLOCK_ENVS = None        # None: the default (256-env) builds of both sides


def set_block_envs(block_envs=None):
    """Every make_pair() from here on pairs the HIP library and the oracle BUILT FOR this SPEC §5 block size (64 / 128 / 256)."""
    global _BLOCK_ENVS
    _BLOCK_ENVS = block_envs
    sc_oracle.use_block_envs(block_envs or 256)


def make_pair(map_name, n_envs, n_options=0, seed=0, env_id_base=0, enabled_mask=0, **hp):
    m = scg.load_map(map_name)
    kw = dict(HP)
    kw.update(hp)
    ctx = ScgContext(n_envs, n_options, m, device=0, seed=seed, env_id_base=env_id_base, block_envs=_BLOCK_ENVS or 256, **kw)      # (explicit: the oracle pairing is per geometry)
    assert ctx.block_envs == (_BLOCK_ENVS or 256) == sc_oracle.lib().sco_block_envs()
    orc = sc_oracle.Oracle(m, SCALE, n_envs=n_envs, n_options=n_options, seed=seed, env_id_base=env_id_base,
                           enabled_mask=enabled_mask, n_threads=8, **kw)
    return ctx, orc, m


def state_to_device(st_np, ctx):
    st = EnvState(len(st_np["x"]), ctx.device, ctx.map)
    for k, v in st_np.items():
        getattr(st, k).copy_(dev(v))
    return st


def assert_state_equal(st_dev, st_np, keys=None, msg=""):
    keys = keys or ("x", "y", "vx", "vy", "option_id", "opt_steps", "ep_steps", "qcache", "action", "reward", "done")
    for k in keys:
        got = getattr(st_dev, k).cpu().numpy()
        assert np.array_equal(got, st_np[k]), f"{msg} field {k}: {np.sum(got != st_np[k])} of {got.size} differ"
