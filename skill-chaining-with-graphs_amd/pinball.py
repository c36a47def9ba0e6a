"""PinballDomain — the vectorized Pinball env (SPEC.md §1). north_star names `PinballDomain.step`; the
reference holds no such code (README.md:1-2 only), so the signature is this build's own."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .core import EnvState, ScgContext
from .maps import PinballMap

ACC_X, ACC_Y, DEC_X, DEC_Y, ACC_NONE = range(5)
STEP_PENALTY, THRUST_PENALTY, END_EPISODE = -1.0, -5.0, 10000.0


class PinballDomain:
    """N env instances as SoA tensors in HBM. `step(actions)` is the un-fused physics entry point
    (scg_pinball_step); agents normally go through SkillChainingAgent.step_batch (fused)."""

    def __init__(self, ctx: ScgContext, state: Optional[EnvState] = None):
        self.ctx = ctx
        self.map: PinballMap = ctx.map
        self.state = state if state is not None else EnvState(ctx.n_envs, ctx.device, ctx.map)

    @property
    def n_envs(self) -> int:
        return self.state.n

    def reset(self, positions: Optional[np.ndarray] = None) -> None:
        """All envs to the first start position (or to given [n,2] positions), zero velocity."""
        st = self.state
        if positions is None:
            st.x.fill_(float(self.map.starts[0][0])); st.y.fill_(float(self.map.starts[0][1]))
        else:
            p = torch.as_tensor(np.ascontiguousarray(positions, np.float32), device=st.x.device)
            st.x.copy_(p[:, 0]); st.y.copy_(p[:, 1])
        st.vx.zero_(); st.vy.zero_()
        st.option_id.zero_(); st.opt_steps.zero_(); st.ep_steps.zero_(); st.qcache.zero_()
        self.ctx.invalidate_order()               # option ids written outside scg_step

    def reset_random(self, seed: int = 0, v_max: float = 1.0) -> None:
        """Collision-free uniform positions and velocities in [-v_max, v_max] (steady-state timing)."""
        rng = np.random.default_rng(seed)
        self.reset(self.map.sample_free(self.n_envs, rng))
        v = torch.as_tensor(rng.uniform(-v_max, v_max, (2, self.n_envs)).astype(np.float32),
                            device=self.state.x.device)
        self.state.vx.copy_(v[0]); self.state.vy.copy_(v[1])

    def step(self, actions: torch.Tensor):
        """Advance every env one step with the given uint8 actions -> (reward[n], goal[n]); in place,
        no reset (SPEC §1.3)."""
        return self.ctx.pinball_step(self.state.state(), actions)
