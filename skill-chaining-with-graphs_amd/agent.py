"""SkillChainingAgent — host driver of the fused step-batch (SPEC.md §4–§5).

north_star names `SkillChainingAgent.q_update` and asks that the Option / SkillChainingAgent Python API
be kept; the reference holds no code (README.md:1-2 only), so the API below is this build's own,
named after north_star. Per-step work is one scg_step launch pair (fused kernel + reduce/apply);
nothing on the per-step path synchronises with or copies to the host."""
from __future__ import annotations

from typing import List, Optional

import torch

from ._lib import CLF_STRIDE, NUM_ACTIONS, NUM_FEATURES
from .core import EnvState, ScgContext
from .maps import PinballMap, load_map
from .option import Option
from .pinball import PinballDomain


class SkillChainingAgent:
    def __init__(self, pmap, n_envs: int, n_options: int = 0, *, device: int = 0, seed: int = 0,
                 env_id_base: int = 0, group=None, **hparams):
        self.map: PinballMap = load_map(pmap) if isinstance(pmap, str) else pmap
        self.ctx = ScgContext(n_envs, n_options, self.map, device=device, seed=seed, env_id_base=env_id_base,
                              **hparams)
        dev = self.ctx.device
        self.n_envs, self.n_options, self.n_vf = n_envs, n_options, n_options + 1
        self.W = torch.zeros((self.n_vf, NUM_ACTIONS, NUM_FEATURES), dtype=torch.float32, device=dev)
        self.clf = torch.zeros((self.n_vf, CLF_STRIDE), dtype=torch.float32, device=dev)
        self.enabled_mask = 0
        self.t = 0
        self.group = group            # torch.distributed group for shared option-Q weights (or None)
        self.domain = PinballDomain(self.ctx)
        self.state: EnvState = self.domain.state
        self.options: List[Option] = [Option(self, k) for k in range(self.n_vf)]

    # ------------------------------------------------------------------ option management (outer loop)
    def enable_option(self, k: int, enabled: bool = True) -> None:
        if not (1 <= k <= self.n_options):
            raise ValueError("option index out of range")
        self.enabled_mask = (self.enabled_mask | (1 << k)) if enabled else (self.enabled_mask & ~(1 << k))

    def init_weights(self, std: float = 1e-3, seed: int = 0) -> None:
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.W.copy_(torch.randn(self.W.shape, generator=g) * std)

    # ------------------------------------------------------------------ the hot path
    def step_batch(self, learn: bool = True) -> None:
        """One fused step-batch over all envs (act, physics, options, features, Q, TD, update)."""
        shared = self.group is not None and learn
        if shared:
            G, n_k = self.ctx.grad_buffers()
        self.ctx.step(self.state, self.W, self.clf, self.enabled_mask, self.t, learn=learn, apply=not shared)
        if shared:
            import torch.distributed as dist
            dist.all_reduce(G, group=self.group)       # RCCL over xGMI: one fused 26 KB x n_vf message
            dist.all_reduce(n_k, group=self.group)
            self.ctx.apply_update(self.W, G, n_k)
        self.t += 1

    def q_update(self, k: int, s, action, r, cont, s_next, apply: bool = True) -> None:
        """Batched intra-option Q-learning update of VF k on explicit transitions (SPEC §5):
        delta = r + cont * max_a' Q_k(s',a') - Q_k(s,a);  W_k[a] += alpha/n * scale * sum delta*phi(s)."""
        self.ctx.q_update(k, s, action, r, cont, s_next, self.W, apply=apply)

    def rollout(self, steps: int, learn: bool = True) -> None:
        for _ in range(steps):
            self.step_batch(learn)
