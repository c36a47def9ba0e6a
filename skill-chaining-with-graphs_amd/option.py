"""Option / InitiationClassifier — host-side handles on the device-resident option tables
(SPEC.md §4, §6). north_star names `Option.{policy,beta,initiation_classifier}`; the reference holds no
such code (README.md:1-2 only), so names follow north_star and signatures are this build's own."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ._lib import CLF_STRIDE, NUM_ACTIONS, NUM_FEATURES
from .core import ScgContext


class InitiationClassifier:
    """Logistic regression on psi(x,y) = (1,u,v,u^2,uv,v^2), u=2x-1, v=2y-1. Row k of the agent's
    classifier table; predict/fit run on the GPU (scg_classifier_predict / scg_fit_initiation)."""

    def __init__(self, ctx: ScgContext, table: torch.Tensor, index: int):
        self.ctx, self.table, self.index = ctx, table, index

    @property
    def weights(self) -> torch.Tensor:
        return self.table[self.index]

    def set_weights(self, w) -> None:
        w = torch.as_tensor(np.asarray(w, np.float32))
        self.table[self.index, : w.numel()] = w.to(self.table.device)

    def set_disc(self, cx: float, cy: float, radius: float) -> None:
        """Closed form: in-set iff (x-cx)^2 + (y-cy)^2 < radius^2 (up to rounding)."""
        uc, vc, r = 2 * cx - 1, 2 * cy - 1, 2 * radius
        self.set_weights([r * r - uc * uc - vc * vc, 2 * uc, 2 * vc, -1.0, 0.0, -1.0])

    def predict(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        return self.ctx.classifier_predict(x, y, self.weights.contiguous())

    def fit(self, xy: torch.Tensor, label: torch.Tensor, iters: int = 300, lr: float = 2.0,
            l2: float = 1e-4, warm_start: bool = False) -> None:
        if not warm_start:
            self.table[self.index].zero_()
        off = torch.tensor([0, label.numel()], dtype=torch.int32, device=self.table.device)
        w = self.table[self.index: self.index + 1]
        self.ctx.fit_initiation(xy.contiguous().view(-1), label, off, w.view(-1), iters, lr, l2)


class Option:
    """Option k: VF k (linear Q over the Fourier basis), initiation classifier k, target = the task goal
    (parent[k] = 0) or the initiation set of option parent[k] (default chain: k - 1). k = 0 is the root policy."""

    def __init__(self, agent: "SkillChainingAgent", index: int):  # noqa: F821
        self.agent, self.index = agent, index
        self.initiation_classifier: Optional[InitiationClassifier] = (
            InitiationClassifier(agent.ctx, agent.clf, index) if index >= 1 else None)

    @property
    def weights(self) -> torch.Tensor:
        return self.agent.W[self.index]

    @property
    def enabled(self) -> bool:
        return self.index == 0 or bool((self.agent.enabled_mask >> self.index) & 1)

    def q_values(self, state) -> torch.Tensor:
        """Q_k(s, .) -> [5, n]."""
        return self.agent.ctx.q_values(state, self.weights.contiguous().view(-1))

    def policy(self, state) -> torch.Tensor:
        """Greedy primitive action per env (first maximum), uint8 [n]."""
        return self.q_values(state).argmax(0).to(torch.uint8)

    def in_initiation_set(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        if self.index == 0:
            return torch.ones_like(x, dtype=torch.uint8)
        if not self.enabled:
            return torch.zeros_like(x, dtype=torch.uint8)
        return self.initiation_classifier.predict(x, y)

    def beta(self, x: torch.Tensor, y: torch.Tensor, goal: Optional[torch.Tensor] = None,
             done: Optional[torch.Tensor] = None, opt_steps: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Termination indicator of SPEC §4.2 at post-physics positions (x, y) = s':
        term = (done != 0) | succ | fail | otime, with succ = goal (parent 0) or in_parent(s'), fail = !succ & !in_k(s'),
        otime = opt_steps + 1 >= max_option_steps. `goal` = the step's goal flags (needed when the option, or k = 0,
        targets the task goal); `done` = the step's done codes (time-limit truncation also terminates); `opt_steps` = the
        option's step counters BEFORE the step. Omitted inputs count as "did not happen". The classifier tests run on
        the GPU (scg_classifier_predict); the rest is mask logic on their uint8 outputs, exactly as in the fused step."""
        ag, k = self.agent, self.index
        z = torch.zeros_like(x, dtype=torch.bool)
        g = goal.bool() if goal is not None else z
        d = (done != 0) if done is not None else z
        if k == 0:
            return (g | d).to(torch.uint8)
        parent = int(ag.ctx.parents[k])               # SPEC §4.2 skill graph (default chain: k - 1)
        succ = g if parent == 0 else ag.options[parent].in_initiation_set(x, y).bool()
        fail = ~succ & ~self.in_initiation_set(x, y).bool()
        otime = (opt_steps + 1 >= int(ag.ctx.cfg.max_option_steps)) if opt_steps is not None else z
        return (d | succ | fail | otime).to(torch.uint8)
