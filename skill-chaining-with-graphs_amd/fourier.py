"""FourierBasis — host mirror of the basis the kernels evaluate (SPEC.md §3).

north_star names `FourierBasis.features`; the reference has no such file to cite (README.md:1-2 only).
`features()` materialises Phi through scg_fourier_features for API completeness and tests; the fused
step never writes Phi to HBM."""
from __future__ import annotations

import numpy as np
import torch

from .core import ScgContext, fourier_scale_table


class FourierBasis:
    def __init__(self, ctx: ScgContext, order: int = 5, n_vars: int = 4):
        if order != 5 or n_vars != 4:
            raise ValueError("the gfx950 kernels are built for order 5 over (x, y, vx, vy) only")
        self.ctx, self.order, self.n_vars = ctx, order, n_vars
        n = order + 1
        idx = np.arange(n ** n_vars)
        self.coefficients = np.stack([(idx // n ** (n_vars - 1 - d)) % n for d in range(n_vars)], 1)
        self.scale = fourier_scale_table(order, n_vars)

    @property
    def num_features(self) -> int:
        return len(self.coefficients)

    def features(self, state) -> torch.Tensor:
        """state = (x, y, vx, vy) device tensors [n] -> Phi [n, 1296]."""
        return self.ctx.features(state)
