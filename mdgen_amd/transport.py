"""Drop-in for the slice of `mdgen.transport` the sampler uses:
`create_transport(...)` -> `Sampler(transport).sample_ode(sampling_method='euler', num_steps=50)`
-> `fn(x0, model_fn, **kw)` (transport.py:408-451, integrators.py:74-114).

velocity prediction + GVP/Linear path => integration interval (t0, t1) = (0, 1) exactly
(transport.py:95-124, 561-563).  Only the fixed-grid Euler solver is provided (dopri5, the SDE
sampler and the likelihood path are outside the accelerated hot path).
"""
from __future__ import annotations

from functools import partial

import torch


class Transport:
    def __init__(self, path_type="GVP", prediction="velocity"):
        if prediction != "velocity":
            raise NotImplementedError("only velocity prediction is on the accelerated path")
        if path_type not in ("GVP", "Linear"):
            raise NotImplementedError("only GVP / Linear paths (t0, t1 = 0, 1)")
        self.path_type, self.prediction = path_type, prediction
        self.train_eps = self.sample_eps = 0

    def check_interval(self, *a, **k):
        return 0, 1


def create_transport(args=None, path_type="GVP", prediction="velocity", loss_weight=None, train_eps=None,
                     sample_eps=None):
    return Transport(path_type, prediction)


class Sampler:
    def __init__(self, transport: Transport):
        self.transport = transport

    def sample_ode(self, *, sampling_method="dopri5", num_steps=50, atol=1e-6, rtol=1e-3, reverse=False):
        """Returns fn(x0, model_fn, **model_kwargs).  `num_steps` is the number of grid points
        (S = num_steps - 1 Euler steps), as in the reference.  The returned tensor holds only the final
        state, shape [1, *x0.shape], so that the caller's `[-1]` (wrapper.py:447) is unchanged."""
        if sampling_method != "euler":
            raise NotImplementedError(
                "mdgen_amd implements the fixed-grid Euler sampler (pass sampling_method='euler'); "
                "adaptive dopri5 is not on the graph-captured path")
        if reverse:
            raise NotImplementedError("reverse-time sampling is not used by the wrapper")
        S = int(num_steps) - 1

        def _sample(x0, model_fn, **model_kwargs):
            from .model import LatentMDGenModel
            fn, kw = model_fn, dict(model_kwargs)
            if isinstance(fn, partial):
                kw = {**fn.keywords, **kw}
                fn = fn.func
            owner = getattr(fn, "__self__", None)
            if isinstance(owner, LatentMDGenModel) and getattr(fn, "__name__", "") in ("forward", "forward_inference"):
                return owner.sample_euler(x0, S, **kw)[None]
            # generic drift (any callable): explicit Euler on the host side, x stays on the device
            tg = torch.linspace(0, 1, S + 1)
            x = x0
            for i in range(S):
                t = torch.ones(x.shape[0], device=x.device) * tg[i]
                x = x + (tg[i + 1] - tg[i]) * model_fn(x, t, **model_kwargs)
            return x[None]

        return _sample
