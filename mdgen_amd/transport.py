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

    # ---- training target and loss, forward only (transport.py:126-189) ----------------------
    def sample(self, x1):
        """transport.py:126-136: x0 ~ N(0, I), t ~ U(0, 1) (velocity + GVP/Linear: t0, t1 = 0, 1), on x1's device."""
        x0 = torch.randn_like(x1)
        t = torch.rand((x1.shape[0],)).to(x1)
        return t, x0, x1

    def plan(self, t, x0, x1):
        """path.py:131-135 `plan`: (t, xt, ut) with xt = alpha x1 + sigma x0, ut = alpha' x1 + sigma' x0, computed by
        `mdgen_path_plan` (GVP: alpha = sin(pi t / 2), sigma = cos(pi t / 2); Linear: alpha = t, sigma = 1 - t)."""
        from ._lib import lib, launch, ptr, require_cuda
        x0 = x0.to(torch.float32).contiguous()
        x1 = x1.to(torch.float32).contiguous()
        t = t.to(torch.float32).contiguous()
        require_cuda(t, x0, x1)
        xt, ut = torch.empty_like(x1), torch.empty_like(x1)
        B = x1.shape[0]
        launch(lib.mdgen_path_plan, x1, B, x1.numel() // B, 1 if self.path_type == "GVP" else 0, ptr(t), ptr(x0), ptr(x1),
               ptr(xt), ptr(ut))
        return t, xt, ut

    def training_losses(self, model, x1, aatype1=None, mask=None, model_kwargs=None, t=None, x0=None):
        """transport.py:138-189 for the velocity model (non-design path): returns {'t', 'pred', 'loss'} with
        loss = mean_flat((model(xt, t) - ut)^2, mask).  `t` / `x0` may be given to reproduce a reference run
        (the reference draws them inside, :126-136).  Forward value only; the differentiated step is `mdgen_amd.train.Trainer.training_step`."""
        from ._lib import lib, launch, ptr, require_cuda
        model_kwargs = model_kwargs or {}
        if t is None or x0 is None:
            t_, x0_, _ = self.sample(x1)
            t = t_ if t is None else t
            x0 = x0_ if x0 is None else x0
        t, xt, ut = self.plan(t, x0, x1)
        pred = model(xt, t, **model_kwargs)
        if pred.shape != xt.shape:
            raise ValueError(f"model output {tuple(pred.shape)} != input {tuple(xt.shape)}")
        m = mask.to(torch.float32).expand_as(xt).contiguous()
        pred = pred.to(torch.float32).contiguous()
        require_cuda(pred, m)
        B = xt.shape[0]
        loss = torch.empty(B, device=xt.device, dtype=torch.float32)
        launch(lib.mdgen_masked_mse, xt, B, xt.numel() // B, ptr(pred), ptr(ut), ptr(m), ptr(loss))
        return {"t": t, "pred": pred, "loss": loss}


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
            use_graph = kw.pop("use_graph", True)   # build-specific keyword: capture / replay the rollout as a hipGraph
            model_kwargs = dict(kw)
            if isinstance(fn, partial):
                kw = {**fn.keywords, **kw}
                fn = fn.func
            owner = getattr(fn, "__self__", None)
            if isinstance(owner, LatentMDGenModel) and getattr(fn, "__name__", "") in ("forward", "forward_inference"):
                return owner.sample_euler(x0, S, use_graph=use_graph, **kw)[None]
            # generic drift (any callable): explicit Euler on the host side, x stays on the device
            tg = torch.linspace(0, 1, S + 1)
            x = x0
            for i in range(S):
                t = torch.ones(x.shape[0], device=x.device) * tg[i]
                x = x + (tg[i + 1] - tg[i]) * model_fn(x, t, **model_kwargs)
            return x[None]

        return _sample
