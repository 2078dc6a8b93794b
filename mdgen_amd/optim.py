"""Optimiser side of the training step (reference: `train.py:46-77`,
`mdgen/wrapper.py:167-172` `configure_optimizers`, `mdgen/ema.py:41-58`, Lightning DDP's gradient averaging).

What is here: flat fp32 parameter / gradient / moment buffers (`FlatParams`), gradient-norm clipping + Adam / AdamW
(`Adam`, kernels in csrc/k_optim.hip, no host round trip for the clip coefficient), the weight EMA (`EMA`) and the
bucketed gradient all-reduce over `torch.distributed` (`GradBucketer`; backend "nccl" is RCCL on ROCm).
The backward pass that fills the gradient buffer is `mdgen_train_forward_backward` (mdgen_amd/train.py).

All arithmetic runs in libmdgen_amd.so; torch carries memory, streams and the process group."""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from ._lib import lib, launch, ptr, require_cuda, MdgenError


class FlatParams:
    """All tensors of a state dict in ONE contiguous fp32 buffer (plus same-shaped buffers for grads / moments on
    request), with named views.  Order = the order of `shapes` (use the reference's parameter order so that bucket
    boundaries follow the backward pass)."""

    def __init__(self, shapes: "OrderedDict[str, Sequence[int]]", device="cuda"):
        self.shapes = OrderedDict((k, tuple(v)) for k, v in shapes.items())
        self.offsets: Dict[str, Tuple[int, int]] = {}
        off = 0
        for k, shp in self.shapes.items():
            n = 1
            for s in shp:
                n *= int(s)
            self.offsets[k] = (off, n)
            off += n
        self.numel = off
        self.device = torch.device(device)
        self.data = torch.zeros(self.numel, dtype=torch.float32, device=self.device)

    def like(self) -> torch.Tensor:
        return torch.zeros_like(self.data)

    def view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        o, n = self.offsets[name]
        return buf[o:o + n].view(self.shapes[name])

    def load_state_dict(self, sd):
        for k in self.shapes:
            self.view(self.data, k).copy_(sd[k].to(torch.float32))
        return self

    def state_dict(self, buf: Optional[torch.Tensor] = None):
        buf = self.data if buf is None else buf
        return OrderedDict((k, self.view(buf, k)) for k in self.shapes)


def adam_state_to_torch(named: dict, order: Sequence[str]) -> dict:
    """The by-name optimiser state of `Adam.state_dict` -> the layout `torch.optim.Adam.state_dict()` has in a Lightning
    checkpoint of the reference (`optimizer_states[0]`): `state[i]` for the i-th parameter of `model.parameters()` (= `order`,
    the reference's registration order, `train.trainable_shapes`) and one param group."""
    state = {i: {"step": torch.tensor(float(named["step"])), "exp_avg": named["exp_avg"][k], "exp_avg_sq": named["exp_avg_sq"][k]}
             for i, k in enumerate(order)}
    group = {"lr": named["lr"], "betas": tuple(named["betas"]), "eps": named["eps"], "weight_decay": named["weight_decay"],
             "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(order)))}
    return {"state": state, "param_groups": [group]}


def adam_state_from_torch(sd: dict, order: Sequence[str]) -> dict:
    """Inverse of `adam_state_to_torch`: accepts what torch.optim.Adam / AdamW saved (a reference checkpoint).  A parameter
    without state (never stepped) gets zero moments; the step count is the largest per-parameter step."""
    ids = [i for g in sd["param_groups"] for i in g["params"]]
    if len(ids) != len(order):
        raise MdgenError(f"optimizer state has {len(ids)} parameters, the model has {len(order)}")
    g0 = sd["param_groups"][0]
    out = {"step": 0, "lr": g0.get("lr"), "betas": tuple(g0.get("betas", (0.9, 0.999))), "eps": g0.get("eps", 1e-8),
           "weight_decay": g0.get("weight_decay", 0.0), "exp_avg": OrderedDict(), "exp_avg_sq": OrderedDict()}
    for i, k in zip(ids, order):
        st = sd["state"].get(i)
        out["exp_avg"][k] = None if st is None else st["exp_avg"]
        out["exp_avg_sq"][k] = None if st is None else st["exp_avg_sq"]
        if st is not None:
            out["step"] = max(out["step"], int(float(st["step"])))
    return out


class Adam:
    """torch.optim.Adam / AdamW over a `FlatParams` (wrapper.py:167-172: `cls(params, lr=args.lr)` with torch's
    defaults betas (0.9, 0.999), eps 1e-8, weight_decay 0 / 0.01), with Lightning's `gradient_clip_val` (train.py:56:
    clip_grad_norm_, 2-norm, coefficient max_norm / (norm + 1e-6) clamped to 1) fused into the step."""

    def __init__(self, params: FlatParams, lr: float, adamw: bool = False, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: Optional[float] = None, grad_clip: Optional[float] = None):
        require_cuda(params.data)
        self.params = params
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.adamw = bool(adamw)
        self.weight_decay = float(weight_decay if weight_decay is not None else (0.01 if adamw else 0.0))
        if not self.adamw and self.weight_decay != 0.0:
            raise MdgenError("coupled (L2) weight decay of torch.optim.Adam is not implemented; the reference uses 0")
        self.grad_clip = grad_clip
        self.exp_avg = params.like()
        self.exp_avg_sq = params.like()
        self.step_count = 0
        self._scratch = torch.empty(1024, dtype=torch.float32, device=params.device)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=params.device)

    def state_dict(self):
        """torch.optim-style: per-parameter moments by name + the step count (resume)."""
        return {"step": self.step_count, "lr": self.lr, "betas": self.betas, "eps": self.eps, "adamw": self.adamw,
                "weight_decay": self.weight_decay,
                "exp_avg": OrderedDict((k, v.detach().cpu().clone()) for k, v in self.params.state_dict(self.exp_avg).items()),
                "exp_avg_sq": OrderedDict((k, v.detach().cpu().clone()) for k, v in self.params.state_dict(self.exp_avg_sq).items())}

    def load_state_dict(self, sd, order: Optional[Sequence[str]] = None):
        """`sd`: this class's by-name layout, or torch.optim's ({"state", "param_groups"}: a reference checkpoint) -- then
        `order` names the parameters in the reference's `model.parameters()` order."""
        if "param_groups" in sd:
            if order is None:
                raise MdgenError("a torch-format optimizer state needs the parameter order")
            sd = adam_state_from_torch(sd, order)
        self.step_count = int(sd["step"])
        for name, buf in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
            for k in self.params.shapes:
                v = sd[name][k]
                if v is None:
                    self.params.view(buf, k).zero_()
                else:
                    self.params.view(buf, k).copy_(v.to(torch.float32).reshape(self.params.shapes[k]))
        return self

    def grad_norm(self, grads: torch.Tensor, grad_scale: float = 1.0) -> torch.Tensor:
        """Device scalar: || grads * grad_scale ||_2 (no host synchronisation)."""
        launch(lib.mdgen_grad_sumsq, grads, grads.numel(), ptr(grads), float(grad_scale), ptr(self._scratch), 1024,
               ptr(self.sumsq))
        return self.sumsq.sqrt()

    def step(self, grads: torch.Tensor, grad_scale: float = 1.0):
        """One update from the flat gradient buffer.  `grad_scale`: e.g. 1 / world_size after a summed all-reduce."""
        if grads.numel() != self.params.numel or grads.dtype != torch.float32:
            raise MdgenError("grads must be the flat fp32 buffer matching the parameters")
        require_cuda(grads)
        self.step_count += 1
        clip = self.grad_clip is not None and self.grad_clip > 0
        if clip:
            launch(lib.mdgen_grad_sumsq, grads, grads.numel(), ptr(grads), float(grad_scale), ptr(self._scratch), 1024,
                   ptr(self.sumsq))
        launch(lib.mdgen_adam_step, grads, self.params.numel, ptr(self.params.data), ptr(grads), ptr(self.exp_avg),
               ptr(self.exp_avg_sq), self.step_count, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
               int(self.adamw), float(grad_scale), ptr(self.sumsq) if clip else None,
               float(self.grad_clip) if clip else 0.0)


class EMA:
    """ema.py:17-58: `stored -= (stored - param) * (1 - decay)` over every entry of the state dict.

    `buffers` / `order`: the model's non-trainable state-dict entries (`pos_embed`, `rot_emb.inv_freq`) and the
    reference's key order.  The reference's ExponentialMovingAverage clones the FULL `model.state_dict()` and
    `load_ema_weights` loads it back strictly (wrapper.py:120-130): with them `state_dict()["params"]` can be dropped into a
    reference checkpoint's `'ema'` entry (the buffers never change, so their average is themselves)."""

    def __init__(self, params: FlatParams, decay: float, buffers=None, order=None):
        self.params, self.decay = params, float(decay)
        self.data = params.data.clone()
        self.buffers = dict(buffers or {})
        self.order = list(order) if order is not None else list(params.shapes) + list(self.buffers)

    def update(self):
        launch(lib.mdgen_ema_update, self.data, self.params.numel, ptr(self.data), ptr(self.params.data), self.decay)

    def state_dict(self):
        own = self.params.state_dict(self.data)
        full = OrderedDict((k, own[k] if k in own else self.buffers[k]) for k in self.order if k in own or k in self.buffers)
        return OrderedDict(params=full, decay=self.decay)

    def load_state_dict(self, sd):
        for k in self.params.shapes:
            self.params.view(self.data, k).copy_(sd["params"][k].to(torch.float32))
        self.decay = float(sd.get("decay", self.decay))
        return self


class GradBucketer:
    """Bucketed gradient all-reduce for DDP (one all-reduce of 34.15 M fp32 gradients = 136.6 MB
    per step, bucketed and overlapped with backward).

    The flat gradient buffer is cut into contiguous buckets of ~`bucket_bytes`, walking the parameters in REVERSE order
    (the order a backward pass produces them).  `mark_ready(name)` records that a parameter's gradient is complete; as
    soon as all parameters of a bucket are ready its all-reduce is launched asynchronously (SUM; the averaging by
    1 / world_size is folded into the optimiser's `grad_scale`).  `finish()` waits for every bucket.
    Bucket size: xGMI is point-to-point (7 links x ~153 GB/s per GPU), a ring all-reduce moves 2 (n-1)/n of the bytes
    per link; buckets close at >= 16 MiB (8 per step for the 34 M-parameter model) -- large enough to stay out of
    the latency regime of a collective (~50 us at 8 GPUs), small enough that 7 of them overlap with the rest of the
    backward pass."""

    def __init__(self, params: FlatParams, grads: torch.Tensor, dist=None, bucket_bytes: int = 16 << 20,
                 single_rank_collectives: bool = False):
        self.params, self.grads, self.dist = params, grads, dist
        # self-test on a one-GPU box: issue every bucket's all-reduce even in a group of ONE rank (an identity), so that the RCCL
        # path of `launch_on_events` executes where no second device exists
        self.single_rank_collectives = single_rank_collectives
        names = list(params.shapes)[::-1]
        self.buckets: List[dict] = []
        cur: List[str] = []
        cur_bytes = 0
        for n in names:
            cur.append(n)
            cur_bytes += params.offsets[n][1] * 4
            if cur_bytes >= bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._close(cur)
        self.bucket_of = {n: i for i, b in enumerate(self.buckets) for n in b["names"]}
        self.launch_order: List[int] = []
        self._handles: List = []
        self.reset()

    def _close(self, names: List[str]):
        lo = min(self.params.offsets[n][0] for n in names)
        hi = max(self.params.offsets[n][0] + self.params.offsets[n][1] for n in names)
        self.buckets.append({"names": list(names), "lo": lo, "hi": hi})

    def reset(self):
        self._pending = [set(b["names"]) for b in self.buckets]
        self.launch_order = []
        self._handles = []

    def world(self) -> int:
        return self.dist.get_world_size() if (self.dist is not None and self.dist.is_initialized()) else 1

    def _reduces(self) -> bool:
        return self.world() > 1 or (self.single_rank_collectives and self.dist is not None and self.dist.is_initialized())

    def mark_ready(self, name: str):
        i = self.bucket_of[name]
        self._pending[i].discard(name)
        if not self._pending[i] and i not in self.launch_order:
            self.launch_order.append(i)
            if self._reduces():
                b = self.buckets[i]
                self._handles.append(self.dist.all_reduce(self.grads[b["lo"]:b["hi"]], op=self.dist.ReduceOp.SUM,
                                                          async_op=True))

    # ---- overlap with a backward pass that runs as ONE library call (mdgen_train_forward_backward) -----------------
    def launch_on_events(self, milestone_of: Callable[[str], int], events, comm_stream, on_bucket=None):
        """The library records `events[k]` on the training stream when the gradients of parameter group k are final
        (include/mdgen_amd.h `mdgen_train_set_milestone_events`).  Every bucket's all-reduce is enqueued on
        `comm_stream` behind the event of the LAST group it contains, in backward order -- so it runs while the
        training stream is still differentiating earlier layers.  `on_bucket(i, view)` (tests) is called inside the
        comm-stream context after the wait.  Call `finish()` afterwards."""
        order = sorted(range(len(self.buckets)), key=lambda i: max(milestone_of(n) for n in self.buckets[i]["names"]))
        for i in order:
            b = self.buckets[i]
            m = max(milestone_of(n) for n in b["names"])
            comm_stream.wait_event(events[m])
            with torch.cuda.stream(comm_stream):
                view = self.grads[b["lo"]:b["hi"]]
                if on_bucket is not None:
                    on_bucket(i, view)
                if self._reduces():
                    self._handles.append(self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, async_op=True))
            self._pending[i] = set()
            self.launch_order.append(i)
        self._comm_stream = comm_stream

    def finish(self) -> float:
        """Wait for all buckets; returns the `grad_scale` (1 / world_size) the optimiser should apply."""
        missing = [i for i, p in enumerate(self._pending) if p]
        if missing:
            raise MdgenError(f"gradients of buckets {missing} were never marked ready")
        for h in self._handles:
            h.wait()
        cs = getattr(self, "_comm_stream", None)
        if cs is not None:                       # the optimiser (training stream) must see the reduced buckets
            torch.cuda.current_stream(self.grads.device).wait_stream(cs)
            self._comm_stream = None
        return 1.0 / self.world()
