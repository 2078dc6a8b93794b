"""Training step (reference: `NewMDGenWrapper.general_step` / `training_step`,
wrapper.py:82-86, 367-403; `train.py:46-77`): flow-matching plan -> forward -> masked MSE -> backward -> bucketed gradient
all-reduce -> gradient clipping + Adam -> EMA, with every piece of arithmetic in libmdgen_amd.so.

Forward and backward run the fp32-operand form of the network (csrc/k_fp32.hip, k_fp32_bwd.hip, train.inc): the
gradients are checked against the reference's autograd at fp32 tolerance.  It is a correct training step, not yet a fast
one (the bf16 MFMA kernels of the sampler have no backward counterparts); see DESIGN.md."""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Optional

import torch

from . import _lib as L
from ._lib import lib, check, ptr, require_cuda
from .config import ModelConfig
from .model import LatentMDGenModel, _frames
from .optim import Adam, EMA, FlatParams, GradBucketer
from .synthetic import state_shapes

FROZEN = ("pos_embed",)          # registered as buffers in the reference (latent_model.py:116-121): no gradient


def trainable_shapes(cfg: ModelConfig) -> "OrderedDict[str, tuple]":
    """The reference's parameters in its own order (what `model.parameters()` yields, minus frozen buffers)."""
    return OrderedDict((k, v) for k, v in state_shapes(cfg).items() if not k.endswith("inv_freq") and k not in FROZEN)


def grad_milestone(name: str, num_layers: int) -> int:
    """Parameter group of the library's backward pass whose completion makes `name`'s gradient final
    (include/mdgen_amd.h `mdgen_train_set_milestone_events`): 0 emb_to_latent | 1 .. nl layers nl-1 .. 0 | nl+1 token
    embedders | nl+2 .. 2nl+1 ipa_layers nl-1 .. 0 | 2nl+2 the rest."""
    nl = num_layers
    head = name.split(".")[0]
    if head == "emb_to_latent":
        return 0
    if head == "layers":
        return 1 + (nl - 1 - int(name.split(".")[1]))
    if head in ("latent_to_emb", "cond_to_emb", "mask_to_emb"):
        return nl + 1
    if head == "ipa_layers":
        return nl + 2 + (nl - 1 - int(name.split(".")[1]))
    return 2 * nl + 2


def flat_order(cfg: ModelConfig) -> "OrderedDict[str, tuple]":
    """`trainable_shapes` re-ordered for the flat parameter / gradient buffers: groups that the backward pass finishes
    LAST come first, so that walking the buffer from its end (what `GradBucketer` does) meets the gradients in the order
    they become final and every bucket can start its all-reduce as early as possible.  (The reference's registration
    order puts `t_embedder` -- final only at the very end -- behind `emb_to_latent`, the first to be ready.)"""
    sh = trainable_shapes(cfg)
    names = sorted(sh, key=lambda k: -grad_milestone(k, cfg.num_layers))   # stable: the reference's order inside a group
    return OrderedDict((k, sh[k]) for k in names)


class TrainableModel:
    """A `LatentMDGenModel` whose trainable tensors live in one flat fp32 buffer (`self.params`) next to a flat
    gradient buffer (`self.grads`); `forward_backward` fills the gradients, `sync_weights` hands the (updated)
    parameters back to the library (bf16 repack for the sampler kernels + fp32 copies for the training kernels)."""

    def __init__(self, cfg: ModelConfig, device="cuda"):
        self.cfg = cfg
        self.model = LatentMDGenModel(cfg, device, precision="fp32")
        self.device = self.model.device
        self.params = FlatParams(flat_order(cfg), device=self.device)
        self.grads = self.params.like()
        self._buffers = {}
        names = self.model.weight_names()
        self._goff = (C.c_int64 * len(names))(*[self.params.offsets[n][0] if n in self.params.offsets else -1 for n in names])
        self._tape = None
        self._tape_key = None

    def load_state_dict(self, sd):
        self.params.load_state_dict(sd)
        self._buffers = {k: v for k, v in sd.items() if k not in self.params.offsets}
        return self.sync_weights()

    def state_dict(self):
        out = OrderedDict(self.params.state_dict())
        out.update(self._buffers)
        return out

    def sync_weights(self):
        self.model.load_state_dict(self.state_dict())
        self.model.set_precision("bf16")          # the sampler's default; forward_backward selects fp32 itself
        return self

    def zero_grad(self):
        self.grads.zero_()

    def forward_backward(self, xt, t, target, loss_mask, mask, start_frames, x_cond, x_cond_mask, aatype, end_frames=None):
        """loss[b] and pred; d mean_b(loss) / d theta is ADDED into self.grads.  `end_frames`: the two-sided (TPS) model's
        second conditioning frames (latent_model.py:193-205)."""
        m = self.model
        B, T, L_, D = xt.shape
        xt = xt.to(torch.float32).contiguous()
        t = t.to(torch.float32).contiguous()
        target = target.to(torch.float32).contiguous()
        loss_mask = loss_mask.to(torch.float32).expand_as(xt).contiguous()
        mask = mask.to(torch.float32).contiguous()
        sr, st = _frames(start_frames)
        if self.cfg.tps_condition and end_frames is None:
            raise L.MdgenError("tps_condition requires end_frames")
        er, et = _frames(end_frames) if self.cfg.tps_condition else (None, None)
        x_cond = x_cond.to(torch.float32).contiguous()
        x_cond_mask = x_cond_mask.to(torch.int64).contiguous()
        aatype = aatype.to(torch.int64).contiguous()
        require_cuda(xt, t, target, loss_mask, mask, sr, st, x_cond, x_cond_mask, aatype)
        ws = m._workspace(B, T, L_, 1, False)
        sh = L.Shape(B, T, L_)
        if self._tape_key != (B, T, L_):
            nbytes = C.c_size_t()
            check(lib.mdgen_train_workspace_bytes(m._ctx, C.byref(sh), C.byref(nbytes)))
            self._tape = torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)
            self._tape_key = (B, T, L_)
        loss = torch.empty(B, device=self.device)
        pred = torch.empty_like(xt)
        with torch.cuda.device(self.device):
            check(lib.mdgen_train_forward_backward(
                m._ctx, C.byref(sh), ptr(xt), ptr(t), ptr(mask), ptr(sr), ptr(st), ptr(er), ptr(et), ptr(x_cond), ptr(x_cond_mask),
                ptr(aatype),
                ptr(target), ptr(loss_mask), ptr(loss), ptr(pred), ptr(self.grads), self._goff, ptr(ws), ws.numel(),
                ptr(self._tape), self._tape.numel(), L.stream_ptr()))
        return loss, pred


class Trainer:
    """`training_step` of the reference (wrapper.py:82-86 -> general_step :367-403) + what Lightning does around it
    (train.py:46-77): zero_grad -> forward/backward -> DDP gradient averaging (bucketed all-reduce over RCCL) ->
    clip_grad_norm_(grad_clip) -> Adam / AdamW step -> EMA update (`on_before_zero_grad`, wrapper.py:78-80)."""

    def __init__(self, wrapper, lr: float = 1e-4, adamw: bool = False, grad_clip: Optional[float] = 1.0,
                 ema_decay: Optional[float] = None, dist=None):
        self.wrapper = wrapper                    # a NewMDGenWrapper whose .model is replaced by the trainable model's
        self.tm = TrainableModel(wrapper.cfg, wrapper.device)
        self.tm.load_state_dict(wrapper.model_state_dict)
        wrapper.model = self.tm.model
        self.opt = Adam(self.tm.params, lr=lr, adamw=adamw, grad_clip=grad_clip)
        self.ema = EMA(self.tm.params, ema_decay) if ema_decay else None
        self.buckets = GradBucketer(self.tm.params, self.tm.grads, dist=dist)
        # gradient milestones of the library's backward pass (include/mdgen_amd.h): one event per parameter group, the
        # buckets' all-reduces wait for them on a communication stream of their own
        nl = wrapper.cfg.num_layers
        self._n_milestones = lib.mdgen_train_num_milestones(self.tm.model._ctx)
        assert self._n_milestones == 2 * nl + 3
        self._events = [torch.cuda.Event() for _ in range(self._n_milestones)]
        with torch.cuda.device(self.tm.device):
            for ev in self._events:
                ev.record()                       # creates the underlying hipEvent_t
            arr = (C.c_void_p * self._n_milestones)(*[ev.cuda_event for ev in self._events])
            check(lib.mdgen_train_set_milestone_events(self.tm.model._ctx, arr, self._n_milestones))
            self._comm_stream = torch.cuda.Stream(device=self.tm.device)
        self.on_bucket = None                     # test hook: called as on_bucket(i, view) on the communication stream

    def milestone_of(self, name: str) -> int:
        return grad_milestone(name, self.wrapper.cfg.num_layers)

    def training_step(self, batch, t=None, x0=None):
        w = self.wrapper
        prep = w.prep_batch(batch)
        x1 = prep["latents"]
        if t is None or x0 is None:
            t_, x0_, _ = w.transport.sample(x1)
            t = t_ if t is None else t
            x0 = x0_ if x0 is None else x0
        t, xt, ut = w.transport.plan(t, x0, x1)
        kw = prep["model_kwargs"]
        self.tm.zero_grad()
        self.buckets.reset()
        loss, _ = self.tm.forward_backward(xt, t, ut, prep["loss_mask"], kw["mask"], kw["start_frames"], kw["x_cond"],
                                           kw["x_cond_mask"], kw["aatype"], end_frames=kw.get("end_frames"))
        # forward_backward has only ENQUEUED the step: the buckets' all-reduces queue up behind the milestone events and
        # overlap with the part of the backward pass that is still running
        self.buckets.launch_on_events(self.milestone_of, self._events, self._comm_stream, on_bucket=self.on_bucket)
        scale = self.buckets.finish()
        self.opt.step(self.tm.grads, grad_scale=scale)
        self.tm.sync_weights()
        if self.ema is not None:
            self.ema.update()
        return loss.mean()
