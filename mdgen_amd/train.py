"""Training step (reference: `NewMDGenWrapper.general_step` / `training_step`,
wrapper.py:82-86, 367-403; `train.py:46-77`): flow-matching plan -> forward -> masked MSE -> backward -> bucketed gradient
all-reduce -> gradient clipping + Adam -> EMA, with every piece of arithmetic in libmdgen_amd.so.

Forward and backward run the fp32-operand form of the network (csrc/k_fp32.hip, k_fp32_bwd.hip, train.inc): the
gradients are checked against the reference's autograd at fp32 tolerance.  It is a correct training step, not yet a fast
one (the bf16 MFMA kernels of the sampler have no backward counterparts); see DESIGN.md.

`python -m mdgen_amd.train` is the launcher (reference: train.py:46-77), one process per GPU under torch.distributed.run."""
from __future__ import annotations

import ctypes as C
import weakref
from collections import OrderedDict
from typing import Optional

import torch

from . import _lib as L
from ._lib import lib, check, ptr, require_cuda
from .config import ModelConfig
from .model import LatentMDGenModel, _frames
from .optim import Adam, EMA, FlatParams, GradBucketer, adam_state_to_torch
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
    gradient buffer (`self.grads`).  After the first load the library's training kernels read their fp32 weights
    straight out of that buffer (`mdgen_train_bind_params`): an optimiser step needs NO hand-back.  The sampler's
    bf16 fragment-packed weights go stale instead; they are re-packed lazily, right before the next network
    evaluation through `self.model` (`mark_updated` / `sync_weights`)."""

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
        self._stale = False
        self.model._pre_run = self._refresh_if_stale

    def load_state_dict(self, sd):
        self.params.load_state_dict(sd)
        self._buffers = {k: v for k, v in sd.items() if k not in self.params.offsets}
        self.sync_weights()
        with torch.cuda.device(self.device):
            check(lib.mdgen_train_bind_params(self.model._ctx, ptr(self.params.data), self._goff))
        return self

    def state_dict(self):
        out = OrderedDict(self.params.state_dict())
        out.update(self._buffers)
        return out

    def sync_weights(self):
        """Re-pack the sampler's bf16 weights from the flat parameters (the training kernels never need this)."""
        self.model.load_state_dict(self.state_dict())
        self.model.set_precision("bf16")          # the sampler's default; forward_backward selects fp32 itself
        self._stale = False
        return self

    # weights whose only consumers in the TRAINING path are the bound fp32 views (attention / MLP / IPA / final linear layers,
    # k_fp32.hip reads them through the context's key -> pointer table); their bf16 fragment packs feed the sampler only
    _PACKED = ("q_proj.weight", "k_proj.weight", "v_proj.weight", "out_proj.weight", "fc1.weight", "fc2.weight",
               "linear_q.weight", "linear_kv.weight", "linear_q_points.weight", "linear_kv_points.weight",
               "linear_out.weight", "emb_to_latent.linear.weight")

    # the parameters whose fp32 values the training kernels read from tables the CONTEXT owns (api.hip setters that copy_f32
    # into wl / bl / wc / bc / mask_emb / aa_emb / t_* / pos_embed / wf7.. / ada_w / ada_b / gamma_beta / head_w); every other
    # parameter is read through its bound pointer.  test_mark_updated_refreshes_every_table_the_training_step_reads holds this
    # list against a full hand-over.
    _TABLE_KEYS = frozenset((
        "latent_to_emb.weight", "latent_to_emb.bias", "cond_to_emb.weight", "cond_to_emb.bias", "mask_to_emb.weight",
        "aatype_to_emb.weight", "t_embedder.mlp.0.weight", "t_embedder.mlp.0.bias", "t_embedder.mlp.2.weight",
        "t_embedder.mlp.2.bias", "pos_embed", "latent_to_emb_f.weight", "latent_to_emb_f.bias", "latent_to_emb_r.weight",
        "latent_to_emb_r.bias"))
    _TABLE_SUFFIXES = ("adaLN_modulation.1.weight", "adaLN_modulation.1.bias", "ipa_norm.weight", "ipa_norm.bias",
                       "ipa.head_weights")

    def mark_updated(self):
        """The flat parameters changed (optimiser step, checkpoint load).  The tensors the training kernels read from the
        context's OWN fp32 tables (time embedder, the concatenated adaLN table, token / residue embeddings, IPA norm and head
        weights, permuted biases) are handed over again right away -- plain device copies from the flat buffer's views, enqueued
        on the stream, no host synchronisation; the bf16 fragment packs of the big matrices (90 % of the bytes, sampler only) are
        left stale until the next network evaluation through `self.model`."""
        self._stale = True
        m = self.model
        sd = self.params.state_dict()
        with torch.cuda.device(self.device):
            s = L.stream_ptr()
            for k, v in sd.items():
                if not (k in self._TABLE_KEYS or k.endswith(self._TABLE_SUFFIXES)):
                    continue   # read in place from the bound flat buffer; its sampler-side copies wait for sync_weights()
                shp = (C.c_int64 * v.dim())(*v.shape)
                check(lib.mdgen_ctx_set_weight(m._ctx, k.encode(), ptr(v), shp, v.dim(), s))
            check(lib.mdgen_ctx_finalize(m._ctx, s))

    def _refresh_if_stale(self):
        if self._stale:
            self.sync_weights()

    def zero_grad(self):
        self.grads.zero_()

    def forward_backward(self, xt, t, target, loss_mask, mask, start_frames, x_cond, x_cond_mask, aatype, end_frames=None,
                         rel_quats=None):
        """loss[b] and pred; d mean_b(loss) / d theta is ADDED into self.grads.  `end_frames`: the two-sided (TPS) model's
        second conditioning frames (latent_model.py:193-205); `rel_quats`: optionally its relative-frame 7-vectors as the
        caller's reference computed them (`LatentMDGenModel._rel7`)."""
        m = self.model
        B, T, L_, D = xt.shape
        rel7 = m._rel7(rel_quats, B, L_)
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
        # (the tape's size also depends on the option train_streams: ask every call -- a host-side sum -- and grow when needed)
        nbytes = C.c_size_t()
        check(lib.mdgen_train_workspace_bytes(m._ctx, C.byref(sh), C.byref(nbytes)))
        if self._tape is None or self._tape.numel() < nbytes.value or self._tape_key != (B, T, L_):
            self._tape = None
            self._tape = torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)
            self._tape_key = (B, T, L_)
        loss = torch.empty(B, device=self.device)
        pred = torch.empty_like(xt)
        with torch.cuda.device(self.device):
            check(lib.mdgen_train_forward_backward(
                m._ctx, C.byref(sh), ptr(xt), ptr(t), ptr(mask), ptr(sr), ptr(st), ptr(er), ptr(et), ptr(rel7), ptr(x_cond),
                ptr(x_cond_mask), ptr(aatype),
                ptr(target), ptr(loss_mask), ptr(loss), ptr(pred), ptr(self.grads), self._goff, ptr(ws), ws.numel(),
                ptr(self._tape), self._tape.numel(), L.stream_ptr()))
        return loss, pred


class Trainer:
    """`training_step` of the reference (wrapper.py:82-86 -> general_step :367-403) + what Lightning does around it
    (train.py:46-77): zero_grad -> forward/backward -> DDP gradient averaging (bucketed all-reduce over RCCL) ->
    clip_grad_norm_(grad_clip) -> Adam / AdamW step -> EMA update (`on_before_zero_grad`, wrapper.py:78-80).

    `dist`: an initialised `torch.distributed` module (backend "nccl" = RCCL on the GPUs; gloo works on CUDA tensors too and
    is what the 2-process test on one GPU uses).  At construction rank 0's parameters, Adam moments and EMA are broadcast
    -- what torch DDP / Lightning do -- so that ranks started from different states cannot silently diverge."""

    def __init__(self, wrapper, lr: float = 1e-4, adamw: bool = False, grad_clip: Optional[float] = 1.0,
                 ema_decay: Optional[float] = None, dist=None, state_dict=None, single_rank_collectives: bool = False):
        self.wrapper = wrapper                    # a NewMDGenWrapper whose .model is replaced by the trainable model's
        # (load_ema_weights caches the trainer's CURRENT weights, not the loaded ones.  A weak reference: wrapper -> trainer ->
        # wrapper would be a cycle, and `__del__` -> `close()`, which detaches the milestone events, would run only at a cyclic GC)
        wrapper.trainer = weakref.ref(self)
        sd = state_dict if state_dict is not None else getattr(wrapper, "model_state_dict", None)
        if sd is None:
            raise L.MdgenError("Trainer needs the model's state dict: load the wrapper with load_model_state_dict() / "
                               "load_from_checkpoint(), or pass state_dict=...")
        self.tm = TrainableModel(wrapper.cfg, wrapper.device)
        self.tm.load_state_dict(sd)
        wrapper.model = self.tm.model
        self.opt = Adam(self.tm.params, lr=lr, adamw=adamw, grad_clip=grad_clip)
        self.ema = EMA(self.tm.params, ema_decay, buffers=self.tm._buffers, order=list(sd.keys())) if ema_decay else None
        # (single_rank_collectives: self-test -- broadcast / all-reduce run in a group of one rank too, see GradBucketer)
        self.dist = dist if (dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or single_rank_collectives)) else None
        if self.dist is not None:
            for buf in [self.tm.params.data, self.opt.exp_avg, self.opt.exp_avg_sq] + ([self.ema.data] if self.ema else []):
                self.dist.broadcast(buf, 0)
            self.tm.mark_updated()
        self.buckets = GradBucketer(self.tm.params, self.tm.grads, dist=self.dist, single_rank_collectives=single_rank_collectives)
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
        self.exposed_comm_ms = 0.0                # host-side wait for the all-reduces after the backward pass was enqueued
        self.global_step = 0

    def close(self):
        """Detach the milestone events from the context: the context (it lives on in `wrapper.model`) must not record on
        hipEvents whose torch owners have been destroyed."""
        ctx = getattr(getattr(self, "tm", None), "model", None)
        if ctx is not None and getattr(ctx, "_ctx", None) and getattr(self, "_events", None):
            lib.mdgen_train_set_milestone_events(ctx._ctx, None, 0)
        self._events = []
        ref = getattr(getattr(self, "wrapper", None), "trainer", None)
        if ref is not None and ref() in (self, None):
            self.wrapper.trainer = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def milestone_of(self, name: str) -> int:
        return grad_milestone(name, self.wrapper.cfg.num_layers)

    def training_step(self, batch, t=None, x0=None):
        import time
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
        t0 = time.perf_counter()
        scale = self.buckets.finish()
        self.exposed_comm_ms = (time.perf_counter() - t0) * 1e3
        self.opt.step(self.tm.grads, grad_scale=scale)
        self.tm.mark_updated()                    # no hand-back: the training kernels read the flat buffer (bind_params)
        if self.ema is not None:
            self.ema.update()
        self.global_step += 1
        return loss.mean()

    @torch.no_grad()
    def validation_loss(self, batches) -> float:
        """`validation_step` of the reference (wrapper.py:88-97, 107-110): the flow-matching loss of `general_step` over
        `batches` with the EMA weights swapped in when an EMA is kept (restored afterwards); forward only."""
        if self.ema is not None:
            sd = self.ema.state_dict()["params"]
            self.tm.model.load_state_dict({k: v for k, v in sd.items()})   # (bound fp32 master parameters are not touched)
            self.tm._stale = False                # the packs now hold the EMA: no lazy re-pack in front of the forward passes
        tot, n = 0.0, 0
        for batch in batches:
            loss, _ = self.wrapper.general_step(batch, stage="val")
            tot += float(loss.mean())
            n += 1
        if self.ema is not None:
            self.tm.mark_updated()                # the packed sampler weights hold the EMA now: re-pack before the next use
        return tot / max(n, 1)

    # ---- checkpoint / resume (the Lightning layout `NewMDGenWrapper.load_from_checkpoint` reads) ----
    def save_checkpoint(self, path):
        sd = self.tm.state_dict()
        ckpt = {"state_dict": OrderedDict(("model." + k, v.detach().cpu().clone()) for k, v in sd.items()),
                "hyper_parameters": {"args": self.wrapper.args},
                # torch.optim.Adam's own layout (what Lightning stores under this key).  NOT a complete Lightning checkpoint: a
                # Lightning `Trainer.fit(ckpt_path=...)` resume also wants `epoch`, `lr_schedulers`, `loops` and
                # `pytorch-lightning_version`; this file is for `load_from_checkpoint` (inference) and `Trainer.load_checkpoint`
                "optimizer_states": [adam_state_to_torch(self.opt.state_dict(), list(trainable_shapes(self.wrapper.cfg)))],
                "global_step": self.global_step}
        if self.ema is not None:
            e = self.ema.state_dict()
            ckpt["ema"] = {"params": OrderedDict((k, v.detach().cpu().clone()) for k, v in e["params"].items()),
                           "decay": e["decay"]}
        torch.save(ckpt, path)

    def load_checkpoint(self, path):
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        sd = {k[len("model."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("model.")}
        self.tm.params.load_state_dict(sd)
        self.tm.mark_updated()
        if ckpt.get("optimizer_states"):
            self.opt.load_state_dict(ckpt["optimizer_states"][0], order=list(trainable_shapes(self.wrapper.cfg)))
        if self.ema is not None and "ema" in ckpt:
            self.ema.load_state_dict(ckpt["ema"])
        self.global_step = int(ckpt.get("global_step", 0))
        return self


def shard_epoch(order, rank: int, world: int, batch_size: int):
    """Rank `rank`'s strided shard of an epoch's (already permuted) item list and its step count -- the SAME count on every
    rank: the list is cut to n * batch_size * world items first (DistributedSampler(drop_last) semantics), so no rank enters a
    bucketed all-reduce the others have already left behind."""
    n = (len(order) // world) // batch_size
    return order[:n * batch_size * world][rank::world], n


def main(argv=None):
    """`python -m mdgen_amd.train ...` -- counterpart of the reference's launcher (train.py:46-77): the flags of
    `parse_train_args` that concern this path, one process per GPU under `torch.distributed.run` (RANK / LOCAL_RANK /
    WORLD_SIZE from the environment, backend nccl = RCCL), DistributedSampler-style sharding of the shuffled epoch,
    checkpoint per epoch in the Lightning layout.  `--synthetic N`: no dataset, N random-trajectory steps per epoch at
    (`--batch_size`, `--num_frames`, `--crop`) -- what the bench's training leg and the multi-GPU smoke run use."""
    import argparse
    import os
    import time
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--train_batches", type=int, default=None)
    ap.add_argument("--batch_size", type=int, default=8, help="per process (Lightning DDP semantics)")
    ap.add_argument("--grad_clip", type=float, default=1.0)
    ap.add_argument("--adamW", action="store_true")
    ap.add_argument("--ema", action="store_true")
    ap.add_argument("--ema_decay", type=float, default=0.999)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--train_split", default=None)
    ap.add_argument("--data_dir", default=None)
    ap.add_argument("--num_frames", type=int, default=50)
    ap.add_argument("--crop", type=int, default=256)
    ap.add_argument("--suffix", default="")
    ap.add_argument("--atlas", action="store_true")
    ap.add_argument("--copy_frames", action="store_true")
    ap.add_argument("--frame_interval", type=int, default=None)
    ap.add_argument("--overfit", action="store_true")
    ap.add_argument("--overfit_frame", action="store_true")
    ap.add_argument("--num_layers", type=int, default=5)
    ap.add_argument("--abs_pos_emb", action="store_true")
    ap.add_argument("--prepend_ipa", action="store_true")
    ap.add_argument("--sim_condition", action="store_true")
    ap.add_argument("--tps_condition", action="store_true")
    ap.add_argument("--out_dir", default=os.environ.get("MODEL_DIR", "."))
    ap.add_argument("--ckpt_freq", type=int, default=1)
    ap.add_argument("--print_freq", type=int, default=100)
    ap.add_argument("--seed", type=int, default=137)
    ap.add_argument("--synthetic", type=int, default=0, metavar="N", help="N synthetic steps per epoch instead of a dataset")
    ap.add_argument("--val_split", default=None)
    ap.add_argument("--val_batches", type=int, default=None)
    ap.add_argument("--no_validate", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--matmul_precision", default="medium", choices=["medium", "highest"],
                    help="medium (what the reference's train.py:13 sets): bf16-operand matrix products, fp32 accumulation / "
                         "weights / activations (library option train_precision = 16); highest: exact fp32 products")
    ap.add_argument("--single_device", action="store_true",
                    help="every rank on cuda:0 (with --backend gloo: lets a one-GPU box run the multi-rank path; RCCL refuses it)")
    a = ap.parse_args(argv)
    if not a.prepend_ipa or a.sim_condition == a.tps_condition:
        raise SystemExit("the accelerated path trains the prepend_ipa models with exactly one of --sim_condition / --tps_condition")
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    if a.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist_.init_process_group(a.backend, **({"device_id": dev} if a.backend == "nccl" else {}))
        dist = dist_
    from .synthetic import synth_state_dict
    from .wrapper import NewMDGenWrapper
    cfg = ModelConfig(crop=a.crop, num_frames=a.num_frames, abs_pos_emb=a.abs_pos_emb, num_layers=a.num_layers,
                      sim_condition=a.sim_condition, tps_condition=a.tps_condition)
    if a.ckpt:
        w = NewMDGenWrapper.load_from_checkpoint(a.ckpt, device=dev)
    else:
        w = NewMDGenWrapper(cfg, device=dev)
        w.load_model_state_dict(synth_state_dict(cfg, a.seed))    # (every rank: same seed; rank 0's values are broadcast anyway)
    tr = Trainer(w, lr=a.lr, adamw=a.adamW, grad_clip=a.grad_clip, ema_decay=a.ema_decay if a.ema else None, dist=dist)
    if a.ckpt:
        tr.load_checkpoint(a.ckpt)
    tr.tm.model.set_option("train_precision", 16 if a.matmul_precision == "medium" else 32)
    import numpy as np
    np.random.seed(a.seed + rank)
    torch.manual_seed(a.seed + rank)
    if a.synthetic:
        from . import synthetic as _bench   # synthetic conditioning batch (frames, torsions) shared with bench.py
        def batches(epoch):
            for i in range(a.synthetic):
                yield _bench.synth_batch(a.batch_size, a.num_frames, a.crop, 16 if a.crop >= 64 else 0, dev,
                                         seed=1000 * epoch + 10 * i + rank, tps=a.tps_condition)
    else:
        from .dataset import MDGenDataset
        ds = MDGenDataset(a, split=a.train_split, device=dev)
        def batches(epoch):
            g = torch.Generator().manual_seed(a.seed + epoch)      # same permutation on every rank, disjoint strided shards
            order, n = shard_epoch(torch.randperm(len(ds), generator=g).tolist(), rank, world, a.batch_size)
            if a.train_batches:
                n = min(n, a.train_batches)
            for i in range(n):
                items = [ds[j] for j in order[i * a.batch_size:(i + 1) * a.batch_size]]
                yield {k: torch.stack([it[k] for it in items]) for k in ("torsions", "torsion_mask", "trans", "rots", "seqres", "mask")}
    val_ds = None
    if not a.no_validate and not a.synthetic and a.val_split:
        from .dataset import MDGenDataset as _DS
        val_ds = _DS(a, split=a.val_split, device=dev)

    def val_batches(epoch):
        if a.synthetic:
            for i in range(min(a.synthetic, a.val_batches or 2)):
                yield _bench.synth_batch(a.batch_size, a.num_frames, a.crop, 16 if a.crop >= 64 else 0, dev,
                                         seed=900000 + 10 * i + rank, tps=a.tps_condition)
        elif val_ds is not None:
            idx, n = shard_epoch(list(range(len(val_ds))), rank, world, a.batch_size)
            if a.val_batches:
                n = min(n, a.val_batches)
            for i in range(n):
                items = [val_ds[j] for j in idx[i * a.batch_size:(i + 1) * a.batch_size]]
                yield {k: torch.stack([it[k] for it in items]) for k in ("torsions", "torsion_mask", "trans", "rots", "seqres", "mask")}

    for epoch in range(a.epochs):
        t0, n, comm = time.perf_counter(), 0, 0.0
        for batch in batches(epoch):
            loss = tr.training_step(batch)
            n += 1
            comm += tr.exposed_comm_ms
            if rank == 0 and n % a.print_freq == 0:
                print(f"epoch {epoch} step {n}: loss {float(loss):.4f}", flush=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        vloss = None
        if not a.no_validate and (a.synthetic or val_ds is not None):
            vloss = tr.validation_loss(val_batches(epoch))
            if dist:
                from .sharding import sum_over_ranks
                vloss = sum_over_ranks(vloss, dist, dev if a.backend == "nccl" else None) / world
        if rank == 0:
            print(f"epoch {epoch}: {n} steps, {dt / max(n, 1) * 1e3:.1f} ms/step, exposed all-reduce wait {comm / max(n, 1):.2f} ms/step, "
                  f"world {world}" + (f", validation loss {vloss:.4f}" if vloss is not None else ""), flush=True)
            if (epoch + 1) % a.ckpt_freq == 0:
                os.makedirs(a.out_dir, exist_ok=True)
                tr.save_checkpoint(os.path.join(a.out_dir, f"epoch={epoch}.ckpt"))
    tr.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
