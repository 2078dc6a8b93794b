"""`NewMDGenWrapper` -- drop-in for the inference surface of `mdgen.wrapper.NewMDGenWrapper`
(wrapper.py:175-221 `__init__`, :283-365 `prep_batch`, :405-484 `inference`) plus
`load_from_checkpoint` for Lightning-format checkpoints (SURVEY.md section 5 "Checkpoint / resume"),
without Lightning.  The training step (`general_step` forward half here; backward, optimiser, EMA, DDP in
`mdgen_amd/train.py`) runs in the same library; Lightning's logging hooks are out of scope."""
from __future__ import annotations

import argparse
import ctypes as C
from functools import partial

import torch

from . import _lib as L
from ._lib import lib, launch, ptr, require_cuda
from .config import ModelConfig
from .geometry import samples_to_atom14
from .model import LatentMDGenModel
from .rigid_utils import Rigid, Rotation
from .transport import Sampler, create_transport

_BACKFILL = ["inpainting", "no_torsion", "hyena", "no_aa_emb", "supervise_all_torsions", "supervise_no_torsions",
             "design_key_frames", "no_design_torsion", "cond_interval", "mpnn", "dynamic_mpnn", "no_offsets",
             "no_frames", "design", "tps_condition", "sim_condition", "abs_pos_emb", "prepend_ipa", "oracle"]
_UNSUPPORTED = ["inpainting", "hyena", "no_aa_emb", "design_key_frames", "mpnn", "dynamic_mpnn", "no_offsets",
                "no_frames", "design", "no_torsion", "no_design_torsion", "interleave_ipa",
                "abs_time_emb", "oracle", "no_rope"]


def default_args(cfg: ModelConfig) -> argparse.Namespace:
    a = argparse.Namespace(**cfg.to_dict())
    a.path_type, a.prediction, a.sampling_method = "GVP", "velocity", "euler"
    return a


class NewMDGenWrapper:
    def __init__(self, args, device="cuda", precision="bf16"):
        if isinstance(args, ModelConfig):
            args = default_args(args)
        for k in _BACKFILL:                       # wrapper.py:178-194: newer flags default to False
            if not hasattr(args, k):
                setattr(args, k, False)
        bad = [k for k in _UNSUPPORTED if getattr(args, k, False)]
        if bad:
            raise L.MdgenError(f"flags outside the accelerated sampler path: {bad}")
        if not getattr(args, "prepend_ipa", False):
            raise L.MdgenError("the accelerated path implements the prepend_ipa models (README.md:48,60)")
        # latent_model.py:183-207 tests sim_condition first and tps_condition second, and only these two branches
        # set up the IPA inputs; wrapper.py:339-342 marks the conditioning frames for exactly one of them.  A
        # checkpoint with both flags or neither would silently run something else here: reject it.
        if bool(args.sim_condition) == bool(args.tps_condition):
            raise L.MdgenError("exactly one of sim_condition / tps_condition must be set "
                               f"(got sim_condition={args.sim_condition}, tps_condition={args.tps_condition})")
        self.args = args
        self.cfg = ModelConfig.from_args(args)
        self.latent_dim = self.cfg.latent_dim
        self.device = torch.device(device)
        self.model = LatentMDGenModel(self.cfg, self.device, precision=precision)
        self.transport = create_transport(args, getattr(args, "path_type", "GVP"), getattr(args, "prediction", "velocity"))
        self.transport_sampler = Sampler(self.transport)

    # Lightning surface used by the drivers (sim_inference.py:129-130)
    @classmethod
    def load_from_checkpoint(cls, path, device="cuda", precision="bf16"):
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        args = ckpt["hyper_parameters"]["args"]
        w = cls(args, device=device, precision=precision)
        sd = {k[len("model."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("model.")}
        w.load_model_state_dict(sd)
        w.ema_state = ckpt.get("ema")            # wrapper.py:120-124 `on_load_checkpoint`
        return w

    # wrapper.py:64-76: validation runs on the EMA weights and restores the raw ones afterwards
    def load_ema_weights(self, ema_params=None):
        """Swap the EMA parameters in (`ema_params`: a full state dict; default: the checkpoint's `ema['params']`)."""
        if ema_params is None:
            if not getattr(self, "ema_state", None):
                raise L.MdgenError("no EMA state: the checkpoint has no 'ema' entry (trained without --ema)")
            ema_params = self.ema_state["params"]
        if not hasattr(self, "model_state_dict"):
            raise L.MdgenError("load_ema_weights needs the raw weights to restore later: use load_model_state_dict()")
        # the weights to restore are the CURRENT ones: a Trainer attached to this wrapper has moved on from what was loaded
        tr = getattr(self, "trainer", None)
        tr = tr() if tr is not None else None          # (`Trainer` leaves a weak reference here)
        self.cached_weights = ({k: v.detach().clone() for k, v in tr.tm.state_dict().items()} if tr is not None
                               else self.model_state_dict)
        self.model.load_state_dict({k: v for k, v in ema_params.items()})
        return self

    def restore_cached_weights(self):
        if getattr(self, "cached_weights", None) is not None:
            self.model.load_state_dict(self.cached_weights)
            self.cached_weights = None
        return self

    def load_model_state_dict(self, sd):
        """Hand the `model.*` tensors to the library and remember them (the training step starts from them)."""
        self.model.load_state_dict(sd)
        self.model_state_dict = {k: v.detach().clone() for k, v in sd.items()}
        return self

    def eval(self):
        return self

    def to(self, device):
        self.model.to(device)
        return self

    # ---------------------------------------------------------------------------------------------
    def prep_batch(self, batch):
        """wrapper.py:283-365 (sim_condition / tps_condition paths)."""
        trans = batch["trans"].to(torch.float32).contiguous()
        rots = batch["rots"].to(torch.float32).contiguous()
        tors = batch["torsions"].to(torch.float32).contiguous()
        require_cuda(trans, rots, tors)
        B, T, L_ = trans.shape[:3]
        tps = bool(self.args.tps_condition)
        D = self.latent_dim
        dev = trans.device
        latents = torch.empty(B, T, L_, D, device=dev)
        x_cond = torch.empty(B, T, L_, D, device=dev)
        cond_mask = torch.empty(B, T, L_, dtype=torch.int64, device=dev)
        sh = L.Shape(B, T, L_)
        launch(lib.mdgen_prep_latents, trans, C.byref(sh), int(tps), int(getattr(self.args, "cond_interval", 0) or 0),
               ptr(rots), ptr(trans), ptr(tors), ptr(latents),
               ptr(x_cond), ptr(cond_mask))
        rigids = Rigid(Rotation(rot_mats=rots), trans)
        mask = batch["mask"].to(torch.float32)
        frame_lm = mask.unsqueeze(-1).expand(-1, -1, 7)
        if tps:
            frame_lm = torch.cat([frame_lm, frame_lm], -1)
        tors_lm = batch["torsion_mask"].unsqueeze(-1).expand(-1, -1, -1, 2).reshape(B, L_, 14)
        if getattr(self.args, "supervise_all_torsions", False):
            tors_lm = torch.ones_like(tors_lm)
        elif getattr(self.args, "supervise_no_torsions", False):
            tors_lm = torch.zeros_like(tors_lm)
        loss_mask = torch.cat([frame_lm, tors_lm.to(frame_lm.dtype)], -1).unsqueeze(1).expand(-1, T, -1, -1)
        return {
            "rigids": rigids,
            "latents": latents,
            "loss_mask": loss_mask,
            "model_kwargs": {
                "start_frames": rigids[:, 0],
                "end_frames": rigids[:, -1],
                "mask": mask.unsqueeze(1).expand(-1, T, -1),
                "aatype": batch["seqres"],
                "x_cond": x_cond,
                "x_cond_mask": cond_mask,
            },
        }

    def general_step(self, batch, stage="val", t=None, x0=None):
        """The forward half of wrapper.py:367-384 `general_step`: `prep_batch` -> `transport.training_losses`
        (flow-matching target, model forward, masked MSE).  Returns the loss per sample (B,) and the term dict.
        The backward pass + optimiser step of the same quantities is `mdgen_amd.train.Trainer.training_step`; `stage` is
        informational; `t` and `x0` may be fixed for reproducibility (the reference draws them inside the transport)."""
        prep = self.prep_batch(batch)
        out = self.transport.training_losses(model=self.model.forward, x1=prep["latents"], aatype1=None,
                                             mask=prep["loss_mask"], model_kwargs=prep["model_kwargs"], t=t, x0=x0)
        return out["loss"], out

    def inference(self, batch, zs=None, num_steps=None, use_graph=True, rel_quats=None):
        """wrapper.py:405-484.  Extra keywords (defaults reproduce the reference): `zs` explicit noise
        (reference: device randn, wrapper.py:439), `num_steps` Euler steps S (reference: 50 grid points = 49
        steps, wrapper.py:441-442 / transport.py:412), `rel_quats` (two-sided models): the relative-frame 7-vectors
        (2,B,L,7) as the caller's reference computes them (`LatentMDGenModel._rel7`; default: w >= 0 convention)."""
        prep = self.prep_batch(batch)
        rigids = prep["rigids"]
        B, T, L_ = rigids.shape
        dev = prep["latents"].device
        if zs is None:
            zs = torch.randn(B, T, L_, self.latent_dim, device=dev)
        # wrapper.py:441 samples with args.sampling_method (argparse default 'dopri5', parsing.py:102) on the
        # solver's own 50-point grid.  Only fixed-grid Euler exists here: without an explicit `num_steps` a
        # non-Euler checkpoint is rejected instead of being silently sampled with a different solver.
        method = getattr(self.args, "sampling_method", "euler")
        if num_steps is None and method != "euler":
            raise L.MdgenError(f"checkpoint args say sampling_method={method!r}; this build implements fixed-grid "
                               "Euler only -- pass num_steps=... (CLI: --num_steps) to sample with Euler explicitly")
        S = 49 if num_steps is None else int(num_steps)
        kw = dict(prep["model_kwargs"])
        kw["mask"] = kw["mask"].contiguous()
        if not self.args.tps_condition:
            kw["end_frames"] = None
        if rel_quats is not None:
            kw["rel_quats"] = rel_quats
        sample_fn = self.transport_sampler.sample_ode(sampling_method="euler", num_steps=S + 1)
        samples = sample_fn(zs, partial(self.model.forward_inference, **kw), use_graph=use_graph)[-1]
        r0 = rigids[:, 0]
        atom14 = samples_to_atom14(samples, r0.get_rots().get_rot_mats(), r0.get_trans(), batch["seqres"],
                                   bool(self.args.tps_condition))
        aa_out = batch["seqres"][:, None].expand(B, T, L_)
        self.last_samples = samples
        return atom14, aa_out

    def rollout(self, batch, num_frames: int, num_rollouts: int, num_steps=None, zs=None, use_graph=True,
                return_next: bool = False):
        """`num_rollouts` chained blocks of `num_frames` frames from the B = 1-frame conditioning batch of
        `sim_inference.get_batch` (torsions (B,1,L,7,2), trans (B,1,L,3), rots (B,1,L,3,3), seqres, mask (B,L)):
        the loop of sim_inference.py:100-113 (`rollout` :61-98 per block) as ONE library call
        (`mdgen_rollout_euler`), captured as one hipGraph.  `zs`: optional noise (num_rollouts, B, T, L, D)
        (reference: a fresh device randn per block, wrapper.py:439).  Returns atom14 (B, num_rollouts*T, L, 14, 3)
        [, the batch that would condition the next block]."""
        from .geometry import residue_tables
        if self.args.tps_condition or getattr(self.args, "cond_interval", None):
            raise L.MdgenError("rollout() is the forward-simulation driver (sim_condition models without cond_interval)")
        method = getattr(self.args, "sampling_method", "euler")
        if num_steps is None and method != "euler":
            raise L.MdgenError(f"checkpoint args say sampling_method={method!r}; pass num_steps=... to sample with Euler")
        S = 49 if num_steps is None else int(num_steps)
        tors = batch["torsions"][:, 0].to(torch.float32)
        trans = batch["trans"][:, 0].to(torch.float32)
        rots = batch["rots"][:, 0].to(torch.float32)
        require_cuda(tors, trans, rots)
        B, L_ = trans.shape[:2]
        T, R = int(num_frames), int(num_rollouts)
        dev = trans.device
        if zs is None:
            zs = torch.randn(R, B, T, L_, self.latent_dim, device=dev)
        mask = batch["mask"].to(torch.float32).unsqueeze(1).expand(-1, T, -1)
        atom14, samples, nxt = self.model.rollout_euler(zs, S, mask, rots, trans, tors, batch["seqres"],
                                                        residue_tables(dev), use_graph=use_graph)
        self.last_samples = samples
        if not return_next:
            return atom14
        new = dict(batch)
        new["trans"], new["rots"], new["torsions"] = nxt["trans"][:, None], nxt["rots"][:, None], nxt["torsions"][:, None]
        return atom14, new
