"""`LatentMDGenModel` -- drop-in for `mdgen.model.latent_model.LatentMDGenModel` (inference path).

`forward` / `forward_inference` keep the reference signature (latent_model.py:212-216, 263-269) and
dispatch to `mdgen_denoiser_forward`; `sample_euler` runs the whole S-step Euler rollout
(`mdgen_sample_euler`, optionally replayed from a hipGraph).  All arithmetic is in libmdgen_amd.so;
this class owns the opaque context, the workspaces and persistent staging buffers (stable pointers
are what make hipGraph replay possible).
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, Optional

import torch

from . import _lib as L
from ._lib import lib, check, ptr, require_cuda
from .config import ModelConfig
from .rigid_utils import Rigid


def _frames(fr):
    """Rigid | (rot, trans) | None -> (rot [B,L,3,3], trans [B,L,3]) contiguous fp32."""
    if fr is None:
        return None, None
    if isinstance(fr, Rigid):
        r, t = fr.get_rots().get_rot_mats(), fr.get_trans()
    else:
        r, t = fr
    return r.to(torch.float32).contiguous(), t.to(torch.float32).contiguous()


class LatentMDGenModel:
    def __init__(self, cfg: ModelConfig, device: Optional[torch.device] = None, precision: str = "bf16"):
        """`precision`: "bf16" (default) -- bf16 MFMA operands, fp32 accumulate / softmax / LayerNorm / residual stream;
        "fp32" -- the reference's own arithmetic on fp32 operands (csrc/k_fp32.hip), a tolerance mode ~10x slower.
        A model built with "fp32" keeps fp32 weight copies and can switch at run time (`set_precision`)."""
        if isinstance(cfg, ModelConfig) is False:
            cfg = ModelConfig.from_args(cfg)
        if precision not in ("bf16", "fp32"):
            raise L.MdgenError(f"precision must be 'bf16' or 'fp32', got {precision!r}")
        self.cfg = cfg
        self.precision = precision
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise L.MdgenError("mdgen_amd.LatentMDGenModel needs a GPU (gfx950); there is no CPU path")
        d = L.ModelDesc(cfg.embed_dim, cfg.mha_heads, cfg.num_layers, cfg.latent_dim, cfg.ipa_heads,
                        cfg.ipa_head_dim, cfg.ipa_qk, cfg.ipa_v, int(cfg.abs_pos_emb), cfg.crop,
                        int(cfg.tps_condition), float(cfg.time_multiplier))
        self._ctx = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.mdgen_ctx_create(C.byref(self._ctx), C.byref(d)))
        if precision == "fp32":
            check(lib.mdgen_ctx_set_option(self._ctx, b"keep_fp32_weights", 1))
        # small LRU caches, keyed independently: a workspace per (B, T, L, S, t_shared) and the persistent staging
        # buffers of the Euler rollout per (B, T, L) -- stable device pointers are what lets a captured hipGraph be
        # replayed, so alternating shapes (ATLAS inference runs on full sequences, L = 39..724) or alternating
        # forward() / sample_euler() calls must not evict each other's buffers on every call.
        self._ws: "OrderedDict[tuple, torch.Tensor]" = OrderedDict()
        self._stage: "OrderedDict[tuple, dict]" = OrderedDict()
        self.max_cached_shapes = 4
        self.poison_workspace = False     # tests: fill new workspaces with 0xFF (NaN in fp32 and bf16)
        self._side = None
        self._loaded = False
        self._pre_run = None              # optional hook run before every network evaluation (train.TrainableModel: lazy weight refresh)

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                lib.mdgen_ctx_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    # ---- weights ----------------------------------------------------------------------------
    def weight_names(self):
        n = lib.mdgen_ctx_num_weights(self._ctx)
        return [lib.mdgen_ctx_weight_name(self._ctx, i).decode() for i in range(n)]

    def load_state_dict(self, sd, strict: bool = True):
        """`sd`: the reference's `LatentMDGenModel.state_dict()` (keys of SURVEY.md section 8(b))."""
        names = set(self.weight_names())
        missing = [k for k in names if k not in sd]
        unexpected = [k for k in sd if k not in names]
        if strict and (missing or unexpected):
            raise L.MdgenError(f"state_dict mismatch: missing={missing[:5]} unexpected={unexpected[:5]}")
        keep = []
        with torch.cuda.device(self.device):
            s = L.stream_ptr()
            for k in names:
                if k not in sd:
                    continue
                t = sd[k].detach().to(device=self.device, dtype=torch.float32).contiguous()
                keep.append(t)
                shp = (C.c_int64 * t.dim())(*t.shape)
                check(lib.mdgen_ctx_set_weight(self._ctx, k.encode(), ptr(t), shp, t.dim(), s))
            check(lib.mdgen_ctx_finalize(self._ctx, s))
            torch.cuda.current_stream().synchronize()   # packing kernels read `keep` asynchronously
        self._loaded = True
        self._state_names = list(sd.keys())
        if self.precision == "fp32":
            self.set_precision("fp32")
        return self

    def set_precision(self, precision: str):
        """Switch between the bf16 MFMA path and the fp32 path (the latter only on a model constructed with
        precision="fp32", which keeps the fp32 weight copies)."""
        if precision not in ("bf16", "fp32"):
            raise L.MdgenError(f"precision must be 'bf16' or 'fp32', got {precision!r}")
        check(lib.mdgen_ctx_set_option(self._ctx, b"precision", 32 if precision == "fp32" else 16))
        self.precision = precision
        return self

    def eval(self):
        return self

    def to(self, device):
        if torch.device(device).type != "cuda":
            raise L.MdgenError("mdgen_amd has no CPU path")
        return self

    def set_option(self, name: str, value: int):
        """Library run-time options (include/mdgen_amd.h `mdgen_ctx_set_option`): "streams", "residue_l4_path",
        "attention_path", "mlp_path", "precision", "keep_fp32_weights"."""
        check(lib.mdgen_ctx_set_option(self._ctx, name.encode(), int(value)))
        return self

    # ---- measurement ------------------------------------------------------------------------
    def profile(self, on: bool):
        check(lib.mdgen_profile_enable(self._ctx, int(on)))

    def phase_trace(self, buf):
        """Arm the one-shot phase trace: the next trunk MLP launch writes [wave][32] s_memtime stamps into the
        int64 CUDA tensor `buf` (None cancels).  Measurement only; use with use_graph=False."""
        if buf is None:
            check(lib.mdgen_profile_phase_trace(self._ctx, None, 0))
        else:
            assert buf.is_cuda and buf.dtype == torch.int64 and buf.is_contiguous()
            check(lib.mdgen_profile_phase_trace(self._ctx, buf.data_ptr(), buf.numel()))

    def profile_report(self) -> dict:
        """{"kernel class": {"count": n, "ms": total}} measured with hipEvents on the launch stream."""
        import json
        buf = C.create_string_buffer(1 << 16)
        with torch.cuda.device(self.device):
            check(lib.mdgen_profile_report(self._ctx, L.stream_ptr(), buf, len(buf)))
        rep = json.loads(buf.value.decode())
        # "@context" is not a kernel class: the placement-probe result (xcd_round_robin), the CU count and the number of
        # split-panel launches (k_mlp8<., 3>) since the last report
        self.context_info = rep.pop("@context", None)
        return rep

    # ---- workspace --------------------------------------------------------------------------
    def workspace_layout(self, B, T, L_, S, t_shared):
        lay = L.WsLayout()
        sh = L.Shape(B, T, L_)
        check(lib.mdgen_workspace_layout(self._ctx, C.byref(sh), S, int(t_shared), C.byref(lay)))
        return lay

    def _workspace(self, B, T, L_, S, t_shared):
        key = (B, T, L_, S, int(t_shared))
        ws = self._ws.get(key)
        lay = self.workspace_layout(B, T, L_, S, t_shared)   # (options may have changed what the call needs: mlp_fold, precision)
        if ws is not None and ws.numel() < lay.total_bytes:
            del self._ws[key]
            ws = None
        if ws is None:
            while len(self._ws) >= self.max_cached_shapes:
                self._ws.popitem(last=False)                  # least recently used
            ws = torch.empty(lay.total_bytes, dtype=torch.uint8, device=self.device)
            if self.poison_workspace:
                ws.fill_(255)
            self._ws[key] = ws
        else:
            self._ws.move_to_end(key)
        return ws

    # ---- forward ----------------------------------------------------------------------------
    def _check_inputs(self, x, mask, x_cond, x_cond_mask, aatype):
        if x_cond is None or x_cond_mask is None or aatype is None:
            raise L.MdgenError("x_cond, x_cond_mask and aatype are required (conditional models only)")
        if x.dim() != 4 or x.shape[-1] != self.cfg.latent_dim:
            raise L.MdgenError(f"x must be (B,T,L,{self.cfg.latent_dim}), got {tuple(x.shape)}")
        B, T, L_, _ = x.shape
        if tuple(mask.shape) != (B, T, L_):
            raise L.MdgenError(f"mask must be (B,T,L)={B, T, L_}, got {tuple(mask.shape)}")
        return B, T, L_

    def _rel7(self, rel_quats, B, L_):
        """Optional relative-frame inputs of the two-sided model, (2,B,L,7): [0] = (start^-1 o end).to_tensor_7(), [1] =
        (end^-1 o start).to_tensor_7() (latent_model.py:193-195) as the CALLER's reference computed them -- their quaternion
        sign is torch.linalg.eigh's (rigid_utils.py:191-210).  None: the library computes them with w >= 0."""
        if rel_quats is None:
            return None
        if not self.cfg.tps_condition:
            raise L.MdgenError("rel_quats is an input of tps_condition models only")
        if tuple(rel_quats.shape) != (2, B, L_, 7):
            raise L.MdgenError(f"rel_quats must be (2,B,L,7)={(2, B, L_, 7)}, got {tuple(rel_quats.shape)}")
        require_cuda(rel_quats)
        return rel_quats.to(torch.float32).contiguous()

    def forward(self, x, t, mask, start_frames=None, end_frames=None, x_cond=None, x_cond_mask=None, aatype=None,
                return_trace: bool = False, rel_quats=None):
        """latent_model.py:212-260 (non-design path).  Returns the velocity (B,T,L,D) fp32."""
        if self._pre_run is not None:
            self._pre_run()
        B, T, L_ = self._check_inputs(x, mask, x_cond, x_cond_mask, aatype)
        require_cuda(x, t, mask, x_cond, x_cond_mask, aatype)
        sr, st = _frames(start_frames)
        er, et = _frames(end_frames)
        if sr is None:
            raise L.MdgenError("start_frames is required (prepend_ipa models)")
        x = x.to(torch.float32).contiguous()
        t = t.to(torch.float32).contiguous()
        mask = mask.to(torch.float32).contiguous()
        x_cond = x_cond.to(torch.float32).contiguous()
        x_cond_mask = x_cond_mask.to(torch.int64).contiguous()
        aatype = aatype.to(torch.int64).contiguous()
        out = torch.empty_like(x)
        ws = self._workspace(B, T, L_, 1, B == 1)
        nl = self.cfg.num_layers
        tr_h = torch.empty(nl + 1, B * T * L_, self.cfg.embed_dim, device=x.device) if return_trace else None
        tr_i = torch.empty(B * L_, self.cfg.embed_dim, device=x.device) if return_trace else None
        sh = L.Shape(B, T, L_)
        rel7 = self._rel7(rel_quats, B, L_)
        with torch.cuda.device(self.device):
            check(lib.mdgen_denoiser_forward(self._ctx, C.byref(sh), ptr(x), ptr(t), ptr(mask), ptr(sr), ptr(st),
                                             ptr(er), ptr(et), ptr(rel7), ptr(x_cond), ptr(x_cond_mask), ptr(aatype), ptr(out),
                                             ptr(tr_h), ptr(tr_i), ptr(ws), ws.numel(), L.stream_ptr()))
        if return_trace:
            C_ = self.cfg.embed_dim
            trace = {"ipa_out": tr_i.view(B, L_, C_)}
            for i in range(nl + 1):
                trace[f"h{i}"] = tr_h[i].view(B, T, L_, C_)
            return out, trace
        return out

    forward_inference = forward
    __call__ = forward

    # ---- Euler rollout ----------------------------------------------------------------------
    def sample_euler(self, zs, num_steps: int, mask=None, start_frames=None, end_frames=None, x_cond=None,
                     x_cond_mask=None, aatype=None, use_graph: bool = True, rel_quats=None):
        """x <- zs; for i < S: x += (t[i+1]-t[i]) * model(x, t[i]) on t = linspace(0,1,S+1); returns x.
        (transport.py:408-451 + integrators.py:95-114 + torchdiffeq fixed-grid Euler.)"""
        if self._pre_run is not None:
            self._pre_run()
        B, T, L_ = self._check_inputs(zs, mask, x_cond, x_cond_mask, aatype)
        require_cuda(zs, mask, x_cond, x_cond_mask, aatype)
        S = int(num_steps)
        sr, st = _frames(start_frames)
        er, et = _frames(end_frames)
        if sr is None:
            raise L.MdgenError("start_frames is required")
        ws = self._workspace(B, T, L_, S, True)
        key = (B, T, L_)
        stg = self._stage.get(key)
        if stg is not None:
            self._stage.move_to_end(key)
        else:   # persistent staging buffers: stable device pointers => hipGraph replay
            while len(self._stage) >= self.max_cached_shapes:
                self._stage.popitem(last=False)
            dev = self.device
            stg = dict(
                x=torch.empty(B, T, L_, self.cfg.latent_dim, device=dev), mask=torch.empty(B, T, L_, device=dev),
                sr=torch.empty(B, L_, 3, 3, device=dev), st=torch.empty(B, L_, 3, device=dev),
                er=torch.empty(B, L_, 3, 3, device=dev), et=torch.empty(B, L_, 3, device=dev),
                x_cond=torch.empty(B, T, L_, self.cfg.latent_dim, device=dev),
                x_cond_mask=torch.empty(B, T, L_, dtype=torch.int64, device=dev),
                aatype=torch.empty(B, L_, dtype=torch.int64, device=dev),
                rel7=torch.empty(2, B, L_, 7, device=dev) if self.cfg.tps_condition else None)
            self._stage[key] = stg
        stg["x"].copy_(zs)
        stg["mask"].copy_(mask)
        stg["sr"].copy_(sr)
        stg["st"].copy_(st)
        if er is not None:
            stg["er"].copy_(er)
            stg["et"].copy_(et)
        stg["x_cond"].copy_(x_cond)
        stg["x_cond_mask"].copy_(x_cond_mask)
        stg["aatype"].copy_(aatype)
        has_end = er is not None
        rel7 = self._rel7(rel_quats, B, L_)
        if rel7 is not None:
            stg["rel7"].copy_(rel7)
        sh = L.Shape(B, T, L_)
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream()
            if use_graph:
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.device)
                self._side.wait_stream(cur)
                stream = self._side
            else:
                stream = cur
            check(lib.mdgen_sample_euler(
                self._ctx, C.byref(sh), S, ptr(stg["x"]), ptr(stg["mask"]), ptr(stg["sr"]), ptr(stg["st"]),
                ptr(stg["er"]) if has_end else None, ptr(stg["et"]) if has_end else None,
                ptr(stg["rel7"]) if rel7 is not None else None, ptr(stg["x_cond"]),
                ptr(stg["x_cond_mask"]), ptr(stg["aatype"]), ptr(ws), ws.numel(), int(use_graph),
                C.c_void_p(stream.cuda_stream)))
            if use_graph:
                cur.wait_stream(self._side)
        return stg["x"].clone()

    # ---- multi-block rollout -----------------------------------------------------------------
    def rollout_euler(self, zs, num_steps: int, mask, cond_rots, cond_trans, cond_torsions, seqres, tables,
                      use_graph: bool = True):
        """`mdgen_rollout_euler`: R = zs.shape[0] chained T-frame blocks in one call (one hipGraph), the driver loop
        of sim_inference.py:100-113.  zs (R,B,T,L,D) noise; mask (B,T,L); cond_* the first conditioning frame
        (B,L,...); seqres (B,L) int64; tables: dict of residue tables on the device (geometry.residue_tables).
        Returns (atom14 (B, R*T, L, 14, 3), samples (R,B,T,L,D), next conditioning frame dict)."""
        if self._pre_run is not None:
            self._pre_run()
        if zs.dim() != 5 or zs.shape[-1] != self.cfg.latent_dim:
            raise L.MdgenError(f"zs must be (R,B,T,L,{self.cfg.latent_dim}), got {tuple(zs.shape)}")
        R, B, T, L_, D = zs.shape
        require_cuda(zs, mask, cond_rots, cond_trans, cond_torsions, seqres)
        S = int(num_steps)
        ws = self._workspace(B, T, L_, S, True)
        key = ("rollout", R, B, T, L_)
        stg = self._stage.get(key)
        if stg is not None:
            self._stage.move_to_end(key)
        else:
            while len(self._stage) >= self.max_cached_shapes:
                self._stage.popitem(last=False)
            dev = self.device
            stg = dict(zs=torch.empty(R, B, T, L_, D, device=dev), mask=torch.empty(B, T, L_, device=dev),
                       rots=torch.empty(B, L_, 3, 3, device=dev), trans=torch.empty(B, L_, 3, device=dev),
                       tors=torch.empty(B, L_, 7, 2, device=dev), seqres=torch.empty(B, L_, dtype=torch.int64, device=dev),
                       x_cond=torch.empty(B, T, L_, D, device=dev),
                       x_cond_mask=torch.empty(B, T, L_, dtype=torch.int64, device=dev),
                       atom14=torch.empty(B, R * T, L_, 14, 3, device=dev))
            self._stage[key] = stg
        stg["zs"].copy_(zs)
        stg["mask"].copy_(mask)
        stg["rots"].copy_(cond_rots)
        stg["trans"].copy_(cond_trans)
        stg["tors"].copy_(cond_torsions)
        stg["seqres"].copy_(seqres)
        tb = L.ResidueTables(*[tables[n].data_ptr() for n in (
            "default_frames", "lit_positions", "atom14_group", "atom14_mask", "atom37_to_atom14", "atom37_mask",
            "chi_atom_indices", "chi_angles_mask")])
        sh = L.Shape(B, T, L_)
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream()
            if use_graph:
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.device)
                self._side.wait_stream(cur)
                stream = self._side
            else:
                stream = cur
            check(lib.mdgen_rollout_euler(
                self._ctx, C.byref(sh), S, R, ptr(stg["zs"]), ptr(stg["mask"]), ptr(stg["rots"]), ptr(stg["trans"]),
                ptr(stg["tors"]), ptr(stg["seqres"]), ptr(stg["x_cond"]), ptr(stg["x_cond_mask"]), C.byref(tb),
                ptr(stg["atom14"]), ptr(ws), ws.numel(), int(use_graph), C.c_void_p(stream.cuda_stream)))
            if use_graph:
                cur.wait_stream(self._side)
        nxt = {"rots": stg["rots"].clone(), "trans": stg["trans"].clone(), "torsions": stg["tors"].clone()}
        return stg["atom14"].clone(), stg["zs"].clone(), nxt
