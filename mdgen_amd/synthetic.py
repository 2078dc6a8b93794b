"""Seeded synthetic weights for the denoiser (there is no network access for checkpoints).

Keys and shapes are exactly the reference's ``LatentMDGenModel.state_dict()`` (probe of
`mdgen/model/latent_model.py:43-128`; key list in SURVEY.md section 8(b)), so the same dict loads
into the reference via ``load_state_dict`` (done in `oracle/gen_golden.py`, this container only)
and into `mdgen_amd.model.LatentMDGenModel`.

Unlike the reference's ``initialize_weights`` (`latent_model.py:130-173`) nothing is
zero-initialised: adaLN modulation, ``emb_to_latent`` and IPA ``linear_out`` get small random
values, otherwise the network is the zero map and parity tests would be vacuous.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

from .config import ModelConfig


def _mha_shapes(prefix, C, head_dim):
    s = OrderedDict()
    s[prefix + "bias_k"] = (1, 1, C)
    s[prefix + "bias_v"] = (1, 1, C)
    for p in ("k_proj", "v_proj", "q_proj", "out_proj"):
        s[prefix + p + ".weight"] = (C, C)
        s[prefix + p + ".bias"] = (C,)
    s[prefix + "rot_emb.inv_freq"] = (head_dim // 2,)
    return s


def state_shapes(cfg: ModelConfig) -> "OrderedDict[str, tuple]":
    C, D = cfg.embed_dim, cfg.latent_dim
    s = OrderedDict()
    s["latent_to_emb.weight"] = (C, D)
    s["latent_to_emb.bias"] = (C,)
    if cfg.tps_condition:
        for n in ("latent_to_emb_f", "latent_to_emb_r"):
            s[n + ".weight"] = (C, 7)
            s[n + ".bias"] = (C,)
    s["cond_to_emb.weight"] = (C, D)
    s["cond_to_emb.bias"] = (C,)
    s["mask_to_emb.weight"] = (2, C)
    if cfg.prepend_ipa:
        s["aatype_to_emb.weight"] = (21, C)
        hc = cfg.ipa_heads * cfg.ipa_head_dim
        for i in range(cfg.num_layers):
            p = f"ipa_layers.{i}."
            s[p + "adaLN_modulation.1.weight"] = (6 * C, C)
            s[p + "adaLN_modulation.1.bias"] = (6 * C,)
            s[p + "ipa_norm.weight"] = (C,)
            s[p + "ipa_norm.bias"] = (C,)
            s[p + "ipa.head_weights"] = (cfg.ipa_heads,)
            s[p + "ipa.linear_q.weight"] = (hc, C)
            s[p + "ipa.linear_q.bias"] = (hc,)
            s[p + "ipa.linear_kv.weight"] = (2 * hc, C)
            s[p + "ipa.linear_kv.bias"] = (2 * hc,)
            s[p + "ipa.linear_q_points.weight"] = (cfg.ipa_heads * cfg.ipa_qk * 3, C)
            s[p + "ipa.linear_q_points.bias"] = (cfg.ipa_heads * cfg.ipa_qk * 3,)
            nkv = cfg.ipa_heads * (cfg.ipa_qk + cfg.ipa_v) * 3
            s[p + "ipa.linear_kv_points.weight"] = (nkv, C)
            s[p + "ipa.linear_kv_points.bias"] = (nkv,)
            cat = cfg.ipa_heads * (cfg.ipa_head_dim + cfg.ipa_v * 4)
            s[p + "ipa.linear_out.weight"] = (C, cat)
            s[p + "ipa.linear_out.bias"] = (C,)
            s.update(_mha_shapes(p + "mha_l.attn.", C, cfg.head_dim))
            s[p + "fc1.weight"] = (4 * C, C)
            s[p + "fc1.bias"] = (4 * C,)
            s[p + "fc2.weight"] = (C, 4 * C)
            s[p + "fc2.bias"] = (C,)
    for i in range(cfg.num_layers):
        p = f"layers.{i}."
        s[p + "adaLN_modulation.1.weight"] = (9 * C, C)
        s[p + "adaLN_modulation.1.bias"] = (9 * C,)
        s.update(_mha_shapes(p + "mha_t.attn.", C, cfg.head_dim))
        s.update(_mha_shapes(p + "mha_l.attn.", C, cfg.head_dim))
        s[p + "fc1.weight"] = (4 * C, C)
        s[p + "fc1.bias"] = (4 * C,)
        s[p + "fc2.weight"] = (C, 4 * C)
        s[p + "fc2.bias"] = (C,)
    s["emb_to_latent.linear.weight"] = (D, C)
    s["emb_to_latent.linear.bias"] = (D,)
    s["emb_to_latent.adaLN_modulation.1.weight"] = (2 * C, C)
    s["emb_to_latent.adaLN_modulation.1.bias"] = (2 * C,)
    s["t_embedder.mlp.0.weight"] = (C, 256)
    s["t_embedder.mlp.0.bias"] = (C,)
    s["t_embedder.mlp.2.weight"] = (C, C)
    s["t_embedder.mlp.2.bias"] = (C,)
    if cfg.abs_pos_emb:
        s["pos_embed"] = (1, cfg.crop, C)
    return s


def sincos_pos_embed(C: int, n: int) -> torch.Tensor:
    """`latent_model.py:22-40,151-153`: [sin | cos] of pos * 10000^(-d/(C/2)), float64 -> float32."""
    omega = 1.0 / 10000 ** (np.arange(C // 2, dtype=np.float64) / (C / 2.0))
    out = np.einsum("m,d->md", np.arange(n, dtype=np.float64), omega)
    return torch.from_numpy(np.concatenate([np.sin(out), np.cos(out)], axis=1)).float().unsqueeze(0)


def rope_inv_freq(head_dim: int) -> torch.Tensor:
    return 1.0 / (10000 ** (torch.arange(0, head_dim, 2).float() / head_dim))


def synth_state_dict(cfg: ModelConfig, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic fp32 weights: one PCG64 stream, tensors drawn in sorted-key order."""
    shapes = state_shapes(cfg)
    rng = np.random.Generator(np.random.PCG64(seed))
    out = OrderedDict()
    for name in sorted(shapes):
        shp = shapes[name]
        if name == "pos_embed":
            t = sincos_pos_embed(cfg.embed_dim, cfg.crop)
        elif name.endswith("rot_emb.inv_freq"):
            t = rope_inv_freq(cfg.head_dim)
        elif name.endswith("ipa_norm.weight"):
            t = torch.from_numpy(1.0 + 0.1 * rng.standard_normal(shp)).float()
        elif name.endswith("head_weights"):
            t = torch.from_numpy(0.5413 + 0.2 * rng.standard_normal(shp)).float()
        elif name.endswith(".bias") or name.endswith("bias_k") or name.endswith("bias_v"):
            std = 0.3 if ("bias_k" in name or "bias_v" in name) else 0.05
            t = torch.from_numpy(std * rng.standard_normal(shp)).float()
        elif len(shp) == 2:
            fan_out, fan_in = shp
            a = math.sqrt(6.0 / (fan_in + fan_out))
            if "adaLN_modulation" in name:
                a *= 0.5
            if "mask_to_emb" in name or "aatype_to_emb" in name:
                a = 0.5
            t = torch.from_numpy(rng.uniform(-a, a, size=shp)).float()
        else:
            t = torch.from_numpy(0.1 * rng.standard_normal(shp)).float()
        assert tuple(t.shape) == tuple(shp), (name, t.shape, shp)
        out[name] = t.contiguous()
    return out
