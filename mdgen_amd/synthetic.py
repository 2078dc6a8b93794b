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


def _quat_to_rot_cpu(q: torch.Tensor) -> torch.Tensor:
    """Hamilton (w, x, y, z) unit quaternion -> rotation matrix (plain torch; input data only)."""
    w, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(*q.shape[:-1], 3, 3)


def synth_forward_inputs(cfg: ModelConfig, B: int, T: int, L: int, n_pad: int, data_seed: int) -> dict:
    """Seeded CPU inputs of one `LatentMDGenModel.forward` call at any size (fixtures of full-size
    configurations store only the seed, an input checksum and sub-sampled outputs): x, per-sample t, mask with
    the last `n_pad` residues of every sample padded (dataset.py:80-89: aatype 0, identity frames), start/end
    frames, conditioning latents on frame 0 (and -1 for two-sided models), aatype."""
    g = torch.Generator().manual_seed(data_seed)
    D = cfg.latent_dim
    x = torch.randn(B, T, L, D, generator=g)
    t = torch.rand(B, generator=g)
    mask = torch.ones(B, L)
    aatype = torch.randint(0, 20, (B, L), generator=g)

    def rot():
        q = torch.randn(B, L, 4, generator=g)
        return _quat_to_rot_cpu(q / q.norm(dim=-1, keepdim=True))

    sR, eR = rot(), rot()
    st = torch.cumsum(2.2 * torch.randn(B, L, 3, generator=g), 1)
    et = st + torch.randn(B, L, 3, generator=g)
    if n_pad:
        mask[:, L - n_pad:] = 0
        aatype[:, L - n_pad:] = 0
        for R_, t_ in ((sR, st), (eR, et)):
            R_[:, L - n_pad:] = torch.eye(3)
            t_[:, L - n_pad:] = 0
    cm = torch.zeros(B, T, L, dtype=torch.long)
    cm[:, 0] = 1
    if cfg.tps_condition:
        cm[:, -1] = 1
    lat = torch.randn(B, T, L, D, generator=g)
    x_cond = torch.where(cm.unsqueeze(-1).bool(), lat, torch.zeros(()))
    return dict(x=x, t=t, mask=mask[:, None].expand(B, T, L).contiguous(), start_rot=sR, start_trans=st,
                end_rot=eR, end_trans=et, x_cond=x_cond, x_cond_mask=cm, aatype=aatype)


def tensor_checksum(d: dict) -> np.ndarray:
    """[sum, sum of |.|] over all tensors of a dict in float64: detects drift of a seeded generator."""
    return np.array([sum(float(v.double().sum()) for v in d.values()),
                     sum(float(v.double().abs().sum()) for v in d.values())])


def synth_batch(B, T, L, n_pad, dev, seed, tps=False):
    """Self-consistent synthetic conditioning batch: random frames + torsions ->
    atom14 (sampler post-processing kernel) -> conditioning frame (rollout-glue kernel), first frame
    expanded over T (sim_inference.py:72-79); trailing `n_pad` residues padded (dataset.py:80-89)."""
    from .geometry import atom14_to_cond, samples_to_atom14
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, L, 4, generator=g)
    q = (q / q.norm(dim=-1, keepdim=True)).to(dev)
    from .rigid_utils import Rotation
    R = Rotation(quats=q).get_rot_mats()
    tr = torch.cumsum(2.2 * torch.randn(B, L, 3, generator=g), 1).to(dev)
    ang = 6.283185307 * torch.rand(B, L, 7, generator=g)
    seqres = torch.randint(0, 20, (B, L), generator=g).to(dev)
    lat = torch.zeros(B, 1, L, 21, device=dev)
    lat[..., 0] = 1.0
    lat[..., 7:21] = torch.stack([ang.sin(), ang.cos()], -1).reshape(B, 1, L, 14).to(dev)
    atom14 = samples_to_atom14(lat, R, tr, seqres, tps=False)[:, 0]
    c = atom14_to_cond(atom14, seqres)
    mask = torch.ones(B, L, device=dev)
    if n_pad:
        mask[:, L - n_pad:] = 0
        seqres[:, L - n_pad:] = 0
    out = {"torsions": c["torsions"][:, None].expand(B, T, L, 7, 2).contiguous(),
           "torsion_mask": c["torsion_mask"], "trans": c["trans"][:, None].expand(B, T, L, 3).contiguous(),
           "rots": c["rots"][:, None].expand(B, T, L, 3, 3).contiguous(), "seqres": seqres, "mask": mask}
    if tps:   # two-sided conditioning: frame -1 is a second, different conformation (tps_inference.py:58-66)
        e = synth_batch(B, 1, L, n_pad, dev, seed + 7919)
        for k in ("torsions", "trans", "rots"):
            out[k][:, -1] = e[k][:, 0]
    return out
