"""CPU oracle for the MDGen denoising-sampler hot path.  *** TEST INFRASTRUCTURE ***

A plain-PyTorch fp32 restatement of the reference's algorithm for the path named by
BASELINE.json `north_star`.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import this module; the product (`mdgen_amd/`) never
does, and fails loudly when its HIP library is missing.

Pinning: the reference has no tests/golden vectors of its own.  This
oracle is pinned against outputs of the reference itself, generated in the development
container by `oracle/gen_golden.py` (reference imported from /root/reference with the stand-in
modules of `oracle/shims/`) and committed under `tests/golden/*.npz`; `tests/test_oracle_cpu.py`
checks every fixture.  Two third-party pieces absent from /root/reference are restated from
their published algorithms and are "parity unpinned" by the reference: fair-esm
RotaryEmbedding (additionally cross-checked against the HF transformers port) and
torchdiffeq fixed-grid Euler.

Every function cites the reference file:line it follows (paths relative to /root/reference).
Weights are a flat dict keyed exactly like `LatentMDGenModel.state_dict()`.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# constants (data dumped from mdgen/residue_constants.py by oracle/gen_residue_tables.py)
# ----------------------------------------------------------------------------------------------
_TABLES = None


def tables():
    global _TABLES
    if _TABLES is None:
        p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mdgen_amd", "data",
                         "residue_tables.npz")
        d = np.load(p)
        _TABLES = {k: (torch.from_numpy(d[k]) if d[k].dtype.kind in "fi" else d[k]) for k in d.files}
    return _TABLES


# ----------------------------------------------------------------------------------------------
# SE(3) algebra  (mdgen/rigid_utils.py) -- functional: a rigid is a pair (R [...,3,3], t [...,3])
# ----------------------------------------------------------------------------------------------
def rot_matmul(a, b):
    """rigid_utils.py:24-61  c_ik = sum_j a_ij b_jk (fp32, written out to dodge AMP)."""
    return torch.einsum("...ij,...jk->...ik", a.float(), b.float())


def rot_vec_mul(r, v):
    """rigid_utils.py:64-86  y_i = sum_j r_ij v_j."""
    return torch.einsum("...ij,...j->...i", r.float(), v.float())


def rigid_compose(R1, t1, R2, t2):
    """rigid_utils.py:1031-1045  (R1 R2, R1 t2 + t1)."""
    return rot_matmul(R1, R2), rot_vec_mul(R1, t2) + t1


def rigid_invert(R, t):
    """rigid_utils.py:1075-1085  (R^T, -R^T t)."""
    Rt = R.transpose(-1, -2)
    return Rt, -rot_vec_mul(Rt, t)


def rigid_apply(R, t, p):
    """rigid_utils.py:1047-1059."""
    return rot_vec_mul(R, p) + t


def rigid_invert_apply(R, t, p):
    """rigid_utils.py:1061-1073  R^T (p - t)."""
    return rot_vec_mul(R.transpose(-1, -2), p - t)


def quat_to_rot(q):
    """rigid_utils.py:156-188 (_QTR_MAT contraction == Hamilton (w,x,y,z) formula)."""
    a, b, c, d = q.unbind(-1)
    rows = [
        a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c),
        2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b),
        2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d,
    ]
    return torch.stack(rows, -1).reshape(q.shape[:-1] + (3, 3))


def rot_to_quat(R):
    """rigid_utils.py:191-210  eigenvector of the largest eigenvalue of the symmetric 4x4 K/3."""
    xx, xy, xz = R[..., 0, 0], R[..., 0, 1], R[..., 0, 2]
    yx, yy, yz = R[..., 1, 0], R[..., 1, 1], R[..., 1, 2]
    zx, zy, zz = R[..., 2, 0], R[..., 2, 1], R[..., 2, 2]
    k = [
        [xx + yy + zz, zy - yz, xz - zx, yx - xy],
        [zy - yz, xx - yy - zz, xy + yx, xz + zx],
        [xz - zx, xy + yx, yy - xx - zz, yz + zy],
        [yx - xy, xz + zx, yz + zy, zz - xx - yy],
    ]
    k = (1.0 / 3.0) * torch.stack([torch.stack(t, dim=-1) for t in k], dim=-2)
    _, vectors = torch.linalg.eigh(k)
    return vectors[..., -1]


def to_tensor_7(R, t, quat_sign="eigh"):
    """rigid_utils.py:1143-1155  [quat(4) | trans(3)].

    quat_sign: the reference takes whatever sign LAPACK's eigh returns (arbitrary, rigid_utils.py:208-210).
    "eigh" reproduces that (same torch build -> same signs; this is what the reference goldens pin);
    "w_nonneg" flips each quaternion so that w >= 0, the convention the HIP kernel (k_se3.hip rot2quat) uses,
    so that device parity can be asserted on the TPS relative-frame path where the sign reaches a Linear."""
    q = rot_to_quat(R)
    if quat_sign == "w_nonneg":
        q = torch.where(q[..., :1] < 0, -q, q)
    return torch.cat([q, t], -1)


def from_tensor_7(x, normalize_quats=True):
    """rigid_utils.py:1157-1173 + Rotation.__init__ :318-325 (q/|q|) + get_rot_mats :500-514."""
    q, t = x[..., :4].float(), x[..., 4:].float()
    if normalize_quats:
        q = q / torch.linalg.norm(q, dim=-1, keepdim=True)
    return quat_to_rot(q), t


def from_3_points(p_neg_x, origin, p_xy, eps=1e-8):
    """rigid_utils.py:1175-1218  Gram-Schmidt frame (columns e0,e1,e2; t = origin)."""
    e0 = origin - p_neg_x
    e1 = p_xy - origin
    e0 = e0 / torch.sqrt((e0 * e0).sum(-1, keepdim=True) + eps)
    dot = (e0 * e1).sum(-1, keepdim=True)
    e1 = e1 - e0 * dot
    e1 = e1 / torch.sqrt((e1 * e1).sum(-1, keepdim=True) + eps)
    e2 = torch.cross(e0, e1, dim=-1)
    R = torch.stack([e0, e1, e2], dim=-1)
    return R, origin


# ----------------------------------------------------------------------------------------------
# geometry (mdgen/geometry.py)
# ----------------------------------------------------------------------------------------------
def atom14_to_frames(atom14):
    """geometry.py:218-231  from_3_points(C, CA, N) right-composed with diag(-1,1,-1)."""
    n, ca, c = atom14[..., 0, :], atom14[..., 1, :], atom14[..., 2, :]
    R, t = from_3_points(c, ca, n)
    flip = torch.tensor([-1.0, 1.0, -1.0], dtype=R.dtype)
    return R * flip, t  # R @ diag(flip) scales columns


def atom14_to_atom37(atom14, aatype):
    """geometry.py:9-27."""
    T = tables()
    idx = T["atom37_to_atom14"][aatype]                      # [...,37]
    g = torch.gather(atom14, -2, idx[..., None].expand(*idx.shape, 3))
    return g * T["atom37_mask"][aatype][..., None]


def atom37_to_torsions(atom37, aatype):
    """geometry.py:82-202  -> (sin,cos)[...,7,2], mask[...,7]; order [pre-omega, phi, psi, chi1-4]."""
    T = tables()
    mask37 = T["atom37_mask"][aatype]
    pad = atom37.new_zeros([*atom37.shape[:-3], 1, 37, 3])
    prev = torch.cat([pad, atom37[..., :-1, :, :]], dim=-3)
    padm = mask37.new_zeros([*mask37.shape[:-2], 1, 37])
    prevm = torch.cat([padm, mask37[..., :-1, :]], dim=-2)
    pre_omega = torch.cat([prev[..., 1:3, :], atom37[..., :2, :]], dim=-2)
    phi = torch.cat([prev[..., 2:3, :], atom37[..., :3, :]], dim=-2)
    psi = torch.cat([atom37[..., :3, :], atom37[..., 4:5, :]], dim=-2)
    pre_omega_m = prevm[..., 1:3].prod(-1) * mask37[..., :2].prod(-1)
    phi_m = prevm[..., 2] * mask37[..., :3].prod(-1)
    psi_m = mask37[..., :3].prod(-1) * mask37[..., 4]
    ai = T["chi_atom_indices"][aatype]                       # [...,4,4]
    flat = ai.reshape(*ai.shape[:-2], 16)
    chis = torch.gather(atom37, -2, flat[..., None].expand(*flat.shape, 3)).reshape(*ai.shape, 3)
    chim = T["chi_angles_mask"][aatype] * torch.gather(mask37, -1, flat).reshape(ai.shape).prod(-1)
    pos = torch.cat([pre_omega[..., None, :, :], phi[..., None, :, :], psi[..., None, :, :], chis], dim=-3)
    tmask = torch.cat([pre_omega_m[..., None], phi_m[..., None], psi_m[..., None], chim], dim=-1)
    R, t = from_3_points(pos[..., 1, :], pos[..., 2, :], pos[..., 0, :], eps=1e-8)
    rel = rigid_invert_apply(R, t, pos[..., 3, :])
    sc = torch.stack([rel[..., 2], rel[..., 1]], dim=-1)
    sc = sc / torch.sqrt((sc * sc).sum(-1, keepdim=True) + 1e-8)
    sign = torch.tensor([1.0, 1.0, -1.0, 1.0, 1.0, 1.0, 1.0])[:, None]
    return sc * sign, tmask


def torsion_angles_to_frames(R, t, alpha, aatype):
    """geometry.py:273-334  8 rigid-group frames per residue in the global frame."""
    T = tables()
    d4 = T["default_frames"][aatype]                          # [...,8,4,4]
    dR, dt = d4[..., :3, :3], d4[..., :3, 3]
    bb = alpha.new_zeros(*alpha.shape[:-2], 1, 2)
    bb[..., 1] = 1
    alpha = torch.cat([bb, alpha], dim=-2)                    # [...,8,2] (sin,cos)
    rot = alpha.new_zeros(*alpha.shape[:-1], 3, 3)
    rot[..., 0, 0] = 1
    rot[..., 1, 1] = alpha[..., 1]
    rot[..., 1, 2] = -alpha[..., 0]
    rot[..., 2, 1] = alpha[..., 0]
    rot[..., 2, 2] = alpha[..., 1]
    fR, ft = rigid_compose(dR, dt, rot, torch.zeros_like(dt))
    # chain chi2..chi4 onto chi1 (groups 4..7)
    c1R, c1t = fR[..., 4, :, :], ft[..., 4, :]
    c2R, c2t = rigid_compose(c1R, c1t, fR[..., 5, :, :], ft[..., 5, :])
    c3R, c3t = rigid_compose(c2R, c2t, fR[..., 6, :, :], ft[..., 6, :])
    c4R, c4t = rigid_compose(c3R, c3t, fR[..., 7, :, :], ft[..., 7, :])
    aR = torch.cat([fR[..., :5, :, :], c2R[..., None, :, :], c3R[..., None, :, :], c4R[..., None, :, :]], dim=-3)
    at = torch.cat([ft[..., :5, :], c2t[..., None, :], c3t[..., None, :], c4t[..., None, :]], dim=-2)
    return rigid_compose(R[..., None, :, :], t[..., None, :], aR, at)


def frames_torsions_to_atom14(R, t, torsions, aatype):
    """geometry.py:61-79 + :236-270  atom14 = group_frame(lit_pos[aatype]) * atom_mask."""
    T = tables()
    gR, gt = torsion_angles_to_frames(R, t, torsions, aatype)         # [...,8,3,3], [...,8,3]
    grp = T["atom14_group"][aatype]                                    # [...,14]
    aR = torch.gather(gR, -3, grp[..., None, None].expand(*grp.shape, 3, 3))
    at = torch.gather(gt, -2, grp[..., None].expand(*grp.shape, 3))
    lit = T["lit_positions"][aatype]
    pos = rot_vec_mul(aR, lit) + at
    return pos * T["atom14_mask"][aatype][..., None]


# ----------------------------------------------------------------------------------------------
# model pieces (mdgen/model/{layers,mha,ipa,latent_model}.py)
# ----------------------------------------------------------------------------------------------
def linear(P, name, x):
    return F.linear(x, P[name + ".weight"], P[name + ".bias"])


def modulate(x, shift, scale):
    """layers.py:14-15."""
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def gelu(x):
    """layers.py:77-84 (exact erf)."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def ln(x, eps=1e-6):
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


def timestep_embedding(t, dim=256, max_period=10000):
    """layers.py:31-50  [cos | sin] of t * exp(-ln(1e4) i/half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def t_embedder(P, t):
    """layers.py:52-55."""
    h = linear(P, "t_embedder.mlp.0", timestep_embedding(t))
    return linear(P, "t_embedder.mlp.2", F.silu(h))


def rope_tables(n_pos, head_dim):
    """esm.rotary_embedding (fair-esm, not vendored): angle[pos, i] = pos * 10000^(-2(i mod d/2)/d)."""
    inv = 1.0 / (10000 ** (torch.arange(0, head_dim, 2).float() / head_dim))
    fr = torch.einsum("i,j->ij", torch.arange(n_pos).float(), inv)
    emb = torch.cat([fr, fr], -1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    x1, x2 = x.chunk(2, dim=-1)
    return torch.cat((-x2, x1), dim=-1)


def mha_rope(P, pre, y, mask, heads):
    """AttentionWithRoPE (latent_model.py:320-329) -> MultiheadAttention.forward live subset
    (mha.py:258-286, 326-336, 356-397).  y [bsz,len,C]; mask [bsz,len] (1 = real key)."""
    bsz, n, C = y.shape
    dh = C // heads
    q = linear(P, pre + "q_proj", y) * dh ** -0.5                       # mha.py:258-263
    k = linear(P, pre + "k_proj", y)
    v = linear(P, pre + "v_proj", y)
    k = torch.cat([k, P[pre + "bias_k"].reshape(1, 1, C).expand(bsz, 1, C)], 1)   # :265-268
    v = torch.cat([v, P[pre + "bias_v"].reshape(1, 1, C).expand(bsz, 1, C)], 1)
    pad = torch.cat([1 - mask, mask.new_zeros(bsz, 1)], 1).bool()      # :274-280 bias key never padded
    q = q.view(bsz, n, heads, dh).transpose(1, 2)                       # [bsz,H,n,dh]
    k = k.view(bsz, n + 1, heads, dh).transpose(1, 2)
    v = v.view(bsz, n + 1, heads, dh).transpose(1, 2)
    cos, sin = rope_tables(n + 1, dh)                                   # :356-357
    q = q * cos[:n] + rotate_half(q) * sin[:n]
    k = k * cos + rotate_half(k) * sin
    s = q @ k.transpose(-1, -2)                                         # :359
    s = s.masked_fill(pad[:, None, None, :], float("-inf"))             # :370-376
    p = F.softmax(s, dim=-1, dtype=torch.float32)                       # :381
    o = (p @ v).transpose(1, 2).reshape(bsz, n, C)                      # :389-396
    return linear(P, pre + "out_proj", o)                               # :397


def ipa(P, pre, s, R, t, mask, H=4, c=32, Pq=8, Pv=8, inf=1e5, eps=1e-8):
    """InvariantPointAttention.forward with c_z=0 (ipa.py:92-255).  s [B,L,C]; R [B,L,3,3]; t [B,L,3]."""
    B, L, _ = s.shape
    q = linear(P, pre + "linear_q", s).view(B, L, H, c)                # :113-118
    kv = linear(P, pre + "linear_kv", s).view(B, L, H, 2 * c)          # :114-124
    k, v = kv[..., :c], kv[..., c:]
    qp = linear(P, pre + "linear_q_points", s)                         # :126-135 [x-block|y-block|z-block]
    qp = torch.stack(torch.split(qp, qp.shape[-1] // 3, dim=-1), dim=-1)
    qp = rigid_apply(R[:, :, None], t[:, :, None], qp).view(B, L, H, Pq, 3)
    kvp = linear(P, pre + "linear_kv_points", s)                       # :137-151
    kvp = torch.stack(torch.split(kvp, kvp.shape[-1] // 3, dim=-1), dim=-1)
    kvp = rigid_apply(R[:, :, None], t[:, :, None], kvp).view(B, L, H, Pq + Pv, 3)
    kp, vp = kvp[..., :Pq, :], kvp[..., Pq:, :]
    a = torch.einsum("bihc,bjhc->bhij", q, k) * math.sqrt(1.0 / (3 * c))            # :161-168
    d2 = ((qp[:, :, None] - kp[:, None]) ** 2).sum(-1)                               # [B,i,j,H,Pq]  :171-175
    hw = F.softplus(P[pre + "head_weights"]) * math.sqrt(1.0 / (3 * (Pq * 9.0 / 2)))  # :176-181
    pt = (d2 * hw[None, None, None, :, None]).sum(-1) * (-0.5)                       # :182-185
    a = a + pt.permute(0, 3, 1, 2)
    sq = inf * (mask[:, :, None] * mask[:, None, :] - 1)                             # :188-190
    a = a + sq[:, None]
    a = torch.softmax(a, dim=-1)                                                     # :203
    o = torch.einsum("bhij,bjhc->bihc", a, v).reshape(B, L, H * c)                   # :209-212
    op = torch.einsum("bhij,bjhpx->bihpx", a, vp)                                    # :216-225
    op = rigid_invert_apply(R[:, :, None, None], t[:, :, None, None], op)            # :226
    opn = torch.sqrt((op ** 2).sum(-1) + eps).reshape(B, L, H * Pv)                  # :229-231
    op = op.reshape(B, L, H * Pv, 3)
    cat = torch.cat([o, op[..., 0], op[..., 1], op[..., 2], opn], dim=-1)            # :250-254
    return linear(P, pre + "linear_out", cat)


def ipa_layer(P, pre, x, t, mask, R, tr, heads, skip=0):
    """IPALayer.forward (latent_model.py:369-384).  `skip` (tests only) drops sub-layers, mirroring the
    library's MDGEN_DEBUG_SKIP bits 3/4/5."""
    C = x.shape[-1]
    mod = linear(P, pre + "adaLN_modulation.1", F.silu(t))
    sh_l, sc_l, g_l, sh_m, sc_m, g_m = mod.chunk(6, dim=-1)
    if not skip & 8:
        xn = F.layer_norm(x, (C,), P[pre + "ipa_norm.weight"], P[pre + "ipa_norm.bias"], 1e-5)
        x = x + ipa(P, pre + "ipa.", xn, R, tr, mask)
    if not skip & 16:
        y = mha_rope(P, pre + "mha_l.attn.", modulate(ln(x), sh_l, sc_l), mask, heads)
        x = x + g_l.unsqueeze(1) * y
    if not skip & 32:
        y = linear(P, pre + "fc2", gelu(linear(P, pre + "fc1", modulate(ln(x), sh_m, sc_m))))
        x = x + g_m.unsqueeze(1) * y
    return x


def trunk_layer(P, pre, x, t, mask, heads, skip=0):
    """LatentMDGenLayer.forward (latent_model.py:446-483).  x [B,T,L,C]; t [B,1,C]; mask [B,T,L].
    `skip` (tests only) mirrors MDGEN_DEBUG_SKIP bits 0/1/2."""
    B, T, L, C = x.shape
    mod = linear(P, pre + "adaLN_modulation.1", F.silu(t))               # [B,1,9C]
    sh_l, sc_l, g_l, sh_t, sc_t, g_t, sh_m, sc_m, g_m = mod.chunk(9, dim=-1)
    if not skip & 1:
        y = modulate(ln(x), sh_l, sc_l)
        y = mha_rope(P, pre + "mha_l.attn.", y.reshape(B * T, L, C), mask.reshape(B * T, L), heads).reshape(B, T, L, C)
        x = x + g_l.unsqueeze(1) * y
    if not skip & 2:
        y = modulate(ln(x), sh_t, sc_t)
        y = mha_rope(P, pre + "mha_t.attn.", y.transpose(1, 2).reshape(B * L, T, C),
                     mask.transpose(1, 2).reshape(B * L, T), heads).reshape(B, L, T, C).transpose(1, 2)
        x = x + g_t.unsqueeze(1) * y
    if not skip & 4:
        y = linear(P, pre + "fc2", gelu(linear(P, pre + "fc1", modulate(ln(x), sh_m, sc_m))))
        x = x + g_m.unsqueeze(1) * y
    return x


def run_ipa(P, cfg, temb, mask_bl, start, end, aatype):
    """LatentMDGenModel.run_ipa (latent_model.py:175-210).  start/end = (R [B,L,3,3], t [B,L,3])."""
    H = cfg["mha_heads"]
    nl = cfg["num_layers"]
    sk = cfg.get("debug_skip", 0)
    aa = P["aatype_to_emb.weight"][aatype]
    if cfg.get("tps_condition", False):
        iR, it = rigid_invert(*start)
        qs = cfg.get("quat_sign", "eigh")
        x_f = to_tensor_7(*rigid_compose(iR, it, *end), quat_sign=qs)    # :194
        iR, it = rigid_invert(*end)
        x_r = to_tensor_7(*rigid_compose(iR, it, *start), quat_sign=qs)  # :195
        x_f = linear(P, "latent_to_emb_f", x_f) + aa
        x_r = linear(P, "latent_to_emb_r", x_r) + aa
        for i in range(nl):
            x_r = ipa_layer(P, f"ipa_layers.{i}.", x_r, temb, mask_bl, start[0], start[1], H, sk)
            x_f = ipa_layer(P, f"ipa_layers.{i}.", x_f, temb, mask_bl, end[0], end[1], H, sk)
        return x_r + x_f
    x = aa
    for i in range(nl):
        x = ipa_layer(P, f"ipa_layers.{i}.", x, temb, mask_bl, start[0], start[1], H, sk)
    return x


def forward(P, cfg, x, t, mask, start_frames, end_frames, x_cond, x_cond_mask, aatype, return_trace=False):
    """LatentMDGenModel.forward (latent_model.py:212-260), non-design path == forward_inference."""
    H = cfg["mha_heads"]
    h = linear(P, "latent_to_emb", x)                                    # :233
    if cfg.get("abs_pos_emb", False):
        h = h + P["pos_embed"]                                           # :234-235 ([1,L,C] bcast over T)
    h = h + linear(P, "cond_to_emb", x_cond) + P["mask_to_emb.weight"][x_cond_mask]   # :240-241
    temb = t_embedder(P, t * cfg.get("time_multiplier", 100.0))[:, None]  # :243
    trace = {}
    if cfg.get("prepend_ipa", True):
        ipa_out = run_ipa(P, cfg, temb[:, 0], mask[:, 0], start_frames, end_frames, aatype)
        trace["ipa_out"] = ipa_out
        h = h + ipa_out[:, None]                                         # :245-246
    trace["h0"] = h
    for i in range(cfg["num_layers"]):
        h = trunk_layer(P, f"layers.{i}.", h, temb, mask, H, cfg.get("debug_skip", 0))   # :248-249
        trace[f"h{i + 1}"] = h
    mod = linear(P, "emb_to_latent.adaLN_modulation.1", F.silu(temb))    # layers.py:70-74
    shift, scale = mod.chunk(2, dim=-1)
    out = linear(P, "emb_to_latent.linear", modulate(ln(h), shift, scale))
    if return_trace:
        return out, trace
    return out


# ----------------------------------------------------------------------------------------------
# wrapper-level pieces (mdgen/wrapper.py, mdgen/utils.py, mdgen/transport/*)
# ----------------------------------------------------------------------------------------------
def get_offsets(ref, rig):
    """utils.py:7-14  ref^-1 o rigids -> [quat|trans]."""
    iR, it = rigid_invert(*ref)
    return to_tensor_7(*rigid_compose(iR, it, *rig))


def prep_batch(batch, cfg):
    """NewMDGenWrapper.prep_batch (wrapper.py:283-365), sim_condition / tps_condition paths."""
    R, t = batch["rots"].float(), batch["trans"].float()
    B, T, L = t.shape[:3]
    off = get_offsets((R[:, 0:1], t[:, 0:1]), (R, t))                    # :307
    off[..., :4] *= torch.where(off[..., 0:1] < 0, -1, 1)                # :309
    frame_lm = batch["mask"].unsqueeze(-1).expand(-1, -1, 7)
    tors_lm = batch["torsion_mask"].unsqueeze(-1).expand(-1, -1, -1, 2).reshape(B, L, 14)
    tps = cfg.get("tps_condition", False)
    if tps:                                                              # :314-318
        off_r = get_offsets((R[:, -1:], t[:, -1:]), (R, t))
        off_r[..., :4] *= torch.where(off_r[..., 0:1] < 0, -1, 1)
        off = torch.cat([off, off_r], -1)
        frame_lm = torch.cat([frame_lm, frame_lm], -1)
    latents = torch.cat([off, batch["torsions"].reshape(B, T, L, 14).float()], -1)   # :327
    loss_mask = torch.cat([frame_lm, tors_lm], -1).unsqueeze(1).expand(-1, T, -1, -1)
    cond_mask = torch.zeros(B, T, L, dtype=torch.long)
    if cfg.get("sim_condition", False):
        cond_mask[:, 0] = 1                                              # :339-340
    if tps:
        cond_mask[:, 0] = 1
        cond_mask[:, -1] = 1                                             # :341-342
    if cfg.get("cond_interval"):
        cond_mask[:, ::int(cfg["cond_interval"])] = 1                    # :343-344
    return {
        "rigids": (R, t),
        "latents": latents,
        "loss_mask": loss_mask,
        "model_kwargs": {
            "start_frames": (R[:, 0], t[:, 0]),
            "end_frames": (R[:, -1], t[:, -1]),
            "mask": batch["mask"].unsqueeze(1).expand(-1, T, -1).float(),
            "aatype": batch["seqres"],
            "x_cond": torch.where(cond_mask.unsqueeze(-1).bool(), latents, 0.0),
            "x_cond_mask": cond_mask,
        },
    }


def sample_euler(P, cfg, zs, model_kwargs, num_steps):
    """Sampler.sample_ode -> ode.sample -> torchdiffeq fixed-grid Euler (transport.py:408-451,
    integrators.py:95-114).  `num_steps` = number of Euler steps S; the grid is linspace(0,1,S+1)
    (the reference hard-codes 50 grid points = 49 steps).  Returns the final state only."""
    tg = torch.linspace(0, 1, num_steps + 1)
    x = zs
    B = zs.shape[0]
    for i in range(num_steps):
        tt = torch.ones(B) * tg[i]                                       # integrators.py:99
        v = forward(P, cfg, x, tt, **model_kwargs)
        x = x + (tg[i + 1] - tg[i]) * v
    return x


def postprocess(samples, rigids, seqres, cfg):
    """NewMDGenWrapper.inference tail (wrapper.py:456-484): latents -> atom14."""
    B, T, L = samples.shape[:3]
    off = samples[..., :7]
    tors = samples[..., 14:28] if cfg.get("tps_condition", False) else samples[..., 7:21]
    oR, ot = from_tensor_7(off, normalize_quats=True)                    # :469
    R0, t0 = rigids[0][:, 0:1], rigids[1][:, 0:1]
    fR, ft = rigid_compose(R0, t0, oR, ot)
    tors = tors.reshape(B, T, L, 7, 2)
    tors = tors / torch.linalg.norm(tors, dim=-1, keepdim=True)          # :474-476
    aatype = seqres[:, None].expand(B, T, L)
    return frames_torsions_to_atom14(fR, ft, tors, aatype), aatype


def inference(P, cfg, batch, zs, num_steps):
    """NewMDGenWrapper.inference (wrapper.py:405-484) with explicit noise `zs` and step count."""
    prep = prep_batch(batch, cfg)
    samples = sample_euler(P, cfg, zs, prep["model_kwargs"], num_steps)
    atom14, aa = postprocess(samples, prep["rigids"], batch["seqres"], cfg)
    return atom14, aa, samples


def get_batch_from_atom14(arr, seqres):
    """sim_inference.get_batch (sim_inference.py:32-59) for an atom14 array [F,L,14,3] (torch)."""
    R, t = atom14_to_frames(arr)
    atom37 = atom14_to_atom37(arr, seqres[None].expand(arr.shape[0], -1)).float()
    tors, tmask = atom37_to_torsions(atom37, seqres[None].expand(arr.shape[0], -1))
    return {"torsions": tors, "torsion_mask": tmask[0], "trans": t, "rots": R, "seqres": seqres,
            "mask": torch.ones(len(seqres))}


def rollout_glue(atom14_last, seqres):
    """sim_inference.rollout tail (sim_inference.py:91-96): last frame -> next conditioning frame.
    atom14_last [B,L,14,3]; seqres [B,L]."""
    R, t = atom14_to_frames(atom14_last)
    atom37 = atom14_to_atom37(atom14_last, seqres)
    tors, _ = atom37_to_torsions(atom37, seqres)
    return {"trans": t[:, None], "rots": R[:, None], "torsions": tors[:, None]}


def path_plan(t, x0, x1, path_type="GVP"):
    """path.py:113-135 `ICPlan.plan` with GVPCPlan coefficients (:177-187) or the linear ICPlan (:28-40):
    xt = alpha_t x1 + sigma_t x0,  ut = alpha_t' x1 + sigma_t' x0,  t expanded over the non-batch dims."""
    te = t.reshape(-1, *([1] * (x1.dim() - 1)))
    if path_type == "GVP":
        a, da = torch.sin(te * math.pi / 2), math.pi / 2 * torch.cos(te * math.pi / 2)
        sg, ds = torch.cos(te * math.pi / 2), -math.pi / 2 * torch.sin(te * math.pi / 2)
    else:
        a, da = te, torch.ones_like(te)
        sg, ds = 1 - te, -torch.ones_like(te)
    return a * x1 + sg * x0, da * x1 + ds * x0


def mean_flat(x, mask):
    """transport.py:13-17: masked mean over all non-batch dimensions."""
    dims = list(range(1, x.dim()))
    return torch.sum(x * mask, dim=dims) / torch.sum(mask, dim=dims)


def training_losses(P, cfg, x1, mask, model_kwargs, t, x0, path_type="GVP"):
    """Transport.training_losses (transport.py:138-189), velocity model, non-design path, with the noise x0 and
    the times t given explicitly (the reference draws them at :126-136)."""
    xt, ut = path_plan(t, x0, x1, path_type)
    pred = forward(P, cfg, xt, t, **model_kwargs)
    return {"t": t, "pred": pred, "loss": mean_flat((pred - ut) ** 2, mask), "xt": xt, "ut": ut}


def cfg_dict(model_config):
    """ModelConfig dataclass -> plain dict used by this module."""
    return dict(model_config.to_dict(), latent_dim=model_config.latent_dim)
