"""GPU parity: the HIP path (through the C-ABI) vs the reference's golden vectors and vs the CPU oracle.

Tolerances: the kernels use bf16 MFMA operands with fp32
accumulation / softmax / LayerNorm statistics / residual stream, so the gate is the bf16 one:
rel-L2 <= 1e-2 per network evaluation (the reference's own bf16-autocast deviates 4e-3);
SE(3)/geometry kernels are fp32: max-abs <= 1e-4 (Angstrom / unit quaternions).
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, weights_for, rel_l2, model_config_from

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
TOL_FWD = 1e-2
# BASELINE.md section 3, end-to-end atom14 of the bf16-operand path at the BASELINE step counts (S = 10 / 49): rms <= 0.02 A,
# max <= 0.5 A.  S = 1 (one Euler step of size 1: the error of a single network evaluation at full weight, not a BASELINE
# configuration) is gated at 0.03 A rms -- measured 0.0196 A.
TOL_RMS, TOL_MAX = 0.02, 0.5


def tol_rms(S):
    return TOL_RMS if S >= 10 else 0.03


def _cuda():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    return torch.device("cuda")


_MODELS = {}


def get_model(cfg, sd, key):
    from mdgen_amd.model import LatentMDGenModel
    if key not in _MODELS:
        _MODELS.clear()
        m = LatentMDGenModel(cfg)
        m.load_state_dict(sd)
        _MODELS[key] = m
    return _MODELS[key]


def _kw(g, dev):
    return dict(
        x=g["x"].to(dev), t=g["t"].to(dev), mask=g["mask"].to(dev),
        start_frames=(g["start_rot"].to(dev), g["start_trans"].to(dev)),
        end_frames=(g["end_rot"].to(dev), g["end_trans"].to(dev)),
        x_cond=g["x_cond"].to(dev), x_cond_mask=g["x_cond_mask"].to(dev), aatype=g["aatype"].to(dev))


def test_native_library_is_loaded():
    import mdgen_amd._lib as L
    assert os.path.exists(L.LIB_PATH)
    assert L.lib.mdgen_abi_version() == L.ABI_VERSION
    with open("/proc/self/maps") as f:
        assert "libmdgen_amd.so" in f.read()


@pytest.mark.parametrize("name", ["fwd_full_sim", "fwd_full_pep", "fwd_full_atlas"])
def test_forward_vs_reference_golden(name):
    """LatentMDGenModel.forward vs the reference's own output (residue axis: micro path L<=8 and
    flash path L=40; padded residues; abs_pos_emb on/off)."""
    dev = _cuda()
    g = load_golden(name)
    cfg, sd = weights_for(g)
    m = get_model(cfg, sd, (name, "w"))
    out, tr = m.forward(**_kw(g, dev), return_trace=True)
    torch.cuda.synchronize()
    nl = cfg.num_layers
    rep = {k: rel_l2(tr[k].cpu(), g[k]) for k in ("ipa_out", "h0", f"h{nl}") if k in g}
    rep["out"] = rel_l2(out.cpu(), g["out"])
    print(name, {k: f"{v:.2e}" for k, v in rep.items()})
    assert torch.isfinite(out).all()
    for k, v in rep.items():
        assert v < TOL_FWD, (k, v)


@pytest.mark.parametrize("shape", [(2, 70, 4, 0), (1, 64, 4, 0), (1, 33, 5, 2), (1, 5, 33, 3), (2, 3, 64, 0),
                                   (1, 130, 9, 1), (1, 64, 64, 4), (2, 32, 96, 5),
                                   (1, 8, 300, 20), (1, 16, 724, 0)])
def test_forward_vs_oracle_shapes(shape):
    """Edge shapes vs the CPU oracle: partial panels/tiles, T or L a multiple of 32/64 (bias key opens a
    new tile), L in {4,5} micro path vs L>8 flash path, padded residues, per-batch t; and sequences longer than the
    training crop -- the reference samples ATLAS on the UNCROPPED chain (sim_inference.py:32-59; splits/atlas_test.csv
    goes up to L = 724: 23 residue-axis key tiles, 724-key tiled IPA attention), here L = 300 with 20 padded residues and L = 724.
    The forward is run on a workspace whose every byte was set to 0xFF first (bf16 / fp32 NaN patterns): whatever
    the kernels read from it must have been written by this call (regression: the bias-only key tile of a
    sequence whose length is a multiple of 64 used to be multiplied in as 0 x stale bytes)."""
    from oracle import mdgen_oracle as O
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    dev = _cuda()
    B, T, L, npad = shape
    cfg = ModelConfig.forward_sim(num_frames=T, crop=max(L, 4))
    sd = synth_state_dict(cfg, 5)
    m = get_model(cfg, sd, ("oracle-shapes", cfg.crop))
    gen = torch.Generator().manual_seed(1000 + B * 7 + T * 3 + L)
    D = cfg.latent_dim
    x = torch.randn(B, T, L, D, generator=gen)
    t = torch.rand(B, generator=gen)
    mask = torch.ones(B, L)
    if npad:
        mask[-1, L - npad:] = 0
    mask = mask[:, None].expand(B, T, L).contiguous()
    q = torch.randn(B, L, 4, generator=gen)
    R = O.quat_to_rot(q / q.norm(dim=-1, keepdim=True))
    tr_ = torch.cumsum(2.2 * torch.randn(B, L, 3, generator=gen), 1)
    cm = torch.zeros(B, T, L, dtype=torch.long)
    cm[:, 0] = 1
    xc = torch.where(cm.unsqueeze(-1).bool(), torch.randn(B, T, L, D, generator=gen), torch.zeros(()))
    aat = torch.randint(0, 20, (B, L), generator=gen)
    kw = dict(x=x, t=t, mask=mask, start_frames=(R, tr_), end_frames=(R, tr_), x_cond=xc, x_cond_mask=cm, aatype=aat)
    ref, rtr = O.forward(sd, O.cfg_dict(cfg), return_trace=True, **kw)
    dkw = {k: (tuple(u.to(dev) for u in v) if isinstance(v, tuple) else v.to(dev)) for k, v in kw.items()}
    m.forward(**dkw)                                   # allocates (and caches) the workspace of this shape
    assert len(m._ws) > 0
    for ws in m._ws.values():
        ws.view(torch.uint8).fill_(0xFF)
    out, tr = m.forward(**dkw, return_trace=True)
    rep = {k: rel_l2(tr[k].cpu(), rtr[k]) for k in ["ipa_out"] + [f"h{i}" for i in range(cfg.num_layers + 1)]}
    rep["out"] = rel_l2(out.cpu(), ref)
    print(shape, {k: f"{v:.2e}" for k, v in rep.items()})
    assert torch.isfinite(out).all()
    assert rep["out"] < TOL_FWD and rep["ipa_out"] < TOL_FWD and rep[f"h{cfg.num_layers}"] < TOL_FWD


def test_forward_tps_vs_oracle_with_reference_inputs():
    """Two-sided (TPS) conditioning: D=28 latents, relative frames start^-1 o end / end^-1 o start through
    rot_to_quat -> latent_to_emb_f/r, and the IPA stack run on both frame sets (latent_model.py:193-207).
    Inputs and weights are the reference golden's (fwd_full_tps).  The reference's quaternion SIGN is whatever
    LAPACK eigh returns (rigid_utils.py:208-210) and reaches a Linear, so (a) with the library's own relative frames the
    expectation is the CPU oracle -- itself pinned bit-for-sign against that golden in test_oracle_cpu -- run with the
    kernel's w >= 0 convention, and (b) with the reference's own to_tensor_7() outputs handed over (`rel_quats` -> the
    `rel7` argument of the C-ABI) the expectation is the REFERENCE's golden output itself."""
    from oracle import mdgen_oracle as O
    dev = _cuda()
    g = load_golden("fwd_full_tps")
    cfg, sd = weights_for(g)
    assert cfg.tps_condition and cfg.latent_dim == 28
    m = get_model(cfg, sd, ("fwd_full_tps", "w"))
    out, tr = m.forward(**_kw(g, dev), return_trace=True)
    torch.cuda.synchronize()
    kw = {k: (tuple(u.cpu() for u in v) if isinstance(v, tuple) else v.cpu()) for k, v in _kw(g, "cpu").items()}
    c = dict(O.cfg_dict(cfg), quat_sign="w_nonneg")
    ref, rtr = O.forward(sd, c, return_trace=True, **kw)
    rep = {k: rel_l2(tr[k].cpu(), rtr[k]) for k in ("ipa_out", "h0", f"h{cfg.num_layers}")}
    rep["out"] = rel_l2(out.cpu(), ref)
    # does the reference's own sign choice coincide with w >= 0 on this fixture?
    iR, it = O.rigid_invert(*kw["start_frames"])
    qf = O.rot_to_quat(O.rigid_compose(iR, it, *kw["end_frames"])[0])
    iR, it = O.rigid_invert(*kw["end_frames"])
    qr = O.rot_to_quat(O.rigid_compose(iR, it, *kw["start_frames"])[0])
    same_sign = bool((qf[..., 0] >= 0).all() and (qr[..., 0] >= 0).all())
    print("fwd_full_tps", {k: f"{v:.2e}" for k, v in rep.items()}, "eigh sign == w>=0:", same_sign)
    assert torch.isfinite(out).all()
    for k in ("ipa_out", "h0", f"h{cfg.num_layers}", "out"):
        assert rep[k] < TOL_FWD, (k, rep[k])
    # (b) the REFERENCE's own output, directly: the caller hands over the reference's to_tensor_7() outputs (fixture key
    # `rel7`, latent_model.py:194-195 run by oracle/gen_golden.py; quaternion sign as eigh chose it there) through the
    # `rel7` argument of mdgen_denoiser_forward -- what a drop-in caller holding a reference-trained checkpoint does
    out_r, tr_r = m.forward(**_kw(g, dev), rel_quats=g["rel7"].to(dev), return_trace=True)
    torch.cuda.synchronize()
    rep_r = {k: rel_l2(tr_r[k].cpu(), g[k]) for k in ("ipa_out", "h0", f"h{cfg.num_layers}") if k in g}
    rep_r["out_vs_reference_golden"] = rel_l2(out_r.cpu(), g["out"])
    print("fwd_full_tps with the reference's rel7:", {k: f"{v:.2e}" for k, v in rep_r.items()})
    for k, v in rep_r.items():
        assert v < TOL_FWD, (k, v)
    # the sign really matters on this fixture (7 of 16 quaternions have w < 0): without rel7 the output differs
    if not same_sign:
        assert rel_l2(out.cpu(), g["out"]) > 5 * TOL_FWD
    # the Euler rollout takes the same argument: S = 1 step of size 1 from x equals x + forward(x, t = 0)
    kw0 = _kw(g, dev)
    kw0["t"] = torch.zeros_like(kw0["t"])
    v0 = m.forward(**kw0, rel_quats=g["rel7"].to(dev))
    kws = {k: v for k, v in kw0.items() if k not in ("x", "t")}
    x1 = m.sample_euler(kw0["x"], 1, **kws, rel_quats=g["rel7"].to(dev), use_graph=False)
    assert rel_l2(x1.cpu(), (kw0["x"] + v0).cpu()) < 1e-5


def test_rigid_ops_fp32():
    from mdgen_amd.rigid_utils import Rigid, Rotation
    dev = _cuda()
    g = load_golden("rigid_ops")
    A = Rigid(Rotation(rot_mats=g["R1"].to(dev)), g["t1"].to(dev))
    Bq = Rigid(Rotation(rot_mats=g["R2"].to(dev)), g["t2"].to(dev))
    c = A.compose(Bq)
    assert torch.allclose(c.get_rots().get_rot_mats().cpu(), g["comp_R"], atol=1e-5)
    assert torch.allclose(c.get_trans().cpu(), g["comp_t"], atol=1e-4)
    inv = A.invert()
    assert torch.allclose(inv.get_rots().get_rot_mats().cpu(), g["inv_R"], atol=1e-6)
    assert torch.allclose(inv.get_trans().cpu(), g["inv_t"], atol=1e-4)
    assert torch.allclose(A.apply(g["p"].to(dev)).cpu(), g["apply"], atol=1e-4)
    assert torch.allclose(A.invert_apply(g["p"].to(dev)).cpu(), g["invert_apply"], atol=1e-4)
    t7 = A.to_tensor_7().cpu()
    sgn = torch.sign((t7[:, :4] * g["tensor7"][:, :4]).sum(-1, keepdim=True))   # reference eigh sign is arbitrary
    assert torch.allclose(t7[:, :4] * sgn, g["tensor7"][:, :4], atol=1e-5)
    assert (t7[:, 0] >= 0).all()
    f7 = Rigid.from_tensor_7(g["q7"].to(dev), normalize_quats=True)
    assert torch.allclose(f7.get_rots().get_rot_mats().cpu(), g["from7_R"], atol=1e-5)
    f3 = Rigid.from_3_points(g["p3a"].to(dev), g["p3b"].to(dev), g["p3c"].to(dev))
    assert torch.allclose(f3.get_rots().get_rot_mats().cpu(), g["f3_R"], atol=1e-5)
    assert torch.allclose(f3.get_trans().cpu(), g["f3_t"], atol=1e-6)
    # map_tensor_fn(sum) over a one-hot-masked axis selects a frame (geometry.py:257-260 usage)
    grp = Rigid(Rotation(rot_mats=g["R1"][:60].reshape(12, 5, 3, 3).to(dev)), g["t1"][:60].reshape(12, 5, 3).to(dev))
    onehot = torch.nn.functional.one_hot(torch.arange(12) % 5, 5).float().to(dev)
    sel = (grp * onehot).map_tensor_fn(lambda x: torch.sum(x, dim=-1))
    pick = torch.arange(12) % 5
    assert torch.allclose(sel.get_rots().get_rot_mats().cpu(), g["R1"][:60].reshape(12, 5, 3, 3)[torch.arange(12), pick], atol=1e-6)
    assert torch.allclose(sel.get_trans().cpu(), g["t1"][:60].reshape(12, 5, 3)[torch.arange(12), pick], atol=1e-6)
    # identities
    ident = A.compose(A.invert())
    assert torch.allclose(ident.get_rots().get_rot_mats().cpu(), torch.eye(3).expand(64, 3, 3), atol=1e-5)
    assert ident.get_trans().abs().max() < 1e-4


@pytest.mark.parametrize("name", ["prep_sim", "prep_tps", "prep_sim_interval"])
def test_prep_batch_vs_reference(name):
    """`prep_batch` (wrapper.py:283-365) against the reference's own outputs, incl. `--cond_interval` (wrapper.py:343-344:
    every k-th frame is a conditioning frame -- the upsampling models)."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.wrapper import NewMDGenWrapper, default_args
    dev = _cuda()
    g = load_golden(name)
    cfg = ModelConfig(**g["cfg"])
    args = default_args(cfg)
    args.cond_interval = int(g["cond_interval"]) or None
    w = NewMDGenWrapper(args)
    batch = {k[3:]: v.to(dev) for k, v in g.items() if k.startswith("in_")}
    prep = w.prep_batch(batch)
    assert torch.allclose(prep["latents"].cpu(), g["latents"], atol=5e-5)
    assert torch.allclose(prep["model_kwargs"]["x_cond"].cpu(), g["x_cond"], atol=5e-5)
    assert torch.equal(prep["model_kwargs"]["x_cond_mask"].cpu(), g["x_cond_mask"])
    assert torch.equal(prep["loss_mask"].float().cpu(), g["loss_mask"].float())
    assert torch.equal(prep["model_kwargs"]["mask"].cpu(), g["mask"])
    assert torch.allclose(prep["model_kwargs"]["start_frames"].get_rots().get_rot_mats().cpu(), g["start_rot"])
    assert torch.allclose(prep["model_kwargs"]["end_frames"].get_trans().cpu(), g["end_trans"])


def test_geometry_kernels_vs_reference():
    from mdgen_amd.geometry import atom14_to_cond, samples_to_atom14
    dev = _cuda()
    g = load_golden("geometry")
    B, T, L = g["atom14"].shape[:3]
    a = g["atom14"].reshape(B * T, L, 14, 3).to(dev)
    sq = g["seqres"][:, None].expand(B, T, L).reshape(B * T, L).to(dev)
    c = atom14_to_cond(a, sq)
    assert torch.allclose(c["rots"].cpu().view(B, T, L, 3, 3), g["frames_R"], atol=1e-5)
    assert torch.allclose(c["trans"].cpu().view(B, T, L, 3), g["frames_t"], atol=1e-5)
    assert torch.allclose(c["torsions"].cpu().view(B, T, L, 7, 2), g["torsions"], atol=1e-4)
    assert torch.allclose(c["torsion_mask"].cpu().view(B, T, L, 7), g["torsion_mask"])
    # frames + torsions -> atom14 through the sampler's post-processing kernel with identity offsets
    samples = torch.zeros(B * T, 1, L, 21, device=dev)
    samples[..., 0] = 1.0
    samples[..., 7:21] = c["torsions"].reshape(B * T, 1, L, 14)
    back = samples_to_atom14(samples, c["rots"], c["trans"], sq, tps=False)
    assert (back.cpu().view(B, T, L, 14, 3) - g["atom14_back"]).abs().max() < 2e-4


def test_inference_end_to_end_vs_reference():
    """NewMDGenWrapper.inference (noise -> S Euler steps -> atom14) with the reference's zs, plus the
    on-device rollout glue, vs the reference's own run (oracle/gen_golden.py gen_inference)."""
    from mdgen_amd.wrapper import NewMDGenWrapper
    from mdgen_amd.geometry import atom14_to_cond
    dev = _cuda()
    g = load_golden("inference_sim")
    cfg, sd = weights_for(g)
    w = NewMDGenWrapper(cfg)
    w.model.load_state_dict(sd)
    batch0 = {k[3:]: v.to(dev) for k, v in g.items() if k.startswith("in_")}
    T = g["S1_b0_zs"].shape[1]
    for S in [int(s) for s in g["steps"]]:
        ex = dict(batch0)
        ex["torsions"] = batch0["torsions"].expand(-1, T, -1, -1, -1)
        ex["trans"] = batch0["trans"].expand(-1, T, -1, -1)
        ex["rots"] = batch0["rots"].expand(-1, T, -1, -1, -1)
        for use_graph in (False, True, True):
            atom14, aa = w.inference(ex, zs=g[f"S{S}_b0_zs"].to(dev), num_steps=S, use_graph=use_graph)
            torch.cuda.synchronize()
            e_s = rel_l2(w.last_samples.cpu(), g[f"S{S}_b0_samples"])
            d = (atom14.cpu() - g[f"S{S}_b0_atom14"]).abs()
            print(f"S={S} graph={use_graph} samples rel-L2 {e_s:.2e}  atom14 rms {d.pow(2).mean().sqrt():.4f} A max {d.max():.4f} A")
            assert e_s < 2e-2
            # bf16-operand kernels are reported against rms <= 0.02 A / max <= 0.5 A
            assert d.pow(2).mean().sqrt() < tol_rms(S) and d.max() < TOL_MAX
            assert torch.equal(aa.cpu(), g["in_seqres"][:, None].expand_as(aa.cpu()))
        nxt = atom14_to_cond(g[f"S{S}_b0_atom14"][:, -1].to(dev), batch0["seqres"])
        assert torch.allclose(nxt["trans"].cpu(), g[f"S{S}_b0_next_trans"][:, 0], atol=1e-5)
        assert torch.allclose(nxt["rots"].cpu(), g[f"S{S}_b0_next_rots"][:, 0], atol=1e-5)
        assert torch.allclose(nxt["torsions"].cpu(), g[f"S{S}_b0_next_torsions"][:, 0], atol=2e-4)


def test_tps_inference_end_to_end_vs_oracle():
    """Two-sided (transition-path) sampling end to end: `tps_inference.get_sample` batch layout (start frame
    expanded over T, frame -1 = end frame; tps_inference.py:43-80) -> prep_batch (D = 28 latents, both offsets,
    cond frames 0 and -1) -> S Euler steps of the TPS model -> atom14, vs the CPU oracle with the kernel's
    w >= 0 quaternion convention for the relative frames (see test_forward_tps_vs_oracle_with_reference_inputs)."""
    from oracle import mdgen_oracle as O
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    from mdgen_amd.wrapper import NewMDGenWrapper
    from mdgen_amd.tps_inference import get_sample, collate
    from mdgen_amd.geometry import restype_order
    dev = _cuda()
    T, S, seqs = 12, 3, ["FLRH", "IMRY"]
    cfg = ModelConfig.tps(num_frames=T, crop=4)
    sd = synth_state_dict(cfg, 9)
    w = NewMDGenWrapper(cfg)
    w.model.load_state_dict(sd)
    gen = torch.Generator().manual_seed(77)
    samples, obatches = [], []
    for sq in seqs:
        seqres = torch.tensor([restype_order[c] for c in sq])
        ends = []
        for _ in range(2):   # two self-consistent conformations: random frames + torsions -> atom14 (oracle geometry)
            q = torch.randn(1, 1, 4, 4, generator=gen)
            R = O.quat_to_rot(q / q.norm(dim=-1, keepdim=True))
            tr = torch.cumsum(2.2 * torch.randn(1, 1, 4, 3, generator=gen), 2)
            ang = torch.randn(1, 1, 4, 7, 2, generator=gen)
            ang = ang / ang.norm(dim=-1, keepdim=True)
            ends.append(O.frames_torsions_to_atom14(R, tr, ang, seqres[None, None])[0])   # [1,L,14,3]
        samples.append(get_sample(ends[0].numpy(), ends[1].numpy(), sq, T, dev))
        ob = O.get_batch_from_atom14(torch.cat([ends[0]] * (T - 1) + [ends[1]], 0), seqres)
        obatches.append({k: (v[None] if torch.is_tensor(v) else v) for k, v in ob.items()})
    batch = collate(samples)
    obatch = {k: torch.cat([b[k] for b in obatches], 0) for k in obatches[0]}
    # the device glue and the oracle geometry agree on the conditioning batch.  Torsions are compared where
    # torsion_mask is set: a masked torsion (e.g. pre-omega of residue 0, whose "previous residue" is all zeros)
    # is a normalised rounding residue of a degenerate frame in the reference -- any unit vector or (0, 0).
    tm = obatch["torsion_mask"].float()
    assert torch.equal(batch["torsion_mask"].cpu().float(), tm)
    for k in ("trans", "rots", "torsions"):
        dk = (batch[k].cpu() - obatch[k].float()).abs()
        if k == "torsions":
            dk = dk * tm[:, None, :, :, None]
        print(f"conditioning batch {k}: max abs diff {dk.max():.3e}")
        assert dk.max() < 2e-4, (k, float(dk.max()))
    # ... and so that both sides consume IDENTICAL conditioning (masked torsions included, they do reach the
    # network through x_cond), the sampler below is fed the oracle's batch
    batch = {k: v.to(dev) for k, v in obatch.items()}
    zs = torch.randn(len(seqs), T, 4, cfg.latent_dim, generator=gen)
    atom14, aa = w.inference(batch, zs=zs.to(dev), num_steps=S, use_graph=False)
    torch.cuda.synchronize()
    c = dict(O.cfg_dict(cfg), quat_sign="w_nonneg")
    ref14, _, ref_samples = O.inference(sd, c, obatch, zs, S)
    e_s = rel_l2(w.last_samples.cpu(), ref_samples)
    d = (atom14.cpu() - ref14).abs()
    print(f"TPS end-to-end S={S}: samples rel-L2 {e_s:.2e}  atom14 rms {d.pow(2).mean().sqrt():.4f} A max {d.max():.4f} A")
    assert torch.isfinite(atom14).all()
    assert e_s < 2e-2
    assert d.pow(2).mean().sqrt() < tol_rms(S) and d.max() < TOL_MAX


def test_attention_fixed_anchor_and_robust_loops_agree():
    """Tiled attention (csrc/k_flash.hip): the default loop keeps ONE softmax shift per query row, anchored on the first
    key tile, and re-runs a (head, 64 queries) job on the moving-shift loop only if a row's denominator overflowed or
    the first tile is fully masked.  (a) On ordinary inputs both loops give the same output up to the bf16 rounding of
    P (the two shifts differ by a non-integer, so the mantissas of P differ): rel-L2 ~2e-3, the size of the bf16 error
    of the whole forward.  (b) With the q / k projections scaled so that scores move by hundreds of log2 units between
    key tiles, the fixed anchor overflows: the fallback must kick in and the result must stay finite and agree with the
    robust loop (the experiment build -DMDGEN_DEV_FLASH_NOFALLBACK fails exactly here with non-finite outputs:
    profiles/r02_flash_variants.txt).  (c) Padded residues (first tile fully masked on the time
    axis) take the robust loop from the start: covered by the n_pad > 0 shapes here and in the ATLAS goldens."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict, synth_forward_inputs
    from mdgen_amd.model import LatentMDGenModel
    dev = _cuda()
    B, T, L, n_pad = 2, 200, 33, 3
    cfg = ModelConfig(crop=L, num_frames=T, num_layers=2, abs_pos_emb=False, sim_condition=True)
    inp = synth_forward_inputs(cfg, B, T, L, n_pad, 11)
    kw = dict(x=inp["x"].to(dev), t=inp["t"].to(dev), mask=inp["mask"].to(dev),
              start_frames=(inp["start_rot"].to(dev), inp["start_trans"].to(dev)), x_cond=inp["x_cond"].to(dev),
              x_cond_mask=inp["x_cond_mask"].to(dev), aatype=inp["aatype"].to(dev))
    for scale, label in ((1.0, "ordinary"), (8.0, "partly overflowing"), (60.0, "overflowing")):
        sd = synth_state_dict(cfg, 4)
        for ax in ("mha_t", "mha_l"):
            for nm in ("q_proj", "k_proj"):
                sd[f"layers.0.{ax}.attn.{nm}.weight"] = sd[f"layers.0.{ax}.attn.{nm}.weight"] * scale
        m = LatentMDGenModel(cfg)
        m.load_state_dict(sd)
        m.set_option("attention_path", 0)
        a = m.forward(**kw).clone()
        m.set_option("attention_path", 1)
        b = m.forward(**kw).clone()
        assert torch.isfinite(a).all() and torch.isfinite(b).all(), label
        e = rel_l2(a.cpu(), b.cpu())
        print(f"attention loops, {label} scores: rel-L2(auto, robust) {e:.2e}; bitwise equal: {torch.equal(a, b)}")
        assert e < 5e-3, (label, e)
        del m


def test_rollout_on_poisoned_workspace():
    """A 2-step rollout at T = L = 64 (both multiples of 64: the learned bias key opens a key tile of its own on both
    axes and in the IPA attention; 4 padded residues: fully masked temporal sequences) on a workspace filled with
    0xFF bytes: finite, and bit-identical to the run on the freshly allocated workspace."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    from mdgen_amd.wrapper import NewMDGenWrapper
    import bench
    dev = _cuda()
    B, T, L, n_pad = 1, 64, 64, 4
    cfg = ModelConfig(crop=L, num_frames=T, num_layers=2, abs_pos_emb=False, sim_condition=True)
    w = NewMDGenWrapper(cfg, device=dev)
    w.model.load_state_dict(synth_state_dict(cfg, 0))
    batch = bench.synth_batch(B, T, L, n_pad, dev, seed=100)
    zs = torch.randn(B, T, L, cfg.latent_dim, generator=torch.Generator().manual_seed(137)).to(dev)
    a0, _ = w.inference(batch, zs=zs, num_steps=2, use_graph=False)
    a0 = a0.clone()
    assert len(w.model._ws) > 0
    for ws in w.model._ws.values():
        ws.view(torch.uint8).fill_(0xFF)
    a1, _ = w.inference(batch, zs=zs, num_steps=2, use_graph=False)
    assert torch.isfinite(a1).all()
    assert torch.equal(a0, a1)


def test_residue_axis_paths_agree():
    """L = 4 has three implementations of the residue-axis sub-layer, selected by the library option
    "residue_l4_path": 2 = the one-kernel default (k_ln_qkv_attn4<true>), 1 = attention fused but projection
    separate, 0 = the general L <= 8 path (k_ln_qkv<SMALL> -> k_proj<2>).  All three must meet the same gate
    against the reference output (fwd_full_pep golden) and agree with each other to bf16-operand noise."""
    dev = _cuda()
    g = load_golden("fwd_full_pep")
    cfg, sd = weights_for(g)
    m = get_model(cfg, sd, ("fwd_full_pep", "paths"))
    outs = {}
    try:
        for mode in (2, 1, 0):
            m.set_option("residue_l4_path", mode)
            outs[mode] = m.forward(**_kw(g, dev)).cpu()
    finally:
        m.set_option("residue_l4_path", 2)
    for mode, o in outs.items():
        e = rel_l2(o, g["out"])
        print(f"residue_l4_path={mode}: rel-L2 vs reference {e:.3e}")
        assert e < TOL_FWD
    assert not torch.equal(outs[0], outs[2])          # the option really switches the code path
    assert rel_l2(outs[1], outs[2]) < 5e-3 and rel_l2(outs[0], outs[2]) < 5e-3
    from mdgen_amd._lib import MdgenError
    with pytest.raises(MdgenError):
        m.set_option("residue_l4_path", 7)
    with pytest.raises(MdgenError):
        m.set_option("no_such_option", 1)


def test_training_losses_vs_reference():
    """Flow-matching target + masked loss (forward only): `Transport.training_losses` with the
    reference's draws of x0 and t on the full-width model, vs the REFERENCE's own output (train_full_sim golden);
    plus the two kernels on their own: `mdgen_path_plan` vs the oracle (fp32), `mdgen_masked_mse` vs torch."""
    from oracle import mdgen_oracle as O
    from mdgen_amd.transport import create_transport
    dev = _cuda()
    g = load_golden("train_full_sim")
    cfg, sd = weights_for(g)
    m = get_model(cfg, sd, ("train_full_sim", "w"))
    tr = create_transport()
    kw = dict(mask=g["mask"].to(dev), start_frames=(g["start_rot"].to(dev), g["start_trans"].to(dev)),
              x_cond=g["x_cond"].to(dev), x_cond_mask=g["x_cond_mask"].to(dev), aatype=g["aatype"].to(dev))
    terms = tr.training_losses(m.forward, g["x1"].to(dev), mask=g["loss_mask"].to(dev), model_kwargs=kw,
                               t=g["t"].to(dev), x0=g["x0"].to(dev))
    torch.cuda.synchronize()
    e_pred = rel_l2(terms["pred"].cpu(), g["pred"])
    e_loss = ((terms["loss"].cpu() - g["loss"]).abs() / g["loss"].abs()).max().item()
    print(f"training_losses: pred rel-L2 {e_pred:.2e}, loss rel err {e_loss:.2e}, loss {terms['loss'].tolist()}")
    assert e_pred < TOL_FWD and e_loss < 1e-2
    # kernels on their own
    for path_type in ("GVP", "Linear"):
        tr2 = create_transport(path_type=path_type)
        _, xt, ut = tr2.plan(g["t"].to(dev), g["x0"].to(dev), g["x1"].to(dev))
        rxt, rut = O.path_plan(g["t"], g["x0"], g["x1"], path_type)
        assert torch.allclose(xt.cpu(), rxt, atol=2e-6) and torch.allclose(ut.cpu(), rut, atol=5e-6), path_type
    from mdgen_amd._lib import lib, check, ptr, stream_ptr
    a, b = torch.randn(3, 1000, device=dev), torch.randn(3, 1000, device=dev)
    mk = (torch.rand(3, 1000, device=dev) > 0.3).float()
    out = torch.empty(3, device=dev)
    check(lib.mdgen_masked_mse(3, 1000, ptr(a), ptr(b), ptr(mk), ptr(out), stream_ptr()))
    torch.cuda.synchronize()
    ref = O.mean_flat(((a - b) ** 2).cpu(), mk.cpu())
    assert torch.allclose(out.cpu(), ref, rtol=1e-5)


def test_graph_replay_matches_eager_bitwise():
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    dev = _cuda()
    cfg = ModelConfig.forward_sim(num_frames=96, crop=4)
    sd = synth_state_dict(cfg, 3)
    m = get_model(cfg, sd, ("graph", 4))
    B, T, L = 2, 96, 4
    gen = torch.Generator().manual_seed(5)
    zs = torch.randn(B, T, L, 21, generator=gen).to(dev)
    mask = torch.ones(B, T, L, device=dev)
    R = torch.eye(3, device=dev).expand(B, L, 3, 3).contiguous()
    tr_ = torch.randn(B, L, 3, generator=gen).to(dev)
    cm = torch.zeros(B, T, L, dtype=torch.long, device=dev)
    cm[:, 0] = 1
    xc = torch.zeros(B, T, L, 21, device=dev)
    aat = torch.randint(0, 20, (B, L), generator=gen).to(dev)
    kw = dict(mask=mask, start_frames=(R, tr_), x_cond=xc, x_cond_mask=cm, aatype=aat)
    a = m.sample_euler(zs, 5, use_graph=False, **kw)
    b = m.sample_euler(zs, 5, use_graph=True, **kw)
    c = m.sample_euler(zs, 5, use_graph=True, **kw)
    torch.cuda.synchronize()
    assert torch.isfinite(a).all()
    assert torch.equal(a, b) and torch.equal(b, c)


def test_dual_stream_split_matches_single_stream():
    """The Euler rollout runs contiguous sub-batch views on concurrent streams (option "streams", default 2;
    DESIGN.md section 3); the result must match the single-stream run to rounding level (panel boundaries move,
    arithmetic does not), eager and graph-replayed alike."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    dev = _cuda()
    cfg = ModelConfig.forward_sim(num_frames=300, crop=4)
    sd = synth_state_dict(cfg, 3)
    m = get_model(cfg, sd, ("dual", 4))
    B, T, L = 5, 300, 4
    gen = torch.Generator().manual_seed(6)
    zs = torch.randn(B, T, L, 21, generator=gen).to(dev)
    mask = torch.ones(B, T, L, device=dev)
    R = torch.eye(3, device=dev).expand(B, L, 3, 3).contiguous()
    tr_ = torch.randn(B, L, 3, generator=gen).to(dev)
    cm = torch.zeros(B, T, L, dtype=torch.long, device=dev)
    cm[:, 0] = 1
    xc = torch.where(cm.unsqueeze(-1).bool(), torch.randn(B, T, L, 21, generator=gen).to(dev), torch.zeros((), device=dev))
    aat = torch.randint(0, 20, (B, L), generator=gen).to(dev)
    kw = dict(mask=mask, start_frames=(R, tr_), x_cond=xc, x_cond_mask=cm, aatype=aat)
    outs = {}
    try:
        for ns in (1, 2, 3):        # 3 streams: views of 2, 2 and 1 samples
            m.set_option("streams", ns)
            for g in (False, True):
                outs[ns, g] = m.sample_euler(zs, 4, use_graph=g, **kw)
    finally:
        m.set_option("streams", 2)
    torch.cuda.synchronize()
    ref = outs[1, False]
    assert torch.isfinite(ref).all()
    assert torch.equal(outs[1, True], ref)
    for ns in (2, 3):
        assert torch.equal(outs[ns, True], outs[ns, False])
        assert rel_l2(outs[ns, False], ref) < 1e-3


def test_full_size_properties_cfg2():
    """BASELINE cfg-2 size (B16 T1000 L4): properties that do not need the (slow) oracle:
    (1) the output is finite and bit-for-bit reproducible run to run (this caught two gfx950 code-generation
        hazards, DESIGN.md "hardware findings");
    (2) batch elements are independent: permuting the batch permutes the output (to fp32-rounding level:
        a token's row position inside an MFMA tile changes, which moves last-bit rounding and very rarely a
        bf16 rounding of an intermediate -- not bit-exact by construction, like any BLAS)."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    dev = _cuda()
    cfg = ModelConfig.forward_sim(num_frames=1000, crop=4)
    sd = synth_state_dict(cfg, 0)
    m = get_model(cfg, sd, ("cfg2", 4))
    B, T, L = 16, 1000, 4
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(B, T, L, 21, generator=gen).to(dev)
    t = torch.full((B,), 0.3, device=dev)
    mask = torch.ones(B, T, L, device=dev)
    q = torch.randn(B, L, 4, generator=gen)
    q = q / q.norm(dim=-1, keepdim=True)
    from mdgen_amd.rigid_utils import Rotation
    R = Rotation(quats=q.to(dev)).get_rot_mats()
    tr_ = torch.cumsum(2.2 * torch.randn(B, L, 3, generator=gen), 1).to(dev)
    cm = torch.zeros(B, T, L, dtype=torch.long, device=dev)
    cm[:, 0] = 1
    xc = torch.where(cm.unsqueeze(-1).bool(), torch.randn(B, T, L, 21, generator=gen).to(dev), torch.zeros((), device=dev))
    aat = torch.randint(0, 20, (B, L), generator=gen).to(dev)
    kw = dict(t=t, mask=mask, start_frames=(R, tr_), x_cond=xc, x_cond_mask=cm, aatype=aat)
    y1 = m.forward(x, **kw)
    y2 = m.forward(x, **kw)
    assert torch.isfinite(y1).all() and torch.equal(y1, y2)
    perm = torch.randperm(B, generator=gen).to(dev)
    kwp = dict(t=t[perm], mask=mask[perm], start_frames=(R[perm].contiguous(), tr_[perm].contiguous()),
               x_cond=xc[perm].contiguous(), x_cond_mask=cm[perm].contiguous(), aatype=aat[perm].contiguous())
    yp = m.forward(x[perm].contiguous(), **kwp)
    assert rel_l2(yp, y1[perm]) < 1e-3 and (yp - y1[perm]).abs().max() < 2e-2


def test_forward_cfg4_full_size_vs_reference_and_oracle():
    """BASELINE.json configs[3] at FULL size: ATLAS crop 256 x 250 frames, B 1, 16 padded residues.  Exercises what
    no small fixture reaches: residue-axis k_flash with 4 q-chunks x 9 key tiles and key padding, k_ipa_attn over
    4 x 256 x 256 logits (ipa.py:161-203), 1000 panels per launch.  (a) vs the REFERENCE's own run, stored
    sub-sampled (fwd_cfg4_atlas_full; inputs regenerated from the seed and checksummed); (b) vs the CPU oracle run
    here on the host cores (every element, every trace)."""
    from oracle import mdgen_oracle as O
    from mdgen_amd.synthetic import synth_forward_inputs, tensor_checksum
    dev = _cuda()
    g = load_golden("fwd_cfg4_atlas_full")
    cfg, sd = weights_for(g)
    B, T, L, n_pad = (int(v) for v in g["shape"])
    assert (B, T, L, n_pad) == (1, 250, 256, 16)
    inp = synth_forward_inputs(cfg, B, T, L, n_pad, int(g["data_seed"]))
    np.testing.assert_allclose(tensor_checksum(inp), g["input_checksum"].numpy(), rtol=1e-12)
    m = get_model(cfg, sd, ("cfg4", "w"))
    kw = dict(x=inp["x"], t=inp["t"], mask=inp["mask"], start_frames=(inp["start_rot"], inp["start_trans"]),
              end_frames=(inp["end_rot"], inp["end_trans"]), x_cond=inp["x_cond"], x_cond_mask=inp["x_cond_mask"],
              aatype=inp["aatype"])
    out, tr = m.forward(**{k: (tuple(u.to(dev) for u in v) if isinstance(v, tuple) else v.to(dev)) for k, v in kw.items()},
                        return_trace=True)
    torch.cuda.synchronize()
    out = out.cpu()
    tr = {k: v.cpu() for k, v in tr.items()}
    assert torch.isfinite(out).all()
    st, sl = (int(v) for v in g["sub"])
    ht, hl = (int(v) for v in g["sub_h"])
    nl = cfg.num_layers
    rep = {"out": rel_l2(out[:, ::st, ::sl], g["out"]), "ipa_out": rel_l2(tr["ipa_out"][:, ::sl], g["ipa_out"]),
           "h0": rel_l2(tr["h0"][:, ::ht, ::hl], g["h0"]), f"h{nl}": rel_l2(tr[f"h{nl}"][:, ::ht, ::hl], g[f"h{nl}"])}
    print("cfg-4 full vs reference (sub-sampled):", {k: f"{v:.2e}" for k, v in rep.items()})
    for k, v in rep.items():
        assert v < TOL_FWD, (k, v)
    ref, rtr = O.forward(sd, O.cfg_dict(cfg), return_trace=True, **kw)
    rep2 = {k: rel_l2(tr[k], rtr[k]) for k in ["ipa_out"] + [f"h{i}" for i in range(nl + 1)]}
    rep2["out"] = rel_l2(out, ref)
    print("cfg-4 full vs oracle (all elements):", {k: f"{v:.2e}" for k, v in rep2.items()})
    for k in ("ipa_out", "h0", f"h{nl}", "out"):
        assert rep2[k] < TOL_FWD, (k, rep2[k])
    # padded residues never influence the valid ones: same call with garbage in the padded inputs
    x2 = inp["x"].clone()
    x2[:, :, L - n_pad:] = 1e3
    kw2 = dict(kw, x=x2)
    # (two untraced calls: without a trace the FinalLayer runs as the last MLP launch's tail -- another summation order than k_final's)
    out1 = m.forward(**{k: (tuple(u.to(dev) for u in v) if isinstance(v, tuple) else v.to(dev)) for k, v in kw.items()}).cpu()
    out2 = m.forward(**{k: (tuple(u.to(dev) for u in v) if isinstance(v, tuple) else v.to(dev)) for k, v in kw2.items()}).cpu()
    assert torch.equal(out2[:, :, :L - n_pad], out1[:, :, :L - n_pad])
    assert rel_l2(out1, out) < 2e-3 and rel_l2(out1, ref) < TOL_FWD


@pytest.mark.parametrize("name", ["fwd_cfg2_T1000", "fwd_cfg1_T100"])
def test_forward_headline_regime_vs_reference_and_oracle(name):
    """BASELINE.json configs[1]'s regime -- tetrapeptide, **1000 frames**: 1001 temporal keys = 32 key tiles per query in
    k_flash (the fixed-anchor softmax runs 31 tiles past its anchor), RoPE positions up to 999 (mha.py:356-396), B 2 with
    distinct t -- and configs[0]'s exact shape (B 1, 100 frames).  (a) vs the REFERENCE's own run (outputs stored
    sub-sampled, inputs regenerated from the seed and checksummed), bf16 operands <= 1e-2 and fp32 mode <= 1e-5;
    (b) vs the CPU oracle run here, every element of the velocity and of every layer's residual stream."""
    from oracle import mdgen_oracle as O
    from mdgen_amd.model import LatentMDGenModel
    from mdgen_amd.synthetic import synth_forward_inputs, tensor_checksum
    dev = _cuda()
    g = load_golden(name)
    cfg, sd = weights_for(g)
    B, T, L, n_pad = (int(v) for v in g["shape"])
    inp = synth_forward_inputs(cfg, B, T, L, n_pad, int(g["data_seed"]))
    np.testing.assert_allclose(tensor_checksum(inp), g["input_checksum"].numpy(), rtol=1e-12)
    kw = dict(x=inp["x"], t=inp["t"], mask=inp["mask"], start_frames=(inp["start_rot"], inp["start_trans"]),
              end_frames=(inp["end_rot"], inp["end_trans"]), x_cond=inp["x_cond"], x_cond_mask=inp["x_cond_mask"],
              aatype=inp["aatype"])
    dkw = {k: (tuple(u.to(dev) for u in v) if isinstance(v, tuple) else v.to(dev)) for k, v in kw.items()}
    st, sl = (int(v) for v in g["sub"])
    ht, hl = (int(v) for v in g["sub_h"])
    nl = cfg.num_layers
    ref, rtr = O.forward(sd, O.cfg_dict(cfg), return_trace=True, **kw)
    # third pass (round 6): the reference's own T = 1000 golden through the HEADLINE's kernels, forced -- by shape this B 2 call takes
    # k_flash + the eight-wave panel kernels; the bench line's B 8 views take k_flash_proj8 / k_mlp_rows / k_ln_qkv<false, false>
    # (small_split 0: at B 2 the L = 4 sub-layer kernel would otherwise take its 32-row form, "attn_L_fused@h32")
    forced = {"flash_proj": 2, "flash_proj_form": 8, "mlp_path": 2, "panel_waves": 4, "small_split": 0}
    for prec, tol in (("bf16", TOL_FWD), ("fp32", 1e-5), ("bf16 headline kernels", TOL_FWD)):
        hk = prec == "bf16 headline kernels"
        if hk and T < 512:
            continue
        m = LatentMDGenModel(cfg, precision="bf16" if hk else prec)
        m.load_state_dict(sd)
        if hk:
            for k, v in forced.items():
                m.set_option(k, v)
            _, _, ran = _profiled_forward(m, dkw)
            for k in ("flash_proj_T@q128", "mlp", "ln_qkv_T", "attn_L_fused"):
                assert ran.get(k) == nl, (k, ran)
        out, tr = m.forward(**dkw, return_trace=True)
        torch.cuda.synchronize()
        out = out.cpu()
        tr = {k: v.cpu() for k, v in tr.items()}
        assert torch.isfinite(out).all()
        rep = {"out": rel_l2(out[:, ::st, ::sl], g["out"]), "ipa_out": rel_l2(tr["ipa_out"][:, ::sl], g["ipa_out"]),
               "h0": rel_l2(tr["h0"][:, ::ht, ::hl], g["h0"]), f"h{nl}": rel_l2(tr[f"h{nl}"][:, ::ht, ::hl], g[f"h{nl}"])}
        print(f"{name} {prec} vs reference (sub-sampled):", {k: f"{v:.2e}" for k, v in rep.items()})
        rep2 = {k: rel_l2(tr[k], rtr[k]) for k in ["ipa_out"] + [f"h{i}" for i in range(nl + 1)]}
        rep2["out"] = rel_l2(out, ref)
        # worst single frame of the velocity: a defect confined to the late key tiles / large RoPE positions cannot hide
        # in the average over 1000 frames
        per_t = ((out - ref).double().pow(2).sum((0, 2, 3)) / ref.double().pow(2).sum((0, 2, 3))).sqrt()
        rep2["out_worst_frame"] = float(per_t.max())
        print(f"{name} {prec} vs oracle (all elements):", {k: f"{v:.2e}" for k, v in rep2.items()})
        for k, v in list(rep.items()) + list(rep2.items()):
            assert v < (3 * tol if k == "out_worst_frame" else tol), (prec, k, v)
        del m
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name", ["inference_cfg2_T1000", "inference_cfg1"])
def test_inference_headline_regime_vs_reference(name):
    """End-to-end `inference()` (noise -> S Euler steps -> atom14) at BASELINE.json configs[1]'s regime (one sample of the 16:
    1000 frames, the reference's 49 Euler steps) and at configs[0]'s exact shape (B 1, 100 frames, 10 steps), against the
    REFERENCE's own run (sim_inference.py:109-115 region; atom14 stored at frames [::sub_t]); gated at BASELINE.md section
    3's bounds: bf16 operands rms <= 0.02 A / max <= 0.5 A, fp32 mode max <= 1e-3 A."""
    from mdgen_amd.wrapper import NewMDGenWrapper
    from mdgen_amd.synthetic import tensor_checksum
    dev = _cuda()
    g = load_golden(name)
    cfg, sd = weights_for(g)
    B, T, L = (int(v) for v in g["shape"])
    S, sub_t = int(g["S"]), int(g["sub_t"])
    zs = torch.randn(B, T, L, cfg.latent_dim, generator=torch.Generator().manual_seed(137))
    np.testing.assert_allclose(tensor_checksum({"zs": zs}), g["zs_checksum"].numpy(), rtol=1e-12)
    batch0 = {k[3:]: v.to(dev) for k, v in g.items() if k.startswith("in_")}
    ex = dict(batch0)
    ex["torsions"] = batch0["torsions"].expand(-1, T, -1, -1, -1)
    ex["trans"] = batch0["trans"].expand(-1, T, -1, -1)
    ex["rots"] = batch0["rots"].expand(-1, T, -1, -1, -1)
    for prec in ("bf16", "fp32"):
        w = NewMDGenWrapper(cfg, precision=prec)
        w.model.load_state_dict(sd)
        atom14, _ = w.inference(ex, zs=zs.to(dev), num_steps=S)
        torch.cuda.synchronize()
        assert torch.isfinite(atom14).all()
        d = (atom14.cpu()[:, ::sub_t] - g["atom14"]).abs()
        rms, mx = float(d.pow(2).mean().sqrt()), float(d.max())
        print(f"{name} S={S} {prec}: atom14 rms {rms:.4f} A max {mx:.4f} A")
        if prec == "bf16":
            assert rms < TOL_RMS and mx < TOL_MAX, (rms, mx)
        else:
            assert mx < 1e-3, mx
        del w
    torch.cuda.empty_cache()


def test_tps_cfg3_size_properties():
    """BASELINE.json configs[2] at its per-GPU size: TPS model (D = 28, two-sided conditioning, dual-stream IPA),
    crop 4, 100 frames, batch 256 / 8 GPUs = 32.  Size-independent properties of the Euler rollout through
    `NewMDGenWrapper.inference`: finite; bit-reproducible run to run (graph replay); a batch of 32 equals the
    concatenation of its two halves sampled separately (samples are independent; rounding level, since panel
    boundaries move); frame 0 / frame -1 conditioning is honoured by prep_batch."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    from mdgen_amd.wrapper import NewMDGenWrapper
    from mdgen_amd.tps_inference import get_sample, collate
    from mdgen_amd.geometry import samples_to_atom14
    from mdgen_amd.rigid_utils import Rotation
    dev = _cuda()
    B, T, L, S = 32, 100, 4, 10
    cfg = ModelConfig.tps(num_frames=T, crop=L)
    w = NewMDGenWrapper(cfg)
    w.model.load_state_dict(synth_state_dict(cfg, 11))
    gen = torch.Generator().manual_seed(303)
    seqs = ["FLRH", "IMRY", "AWKD", "GSTV"]
    samples = []
    for b in range(B):
        ends = []
        for _ in range(2):
            q = torch.randn(1, L, 4, generator=gen)
            R = Rotation(quats=(q / q.norm(dim=-1, keepdim=True)).to(dev)).get_rot_mats()
            tr_ = torch.cumsum(2.2 * torch.randn(1, L, 3, generator=gen), 1).to(dev)
            ang = 6.2831853 * torch.rand(1, L, 7, generator=gen)
            lat = torch.zeros(1, 1, L, 21, device=dev)
            lat[..., 0] = 1.0
            lat[..., 7:21] = torch.stack([ang.sin(), ang.cos()], -1).reshape(1, 1, L, 14).to(dev)
            from mdgen_amd.geometry import restype_order
            sq = torch.tensor([[restype_order[c] for c in seqs[b % 4]]], device=dev)
            ends.append(samples_to_atom14(lat, R, tr_, sq, tps=False)[0].cpu().numpy())   # [1,L,14,3]
        samples.append(get_sample(ends[0], ends[1], seqs[b % 4], T, dev))
    batch = collate(samples)
    prep = w.prep_batch(batch)
    cm = prep["model_kwargs"]["x_cond_mask"]
    assert cm[:, 0].all() and cm[:, -1].all() and not cm[:, 1:-1].any()
    assert prep["latents"].shape == (B, T, L, 28)
    zs = torch.randn(B, T, L, 28, generator=gen).to(dev)
    a1, _ = w.inference(batch, zs=zs, num_steps=S)
    s1 = w.last_samples.clone()
    a2, _ = w.inference(batch, zs=zs, num_steps=S)
    torch.cuda.synchronize()
    assert torch.isfinite(a1).all() and torch.equal(a1, a2)
    # independence of the samples: (i) the batch on ONE stream (full-batch launches) against the default two sub-batch
    # streams, which already run it as 2 x 16 (so comparing those with two separate 16-sample calls would be vacuous);
    # (ii) three unequal pieces 5 + 16 + 11 sampled separately -- panel / tile boundaries fall elsewhere in every piece
    w.model.set_option("streams", 1)
    w.inference(batch, zs=zs, num_steps=S)
    one = w.last_samples.clone()
    w.model.set_option("streams", 2)
    e1 = rel_l2(one, s1)
    pieces = []
    for lo, hi in ((0, 5), (5, 21), (21, 32)):
        hb = {k: v[lo:hi].contiguous() for k, v in batch.items()}
        w.inference(hb, zs=zs[lo:hi].contiguous(), num_steps=S)
        pieces.append(w.last_samples.clone())
    e2 = rel_l2(torch.cat(pieces, 0), s1)
    print(f"cfg-3 size TPS: one stream vs two sub-batch streams rel-L2 {e1:.2e}; batch 32 vs pieces 5 + 16 + 11 rel-L2 {e2:.2e}")
    assert e1 < 2e-3 and e2 < 2e-3


def test_inference_S49_error_growth():
    """The product default and the bench use S = 49 Euler steps (the reference's hard-coded 50-point grid,
    wrapper.py:441-442).  Report the error against the reference's own S = 1 / 10 / 49 runs (inference_sim golden)
    and gate S = 49 at the bf16-operand bounds of BASELINE.md section 3."""
    from mdgen_amd.wrapper import NewMDGenWrapper
    dev = _cuda()
    g = load_golden("inference_sim")
    assert 49 in [int(s) for s in g["steps"]]
    cfg, sd = weights_for(g)
    w = NewMDGenWrapper(cfg)
    w.model.load_state_dict(sd)
    batch0 = {k[3:]: v.to(dev) for k, v in g.items() if k.startswith("in_")}
    T = g["S49_b0_zs"].shape[1]
    ex = dict(batch0)
    ex["torsions"] = batch0["torsions"].expand(-1, T, -1, -1, -1)
    ex["trans"] = batch0["trans"].expand(-1, T, -1, -1)
    ex["rots"] = batch0["rots"].expand(-1, T, -1, -1, -1)
    rows = {}
    for S in (1, 10, 49):
        atom14, _ = w.inference(ex, zs=g[f"S{S}_b0_zs"].to(dev), num_steps=S)
        d = (atom14.cpu() - g[f"S{S}_b0_atom14"]).abs()
        rows[S] = (rel_l2(w.last_samples.cpu(), g[f"S{S}_b0_samples"]), float(d.pow(2).mean().sqrt()), float(d.max()))
        print(f"S={S:2d}: samples rel-L2 {rows[S][0]:.2e}  atom14 rms {rows[S][1]:.4f} A  max {rows[S][2]:.4f} A")
    assert rows[49][0] < 2e-2 and rows[49][1] < TOL_RMS and rows[49][2] < TOL_MAX
    assert rows[10][1] < TOL_RMS and rows[10][2] < TOL_MAX and rows[1][1] < tol_rms(1)


def test_multi_block_rollout_one_graph():
    """`NewMDGenWrapper.rollout` = `mdgen_rollout_euler`: R chained blocks (prep -> S Euler steps -> atom14 -> next
    conditioning frame) in one library call / one hipGraph (sim_inference.py:100-113).  (a) identical, bit for bit,
    to chaining `inference()` + `atom14_to_cond` from Python block by block (same kernels, same order); (b) graph
    replay == eager; (c) vs the reference's own two chained blocks (inference_sim golden, S = 10): block 0 at the
    single-block gate, block 1 (conditioned on OUR block-0 end frame, i.e. error carried over) reported and gated
    at twice that."""
    from mdgen_amd.wrapper import NewMDGenWrapper
    from mdgen_amd.sim_inference import rollout as py_rollout
    dev = _cuda()
    g = load_golden("inference_sim")
    cfg, sd = weights_for(g)
    w = NewMDGenWrapper(cfg)
    w.model.load_state_dict(sd)
    batch0 = {k[3:]: v.to(dev) for k, v in g.items() if k.startswith("in_")}
    S, R = 10, 2
    T = g[f"S{S}_b0_zs"].shape[1]
    zs = torch.stack([g[f"S{S}_b{r}_zs"] for r in range(R)]).to(dev)
    chained, cur = [], dict(batch0)
    for r in range(R):
        a, cur = py_rollout(w, cur, T, S, zs=zs[r])
        chained.append(a)
    chained = torch.cat(chained, 1)
    one_e, nxt_e = w.rollout(batch0, T, R, num_steps=S, zs=zs, use_graph=False, return_next=True)
    one_g = w.rollout(batch0, T, R, num_steps=S, zs=zs, use_graph=True)
    one_g2 = w.rollout(batch0, T, R, num_steps=S, zs=zs, use_graph=True)
    torch.cuda.synchronize()
    assert torch.isfinite(one_e).all()
    assert torch.equal(one_e, chained)
    assert torch.equal(one_g, one_e) and torch.equal(one_g2, one_e)
    for k in ("trans", "rots", "torsions"):
        assert torch.equal(nxt_e[k], cur[k]), k
    for r in range(R):
        d = (one_g[:, r * T:(r + 1) * T].cpu() - g[f"S{S}_b{r}_atom14"]).abs()
        rms, mx = float(d.pow(2).mean().sqrt()), float(d.max())
        print(f"rollout block {r} vs reference: atom14 rms {rms:.4f} A max {mx:.4f} A")
        assert rms < TOL_RMS * (r + 1) and mx < TOL_MAX * (r + 1)


def test_many_views_beyond_the_32bit_offset_limit():
    """One launch addresses the residual stream with 32-bit byte offsets (token * 1536): at most 2 796 202 token rows.
    A batch of 45 ATLAS-size samples (2.88 M tokens) must therefore run as two sub-batch launch views
    (`mdgen_debug_view_plan`) and give, for every sample, what that sample gives alone (rounding level)."""
    import ctypes as C
    import mdgen_amd._lib as L
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict, synth_forward_inputs
    dev = _cuda()
    B, T, L_, n_pad = 45, 250, 256, 16
    nv, per = C.c_int32(), C.c_int32()
    L.check(L.lib.mdgen_debug_view_plan(C.byref(L.Shape(B, T, L_)), 1, C.byref(nv), C.byref(per)))
    assert nv.value == 2 and per.value == 23
    cfg = ModelConfig.atlas(num_frames=T, crop=L_)
    sd = synth_state_dict(cfg, 6)
    m = get_model(cfg, sd, ("cfg4", "w"))
    one = synth_forward_inputs(cfg, 3, T, L_, n_pad, 91)          # three distinct samples, tiled over the batch
    idx = torch.arange(B) % 3

    def kw_for(sel):
        k = {n: one[n][sel].contiguous() for n in ("x", "t", "mask", "x_cond", "x_cond_mask", "aatype")}
        k["start_frames"] = (one["start_rot"][sel].contiguous().to(dev), one["start_trans"][sel].contiguous().to(dev))
        return {n: (v if isinstance(v, tuple) else v.to(dev)) for n, v in k.items()}
    # (the same kernel forms in both calls: with `small_split` the IPA stack's MLP of the 3-sample call -- 12 panels -- would take
    # the three-workgroup form and the 45-sample call's -- 180 panels -- not: 2.4e-3 of summation-order rounding, measured)
    m.set_option("small_split", 0)
    try:
        big = m.forward(**kw_for(idx))
        torch.cuda.synchronize()
        assert torch.isfinite(big).all()
        small = m.forward(**kw_for(torch.arange(3)))
        for b in (0, 1, 22, 23, 44):          # both views, both ends of each
            assert rel_l2(big[b], small[b % 3]) < 1e-3, b
    finally:
        m.set_option("small_split", 1)        # (the model is shared with other tests)
    del big
    m._ws.clear()
    torch.cuda.empty_cache()


def test_checkpoint_and_cli_end_to_end(tmp_path):
    """The boundary's file half: a Lightning-layout checkpoint ({'state_dict': {'model.*'}, 'hyper_parameters':
    {'args': Namespace}}, wrapper.py:50,120-130) -> `NewMDGenWrapper.load_from_checkpoint` -> the
    `sim_inference`-compatible CLI on a synthetic fp16 `.npy` trajectory + split CSV -> `{name}.pdb`, equal to what
    the Python API gives for the same seed; a checkpoint whose args say `sampling_method='dopri5'` (the reference's
    argparse default) is REFUSED unless `--num_steps` is given."""
    import argparse
    import pandas as pd
    from mdgen_amd._lib import MdgenError
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.geometry import restype_order, samples_to_atom14
    from mdgen_amd.rigid_utils import Rotation
    from mdgen_amd.synthetic import synth_state_dict
    from mdgen_amd.wrapper import NewMDGenWrapper
    from mdgen_amd import sim_inference as cli
    from mdgen_amd.pdb import frames_to_pdb_string
    dev = _cuda()
    T, R, S = 24, 2, 3
    cfg = ModelConfig.forward_sim(num_frames=T, crop=4)
    sd = synth_state_dict(cfg, 21)

    def write_ckpt(path, sampling_method):
        a = argparse.Namespace(**cfg.to_dict(), path_type="GVP", prediction="velocity", sampling_method=sampling_method,
                               lr=1e-4, batch_size=8, ema=False)
        torch.save({"state_dict": {"model." + k: v for k, v in sd.items()}, "hyper_parameters": {"args": a},
                    "epoch": 3, "global_step": 1234}, path)
    ck_euler, ck_dopri = str(tmp_path / "euler.ckpt"), str(tmp_path / "dopri5.ckpt")
    write_ckpt(ck_euler, "euler")
    write_ckpt(ck_dopri, "dopri5")
    # synthetic MD data: {data_dir}/{name}.npy fp16 [frames, L, 14, 3] (scripts/prep_sims.py:54-77), split CSV name,seqres
    names = {"FLRH": "FLRH", "IMRY": "IMRY", "AWKD": "AWKD"}
    gen = torch.Generator().manual_seed(5)
    data = tmp_path / "data"
    data.mkdir()
    for n, sq in names.items():
        q = torch.randn(1, 4, 4, generator=gen)
        Rm = Rotation(quats=(q / q.norm(dim=-1, keepdim=True)).to(dev)).get_rot_mats()
        tr_ = torch.cumsum(2.2 * torch.randn(1, 4, 3, generator=gen), 1).to(dev)
        ang = 6.2831853 * torch.rand(1, 4, 7, generator=gen)
        lat = torch.zeros(1, 1, 4, 21, device=dev)
        lat[..., 0] = 1.0
        lat[..., 7:21] = torch.stack([ang.sin(), ang.cos()], -1).reshape(1, 1, 4, 14).to(dev)
        a14 = samples_to_atom14(lat, Rm, tr_, torch.tensor([[restype_order[c] for c in sq]], device=dev), tps=False)[0]
        np.save(data / f"{n}.npy", a14.cpu().numpy().repeat(3, 0).astype(np.float16))
    split = tmp_path / "split.csv"
    pd.DataFrame({"name": list(names), "seqres": list(names.values())}).to_csv(split, index=False)
    out = tmp_path / "out"
    base = ["--data_dir", str(data), "--split", str(split), "--out_dir", str(out), "--num_frames", str(T),
            "--num_rollouts", str(R), "--npy"]
    # (1) dopri5 checkpoint without --num_steps: refused, loudly
    with pytest.raises(MdgenError, match="dopri5"):
        cli.main(["--sim_ckpt", ck_dopri] + base)
    # (2) the same checkpoint with explicit Euler steps runs; (3) Euler checkpoint, batched, one device rollout
    torch.manual_seed(1234)
    res = cli.main(["--sim_ckpt", ck_dopri, "--num_steps", str(S), "--batch", "3"] + base)
    assert res["names"] == list(names) and res["frames"] == 3 * R * T
    got = {n: np.load(out / f"{n}.npy") for n in names}
    for n in names:
        assert got[n].shape == (R * T, 4, 14, 3) and np.isfinite(got[n]).all()
        text = open(out / f"{n}.pdb").read()
        assert text.count("MODEL") == R * T
        assert text == frames_to_pdb_string(got[n], np.array([restype_order[c] for c in names[n]]))
    # the same through the Python API (same seed -> same device noise)
    w = NewMDGenWrapper.load_from_checkpoint(ck_euler)
    assert w.cfg == cfg and w.args.sampling_method == "euler"
    batch = cli.collate([cli.get_batch(np.load(data / f"{n}.npy"), names[n], dev) for n in names])
    torch.manual_seed(1234)
    api = w.rollout(batch, T, R, num_steps=S).cpu().numpy()
    for i, n in enumerate(names):
        assert np.array_equal(api[i], got[n]), n
    # (4) --chunk_idx / --n_chunks (tps_inference.py:160-161) and --pdb_id select the work; B = 1 per call
    res = cli.main(["--sim_ckpt", ck_euler, "--chunk_idx", "1", "--n_chunks", "2", "--per_block"] + base)
    assert res["names"] == ["AWKD"]
    res = cli.main(["--sim_ckpt", ck_euler, "--pdb_id", "IMRY"] + base)
    assert res["names"] == ["IMRY"]
    # flags of models outside the path are refused at load time
    bad = argparse.Namespace(**cfg.to_dict(), sampling_method="euler", design=True)
    torch.save({"state_dict": {}, "hyper_parameters": {"args": bad}}, tmp_path / "bad.ckpt")
    with pytest.raises(MdgenError, match="design"):
        NewMDGenWrapper.load_from_checkpoint(str(tmp_path / "bad.ckpt"))
    both = argparse.Namespace(**dict(cfg.to_dict(), tps_condition=True), sampling_method="euler")
    torch.save({"state_dict": {}, "hyper_parameters": {"args": both}}, tmp_path / "both.ckpt")
    with pytest.raises(MdgenError, match="exactly one"):
        NewMDGenWrapper.load_from_checkpoint(str(tmp_path / "both.ckpt"))


def test_use_graph_flag_reaches_the_library():
    """`inference(use_graph=False)` must launch eagerly (the flag used to be dropped): with profiling
    off, an eager call leaves the context's graph cache untouched, a graph call adds exactly one entry -- observed
    through the launch counts of the profiling report, which only eager launches feed."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    from mdgen_amd.wrapper import NewMDGenWrapper
    dev = _cuda()
    cfg = ModelConfig.forward_sim(num_frames=40, crop=4)
    w = NewMDGenWrapper(cfg)
    w.model.load_state_dict(synth_state_dict(cfg, 2))
    seen = []
    orig = w.model.sample_euler

    def spy(*a, **k):
        seen.append(k.get("use_graph"))
        return orig(*a, **k)
    w.model.sample_euler = spy
    gen = torch.Generator().manual_seed(0)
    B, T, L_ = 2, 40, 4
    from mdgen_amd.rigid_utils import Rotation
    q = torch.randn(B, 1, L_, 4, generator=gen)
    Rm = Rotation(quats=(q / q.norm(dim=-1, keepdim=True)).to(dev)).get_rot_mats()
    ang = 6.2831853 * torch.rand(B, 1, L_, 7, generator=gen)
    batch = {"torsions": torch.stack([ang.sin(), ang.cos()], -1).expand(B, T, L_, 7, 2).contiguous().to(dev),
             "torsion_mask": torch.ones(B, L_, 7, device=dev), "trans": torch.randn(B, 1, L_, 3, generator=gen).expand(B, T, L_, 3).contiguous().to(dev),
             "rots": Rm.expand(B, T, L_, 3, 3).contiguous(), "seqres": torch.randint(0, 20, (B, L_), generator=gen).to(dev),
             "mask": torch.ones(B, L_, device=dev)}
    zs = torch.randn(B, T, L_, 21, generator=gen).to(dev)
    a, _ = w.inference(batch, zs=zs, num_steps=2, use_graph=False)
    b, _ = w.inference(batch, zs=zs, num_steps=2, use_graph=True)
    assert seen == [False, True]
    assert torch.equal(a, b)


TOL_FP32 = 1e-5


@pytest.mark.parametrize("name", ["fwd_full_sim", "fwd_full_pep", "fwd_full_atlas", "fwd_full_tps"])
def test_fp32_mode_forward_vs_reference_golden(name):
    """Option "precision" = 32 (csrc/k_fp32.hip): fp32 operands on v_mfma_f32_32x32x2_f32, the reference's own
    arithmetic, gated at BASELINE.md section 3's fp32 bound rel-L2 <= 1e-5 against the REFERENCE's outputs -- the
    two-sided model included: the reference's own relative-frame 7-vectors (fixture key `rel7`) are handed over, see
    test_forward_tps_vs_oracle_with_reference_inputs."""
    from oracle import mdgen_oracle as O
    from mdgen_amd.model import LatentMDGenModel
    dev = _cuda()
    g = load_golden(name)
    cfg, sd = weights_for(g)
    m = LatentMDGenModel(cfg, precision="fp32")
    m.load_state_dict(sd)
    # the two-sided model against the REFERENCE's golden too: its own relative-frame 7-vectors are handed over (`rel7`)
    extra = {"rel_quats": g["rel7"].to(dev)} if cfg.tps_condition else {}
    out, tr = m.forward(**_kw(g, dev), **extra, return_trace=True)
    torch.cuda.synchronize()
    nl = cfg.num_layers
    want = {k: g[k] for k in ("ipa_out", "h0", f"h{nl}", "out")}
    got = {"ipa_out": tr["ipa_out"], "h0": tr["h0"], f"h{nl}": tr[f"h{nl}"], "out": out}
    rep = {k: rel_l2(got[k].cpu(), want[k]) for k in want}
    print(name, "fp32 mode:", {k: f"{v:.2e}" for k, v in rep.items()})
    assert torch.isfinite(out).all()
    for k, v in rep.items():
        assert v < TOL_FP32, (k, v)
    # the same context switched to bf16 operands gives the (different) bf16-gate result
    m.set_precision("bf16")
    out16 = m.forward(**_kw(g, dev), **extra)
    assert not torch.equal(out16, out)
    e16 = rel_l2(out16.cpu(), want["out"])
    assert TOL_FP32 < e16 < TOL_FWD, e16


def test_fp32_mode_inference_end_to_end_vs_reference():
    """fp32 mode through `NewMDGenWrapper.inference`, S = 1 / 10 / 49 Euler steps with the reference's noise:
    BASELINE.md section 3 gate for fp32 kernels, end-to-end atom14 max-abs <= 1e-3 A (and torsion / offset samples
    to rel-L2 <= 1e-5 x steps), eager and graph-replayed alike."""
    from mdgen_amd.wrapper import NewMDGenWrapper
    dev = _cuda()
    g = load_golden("inference_sim")
    cfg, sd = weights_for(g)
    w = NewMDGenWrapper(cfg, precision="fp32")
    w.model.load_state_dict(sd)
    batch0 = {k[3:]: v.to(dev) for k, v in g.items() if k.startswith("in_")}
    T = g["S1_b0_zs"].shape[1]
    ex = dict(batch0)
    ex["torsions"] = batch0["torsions"].expand(-1, T, -1, -1, -1)
    ex["trans"] = batch0["trans"].expand(-1, T, -1, -1)
    ex["rots"] = batch0["rots"].expand(-1, T, -1, -1, -1)
    for S in (1, 10, 49):
        for use_graph in (False, True):
            atom14, _ = w.inference(ex, zs=g[f"S{S}_b0_zs"].to(dev), num_steps=S, use_graph=use_graph)
            e_s = rel_l2(w.last_samples.cpu(), g[f"S{S}_b0_samples"])
            d = (atom14.cpu() - g[f"S{S}_b0_atom14"]).abs()
            print(f"fp32 mode S={S:2d} graph={use_graph}: samples rel-L2 {e_s:.2e}  atom14 max {float(d.max()):.2e} A")
            assert e_s < 1e-5 * max(S, 3) and float(d.max()) < 1e-3


def test_fp32_mode_cfg4_full_size_vs_reference():
    """fp32 mode at BASELINE.json configs[3]'s full size (ATLAS 256 x 250, 16 padded residues) against the
    reference's sub-sampled outputs: rel-L2 <= 1e-5 on the velocity, the IPA table and the residual stream."""
    from mdgen_amd.model import LatentMDGenModel
    from mdgen_amd.synthetic import synth_forward_inputs
    dev = _cuda()
    g = load_golden("fwd_cfg4_atlas_full")
    cfg, sd = weights_for(g)
    B, T, L, n_pad = (int(v) for v in g["shape"])
    inp = synth_forward_inputs(cfg, B, T, L, n_pad, int(g["data_seed"]))
    m = LatentMDGenModel(cfg, precision="fp32")
    m.load_state_dict(sd)
    out, tr = m.forward(x=inp["x"].to(dev), t=inp["t"].to(dev), mask=inp["mask"].to(dev),
                        start_frames=(inp["start_rot"].to(dev), inp["start_trans"].to(dev)),
                        x_cond=inp["x_cond"].to(dev), x_cond_mask=inp["x_cond_mask"].to(dev), aatype=inp["aatype"].to(dev),
                        return_trace=True)
    torch.cuda.synchronize()
    st, sl = (int(v) for v in g["sub"])
    ht, hl = (int(v) for v in g["sub_h"])
    nl = cfg.num_layers
    rep = {"out": rel_l2(out.cpu()[:, ::st, ::sl], g["out"]), "ipa_out": rel_l2(tr["ipa_out"].cpu()[:, ::sl], g["ipa_out"]),
           "h0": rel_l2(tr["h0"].cpu()[:, ::ht, ::hl], g["h0"]), f"h{nl}": rel_l2(tr[f"h{nl}"].cpu()[:, ::ht, ::hl], g[f"h{nl}"])}
    print("cfg-4 full, fp32 mode vs reference:", {k: f"{v:.2e}" for k, v in rep.items()})
    for k, v in rep.items():
        assert v < TOL_FP32, (k, v)
    del m
    torch.cuda.empty_cache()


def test_dataset_window_crop_pad_vs_reference(tmp_path):
    """`MDGenDataset.__getitem__` (dataset.py:19-100) against the reference's own items (tests/golden/dataset.npz,
    oracle/gen_golden_dataset.py): same numpy seeds -> same replica / window / crop, ATLAS crop (L 14 > 8), ATLAS padding
    (L 6 < 8: identity frames, zero torsions, mask 0) and the uncropped tetrapeptide case; geometry on the GPU."""
    import argparse
    import pandas as pd
    from mdgen_amd.dataset import MDGenDataset
    _cuda()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dataset.npz"))
    for k in g.files:
        if k.startswith("arr_"):
            np.save(tmp_path / (k[4:] + ".npy"), g[k])
    seqs = dict(zip([str(x) for x in g["seq_names"]], [str(x) for x in g["seq_strings"]]))
    done = 0
    for tag, names, atlas, crop in (("atlas", ["pLong", "pShort"], True, 8), ("pep", ["FLRH"], False, 4)):
        split = tmp_path / f"{tag}.csv"
        pd.DataFrame({"name": names, "seqres": [seqs[n] for n in names]}).to_csv(split, index=False)
        args = argparse.Namespace(data_dir=str(tmp_path), suffix="", atlas=atlas, crop=crop, num_frames=4, overfit=False,
                                  overfit_peptide=None, overfit_frame=False, frame_interval=None, copy_frames=False,
                                  no_frames=False)
        ds = MDGenDataset(args, str(split), repeat=2)
        assert len(ds) == 2 * len(names)
        for seed in (0, 1, 2, 3):
            for idx in range(len(ds)):
                np.random.seed(100 * seed + idx)
                it = ds[idx]
                key = f"{tag}_s{seed}_i{idx}"
                assert it["name"] == str(g[key + "_name"]) and int(it["frame_start"]) == int(g[key + "_frame_start"]), key
                assert np.array_equal(it["seqres"].cpu().numpy(), g[key + "_seqres"]), key
                assert np.array_equal(it["mask"].cpu().numpy(), g[key + "_mask"]), key
                assert np.array_equal(it["torsion_mask"].cpu().numpy(), g[key + "_torsion_mask"]), key
                assert np.abs(it["rots"].cpu().numpy() - g[key + "_rots"]).max() < 1e-5, key
                assert np.abs(it["trans"].cpu().numpy() - g[key + "_trans"]).max() < 1e-5, key
                tm = g[key + "_torsion_mask"][None, :, :, None]            # masked torsions are degenerate in the reference
                assert (np.abs(it["torsions"].cpu().numpy() - g[key + "_torsions"]) * tm).max() < 2e-4, key
                done += 1
    assert done == 24
    # the items collate and feed prep_batch (the training step's first stage)
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.wrapper import NewMDGenWrapper
    loader = torch.utils.data.DataLoader(ds, batch_size=2, num_workers=0)
    batch = next(iter(loader))
    w = NewMDGenWrapper(ModelConfig.forward_sim(num_frames=4, crop=4))
    prep = w.prep_batch(batch)
    assert prep["latents"].shape == (2, 4, 4, 21) and torch.isfinite(prep["latents"]).all()


def test_optimizer_kernels_vs_torch():
    """csrc/k_optim.hip against torch itself on the CPU: gradient-norm clipping (clip_grad_norm_, what Lightning's
    gradient_clip_val does, train.py:56) + torch.optim.Adam / AdamW (wrapper.py:167-172) over several steps, DDP's
    1 / world averaging folded in, and the weight EMA (ema.py:41-58)."""
    from collections import OrderedDict
    from mdgen_amd.optim import FlatParams, Adam, EMA
    dev = _cuda()
    shapes = OrderedDict([("a.weight", (257, 129)), ("a.bias", (257,)), ("b.weight", (1000, 999)), ("c", (3, 5, 7))])
    gen = torch.Generator().manual_seed(0)
    sd = OrderedDict((k, torch.randn(*v, generator=gen)) for k, v in shapes.items())
    for adamw, clip, world in ((False, 1.0, 1), (True, 0.5, 8), (False, None, 2)):
        fp = FlatParams(shapes, device=dev).load_state_dict(sd)
        opt = Adam(fp, lr=1e-2, adamw=adamw, grad_clip=clip)
        ema = EMA(fp, 0.99)
        ref_p = [torch.nn.Parameter(v.clone()) for v in sd.values()]
        ref_opt = (torch.optim.AdamW if adamw else torch.optim.Adam)(ref_p, lr=1e-2)
        ref_ema = [p.detach().clone() for p in ref_p]
        for step in range(6):
            gs = [torch.randn(*v, generator=gen) * (10.0 if step % 2 else 0.01) for v in shapes.values()]   # clipped / not
            flat = torch.cat([x.reshape(-1) for x in gs]).to(dev)
            n_dev = opt.grad_norm(flat, 1.0 / world).item()
            for p, x in zip(ref_p, gs):
                p.grad = x / world                                # DDP: mean over ranks of the summed gradient
            if clip is not None:
                n_ref = float(torch.nn.utils.clip_grad_norm_(ref_p, clip))
                assert abs(n_dev - n_ref) / n_ref < 3e-5      # torch's own fp32 norm-of-norms carries ~1e-5
            ref_opt.step()
            opt.step(flat, grad_scale=1.0 / world)
            ema.update()
            for e, p in zip(ref_ema, ref_p):                      # ema.py:47-51
                diff = e - p.detach()
                diff *= 1 - 0.99
                e -= diff
        got = fp.state_dict()
        for (k, v), p, e in zip(got.items(), ref_p, ref_ema):
            assert torch.allclose(v.cpu(), p.detach(), rtol=2e-5, atol=1e-6), (k, adamw, clip)
            assert torch.allclose(ema.state_dict()["params"][k].cpu(), e, rtol=2e-5, atol=1e-6), k
        assert opt.step_count == 6


def _train_case(cfg, B, T, L, seed):
    """Seeded inputs of one training step: flow-matching pair (xt, ut) from the library's plan, random loss mask with
    one padded residue in the last sample, per-sample t."""
    from oracle import mdgen_oracle as O
    gen = torch.Generator().manual_seed(seed)
    D = cfg.latent_dim
    x1 = torch.randn(B, T, L, D, generator=gen)
    x0 = torch.randn(B, T, L, D, generator=gen)
    t = torch.rand(B, generator=gen)
    mask = torch.ones(B, L)
    mask[-1, L - 1:] = 0
    mask_btl = mask[:, None].expand(B, T, L).contiguous()
    loss_mask = (torch.rand(B, T, L, D, generator=gen) > 0.2).float() * mask_btl[..., None]
    aatype = torch.randint(0, 20, (B, L), generator=gen)
    q = torch.randn(B, L, 4, generator=gen)
    sR = O.quat_to_rot(q / q.norm(dim=-1, keepdim=True))
    st = torch.cumsum(2.2 * torch.randn(B, L, 3, generator=gen), 1)
    cm = torch.zeros(B, T, L, dtype=torch.long)
    cm[:, 0] = 1
    x_cond = torch.where(cm.unsqueeze(-1).bool(), x1, torch.zeros(()))
    return dict(x1=x1, x0=x0, t=t, mask=mask_btl, loss_mask=loss_mask, aatype=aatype, sR=sR, st=st, cm=cm, x_cond=x_cond)


@pytest.mark.parametrize("shape", [(2, 6, 5, 2), (1, 40, 33, 1)])
def test_training_step_gradients_vs_autograd(shape):
    """`mdgen_train_forward_backward` (fp32 forward with tape + backward kernels) against torch autograd through the CPU
    oracle (itself pinned to the reference's forward, loss AND gradients: tests/test_oracle_cpu.py) for EVERY trainable
    tensor of the full-width model: rel-L2 <= 2e-4 per tensor (fp32 summation-order noise), loss to 1e-5.  Shapes:
    micro residue axis (L = 5) with a padded residue; flash-sized residue axis (L = 33) with partial tiles."""
    from oracle import mdgen_oracle as O
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    from mdgen_amd.train import TrainableModel, trainable_shapes
    dev = _cuda()
    B, T, L, nl = shape
    cfg = ModelConfig(crop=L, num_frames=T, num_layers=nl, abs_pos_emb=True, sim_condition=True)
    sd = synth_state_dict(cfg, 17)
    c = _train_case(cfg, B, T, L, 1000 + T)
    # reference gradients: autograd through the oracle
    P = {k: (v.clone().requires_grad_(True) if k in trainable_shapes(cfg) else v) for k, v in sd.items()}
    kw = dict(mask=c["mask"], start_frames=(c["sR"], c["st"]), end_frames=(c["sR"], c["st"]), x_cond=c["x_cond"],
              x_cond_mask=c["cm"], aatype=c["aatype"])
    with torch.enable_grad():
        ref = O.training_losses(P, O.cfg_dict(cfg), c["x1"], c["loss_mask"], kw, c["t"], c["x0"])
        ref["loss"].mean().backward()
    xt, ut = O.path_plan(c["t"], c["x0"], c["x1"], "GVP")
    tm = TrainableModel(cfg, dev).load_state_dict(sd)
    tm.zero_grad()
    loss, pred = tm.forward_backward(xt.to(dev), c["t"].to(dev), ut.to(dev), c["loss_mask"].to(dev), c["mask"].to(dev),
                                     (c["sR"].to(dev), c["st"].to(dev)), c["x_cond"].to(dev), c["cm"].to(dev), c["aatype"].to(dev))
    torch.cuda.synchronize()
    assert torch.allclose(loss.cpu(), ref["loss"].detach(), rtol=1e-5)
    assert rel_l2(pred.cpu(), ref["pred"].detach()) < 1e-5
    got = tm.params.state_dict(tm.grads)
    worst = []
    for k in trainable_shapes(cfg):
        g_ref = P[k].grad
        assert g_ref is not None, k
        e = rel_l2(got[k].cpu(), g_ref) if float(g_ref.norm()) > 0 else float(got[k].abs().max())
        worst.append((e, k))
    worst.sort(reverse=True)
    print("worst gradient rel-L2:", [(f"{e:.1e}", k) for e, k in worst[:6]])
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                           f"train_grad_errors_L{L}.txt"), "w") as f:
        for e, k in worst:
            f.write(f"{e:.3e} {k} |g_ref| {float(P[k].grad.norm()):.3e}\n")
    bad = [(e, k) for e, k in worst if not e < 2e-4]
    assert not bad, bad[:10]
    # a second call ADDS (gradient accumulation) and is bit-reproducible -- on a tape and a workspace filled with 0xFF
    # bytes (NaN patterns): whatever the step reads from them it has written itself
    g1 = tm.grads.clone()
    tm._tape.fill_(0xFF)
    for ws in tm.model._ws.values():
        ws.view(torch.uint8).fill_(0xFF)
    tm.forward_backward(xt.to(dev), c["t"].to(dev), ut.to(dev), c["loss_mask"].to(dev), c["mask"].to(dev),
                        (c["sR"].to(dev), c["st"].to(dev)), c["x_cond"].to(dev), c["cm"].to(dev), c["aatype"].to(dev))
    assert torch.isfinite(tm.grads).all()
    assert torch.allclose(tm.grads, 2 * g1, rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("shape", [(2, 6, 5, 2), (1, 24, 33, 1)])
def test_training_step_gradients_tps_vs_autograd(shape):
    """The two-sided (TPS) model's training step: D = 28 latents, the IPA stack run twice on shared weights (x_r stream on
    the start frames, x_f stream on the end frames, latent_model.py:193-205) -- two tapes, two backward passes into the
    same gradients, plus latent_to_emb_f / _r.  Against torch autograd through the oracle run with the kernel's
    quaternion convention (w >= 0; the oracle itself is pinned to the reference's TPS gradients with the reference's
    own sign in tests/test_oracle_cpu.py): every one of the 128 trainable tensors to 2e-4."""
    from oracle import mdgen_oracle as O
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    from mdgen_amd.train import TrainableModel, trainable_shapes
    dev = _cuda()
    B, T, L, nl = shape
    cfg = ModelConfig(crop=L, num_frames=T, num_layers=nl, abs_pos_emb=True, sim_condition=False, tps_condition=True)
    assert cfg.latent_dim == 28
    sd = synth_state_dict(cfg, 19)
    c = _train_case(cfg, B, T, L, 2000 + T)
    gen = torch.Generator().manual_seed(77)
    q = torch.randn(B, L, 4, generator=gen)
    eR = O.quat_to_rot(q / q.norm(dim=-1, keepdim=True))
    et = c["st"] + 0.7 * torch.randn(B, L, 3, generator=gen)
    cm = c["cm"].clone()
    cm[:, -1] = 1
    x_cond = torch.where(cm.unsqueeze(-1).bool(), c["x1"], torch.zeros(()))
    names = list(trainable_shapes(cfg))
    assert len(names) == 128 if nl == 2 else True
    P = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    kw = dict(mask=c["mask"], start_frames=(c["sR"], c["st"]), end_frames=(eR, et), x_cond=x_cond, x_cond_mask=cm,
              aatype=c["aatype"])
    with torch.enable_grad():
        ref = O.training_losses(P, dict(O.cfg_dict(cfg), quat_sign="w_nonneg"), c["x1"], c["loss_mask"], kw, c["t"], c["x0"])
        ref["loss"].mean().backward()
    xt, ut = O.path_plan(c["t"], c["x0"], c["x1"], "GVP")
    tm = TrainableModel(cfg, dev).load_state_dict(sd)
    tm.zero_grad()
    loss, pred = tm.forward_backward(xt.to(dev), c["t"].to(dev), ut.to(dev), c["loss_mask"].to(dev), c["mask"].to(dev),
                                     (c["sR"].to(dev), c["st"].to(dev)), x_cond.to(dev), cm.to(dev), c["aatype"].to(dev),
                                     end_frames=(eR.to(dev), et.to(dev)))
    torch.cuda.synchronize()
    assert torch.allclose(loss.cpu(), ref["loss"].detach(), rtol=1e-5)
    assert rel_l2(pred.cpu(), ref["pred"].detach()) < 1e-5
    got = tm.params.state_dict(tm.grads)
    worst = []
    for k in names:
        g_ref = P[k].grad
        assert g_ref is not None, k
        e = rel_l2(got[k].cpu(), g_ref) if float(g_ref.norm()) > 0 else float(got[k].abs().max())
        worst.append((e, k))
    worst.sort(reverse=True)
    print("TPS worst gradient rel-L2:", [(f"{e:.1e}", k) for e, k in worst[:6]])
    bad = [(e, k) for e, k in worst if not e < 2e-4]
    assert not bad, bad[:10]
    for k in ("latent_to_emb_f.weight", "latent_to_emb_r.weight", "latent_to_emb_f.bias", "latent_to_emb_r.bias"):
        assert float(got[k].abs().max()) > 0


def test_training_gradients_vs_reference_fixture():
    """The same gradients against the REFERENCE's own backward pass directly (tests/golden/train_grads_sim.npz: norms and
    strided samples of every parameter's gradient from `loss.mean().backward()` in the reference)."""
    from oracle import mdgen_oracle as O
    from mdgen_amd.train import TrainableModel
    dev = _cuda()
    g = load_golden("train_grads_sim")
    cfg, sd = weights_for(g)
    xt, ut = O.path_plan(g["t"], g["x0"], g["x1"], "GVP")
    tm = TrainableModel(cfg, dev).load_state_dict(sd)
    tm.zero_grad()
    loss, _ = tm.forward_backward(xt.to(dev), g["t"].to(dev), ut.to(dev), g["loss_mask"].to(dev), g["mask"].to(dev),
                                  (g["start_rot"].to(dev), g["start_trans"].to(dev)), g["x_cond"].to(dev),
                                  g["x_cond_mask"].to(dev), g["aatype"].to(dev))
    assert torch.allclose(loss.cpu(), g["loss"], rtol=2e-5)
    got = tm.params.state_dict(tm.grads)
    names = [str(n) for n in g["grad_names"]]
    assert sorted(names) == sorted(got.keys())
    worst = 0.0
    for k in names:
        gr = got[k].reshape(-1).cpu()
        stride = int(g["gstride_" + k])
        e = rel_l2(gr[::stride][:2048], g["gsamp_" + k])
        worst = max(worst, e)
        assert e < 2e-4, (k, e)
        assert abs(float(gr.double().norm()) - float(g["gnorm_" + k])) <= 2e-4 * float(g["gnorm_" + k]) + 1e-9, k
    print(f"gradients vs reference autograd: worst rel-L2 {worst:.2e} over {len(names)} tensors")


def test_trainer_steps_vs_torch_adam():
    """Three full training steps (`Trainer.training_step`: prep_batch -> GVP plan -> forward/backward -> gradient clipping
    -> Adam -> EMA -> weights handed back to the library) against the same steps done with the oracle's autograd and
    torch.optim.Adam + clip_grad_norm_ on the CPU: losses to 1e-4, parameters after three steps to 1 % of their update
    over the entries whose gradient is above the noise floor (Adam divides by sqrt(v): an entry whose exact gradient is
    zero -- e.g. the key bias of an attention, to which softmax is invariant -- has a rounding-noise gradient in BOTH
    implementations and gets a full-size update of arbitrary sign; those entries are excluded, and are < 1e-3 of the
    largest gradient of their tensor in all three steps)."""
    from oracle import mdgen_oracle as O
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    from mdgen_amd.train import Trainer, trainable_shapes
    from mdgen_amd.wrapper import NewMDGenWrapper
    dev = _cuda()
    B, T, L = 2, 6, 5
    cfg = ModelConfig(crop=L, num_frames=T, num_layers=1, abs_pos_emb=True, sim_condition=True)
    sd = synth_state_dict(cfg, 23)
    w = NewMDGenWrapper(cfg)
    w.load_model_state_dict(sd)
    tr = Trainer(w, lr=1e-3, adamw=False, grad_clip=1.0, ema_decay=0.9)
    # CPU twin
    names = list(trainable_shapes(cfg))
    P = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    opt = torch.optim.Adam([P[k] for k in names], lr=1e-3)
    gen = torch.Generator().manual_seed(3)
    g0 = load_golden("prep_sim")                         # a self-consistent conditioning batch (B 2, T 6, L 5)
    batch = {k[3:]: v for k, v in g0.items() if k.startswith("in_")}
    cd = O.cfg_dict(cfg)
    signif = {}
    for step in range(3):
        t = torch.rand(B, generator=gen)
        x0 = torch.randn(B, T, L, cfg.latent_dim, generator=gen)
        loss = tr.training_step({k: v.to(dev) for k, v in batch.items()}, t=t.to(dev), x0=x0.to(dev))
        prep = O.prep_batch(batch, cd)
        with torch.enable_grad():
            ref = O.training_losses(P, cd, prep["latents"], prep["loss_mask"], prep["model_kwargs"], t, x0)
            opt.zero_grad()
            ref["loss"].mean().backward()
        for k in names:
            gk = P[k].grad.abs()
            m = gk > 1e-3 * gk.max()
            signif[k] = m if k not in signif else (signif[k] & m)
        torch.nn.utils.clip_grad_norm_([P[k] for k in names], 1.0)
        opt.step()
        print(f"step {step}: loss {float(loss):.6f} (reference {float(ref['loss'].mean()):.6f})")
        assert abs(float(loss) - float(ref["loss"].mean())) < 1e-4 * abs(float(ref["loss"].mean()))
    got = tr.tm.params.state_dict()
    worst = 0.0
    for k in names:
        m = signif[k]
        upd = ((P[k].detach() - sd[k]) * m).norm()
        err = ((got[k].cpu() - P[k].detach()) * m).norm()
        worst = max(worst, float(err) / (float(upd) + 1e-12))
        assert float(err) <= 1e-2 * float(upd) + 1e-7, (k, float(err), float(upd))
    print(f"parameters after 3 steps: worst |difference| / |update| {worst:.2e}")
    # the sampler now runs on the updated weights (bf16 path) and the EMA tracks them
    out = w.model.forward(x=torch.zeros(B, T, L, cfg.latent_dim, device=dev), t=torch.zeros(B, device=dev),
                          **{k: (v.to(dev) if torch.is_tensor(v) else (v[0].to(dev), v[1].to(dev)))
                             for k, v in dict(mask=prep["model_kwargs"]["mask"].contiguous(),
                                              start_frames=prep["model_kwargs"]["start_frames"],
                                              x_cond=prep["model_kwargs"]["x_cond"], x_cond_mask=prep["model_kwargs"]["x_cond_mask"],
                                              aatype=prep["model_kwargs"]["aatype"]).items()})
    assert torch.isfinite(out).all()
    assert tr.ema is not None and not torch.equal(tr.ema.data, tr.tm.params.data)


def test_trainer_step_tps():
    """One `Trainer.training_step` of the two-sided (TPS) model on the reference-generated TPS batch (prep_tps.npz:
    prep_batch -> plan -> forward / backward over both IPA streams -> clip -> AdamW -> weights handed back): loss equals
    the oracle's (kernel quaternion convention) to 1e-4, every tensor moved, sampler still finite on the new weights."""
    from oracle import mdgen_oracle as O
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    from mdgen_amd.train import Trainer, trainable_shapes
    from mdgen_amd.wrapper import NewMDGenWrapper
    dev = _cuda()
    g0 = load_golden("prep_tps")
    batch = {k[3:]: v for k, v in g0.items() if k.startswith("in_")}
    B, T, L = batch["torsions"].shape[:3]
    cfg = ModelConfig(crop=L, num_frames=T, num_layers=1, abs_pos_emb=True, sim_condition=False, tps_condition=True)
    sd = synth_state_dict(cfg, 29)
    w = NewMDGenWrapper(cfg)
    w.load_model_state_dict(sd)
    tr = Trainer(w, lr=1e-3, adamw=True, grad_clip=1.0, ema_decay=0.99)
    gen = torch.Generator().manual_seed(4)
    t = torch.rand(B, generator=gen)
    x0 = torch.randn(B, T, L, cfg.latent_dim, generator=gen)
    before = tr.tm.params.data.clone()
    loss = tr.training_step({k: v.to(dev) for k, v in batch.items()}, t=t.to(dev), x0=x0.to(dev))
    cd = dict(O.cfg_dict(cfg), quat_sign="w_nonneg")
    prep = O.prep_batch(batch, cd)
    ref = O.training_losses(sd, cd, prep["latents"], prep["loss_mask"], prep["model_kwargs"], t, x0)
    print(f"TPS training step: loss {float(loss):.6f} (oracle {float(ref['loss'].mean()):.6f})")
    assert abs(float(loss) - float(ref["loss"].mean())) < 1e-4 * abs(float(ref["loss"].mean()))
    after = tr.tm.params.state_dict()
    b4 = tr.tm.params.state_dict(before)
    still = [k for k in trainable_shapes(cfg) if torch.equal(after[k], b4[k])]
    assert not still, still
    out, _ = w.inference({k: v.to(dev) for k, v in batch.items()}, num_steps=2)
    assert torch.isfinite(out).all()


def test_gradient_buckets_are_final_at_their_milestones():
    """DDP overlap: the library records an event per parameter group as soon as that group's gradients are final, and
    `Trainer` enqueues every bucket's all-reduce on a communication stream behind the event of the last group the
    bucket contains.  Here (one GPU) the all-reduce is replaced by a snapshot of the bucket taken at that point of the
    communication stream: every snapshot must equal the bucket's content after the whole backward pass bit for bit --
    nothing the backward pass does after a milestone touches a gradient released by it -- for the one-sided and the
    two-sided model (whose IPA weights collect gradients from two passes)."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    from mdgen_amd.train import Trainer
    from mdgen_amd.wrapper import NewMDGenWrapper
    dev = _cuda()
    for name, tps in (("prep_sim", False), ("prep_tps", True)):
        g0 = load_golden(name)
        batch = {k[3:]: v.to(dev) for k, v in g0.items() if k.startswith("in_")}
        B, T, L = batch["torsions"].shape[:3]
        cfg = ModelConfig(crop=L, num_frames=T, num_layers=3, abs_pos_emb=True, sim_condition=not tps, tps_condition=tps)
        w = NewMDGenWrapper(cfg)
        w.load_model_state_dict(synth_state_dict(cfg, 31))
        tr = Trainer(w, lr=1e-3, grad_clip=1.0)
        from mdgen_amd.optim import GradBucketer
        tr.buckets = GradBucketer(tr.tm.params, tr.tm.grads, dist=None, bucket_bytes=4 << 20)   # ~20 buckets
        snaps = {}
        tr.on_bucket = lambda i, view: snaps.__setitem__(i, view.clone())
        gen = torch.Generator().manual_seed(8)
        tr.training_step(batch, t=torch.rand(B, generator=gen).to(dev), x0=torch.randn(B, T, L, cfg.latent_dim, generator=gen).to(dev))
        torch.cuda.synchronize()
        assert len(snaps) == len(tr.buckets.buckets) >= 10
        ms = [max(tr.milestone_of(n) for n in b["names"]) for b in tr.buckets.buckets]
        assert ms == sorted(ms), "buckets in backward order become ready in backward order"
        assert tr.buckets.launch_order == sorted(range(len(ms)), key=lambda i: ms[i])
        for i, b in enumerate(tr.buckets.buckets):
            assert torch.equal(snaps[i], tr.tm.grads[b["lo"]:b["hi"]]), (name, i, b["names"][:2])
        assert float(tr.tm.grads.abs().max()) > 0


def test_training_step_cfg5_size():
    """BASELINE.json configs[4] at its per-GPU size: ATLAS crop 256 x 250 frames, batch 1 per GPU, the full 5-layer
    model (34.15 M parameters): one forward + backward.  Size-independent checks: loss equals the fp32 forward's loss,
    gradients finite and bit-reproducible, gradient of a frozen-out tensor untouched, and the Adam step changes every
    tensor.  Timing is printed (fp32 unfused kernels: a correct step, not a fast one)."""
    import time
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.optim import Adam
    from mdgen_amd.synthetic import synth_state_dict, synth_forward_inputs
    from mdgen_amd.train import TrainableModel
    dev = _cuda()
    B, T, L = 1, 250, 256
    cfg = ModelConfig.atlas(num_frames=T, crop=L)
    sd = synth_state_dict(cfg, 6)
    inp = synth_forward_inputs(cfg, B, T, L, 16, 27)
    gen = torch.Generator().manual_seed(5)
    ut = torch.randn(B, T, L, cfg.latent_dim, generator=gen)
    lm = (torch.rand(B, T, L, cfg.latent_dim, generator=gen) > 0.1).float() * inp["mask"][..., None]
    tm = TrainableModel(cfg, dev).load_state_dict(sd)
    args = (inp["x"].to(dev), inp["t"].to(dev), ut.to(dev), lm.to(dev), inp["mask"].to(dev),
            (inp["start_rot"].to(dev), inp["start_trans"].to(dev)), inp["x_cond"].to(dev), inp["x_cond_mask"].to(dev),
            inp["aatype"].to(dev))
    tm.zero_grad()
    loss, pred = tm.forward_backward(*args)
    torch.cuda.synchronize()
    g1 = tm.grads.clone()
    tm.zero_grad()
    t0 = time.time()
    loss2, _ = tm.forward_backward(*args)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"cfg-5 size (B1 T250 L256, 34.15 M parameters): forward + backward {dt * 1e3:.0f} ms; loss {float(loss):.5f}; "
          f"|grad| {float(tm.grads.norm()):.4f}")
    assert torch.isfinite(loss).all() and torch.isfinite(tm.grads).all()
    assert torch.equal(tm.grads, g1) and torch.equal(loss, loss2)
    tm.model.set_precision("fp32")
    out = tm.model.forward(x=args[0], t=args[1], mask=args[4], start_frames=args[5], x_cond=args[6], x_cond_mask=args[7],
                           aatype=args[8])
    tm.model.set_precision("bf16")
    e_fwd = rel_l2(out.cpu(), pred.cpu())
    print(f"training forward vs the fp32 sampler forward: rel-L2 {e_fwd:.2e}")
    # same fp32 kernels, except that the training step slices the IPA attention's key loop over workgroups (B = 1: four
    # workgroups otherwise) and merges the slices' softmax states: a different, equally valid summation order
    assert e_fwd < 5e-6
    nz = [k for k, v in tm.params.state_dict(tm.grads).items() if float(v.abs().max()) == 0.0]
    assert not nz, nz
    before = tm.params.data.clone()
    Adam(tm.params, lr=1e-4, grad_clip=1.0).step(tm.grads)
    assert torch.isfinite(tm.params.data).all() and not torch.equal(before, tm.params.data)
    del tm
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_ddp_two_processes_match_one(tmp_path):
    """Data-parallel training through the code path `Trainer` really uses with world > 1 (`GradBucketer.launch_on_events`:
    bucketed all-reduce on a communication stream behind the library's gradient milestones, averaging folded into Adam's
    grad_scale, construction-time broadcast of rank 0's parameters): two processes with one item each (tests/ddp_worker.py,
    gloo on CUDA tensors, both on this GPU; rank 1 is started from DIFFERENT weights) must end two steps with the parameters
    and EMA of one process that trained on the two-item batch.  Reference behaviour: Lightning DDP under train.py:46-77."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    worker = os.path.join(ROOT, "tests", "ddp_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = str(tmp_path / "one.pt")
    subprocess.run([sys.executable, worker, one], check=True, timeout=600, env=dict(env, RANK="0", WORLD_SIZE="1"))
    two = str(tmp_path / "two.pt")
    procs = [subprocess.Popen([sys.executable, worker, two], env=dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0"))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    a, b = torch.load(one), torch.load(two)
    assert a["world"] == 1 and b["world"] == 2
    upd = None
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    from mdgen_amd.train import flat_order
    cfg = ModelConfig(crop=5, num_frames=6, num_layers=1, abs_pos_emb=True, sim_condition=True)
    sd = synth_state_dict(cfg, 23)
    start = torch.cat([sd[k].reshape(-1).float() for k in flat_order(cfg)])
    upd = (a["params"] - start).norm()
    err = (a["params"] - b["params"]).norm()
    print(f"2 ranks x 1 item vs 1 rank x 2 items after two steps: |difference| / |update| {float(err / upd):.2e}; "
          f"EMA rel-L2 {rel_l2(b['ema'], a['ema']):.2e}")
    # the loss of a step is the mean over items either way; the two-rank gradient is the average of two fp32 sums (another
    # summation order) and Adam normalises by sqrt(v): entries at the gradient noise floor move by a full-size step of either
    # sign, so the bound is on the norm, not per entry
    assert float(err) <= 2e-2 * float(upd)
    assert rel_l2(b["ema"], a["ema"]) < 1e-4


def test_ddp_over_rccl_two_gpus(tmp_path):
    """The same two-process data-parallel step over **RCCL** (backend "nccl"), one rank per GPU -- the production path of
    `python -m mdgen_amd.train` under torch.distributed.run (reference: Lightning DDP = NCCL, train.py:46-77; BASELINE.json
    configs[4]).  Needs >= 2 GPUs: skipped on the one-GPU boxes of this pool, runs wherever the driver gets a multi-GPU node.
    Asserts (a) two ranks x one item == one rank x two items (as the gloo test), (b) at configs[4]'s per-GPU shape (B 1, T 250,
    L 256) the all-reduce wait left exposed after the backward pass was enqueued is < 10 % of the step."""
    import socket
    import subprocess
    _cuda()
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    worker = os.path.join(ROOT, "tests", "ddp_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = str(tmp_path / "one.pt")
    subprocess.run([sys.executable, worker, one], check=True, timeout=900, env=dict(env, RANK="0", WORLD_SIZE="1"))
    two = str(tmp_path / "two.pt")
    procs = [subprocess.Popen([sys.executable, worker, two],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", DDP_BACKEND="nccl", DDP_TIMING="1"))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    a, b = torch.load(one), torch.load(two)
    assert a["world"] == 1 and b["world"] == 2 and b["backend"] == "nccl"
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    from mdgen_amd.train import flat_order
    cfg = ModelConfig(crop=5, num_frames=6, num_layers=1, abs_pos_emb=True, sim_condition=True)
    sd = synth_state_dict(cfg, 23)
    start = torch.cat([sd[k].reshape(-1).float() for k in flat_order(cfg)])
    upd, err = (a["params"] - start).norm(), (a["params"] - b["params"]).norm()
    print(f"RCCL, 2 ranks x 1 item vs 1 rank x 2 items: |difference| / |update| {float(err / upd):.2e}; "
          f"cfg-5 per-GPU step {b['step_ms']:.1f} ms, exposed all-reduce wait {b['exposed_comm_ms']:.2f} ms")
    assert float(err) <= 2e-2 * float(upd)
    assert rel_l2(b["ema"], a["ema"]) < 1e-4
    assert b["exposed_comm_ms"] < 0.10 * b["step_ms"], (b["exposed_comm_ms"], b["step_ms"])


@pytest.mark.parametrize("shape", [(2, 40, 72, False), (1, 250, 256, False), (2, 24, 8, True)])
def test_training_weight_gradients_on_second_stream_are_identical(shape):
    """Option `train_streams` (default 2): the weight / bias gradients of the linear layers, the bias-key / value sums, the adaLN
    heads and the token embedders' gradients run on a second stream beside the backward pass's critical path (dX products,
    attention backward, element-wise passes), reading alternating instances of the scratch the main stream would overwrite
    (csrc/train.inc `Train::begin_sub / fork`).  Same kernels, same summation orders: gradients and loss must be BIT-identical
    to the one-stream run, in both operand modes, after repeated steps on the same buffers (a missed hazard shows up as a
    difference here).  Shapes: a mid-size ATLAS-like case, cfg-5's per-GPU size, and the two-sided model (two IPA passes adding
    into the same gradients)."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict, synth_forward_inputs
    from mdgen_amd.train import TrainableModel
    dev = _cuda()
    B, T, L, tps = shape
    cfg = (ModelConfig(crop=L, num_frames=T, num_layers=2, tps_condition=True, abs_pos_emb=True) if tps
           else ModelConfig.atlas(num_frames=T, crop=L))
    sd = synth_state_dict(cfg, 6)
    inp = synth_forward_inputs(cfg, B, T, L, 3, 27)
    gen = torch.Generator().manual_seed(5)
    ut = torch.randn(B, T, L, cfg.latent_dim, generator=gen)
    lm = (torch.rand(B, T, L, cfg.latent_dim, generator=gen) > 0.1).float() * inp["mask"][..., None]
    args = (inp["x"].to(dev), inp["t"].to(dev), ut.to(dev), lm.to(dev), inp["mask"].to(dev),
            (inp["start_rot"].to(dev), inp["start_trans"].to(dev)), inp["x_cond"].to(dev), inp["x_cond_mask"].to(dev),
            inp["aatype"].to(dev))
    kw = {"end_frames": (inp["end_rot"].to(dev), inp["end_trans"].to(dev))} if tps else {}
    for prec in (16, 32):
        res = {}
        for ns in (1, 2):
            tm = TrainableModel(cfg, dev).load_state_dict(sd)
            tm.model.set_option("train_precision", prec)
            tm.model.set_option("train_streams", ns)
            for _ in range(3):
                tm.zero_grad()
                loss, _ = tm.forward_backward(*args, **kw)
            torch.cuda.synchronize()
            res[ns] = (loss.clone(), tm.grads.clone())
            del tm
        assert torch.isfinite(res[2][1]).all()
        assert torch.equal(res[1][0], res[2][0]), (shape, prec)
        assert torch.equal(res[1][1], res[2][1]), (shape, prec, float((res[1][1] - res[2][1]).abs().max()))


def test_trainer_checkpoint_resume_round_trip(tmp_path):
    """`Trainer.save_checkpoint` -> `load_checkpoint` (Lightning layout; the optimiser state in torch.optim.Adam's own
    state_dict layout, i.e. what a checkpoint of the reference holds under `optimizer_states[0]`): a run resumed from the
    checkpoint after two steps takes a third step identical, bit for bit, to the uninterrupted run's; torch.optim.Adam
    itself accepts the stored optimiser state."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    from mdgen_amd.train import Trainer, trainable_shapes
    from mdgen_amd.wrapper import NewMDGenWrapper
    dev = _cuda()
    B, T, L = 2, 6, 5
    cfg = ModelConfig(crop=L, num_frames=T, num_layers=1, abs_pos_emb=True, sim_condition=True)
    g0 = load_golden("prep_sim")
    batch = {k[3:]: v.to(dev) for k, v in g0.items() if k.startswith("in_")}
    gen = torch.Generator().manual_seed(5)
    draws = [(torch.rand(B, generator=gen).to(dev), torch.randn(B, T, L, cfg.latent_dim, generator=gen).to(dev)) for _ in range(3)]
    w = NewMDGenWrapper(cfg)
    w.load_model_state_dict(synth_state_dict(cfg, 23))
    tr = Trainer(w, lr=1e-3, grad_clip=1.0, ema_decay=0.9)
    for t, x0 in draws[:2]:
        tr.training_step(batch, t=t, x0=x0)
    path = str(tmp_path / "resume.ckpt")
    tr.save_checkpoint(path)
    tr.training_step(batch, t=draws[2][0], x0=draws[2][1])
    torch.cuda.synchronize()
    want = (tr.tm.params.data.clone(), tr.opt.exp_avg.clone(), tr.opt.exp_avg_sq.clone(), tr.ema.data.clone(), tr.opt.step_count)
    tr.close()
    ck = torch.load(path, map_location="cpu", weights_only=False)
    osd = ck["optimizer_states"][0]
    assert set(osd) == {"state", "param_groups"} and len(osd["state"]) == len(trainable_shapes(cfg))
    probe = torch.optim.Adam([torch.nn.Parameter(torch.zeros(*s)) for s in trainable_shapes(cfg).values()], lr=1.0)
    probe.load_state_dict(osd)                                   # torch's own loader takes it
    assert float(probe.state_dict()["state"][0]["step"]) == 2.0
    w2 = NewMDGenWrapper.load_from_checkpoint(path)              # another process would start here (sim_inference.py:129-130)
    tr2 = Trainer(w2, lr=1e-3, grad_clip=1.0, ema_decay=0.9).load_checkpoint(path)
    assert tr2.opt.step_count == 2 and tr2.global_step == 2
    tr2.training_step(batch, t=draws[2][0], x0=draws[2][1])
    torch.cuda.synchronize()
    got = (tr2.tm.params.data, tr2.opt.exp_avg, tr2.opt.exp_avg_sq, tr2.ema.data, tr2.opt.step_count)
    for a_, b_ in zip(want[:4], got[:4]):
        assert torch.equal(a_, b_)
    assert got[4] == want[4] == 3
    tr2.close()


def test_integration_md_level2_stub_runs():
    """INTEGRATION.md "Level 2": the ctypes stub a maintainer of the reference would add is EXECUTED as written (the python
    block is cut out of the document; only the library path is made absolute) against a duck-typed reference wrapper
    (`.args`, `.latent_dim`, `.model.state_dict()`), for a forward-simulation and a two-sided model; its Euler rollout must
    equal `LatentMDGenModel.sample_euler` of this package bit for bit (same library, same kernels)."""
    import re
    import types
    import mdgen_amd._lib as L
    from mdgen_amd.model import LatentMDGenModel
    from mdgen_amd.rigid_utils import Rigid, Rotation
    from mdgen_amd.wrapper import default_args
    dev = _cuda()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(# mdgen/amd_backend.py.*?)```", doc, re.S).group(1)
    assert 'C.CDLL("libmdgen_amd.so")' in code
    ns = {}
    exec(compile(code.replace('C.CDLL("libmdgen_amd.so")', f'C.CDLL({L.LIB_PATH!r})'), "INTEGRATION.md:level2", "exec"), ns)
    for name in ("fwd_full_pep", "fwd_full_tps"):
        g = load_golden(name)
        cfg, sd = weights_for(g)
        duck = types.SimpleNamespace(args=default_args(cfg), latent_dim=cfg.latent_dim,
                                     model=types.SimpleNamespace(state_dict=lambda sd=sd: {k: v.to(dev) for k, v in sd.items()}))
        ns["attach"](duck)
        kw = _kw(g, dev)
        sf = Rigid(Rotation(rot_mats=kw["start_frames"][0]), kw["start_frames"][1])
        ef = Rigid(Rotation(rot_mats=kw["end_frames"][0]), kw["end_frames"][1])
        S = 3
        got = duck.amd_sample(kw["x"], S, kw["mask"], sf, ef, kw["x_cond"], kw["x_cond_mask"], kw["aatype"])
        torch.cuda.synchronize()
        m = LatentMDGenModel(cfg)
        m.load_state_dict(sd)
        skw = {k: v for k, v in kw.items() if k not in ("x", "t")}
        if not cfg.tps_condition:
            skw.pop("end_frames")
        want = m.sample_euler(kw["x"], S, **skw, use_graph=False)
        assert torch.isfinite(got).all()
        assert torch.equal(got, want), (name, rel_l2(got.cpu(), want.cpu()))
        del m


def test_ten_chained_blocks_error_growth():
    """The README run chains 10 blocks (sim_inference.py:110-113, README.md:72 `--num_rollouts 10`).  rollout10_sim holds the
    reference's own ten chained blocks (S = 10, B 1, T 8, L 4, full-width model); `NewMDGenWrapper.rollout` chains ten blocks
    in one graph on ITS OWN end frames, in both operand precisions.  Per-block rms / max deviation is reported.  What can be
    gated: tests/test_oracle_cpu.py::test_ten_chained_blocks_vs_reference shows that on these (random-weight, unphysical)
    structures the rollout glue amplifies even fp32 summation-order noise to ~0.1 A max by block 2 and ~0.4 A by block 9 --
    the reference does not track ITSELF more closely than that -- so the per-block bounds below are the single-block bf16
    bound for block 0, an fp32-tight bound for the fp32 mode's block 0, and an rms bound afterwards."""
    from mdgen_amd.wrapper import NewMDGenWrapper
    dev = _cuda()
    g = load_golden("rollout10_sim")
    cfg, sd = weights_for(g)
    S, R = 10, 10
    T = g["S10_b0_zs"].shape[1]
    zs = torch.stack([g[f"S{S}_b{r}_zs"] for r in range(R)]).to(dev)
    batch0 = {k[3:]: v.to(dev) for k, v in g.items() if k.startswith("in_")}
    for prec in ("bf16", "fp32"):
        w = NewMDGenWrapper(cfg, precision=prec)
        w.model.load_state_dict(sd)
        a = w.rollout(batch0, T, R, num_steps=S, zs=zs)
        torch.cuda.synchronize()
        assert torch.isfinite(a).all()
        left = None
        for r in range(R):
            d = (a[:, r * T:(r + 1) * T].cpu() - g[f"S{S}_b{r}_atom14"]).abs()
            rms, mx = float(d.pow(2).mean().sqrt()), float(d.max())
            if left is None and mx > 0.5:
                left = r
            print(f"{prec} operands, block {r}: atom14 rms {rms:.4f} A max {mx:.4f} A")
            if prec == "fp32":   # block 1 already starts from OUR block-0 end frame: the glue's amplification applies from there on
                assert (mx < 1e-3 if r < 1 else rms < 0.1), (prec, r, rms, mx)
            else:
                assert (rms < TOL_RMS and mx < TOL_MAX) if r == 0 else rms < 0.25, (prec, r, rms, mx)
        print(f"{prec} operands: first block whose max deviation exceeds 0.5 A: {left}")
        del w


def test_time_embedding_and_adaln_table_all_steps():
    """Row a-1 in isolation: `TimestepEmbedder` (layers.py:17-55) + every adaLN_modulation Linear (latent_model.py:349-352,
    408-411, layers.py:66-69) for ALL S Euler steps -- the step-invariant table `k_temb` / `k_adaln` build once per call --
    read back from the workspace after an S = 49 rollout and compared with the oracle row by row (fp32 kernels: 1e-5)."""
    import ctypes as C
    import torch.nn.functional as F
    from oracle import mdgen_oracle as O
    from mdgen_amd import _lib as L
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict, synth_forward_inputs
    from mdgen_amd.model import LatentMDGenModel
    dev = _cuda()
    B, T, L_, S = 2, 6, 5, 49
    cfg = ModelConfig.forward_sim(num_frames=T, crop=L_)
    sd = synth_state_dict(cfg, 5)
    m = LatentMDGenModel(cfg)
    m.load_state_dict(sd)
    inp = synth_forward_inputs(cfg, B, T, L_, 1, 77)
    x = m.sample_euler(inp["x"].to(dev), S, mask=inp["mask"].to(dev), start_frames=(inp["start_rot"].to(dev), inp["start_trans"].to(dev)),
                       x_cond=inp["x_cond"].to(dev), x_cond_mask=inp["x_cond_mask"].to(dev), aatype=inp["aatype"].to(dev),
                       use_graph=False)
    torch.cuda.synchronize()
    assert torch.isfinite(x).all()
    lay = m.workspace_layout(B, T, L_, S, True)
    ws = m._ws[(B, T, L_, S, 1)]
    C_, nl = cfg.embed_dim, cfg.num_layers
    modrow = nl * 15 * C_ + 2 * C_
    tab = ws[lay.mod:lay.mod + S * modrow * 4].view(torch.float32).view(S, modrow).cpu()
    tg = torch.linspace(0, 1, S + 1)[:S]
    temb = O.t_embedder(sd, tg * cfg.time_multiplier)              # (S, C); latent_model.py:243
    act = F.silu(temb)
    worst = 0.0
    for i in range(nl):                                            # library row layout: trunk i at 9C i, IPA i at 9C nl + 6C i, final at 15C nl
        for name, off, n in ((f"layers.{i}.adaLN_modulation.1", i * 9 * C_, 9 * C_),
                             (f"ipa_layers.{i}.adaLN_modulation.1", nl * 9 * C_ + i * 6 * C_, 6 * C_)):
            ref = O.linear(sd, name, act)
            worst = max(worst, rel_l2(tab[:, off:off + n], ref))
    ref = O.linear(sd, "emb_to_latent.adaLN_modulation.1", act)
    worst = max(worst, rel_l2(tab[:, nl * 15 * C_:], ref))
    print(f"adaLN table, {S} steps x {modrow} floats: worst block rel-L2 vs the oracle {worst:.2e}")
    assert worst < 1e-5


def test_training_gradients_bf16_operands_vs_reference_fixture():
    """Option train_precision = 16: the training step's matrix products (linear layers, weight gradients, the attention's
    q k^T / p v and their backward) multiply bf16-rounded operands on the bf16 MFMA with fp32 accumulation, fp32 master weights
    and activations -- the precision class the reference trains with (train.py:13 `set_float32_matmul_precision('medium')`);
    LayerNorm, softmax, reductions stay fp32 (GELU to 5e-6).  Gate against the
    reference's own fp32 autograd gradients (train_grads_sim.npz): loss to 1e-2 relative, every tensor's gradient samples to
    rel-L2 5e-2 and cosine 0.999 (tensors whose gradient is at the noise floor excepted), the exact mode (32) stays as tested
    above."""
    from oracle import mdgen_oracle as O
    from mdgen_amd.train import TrainableModel
    dev = _cuda()
    g = load_golden("train_grads_sim")
    cfg, sd = weights_for(g)
    xt, ut = O.path_plan(g["t"], g["x0"], g["x1"], "GVP")
    tm = TrainableModel(cfg, dev).load_state_dict(sd)
    tm.model.set_option("train_precision", 16)
    tm.zero_grad()
    loss, _ = tm.forward_backward(xt.to(dev), g["t"].to(dev), ut.to(dev), g["loss_mask"].to(dev), g["mask"].to(dev),
                                  (g["start_rot"].to(dev), g["start_trans"].to(dev)), g["x_cond"].to(dev),
                                  g["x_cond_mask"].to(dev), g["aatype"].to(dev))
    assert torch.allclose(loss.cpu(), g["loss"], rtol=1e-2), (loss.cpu(), g["loss"])
    got = tm.params.state_dict(tm.grads)
    names = [str(n) for n in g["grad_names"]]
    gmax = max(float(g["gnorm_" + k]) for k in names)
    rep = []
    for k in names:
        gr = got[k].reshape(-1).cpu()
        ref = g["gsamp_" + k]
        mine = gr[::int(g["gstride_" + k])][:2048]
        e = rel_l2(mine, ref)
        cos = float((mine.double() @ ref.double()) / (mine.double().norm() * ref.double().norm() + 1e-300))
        rep.append((e, cos, k, float(g["gnorm_" + k])))
    rep.sort(reverse=True)
    print("bf16-operand training step, worst gradients (rel-L2, cosine, tensor):", [(f"{e:.1e}", f"{c:.4f}", k) for e, c, k, _ in rep[:6]])
    for e, cos, k, nrm in rep:
        if nrm < 1e-4 * gmax:   # a gradient at the noise floor (e.g. an attention's key bias: softmax is invariant to it)
            continue
        assert e < 5e-2 and cos > 0.999, (k, e, cos)
    tm.model.set_option("train_precision", 32)


def test_training_turned_weights_ahead_of_time():
    """Option `train_turn_ahead` (default on): the turned weights of the small launches' dX products are computed on the second stream
    at the start of a call, from the request list the previous call recorded.  Call 1 records (everything in place), call 2 uses the
    images, a call at another shape deviates from the list (falls back in place, records anew), the call after that uses the new
    list: every call's loss and gradients are bit-identical to the same call with the option off -- also after the weights were
    changed in place between two calls (the images are recomputed every call)."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict, synth_forward_inputs
    from mdgen_amd.train import TrainableModel
    dev = _cuda()

    def make_args(cfg, B, T, L, npad, seed):
        inp = synth_forward_inputs(cfg, B, T, L, npad, seed)
        gen = torch.Generator().manual_seed(seed)
        ut = torch.randn(B, T, L, cfg.latent_dim, generator=gen)
        lm = (torch.rand(B, T, L, cfg.latent_dim, generator=gen) > 0.1).float() * inp["mask"][..., None]
        return (inp["x"].to(dev), inp["t"].to(dev), ut.to(dev), lm.to(dev), inp["mask"].to(dev),
                (inp["start_rot"].to(dev), inp["start_trans"].to(dev)), inp["x_cond"].to(dev), inp["x_cond_mask"].to(dev),
                inp["aatype"].to(dev))

    results = {}
    for ahead in (1, 0):
        out = []
        tms = {}
        for T, L in ((12, 40), (6, 24)):        # two models (num_frames is part of the config), sharing nothing
            cfg = ModelConfig.atlas(num_frames=T, crop=L)
            tm = TrainableModel(cfg, dev).load_state_dict(synth_state_dict(cfg, 11))
            tm.model.set_option("train_precision", 16)
            tm.model.set_option("train_turn_ahead", ahead)
            tms[(T, L)] = (tm, make_args(cfg, 1, T, L, 3, 5))
        tm, args = tms[(12, 40)]
        for call in range(4):
            if call == 2:                       # the weights change in place (as after an optimiser step)
                with torch.no_grad():
                    tm.params.data.mul_(1.001)
                tm.mark_updated()
            tm.zero_grad()
            loss, _ = tm.forward_backward(*args)
            torch.cuda.synchronize()
            out.append((float(loss), {k: v.detach().float().cpu().clone() for k, v in tm.params.state_dict(tm.grads).items()}))
        tm2, args2 = tms[(6, 24)]               # a second context: its own list
        for call in range(2):
            tm2.zero_grad()
            loss, _ = tm2.forward_backward(*args2)
            torch.cuda.synchronize()
            out.append((float(loss), {k: v.detach().float().cpu().clone() for k, v in tm2.params.state_dict(tm2.grads).items()}))
        for tm_, _ in tms.values():
            tm_.model.set_option("train_precision", 32)
        results[ahead] = out
    for i, ((l1, g1), (l0, g0)) in enumerate(zip(results[1], results[0])):
        assert l1 == l0, (i, l1, l0)
        for k in g0:
            assert torch.equal(g1[k], g0[k]), (i, k)
    # call 1 (recorded in place) against call 2 (images computed ahead): the same inputs, the same bits
    assert results[1][0][0] == results[1][1][0]
    for k in results[1][0][1]:
        assert torch.equal(results[1][0][1][k], results[1][1][1][k]), k


def test_training_round6_options_agree():
    """Options `train_defer_gate` (the trunk's gated residual updates formed by the next sub-layer's LayerNorm launch) and
    `train_attn_form` (sequence-resident attention kernels, RoPE inside, for axes of 129 .. 256 positions), each against its off
    value on one step of 160 frames x 130 residues with bf16 operands: deferring the update changes no arithmetic -- loss and every
    gradient bit-identical; `train_y_bf16` (the trunk's taped LayerNorm outputs stored as bf16 rows: the kernels that read them round
    them to bf16 anyway) likewise bit-identical; the attention forms round q at a different point (q log2 e instead of q) -- loss to 2e-3, every
    gradient above the noise floor to rel-L2 2e-2."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict, synth_forward_inputs
    from mdgen_amd.train import TrainableModel
    dev = _cuda()
    B, T, L, npad = 1, 160, 130, 9
    cfg = ModelConfig.atlas(num_frames=T, crop=L)
    sd = synth_state_dict(cfg, 11)
    inp = synth_forward_inputs(cfg, B, T, L, npad, 5)
    gen = torch.Generator().manual_seed(9)
    ut = torch.randn(B, T, L, cfg.latent_dim, generator=gen)
    lm = (torch.rand(B, T, L, cfg.latent_dim, generator=gen) > 0.1).float() * inp["mask"][..., None]
    args = (inp["x"].to(dev), inp["t"].to(dev), ut.to(dev), lm.to(dev), inp["mask"].to(dev),
            (inp["start_rot"].to(dev), inp["start_trans"].to(dev)), inp["x_cond"].to(dev), inp["x_cond_mask"].to(dev),
            inp["aatype"].to(dev))
    res = {}
    for name, opts in (("default", {}), ("gate_now", {"train_defer_gate": 0}), ("y_fp32", {"train_y_bf16": 0}),
                       ("dqkv_fp32", {"train_dqkv_bf16": 0}), ("du_fp32", {"train_du_bf16": 0}), ("dhid_fp32", {"train_dhid_bf16": 0}),
                       ("chunked", {"train_attn_form": 0})):
        tm = TrainableModel(cfg, dev).load_state_dict(sd)
        tm.model.set_option("train_precision", 16)
        for k, v in opts.items():
            tm.model.set_option(k, v)
        tm.zero_grad()
        loss, _ = tm.forward_backward(*args)
        torch.cuda.synchronize()
        res[name] = (float(loss), {k: v.detach().float().cpu().clone() for k, v in tm.params.state_dict(tm.grads).items()})
        tm.model.set_option("train_precision", 32)
    l0, g0 = res["default"]
    for other in ("gate_now", "y_fp32"):
        l1, g1 = res[other]
        assert l0 == l1, (other, l0, l1)
        for k in g0:
            assert torch.equal(g0[k], g1[k]), (other, k)
    # `train_dqkv_bf16` (the q | k | v gradients stored as bf16 rows): the dX product and the weight gradient round them to bf16 anyway
    # -- bit-identical -- but the q / k / v BIAS gradients are column sums of the stored values: rounded then, to 5e-3
    # `train_du_bf16` (the gated gradients du = gate * dh) likewise: only the out-projection / fc2 bias gradients may differ
    # ... and `train_dhid_bf16` (d pre = d hid * gelu'(pre) of the MLPs): only the fc1 bias gradients
    for other, biases in (("dqkv_fp32", ("q_proj.bias", "k_proj.bias", "v_proj.bias")), ("du_fp32", ("out_proj.bias", "fc2.bias")),
                          ("dhid_fp32", ("fc1.bias",))):
        l3, g3 = res[other]
        assert l0 == l3, (other, l0, l3)
        for k in g0:
            if torch.equal(g0[k], g3[k]):
                continue
            assert k.endswith(biases), (other, k)
            e = float((g0[k].double() - g3[k].double()).norm() / (g3[k].double().norm() + 1e-300))
            assert e < 5e-3, (other, k, e)
    l2, g2 = res["chunked"]
    assert abs(l0 - l2) <= 2e-3 * abs(l2), (l0, l2)
    gmax = max(float(v.norm()) for v in g2.values())
    for k, ref in g2.items():
        if float(ref.norm()) < 1e-4 * gmax:
            continue
        e = float((g0[k].double() - ref.double()).norm() / ref.double().norm())
        assert e < 2e-2, (k, e)


@pytest.mark.parametrize("shape", [(1, 24, 203, 37), (1, 300, 5, 1), (1, 160, 130, 9), (1, 250, 256, 16)],
                         ids=["T24_L203", "T300_L5", "T160_L130", "T250_L256_bench_shape"])
def test_training_bf16_operand_kernels_vs_exact_mode_at_tile_sizes(shape):
    """The golden-fixture gradient tests above run shapes of a few dozen tokens, which the launchers route to the general
    kernels.  This one is sized for the kernels the real workload runs -- the 128 x 384-tile linear layer and weight gradient
    (k_wide16.hip: >= 1024 / 4096 token rows, with row tails), the one-pass q|k|v forms, the MFMA attention forward and its
    two backward passes (k_attn16.hip) on both axes with ragged tiles (24 frames, 203 residues: partial 32-row tiles, the bias
    key inside a tile; 300 frames: three 128-query blocks, five 64-key chunks; 5 residues: one partial tile) and key padding,
    the bf16-stored GELU output (>= 4096 rows); (round 6) 203 residues and 160 frames x 130 residues: axes of 129 .. 256 positions
    take the sequence-resident attention kernels with RoPE inside (no rotation pass; `train_attn_form`), and every trunk sub-layer's gated
    update rides in the next LayerNorm launch (`train_defer_gate`); 250 frames x 256 residues = the shape bench.py's training leg
    times (len 256: the bias key opens a ninth key tile) -- and compares train_precision 16 against the exact fp32 mode (itself gated against
    the reference's autograd above) on identical inputs: loss to 1e-2 relative, every parameter's gradient to rel-L2 5e-2 and
    cosine 0.999 (gradients at the noise floor excepted)."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict, synth_forward_inputs
    from mdgen_amd.train import TrainableModel
    dev = _cuda()
    B, T, L, npad = shape
    cfg = ModelConfig.atlas(num_frames=T, crop=L)
    sd = synth_state_dict(cfg, 11)
    inp = synth_forward_inputs(cfg, B, T, L, npad, 5)        # the last `npad` residues are padding
    gen = torch.Generator().manual_seed(9)
    ut = torch.randn(B, T, L, cfg.latent_dim, generator=gen)
    lm = (torch.rand(B, T, L, cfg.latent_dim, generator=gen) > 0.1).float() * inp["mask"][..., None]
    args = (inp["x"].to(dev), inp["t"].to(dev), ut.to(dev), lm.to(dev), inp["mask"].to(dev),
            (inp["start_rot"].to(dev), inp["start_trans"].to(dev)), inp["x_cond"].to(dev), inp["x_cond_mask"].to(dev),
            inp["aatype"].to(dev))
    res = {}
    for prec in (32, 16):
        tm = TrainableModel(cfg, dev).load_state_dict(sd)
        tm.model.set_option("train_precision", prec)
        tm.zero_grad()
        loss, _ = tm.forward_backward(*args)
        torch.cuda.synchronize()
        res[prec] = (float(loss), {k: v.detach().float().cpu().clone() for k, v in tm.params.state_dict(tm.grads).items()})
        tm.model.set_option("train_precision", 32)
    l32, g32 = res[32]
    l16, g16 = res[16]
    assert abs(l16 - l32) <= 1e-2 * abs(l32), (l16, l32)
    gmax = max(float(v.norm()) for v in g32.values())
    rep = []
    for k, ref in g32.items():
        mine = g16[k].reshape(-1).double()
        r = ref.reshape(-1).double()
        assert torch.isfinite(mine).all(), k
        e = float((mine - r).norm() / (r.norm() + 1e-300))
        cos = float((mine @ r) / (mine.norm() * r.norm() + 1e-300))
        rep.append((e, cos, k, float(r.norm())))
    rep.sort(reverse=True)
    print(f"train_precision 16 vs 32 at B{B} T{T} L{L}: loss {l16:.5f} vs {l32:.5f}; worst gradients (rel-L2, cosine, tensor):",
          [(f"{e:.1e}", f"{c:.5f}", k) for e, c, k, _ in rep[:6]])
    for e, cos, k, nrm in rep:
        if nrm < 1e-4 * gmax:
            continue
        assert e < 5e-2 and cos > 0.999, (k, e, cos)


def test_mark_updated_refreshes_every_table_the_training_step_reads():
    """`TrainableModel.mark_updated()` hands over only the parameters that the training kernels read through context-owned
    tables (time embedder, adaLN table, embeddings, IPA norm / head weights); everything else is read in place from the bound
    flat buffer.  A parameter missing from that list would train on a stale copy without any error, so: perturb EVERY
    parameter in place, mark_updated(), and require loss and all gradients to be bit-identical to a fresh model loaded from
    the perturbed state dict (a full hand-over) -- for both model kinds and both operand precisions."""
    from oracle import mdgen_oracle as O
    from mdgen_amd.train import TrainableModel
    dev = _cuda()
    for name in ("train_grads_sim", "train_grads_tps"):
        g = load_golden(name)
        cfg, sd = weights_for(g)
        xt, ut = O.path_plan(g["t"], g["x0"], g["x1"], "GVP")
        ends = (g["end_rot"].to(dev), g["end_trans"].to(dev)) if "end_rot" in g else None
        args = (xt.to(dev), g["t"].to(dev), ut.to(dev), g["loss_mask"].to(dev), g["mask"].to(dev),
                (g["start_rot"].to(dev), g["start_trans"].to(dev)), g["x_cond"].to(dev), g["x_cond_mask"].to(dev), g["aatype"].to(dev))
        kw = {"end_frames": ends} if ends is not None else {}
        for prec in (32, 16):
            tm = TrainableModel(cfg, dev).load_state_dict(sd)
            tm.model.set_option("train_precision", prec)
            tm.zero_grad()
            tm.forward_backward(*args, **kw)              # tables in use once, as in a running job
            gen = torch.Generator().manual_seed(3)
            flat = tm.params.data
            flat += 0.02 * flat.abs().mean() * torch.randn(flat.numel(), generator=gen).to(flat)
            tm.mark_updated()
            tm.zero_grad()
            loss, _ = tm.forward_backward(*args, **kw)
            torch.cuda.synchronize()
            got = {k: v.detach().cpu().clone() for k, v in tm.params.state_dict(tm.grads).items()}
            sd2 = {k: v.detach().cpu().clone() for k, v in tm.params.state_dict().items()}
            for k, v in sd.items():   # frozen buffers etc.
                sd2.setdefault(k, v)
            ref = TrainableModel(cfg, dev).load_state_dict(sd2)
            ref.model.set_option("train_precision", prec)
            ref.zero_grad()
            loss_ref, _ = ref.forward_backward(*args, **kw)
            torch.cuda.synchronize()
            want = ref.params.state_dict(ref.grads)
            assert torch.equal(loss.cpu(), loss_ref.cpu()), (name, prec, loss, loss_ref)
            for k in got:
                assert torch.equal(got[k], want[k].cpu()), (name, prec, k)
            tm.model.set_option("train_precision", 32)
            ref.model.set_option("train_precision", 32)


def _bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("shape", [
    (64000 // 8 + 37, 384, 384, "streamed 128 x 384 tiles, row tail"),
    (4100, 1536, 384, "streamed, four column groups"),
    (5000, 384, 1536, "streamed, 24 k-steps"),
    (3000, 21, 384, "128 x 128 branch-free kernel (more than 2048 rows, narrow output)"),
    (300, 672, 384, "one wave per 32 x 32 tile"),
    (257, 96, 256, "one wave per tile, ragged"),
    (77, 50, 100, "general kernel (k not a multiple of 64)"),
], ids=lambda s: f"n{s[0]}_m{s[1]}_k{s[2]}")
def test_training_linear_kernels_unit(shape):
    """Every kernel family behind the training step's linear layer (`mdgen_debug_train_linear`: the dispatch of
    `mdgen_train_forward_backward`), alone, against torch on the same operands: exact mode (fp32 products) to 5e-6 relative,
    bf16-operand mode against the product of the bf16-ROUNDED operands accumulated in fp64 (the only difference left is the fp32
    summation order: 1e-5).  Shapes pick each kernel: the streamed 128 x 384-tile kernel with its LDS-DMA weight stream (row
    tails, one to four column groups, 6 / 24 k-steps), the branch-free 128 x 128 kernel, the one-wave tiles, the general one."""
    from mdgen_amd import _lib as L
    dev = _cuda()
    n, m, k, _ = shape
    gen = torch.Generator().manual_seed(n + m + k)
    a = torch.randn(n, k, generator=gen).to(dev)
    w = (torch.randn(m, k, generator=gen) / k ** 0.5).to(dev)
    bias = torch.randn(m, generator=gen).to(dev)
    scratch = torch.empty(m * k, dtype=torch.bfloat16, device=dev)
    s = L.stream_ptr()
    for prec in (32, 16):
        c = torch.full((n, m), float("nan"), device=dev)
        L.check(L.lib.mdgen_debug_train_linear(prec, L.ptr(a), k, L.ptr(w), k, L.ptr(bias), n, m, k, L.ptr(c), m, L.ptr(scratch), s))
        torch.cuda.synchronize()
        if prec == 32:
            ref = a.double() @ w.double().T + bias.double()
            tol = 5e-6
        else:
            ref = _bf16_round(a).double() @ _bf16_round(w).double().T + bias.double()
            tol = 1e-5
        assert torch.isfinite(c).all()
        e = float((c.double() - ref).norm() / ref.norm())
        assert e < tol, (shape, prec, e)


@pytest.mark.parametrize("shape", [
    (8000 + 13, 384, 384, "wide tiles, row tail"),
    (5000, 1152, 384, "wide, nine row tiles (q | k | v)"),
    (4500, 384, 1536, "wide, four column groups"),
    (1500, 384, 384, "128 x 128 split kernel"),
    (257, 96, 256, "small"),
    (77, 50, 100, "general (unaligned)"),
], ids=lambda s: f"n{s[0]}_m{s[1]}_k{s[2]}")
def test_training_weight_gradient_kernels_unit(shape):
    """The weight / bias gradient kernels (`mdgen_debug_train_dw`) alone: dW += dY^T X, db += colsum(dY), against torch in fp64 on
    the same (exact mode) or the bf16-rounded (bf16-operand mode; the bias gradient sums the unrounded dY) operands, starting
    from non-zero accumulators; run twice for bit-reproducibility of the split reduction."""
    from mdgen_amd import _lib as L
    dev = _cuda()
    n, m, k, _ = shape
    gen = torch.Generator().manual_seed(n * 3 + m + k)
    dy = torch.randn(n, m, generator=gen).to(dev)
    x = torch.randn(n, k, generator=gen).to(dev)
    dw0 = torch.randn(m, k, generator=gen).to(dev)
    db0 = torch.randn(m, generator=gen).to(dev)
    part = torch.empty(16 << 20, device=dev)
    s = L.stream_ptr()
    for prec in (32, 16):
        outs = []
        for _ in range(2):
            dw, db = dw0.clone(), db0.clone()
            L.check(L.lib.mdgen_debug_train_dw(prec, L.ptr(dy), m, L.ptr(x), k, n, m, k, L.ptr(dw), L.ptr(db), L.ptr(part), part.numel(), s))
            torch.cuda.synchronize()
            outs.append((dw, db))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        dyr, xr = (dy, x) if prec == 32 else (_bf16_round(dy), _bf16_round(x))
        ref_w = dw0.double() + dyr.double().T @ xr.double()
        ref_b = db0.double() + dy.double().sum(0)
        ew = float((outs[0][0].double() - ref_w).norm() / ref_w.norm())
        eb = float((outs[0][1].double() - ref_b).norm() / ref_b.norm())
        assert ew < (5e-6 if prec == 32 else 1e-5) and eb < 5e-6, (shape, prec, ew, eb)


def _attn_axis_reference(qkv, mask, bias_k, bias_v, inv_freq, dout, tok):
    """torch fp64 reference of one attention axis as the training kernels see it (mha.py:258-268, 359-396): q (scaled, rotated) |
    k (rotated) | v per token, learned bias key (rotated at position len) / value appended, padded keys masked; autograd for the
    backward, then d q / d k taken back through RoPE (d q also through the q scale) as the kernels return them.
    tok[s][i] = token index.  Returns out, lse, dqkv, dbias."""
    with torch.enable_grad():      # (this module runs with gradients off)
        return _attn_axis_reference_impl(qkv, mask, bias_k, bias_v, inv_freq, dout, tok)


def _attn_axis_reference_impl(qkv, mask, bias_k, bias_v, inv_freq, dout, tok):
    nseq, ln = tok.shape
    H, DH = 16, 24
    qkv = qkv.double().clone().requires_grad_(True)
    bk = bias_k.double().clone().requires_grad_(True)
    bv = bias_v.double().clone().requires_grad_(True)
    f = inv_freq.double()

    def rot(x, pos, sign=1.0):      # x [..., 24], rotate-half RoPE by angle sign * pos * inv_freq
        ang = pos[..., None] * f
        c, sn = torch.cos(ang), torch.sin(ang) * sign
        x1, x2 = x[..., :12], x[..., 12:]
        return torch.cat([x1 * c - x2 * sn, x2 * c + x1 * sn], -1)

    out = torch.zeros(qkv.shape[0], H * DH, dtype=torch.float64)
    lse = torch.zeros(qkv.shape[0], H, dtype=torch.float64)
    outs, lses = [], []
    for s_ in range(nseq):
        t = tok[s_]
        q = qkv[t, 0:384].view(ln, H, DH)
        k = qkv[t, 384:768].view(ln, H, DH)
        v = qkv[t, 768:1152].view(ln, H, DH)
        kb = rot(bk.view(H, DH), torch.full((H,), float(ln), dtype=torch.float64))
        K = torch.cat([k, kb[None]], 0)
        V = torch.cat([v, bv.view(1, H, DH)], 0)
        valid = torch.cat([mask[t] != 0, torch.ones(1, dtype=torch.bool)])
        logit = torch.einsum("ihd,jhd->hij", q, K).masked_fill(~valid[None, None, :], float("-inf"))
        outs.append((t, torch.einsum("hij,jhd->ihd", torch.softmax(logit, -1), V).reshape(ln, H * DH)))
        lses.append((t, torch.logsumexp(logit, -1).T))
    out = torch.zeros(qkv.shape[0], H * DH, dtype=torch.float64)
    lse = torch.zeros(qkv.shape[0], H, dtype=torch.float64)
    loss = 0.0
    for (t, o), (_, l_) in zip(outs, lses):
        loss = loss + (o * dout.double()[t]).sum()
        out[t] = o.detach()
        lse[t] = l_.detach()
    # per-sequence bias gradients: differentiate each sequence's contribution separately
    dbias = torch.zeros(nseq, 768, dtype=torch.float64)
    for s_, (t, o) in enumerate(outs):
        gk, gv = torch.autograd.grad((o * dout.double()[t]).sum(), (bk, bv), retain_graph=True)
        dbias[s_, :384] = gk
        dbias[s_, 384:] = gv
    (g,) = torch.autograd.grad(loss, qkv)
    dq = torch.zeros_like(g)
    for s_ in range(nseq):
        t = tok[s_]
        pos = torch.arange(ln, dtype=torch.float64)[:, None].expand(ln, H)
        dq[t, 0:384] = (rot(g[t, 0:384].view(ln, H, DH), pos, -1.0) * 24 ** -0.5).reshape(ln, 384)
        dq[t, 384:768] = rot(g[t, 384:768].view(ln, H, DH), pos, -1.0).reshape(ln, 384)
        dq[t, 768:] = g[t, 768:]
    return out, lse, dq, dbias


@pytest.mark.parametrize("ln", [1, 5, 31, 32, 33, 64, 65, 127, 128, 129, 160, 192, 224, 250, 255, 256, 257, 300, 1000, 1001])
@pytest.mark.parametrize("layout", ["residue", "temporal"])
def test_training_attention_kernels_unit(ln, layout):
    """The attention kernels of the training step alone (`mdgen_debug_train_attention`), forward and backward, both precisions,
    against a torch fp64 reference with autograd: sequence lengths around every tile / chunk / block boundary (32-row tiles, 64-row
    chunks, 128- and 256-row workgroups; the bias key first / last in a tile; 1000 / 1001 = the tetrapeptide headline's temporal length), both token layouts of the trunk, random key
    padding plus a sequence whose first 40 keys are all padded (whole masked tiles) and one with every real key padded (only
    the bias key left).  Exact mode to 2e-5; bf16 operands: output 1e-2, gradients 3e-2, the bias key's 1e-1 (rel-L2 per tensor).
    Lengths 129 .. 256 take the sequence-resident kernels (round 6, option `train_attn_form`: one workgroup per (sequence, head), the
    whole sequence in LDS; 256 = the bias key opens a ninth tile that is shared out over the waves; 160 / 192 / 224: it is the first
    key of an owned tile) -- there the chunked kernels run as a third leg (precision 160) and the two forms must agree to 1e-2 (the sequence-resident
    forms round q log2(e) to bf16 where the chunked ones round q: measured 4e-3)."""
    from mdgen_amd import _lib as L
    dev = _cuda()
    nseq = 4
    ntok = nseq * ln
    gen = torch.Generator().manual_seed(1000 + ln)
    if layout == "residue":     # token = s * len + i
        tok = torch.arange(ntok).view(nseq, ln)
        ax = (nseq, ln, 1, ln, 0, 1)
    else:                       # token = i * nseq + s
        tok = torch.arange(ntok).view(ln, nseq).T.contiguous()
        ax = (nseq, ln, nseq, ln * nseq, 1, nseq)
    qkv = torch.randn(ntok, 1152, generator=gen)
    qkv[:, :384] *= 24 ** -0.5 * 2.0
    mask = (torch.rand(ntok, generator=gen) > 0.2).float()
    mask[tok[1][:40]] = 0.0
    mask[tok[2]] = 0.0
    bias_k = torch.randn(384, generator=gen)
    bias_v = torch.randn(384, generator=gen)
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, 24, 2).float() / 24))
    dout = torch.randn(ntok, 384, generator=gen)
    r_out, r_lse, r_dqkv, r_dbias = _attn_axis_reference(qkv, mask, bias_k, bias_v, inv_freq, dout, tok)
    # precision 161: q, k handed over UNROTATED, the kernels apply RoPE themselves (what the training step does on such an axis)
    pos_of = torch.empty(ntok, dtype=torch.long)
    pos_of[tok.reshape(-1)] = torch.arange(ln).repeat(nseq)
    ang = pos_of[:, None].float() * inv_freq[None, :]                      # [ntok][12]
    co, si = torch.cos(ang)[:, None, :], torch.sin(ang)[:, None, :]
    def unrot(x):                                                          # inverse of y1 = x1 c - x2 s, y2 = x2 c + x1 s, per head
        y = x.view(ntok, 16, 24)
        y1, y2 = y[..., :12], y[..., 12:]
        return torch.cat([y1 * co + y2 * si, y2 * co - y1 * si], -1).reshape(ntok, 384)
    qkv_raw = torch.cat([unrot(qkv[:, :384]), unrot(qkv[:, 384:768]), qkv[:, 768:]], 1)
    d = lambda t: t.to(dev).contiguous()
    g = dict(qkv=d(qkv), qkv_raw=d(qkv_raw), mask=d(mask), bk=d(bias_k), bv=d(bias_v), f=d(inv_freq), dout=d(dout))
    s = L.stream_ptr()
    legs = ((32, 2e-5, 2e-5), (16, 1e-2, 3e-2)) + (((160, 1e-2, 3e-2), (161, 1e-2, 3e-2)) if 128 < ln <= 256 else ())
    seen = {}
    for prec, tol_o, tol_g in legs:
        out = torch.full((ntok, 384), float("nan"), device=dev)
        lse = torch.full((ntok, 16), float("nan"), device=dev)
        dqkv = torch.full((ntok, 1152), float("nan"), device=dev)
        dbias = torch.full((nseq, 768), float("nan"), device=dev)
        stats = torch.empty(ntok, 16, 2, device=dev)
        L.check(L.lib.mdgen_debug_train_attention(prec, L.ptr(g["qkv_raw" if prec == 161 else "qkv"]), ntok, *ax, L.ptr(g["mask"]), L.ptr(g["bk"]), L.ptr(g["bv"]),
                                                  L.ptr(g["f"]), L.ptr(g["dout"]), L.ptr(out), L.ptr(lse), L.ptr(dqkv), L.ptr(dbias),
                                                  L.ptr(stats), s))
        torch.cuda.synchronize()
        for name, got, ref, tol in (("out", out, r_out, tol_o), ("lse", lse, r_lse, tol_o), ("dq", dqkv[:, :384], r_dqkv[:, :384], tol_g),
                                    ("dk", dqkv[:, 384:768], r_dqkv[:, 384:768], tol_g), ("dv", dqkv[:, 768:], r_dqkv[:, 768:], tol_g),
                                    # the bias key is attended by EVERY query: its d k is the longest, most cancelling sum of
                                    # bf16-rounded d s terms (3e-2 .. 6e-2 on these random inputs, growing with the length)
                                    # (measured 0.17 at len 1000, where the exact mode is at 2e-5: rounding, not logic; the whole-step
                                    # gradients against the reference's autograd are gated separately)
                                    ("dbias_k", dbias[:, :384], r_dbias[:, :384], tol_g if prec == 32 else (1e-1 if ln <= 300 else 2.5e-1)),
                                    ("dbias_v", dbias[:, 384:], r_dbias[:, 384:], tol_g)):
            assert torch.isfinite(got).all(), (name, prec)
            e = float((got.double().cpu() - ref).norm() / (ref.norm() + 1e-300))
            assert e < tol, (ln, layout, prec, name, e)
            seen[(prec, name)] = got.double().cpu()
    if 128 < ln <= 256:   # the two bf16-operand forms: same products, same operand rounding, different summation grouping
        for name in ("out", "lse", "dq", "dk", "dv", "dbias_k", "dbias_v"):
            for leg in (16, 161):
                a, b = seen[(leg, name)], seen[(160, name)]
                e = float((a - b).norm() / (b.norm() + 1e-300))
                assert e < (3e-2 if name == "dbias_k" else 1e-2), (ln, layout, leg, name, e)


def test_row_owner_mlp_paths_agree():
    """The MLP block has these forms: the 64-row resident-panel kernel with four waves (k_mlp<3>) or eight (k_mlp8; `panel_waves`),
    each with or without the temporal out-projection as a prologue phase (`fuse_proj` 2), the row-owner kernel (`mlp_path` 2:
    activations in registers, LDS-DMA weight stream) with or without the out-projection fused in front (`fuse_proj` 1); the
    temporal LN -> q, k, v kernel has a four- and an eight-wave form too (k_ln_qkv<false> / k_ln_qkv8) and, on the tiled residue
    axis, the residue out-projection as a prologue phase (`fuse_proj_qkv`).  Every form against the reference golden at the bf16
    gate and against each other (summation orders / GELU polynomial: a few 1e-3) -- and the profile report must name the
    kernel the options asked for (`@p4` / `@p8` tags): at these sizes every launch is <= 256 panels, so without `panel_waves` 4 the
    four-wave kernels would not run at all (round-4 verdict, weak #1).  The IPA stack (S*B*L rows: partial tiles, idle waves)
    takes the same kernels."""
    from mdgen_amd.model import LatentMDGenModel
    dev = _cuda()
    P4, P8 = {"panel_waves": 4}, {"panel_waves": 8, "small_split": 0}
    X8 = {"panel_waves": 8, "small_split": 1}   # (round 5) a panel's work over several workgroups: k_mlp8<., 3>, k_ln_qkv8<true>
    forms = (("panel4", dict(P4, mlp_path=0, fuse_proj=0), "mlp@p4"), ("panel8", dict(P8, mlp_path=0, fuse_proj=0), "mlp@p8"),
             ("panel4+proj", dict(P4, mlp_path=0, fuse_proj=2), "proj_mlp@p4"), ("panel8+proj", dict(P8, mlp_path=0, fuse_proj=2), "proj_mlp@p8"),
             ("panel8 split", dict(X8, mlp_path=0, fuse_proj=0), "mlp@p8x3"), ("panel8+proj split", dict(X8, mlp_path=0, fuse_proj=2), "proj_mlp@p8x3"),
             ("rows", {"mlp_path": 2, "fuse_proj": 0, "mlp_fold": 0}, "mlp"), ("rows+proj", {"mlp_path": 2, "fuse_proj": 1}, "proj_mlp"),
             # (round 6) a forward of B = 1 shares t: the gate-folded form of the row-owner kernel (option mlp_fold, default on)
             ("rows fold", {"mlp_path": 2, "fuse_proj": 0}, "mlp@fold|mlp"),
             ("no-qkv-prologue p4", dict(P4, mlp_path=0, fuse_proj=0, fuse_proj_qkv=0), "ln_qkv_T"),
             ("no-qkv-prologue p8", dict(P8, mlp_path=0, fuse_proj=0, fuse_proj_qkv=0), "ln_qkv_T@p8"),
             ("no-qkv-prologue p8 split", dict(X8, mlp_path=0, fuse_proj=0, fuse_proj_qkv=0), "ln_qkv_T@p8x2"),
             ("qkv-prologue", dict(P4, mlp_path=0, fuse_proj=0, fuse_proj_qkv=1), None), ("defaults", {}, None))
    for name in ("fwd_full_pep", "fwd_full_atlas"):
        g = load_golden(name)
        cfg, sd = weights_for(g)
        outs = {}
        for key, opts, want in forms:
            m = LatentMDGenModel(cfg)
            m.load_state_dict(sd)
            m.set_option("flash_proj", 0)   # (this test is about the projection-carrying forms of the MLP / q, k, v kernels)
            for k, v in opts.items():
                m.set_option(k, v)
            kw = _kw(g, dev)
            kw.pop("end_frames")
            m.profile(True)
            out = m.forward(**kw)
            torch.cuda.synchronize()
            ran = {k: v["count"] for k, v in m.profile_report().items()}
            m.profile(False)
            assert torch.isfinite(out).all()
            outs[key] = out.cpu()
            e = rel_l2(outs[key], g["out"])
            print(f"{name} {key}: rel-L2 vs reference {e:.2e}   {sorted(k for k in ran if 'mlp' in k or 'qkv' in k)}")
            assert e < TOL_FWD
            T_, L_ = g["x"].shape[1:3]
            if want is not None:
                if "|" in want:   # by batch size: B = 1 -> the first name, else the second
                    want = want.split("|")[0 if g["x"].shape[0] == 1 else 1]
                # (an untraced forward: the folded form's last launch also runs the FinalLayer, "mlp@fold+final")
                # ("@h32x2": the split q, k | v kernel on 32-position workgroups, where even those fit one per CU -- round 6)
                alt = "ln_qkv_T@h32x2" if want == "ln_qkv_T@p8x2" else want
                assert ran.get(want, 0) + ran.get(want + "+final", 0) + (ran.get(alt, 0) if alt != want else 0) == cfg.num_layers, (key, want, ran)
                family = [k for k in ran if k.split("@")[0] == want.split("@")[0] and k not in (want, want + "+final", alt)]   # no other form of the same kernel ran
                assert not family, (key, want, ran)
                if "split" in key:   # the IPA stack's MLP (S B L rows: one panel here) takes the split form too
                    assert any(k == "ipa.mlp@p8x3" for k in ran), ran
            if key == "qkv-prologue":
                assert ("projL_qkvT" in ran) == (L_ > 8 and T_ > 8), ran   # (neither golden has both: test_panel_kernels_257_to_383_panels_vs_oracle does)
            del m
        assert rel_l2(outs["panel8"], outs["panel4"]) < 6e-3 and rel_l2(outs["panel8+proj"], outs["panel4+proj"]) < 6e-3
        assert rel_l2(outs["rows"], outs["panel4"]) < 6e-3 and rel_l2(outs["rows+proj"], outs["rows"]) < 6e-3
        assert rel_l2(outs["rows fold"], outs["rows"]) < 3e-3
        assert rel_l2(outs["panel4+proj"], outs["panel4"]) < 6e-3
        assert rel_l2(outs["no-qkv-prologue p8"], outs["no-qkv-prologue p4"]) < 6e-3
        # the split forms: k_ln_qkv8<true> computes the same products with the same operands (same bits as k_ln_qkv8<false>);
        # k_mlp8<., 3> adds three partial fc2 sums instead of two (fp32 rounding)
        assert torch.equal(outs["no-qkv-prologue p8 split"], outs["no-qkv-prologue p8"]) or \
            rel_l2(outs["no-qkv-prologue p8 split"], outs["no-qkv-prologue p8"]) < 2e-3   # (its MLP is the split form too)
        assert rel_l2(outs["panel8 split"], outs["panel8"]) < 2e-3 and rel_l2(outs["panel8+proj split"], outs["panel8+proj"]) < 2e-3
        assert rel_l2(outs["qkv-prologue"], outs["no-qkv-prologue p4"]) < 6e-3
        # the defaults at this size: eight-wave panel MLP kernel (at most one workgroup per CU), split over three workgroups, with
        # the out-projection in front
        assert rel_l2(outs["defaults"], outs["panel8+proj split"]) < 6e-3


def test_validation_on_ema_weights_leaves_the_master_parameters_alone(tmp_path):
    """wrapper.py:64-76, 88-97: validation runs on the EMA weights.  `Trainer.validation_loss` swaps them into the library (the
    training kernels' fp32 weights are BOUND to the flat parameter buffer: the swap must not write there), evaluates
    `general_step`, and the next training step continues from the raw parameters: two steps with a validation pass in
    between equal two steps without one, bit for bit.  Also: checkpoint -> `load_from_checkpoint` -> `load_ema_weights` /
    `restore_cached_weights` on the inference wrapper."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    from mdgen_amd.train import Trainer
    from mdgen_amd.wrapper import NewMDGenWrapper
    dev = _cuda()
    B, T, L = 2, 6, 5
    cfg = ModelConfig(crop=L, num_frames=T, num_layers=1, abs_pos_emb=True, sim_condition=True)
    sd = synth_state_dict(cfg, 23)
    g0 = load_golden("prep_sim")
    batch = {k[3:]: v.to(dev) for k, v in g0.items() if k.startswith("in_")}
    res = []
    for validate in (False, True):
        w = NewMDGenWrapper(cfg)
        w.load_model_state_dict(sd)
        tr = Trainer(w, lr=1e-3, grad_clip=1.0, ema_decay=0.5)
        gen = torch.Generator().manual_seed(3)
        for step in range(2):
            t = torch.rand(B, generator=gen).to(dev)
            x0 = torch.randn(B, T, L, cfg.latent_dim, generator=gen).to(dev)
            tr.training_step(batch, t=t, x0=x0)
            if validate and step == 0:
                v_ema = tr.validation_loss([batch])
                assert v_ema == v_ema and v_ema > 0
        torch.cuda.synchronize()
        res.append((tr.tm.params.data.clone(), tr.ema.data.clone()))
        if validate:
            path = str(tmp_path / "t.ckpt")
            tr.save_checkpoint(path)
        tr.close()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    wi = NewMDGenWrapper.load_from_checkpoint(path)
    kw = dict(x=torch.zeros(B, T, L, cfg.latent_dim, device=dev), t=torch.zeros(B, device=dev))
    prep = wi.prep_batch(batch)
    mk = dict(prep["model_kwargs"]); mk["mask"] = mk["mask"].contiguous(); mk.pop("end_frames")
    raw = wi.model.forward(**kw, **mk)
    wi.load_ema_weights()
    ema = wi.model.forward(**kw, **mk)
    wi.restore_cached_weights()
    raw2 = wi.model.forward(**kw, **mk)
    assert torch.equal(raw, raw2) and not torch.equal(raw, ema) and torch.isfinite(ema).all()


# ---- round 5: k_flash_proj, the 257..383-panel window, four- vs eight-wave panel kernels ---------------------------------------
def _fwd_case(B, T, L, n_pad, seed, weights_seed=5, tps=False):
    """(cfg, sd, host kwargs, device kwargs) of one synthetic forward call (forward-simulation model, crop max(L, 4); tps: the
    two-sided model, D = 28)."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_forward_inputs, synth_state_dict
    dev = _cuda()
    cfg = ModelConfig(crop=max(L, 4), num_frames=T, abs_pos_emb=True, sim_condition=False, tps_condition=True) if tps else \
        ModelConfig.forward_sim(num_frames=T, crop=max(L, 4))
    sd = synth_state_dict(cfg, weights_seed)
    inp = synth_forward_inputs(cfg, B, T, L, n_pad, seed)
    kw = dict(x=inp["x"], t=inp["t"], mask=inp["mask"], start_frames=(inp["start_rot"], inp["start_trans"]),
              end_frames=(inp["end_rot"], inp["end_trans"]), x_cond=inp["x_cond"], x_cond_mask=inp["x_cond_mask"],
              aatype=inp["aatype"])
    dkw = {k: (tuple(u.to(dev) for u in v) if isinstance(v, tuple) else v.to(dev)) for k, v in kw.items()}
    return cfg, sd, kw, dkw


def _profiled_forward(m, dkw, poison=True):
    """One forward on a 0xFF-filled workspace with the hipEvent profile on: (out, trace, {kernel class: launches})."""
    m.forward(**dkw)                       # allocates (and caches) the workspace of this shape
    if poison:
        for ws in m._ws.values():
            ws.view(torch.uint8).fill_(0xFF)
    m.profile(True)
    try:
        out, tr = m.forward(**dkw, return_trace=True)
        torch.cuda.synchronize()
        rep = m.profile_report()
    finally:
        m.profile(False)
    return out, tr, {k: v["count"] for k, v in rep.items()}


@pytest.mark.parametrize("shape", [(1, 130, 9, 1), (2, 64, 33, 0), (1, 40, 96, 5), (2, 1000, 4, 0), (1, 250, 64, 3)],
                         ids=["T130_L9", "T64_L33", "T40_L96", "T1000_L4", "T250_L64"])
def test_flash_proj_kernel_vs_separate_kernels_and_oracle(shape):
    """Option `flash_proj` (round 5): the tiled attention of all 16 heads for 64 queries of a sequence + the sub-layer's
    out-projection + gated residual in ONE launch (k_flash_proj; mha.py:359-397, latent_model.py:462,476) against (a) the CPU oracle
    at the bf16 gate, every trace, and (b) the separate kernels it replaces (k_flash, then k_proj<0> or a projection deferred into
    the next kernel): same operands, same summation order -> equal to fp32 rounding.  Both key-tile walk orders (`flash_rotate`)
    and both softmax loops (`attention_path` 0 / 1).  Shapes: a partial last 64-query chunk whose second tile is past the
    sequence (T 130, L 9), T a multiple of 64 (the bias key opens a key tile of its own), L 33 / 96 / 64 on the residue axis,
    padded residues (their temporal sequences see only the learned bias key: the direct bias_v path), the headline's T 1000.
    Workspace filled with 0xFF bytes; the profile report says which kernels ran."""
    from oracle import mdgen_oracle as O
    from mdgen_amd.model import LatentMDGenModel
    B, T, L, n_pad = shape
    cfg, sd, kw, dkw = _fwd_case(B, T, L, n_pad, 500 + T + L)
    ref, rtr = O.forward(sd, O.cfg_dict(cfg), return_trace=True, **kw)
    outs = {}
    F4, F8 = {"flash_proj": 2, "flash_proj_form": 4}, {"flash_proj": 2, "flash_proj_form": 8}
    for key, opts in (("separate", {"flash_proj": 0}), ("fused", F4), ("fused, natural tile order", dict(F4, flash_rotate=0)),
                      ("fused, robust loop", dict(F4, attention_path=1)), ("fused 128", F8),
                      ("fused 128, natural tile order", dict(F8, flash_rotate=0)), ("fused 128, robust loop", dict(F8, attention_path=1)),
                      ("separate, natural tile order", {"flash_proj": 0, "flash_rotate": 0})):
        m = LatentMDGenModel(cfg)
        m.load_state_dict(sd)
        for k, v in opts.items():
            m.set_option(k, v)
        out, tr, ran = _profiled_forward(m, dkw)
        assert torch.isfinite(out).all(), key
        rep = {k: rel_l2(tr[k].cpu(), rtr[k]) for k in ["ipa_out"] + [f"h{i}" for i in range(cfg.num_layers + 1)]}
        rep["out"] = rel_l2(out.cpu(), ref)
        print(shape, key, {k: f"{v:.2e}" for k, v in rep.items()}, {k: v for k, v in ran.items() if "flash" in k or "proj" in k})
        for k, v in rep.items():
            assert v < TOL_FWD, (key, k, v)
        fused = not key.startswith("separate")
        tag = "@q128" if "128" in key else "@q64"      # the report names the fused form: k_flash_proj8 / k_flash_proj
        assert ("flash_proj_T" + tag in ran) == fused and ("flash_T" in ran) != fused, ran
        assert not any(k.startswith("flash_proj") and not k.endswith(tag) for k in ran), ran
        if L > 8:
            assert ("flash_proj_L" + tag in ran) == fused and ("ipa.flash_proj" + tag in ran) == fused, ran
        if fused:   # nothing left to project: no k_proj<0>, no projection deferred into the next kernel
            assert not any(k.startswith(("proj_T", "proj_L", "projL_qkvT", "proj_mlp")) for k in ran), ran
        outs[key] = (out.cpu(), tr[f"h{cfg.num_layers}"].cpu())
        assert torch.equal(m.forward(**dkw), out), key   # a second call on the same context: the same bits
        del m
    for key in ("fused", "fused, natural tile order", "fused 128"):
        e_out, e_h = rel_l2(outs[key][0], outs["separate"][0]), rel_l2(outs[key][1], outs["separate"][1])
        print(shape, f"{key} vs separate kernels: out {e_out:.2e} h {e_h:.2e} equal {torch.equal(outs[key][0], outs['separate'][0])}")
        # (`flash_rotate` changes the order in which a query's keys are summed: fp32 rounding, now and then one bf16 step of an
        # attention output; k_flash and k_flash_proj walk the tiles in the same rotated order, the 128-row form rotates by
        # 128-query chunks)
        assert (e_out < 2e-5 and e_h < 2e-5) if key == "fused" else (e_out < 2e-3 and e_h < 2e-3)
    # in natural tile order all three forms sum every query's keys in the same order: the same bits
    assert torch.equal(outs["fused, natural tile order"][0], outs["separate, natural tile order"][0])
    e8 = rel_l2(outs["fused 128, natural tile order"][0], outs["separate, natural tile order"][0])
    print(shape, f"128-row form vs separate kernels, natural tile order: {e8:.2e}")
    assert e8 < 2e-5
    assert rel_l2(outs["fused, robust loop"][0], outs["fused"][0]) < 6e-3   # (P rounded to bf16 around a different shift)
    assert rel_l2(outs["fused 128, robust loop"][0], outs["fused 128"][0]) < 6e-3
    if n_pad:   # padded residues never influence the valid ones
        m = LatentMDGenModel(cfg)
        m.load_state_dict(sd)
        m.set_option("flash_proj", 2)
        m.set_option("flash_proj_form", 8)
        x2 = kw["x"].clone()
        x2[:, :, L - n_pad:] = 1e3
        a = m.forward(**dkw)
        b2 = m.forward(**dict(dkw, x=x2.to(a.device)))
        assert torch.equal(a[:, :, :L - n_pad], b2[:, :, :L - n_pad])


def test_flash_proj_is_the_default_where_the_launch_fills_the_chip():
    """`flash_proj` 1 (default): the fused kernel for launches of >= 512 (sequence, 64-query chunk) workgroups -- B 8 x T 1000 x L 4 is
    32 sequences x 16 chunks = 512 (and 32 x 8 = 256 128-query chunks: one per CU, so `flash_proj_form` 0 picks the 128-row form
    k_flash_proj8 there) -- and the separate kernels below (B 7: 448).  Both against the separate kernels' output."""
    from mdgen_amd.model import LatentMDGenModel
    for B, want in ((8, True), (7, False)):
        cfg, sd, kw, dkw = _fwd_case(B, 1000, 4, 0, 77)
        m = LatentMDGenModel(cfg)
        m.load_state_dict(sd)
        out, _, ran = _profiled_forward(m, dkw, poison=False)
        assert ("flash_proj_T@q128" in ran) == want and ("flash_T" in ran) != want, (B, ran)
        m.set_option("flash_proj", 0)
        out0, _, ran0 = _profiled_forward(m, dkw, poison=False)
        assert "flash_T" in ran0 and not any(k.startswith("flash_proj") for k in ran0)
        e = rel_l2(out, out0)
        print(f"B {B}: default {sorted(k for k in ran if 'flash' in k)} vs flash_proj 0: {e:.2e}")
        assert torch.isfinite(out).all() and e < 2e-3   # (the 128-row form rotates its key-tile walk by 128-query chunks: fp32 rounding)
        del m


@pytest.mark.parametrize("shape", [(1, 1000, 4, 0), (1, 130, 4, 0), (1, 40, 96, 5), (2, 100, 9, 1)],
                         ids=["B1_T1000_L4", "B1_T130_L4", "B1_T40_L96_pad", "B2_T100_L9_pad"])
def test_small_launches_split_a_panel_over_workgroups_vs_oracle(shape):
    """Option `small_split` (round 5, default on): launches far below one workgroup per CU -- the CLI's B = 1 (sim_inference.py:100-113),
    the IPA stack -- give a 64-row panel to SEVERAL workgroups: k_mlp8<., 3> (latent_model.py:477-481: the twelve hidden chunks over
    three workgroups on one XCD; fp32 partials of the fc2 product meet in L2 and the last arriver runs the gated residual epilogue;
    with the temporal out-projection fused in front each workgroup keeps a private copy of the updated rows) and k_ln_qkv8<true>
    (q, k | v over two workgroups).  Against the CPU oracle at the bf16 gate (every trace), against the unsplit kernels (fp32
    rounding of a three-term instead of a two-term sum), bit-reproducible over repeated calls on a 0xFF-filled workspace (the arrival
    counters are re-zeroed per call; a race between the workgroups of a panel would show as differing bits), and through a replayed
    hipGraph rollout.  Shapes: 63 panels (B 1 x T 1000 x L 4: the headline model at B = 1), 9 panels with a partial last one and a
    grid padded to whole XCD groups (T 130), the tiled residue axis with padded residues (L 96), L 9."""
    from oracle import mdgen_oracle as O
    from mdgen_amd.model import LatentMDGenModel
    B, T, L, n_pad = shape
    cfg, sd, kw, dkw = _fwd_case(B, T, L, n_pad, 1200 + T + L)
    ref, rtr = O.forward(sd, O.cfg_dict(cfg), return_trace=True, **kw)
    outs = {}
    for key, opts in (("split", {}), ("unsplit", {"small_split": 0})):
        m = LatentMDGenModel(cfg)
        m.load_state_dict(sd)
        for k, v in opts.items():
            m.set_option(k, v)
        out, tr, ran = _profiled_forward(m, dkw)
        rep = {k: rel_l2(tr[k].cpu(), rtr[k]) for k in ["ipa_out"] + [f"h{i}" for i in range(cfg.num_layers + 1)]}
        rep["out"] = rel_l2(out.cpu(), ref)
        print(shape, key, {k: f"{v:.2e}" for k, v in rep.items()}, {k: v for k, v in ran.items() if "mlp" in k or "qkv" in k})
        assert torch.isfinite(out).all()
        for k, v in rep.items():
            assert v < TOL_FWD, (key, k, v)
        split = key == "split"
        nl = cfg.num_layers
        assert (ran.get("proj_mlp@p8x3", 0) + ran.get("mlp@p8x3", 0) == nl) == split, ran
        assert ("ipa.mlp@p8x3" in ran) == split, ran
        qkv = [k for k in ran if k.startswith("ln_qkv_T")]   # (none on the tiled residue axis: projL_qkvT carries the projection)
        assert all((k in ("ln_qkv_T@p8x2", "ln_qkv_T@h32x2")) == split for k in qkv) and (qkv or L > 8), ran   # (@h32x2: 32-position workgroups)
        if split:   # repeated calls, each on a poisoned workspace: the same bits (profile off: the product's launch path)
            for rep_i in range(6):
                for ws in m._ws.values():
                    ws.view(torch.uint8).fill_(0xFF)
                again = m.forward(**dkw)
                assert torch.equal(again, out), (shape, rep_i, rel_l2(again.cpu(), out.cpu()))
        outs[key] = out.cpu()
        del m
    e = rel_l2(outs["split"], outs["unsplit"])
    print(shape, f"split vs unsplit kernels: {e:.2e}")
    assert e < 2e-3


def test_small_split_in_a_replayed_graph_rollout():
    """The split kernels inside the hipGraph Euler rollout (B 1 x T 1000 x L 4, 3 steps): replays give the same bits (counters zeroed
    ahead of every launch of the graph, partial sums added in a fixed order) and the trajectory matches the unsplit kernels'."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    from mdgen_amd.wrapper import NewMDGenWrapper
    import bench
    dev = _cuda()
    B, T, L, abs_pos, n_pad = bench.WORKLOADS["tetrapeptide_fwdsim_crop4_T1000_B1"]
    cfg = ModelConfig(crop=L, num_frames=T, abs_pos_emb=abs_pos, sim_condition=True)
    sd = synth_state_dict(cfg, 0)
    batch = bench.synth_batch(B, T, L, n_pad, dev, seed=100)
    zs = torch.randn(1, 1000, 4, 21, generator=torch.Generator().manual_seed(137)).to(dev)
    res = {}
    for key, v in (("split", 1), ("unsplit", 0)):
        w = NewMDGenWrapper(cfg, device=dev)
        w.model.load_state_dict(sd)
        w.model.set_option("small_split", v)
        a, _ = w.inference(batch, zs=zs, num_steps=3, use_graph=True)
        for _ in range(3):
            b, _ = w.inference(batch, zs=zs, num_steps=3, use_graph=True)   # replays
            assert torch.equal(a, b), key
        e, _ = w.inference(batch, zs=zs, num_steps=3, use_graph=False)
        assert torch.equal(a, e), key                                        # eager launches: the same kernels, the same bits
        res[key] = a.cpu()
        del w
    d = rel_l2(res["split"], res["unsplit"])
    print(f"rollout, split vs unsplit kernels: {d:.2e}")
    assert torch.isfinite(res["split"]).all() and d < 5e-3


@pytest.mark.parametrize("case", ["B5_T1000_L4", "B1_T250_L80_pad"])
def test_panel_kernels_257_to_383_panels_vs_oracle(case):
    """Launches of 257..383 64-row panels take the FOUR-wave panel kernels (more than one workgroup per CU, fewer than the 768 row
    tiles the row-owner MLP wants): k_mlp<3, true> (temporal out-projection as its prologue phase, `fuse_proj` 3) and, on the
    tiled residue axis, k_ln_qkv<false, true> (`fuse_proj_qkv`).  That window is what two sub-batch streams make of B = 9..12 at
    T 1000 L 4 and what ATLAS chains of L ~ 66..98 are at T 250.  Default options, every element against the CPU oracle
    (latent_model.py:446-481), 313 panels each; the profile report names the kernels that ran."""
    from oracle import mdgen_oracle as O
    from mdgen_amd.model import LatentMDGenModel
    B, T, L, n_pad = (5, 1000, 4, 0) if case == "B5_T1000_L4" else (1, 250, 80, 10)
    assert 257 <= (B * T * L + 63) // 64 <= 383
    cfg, sd, kw, dkw = _fwd_case(B, T, L, n_pad, 900 + L)
    ref, rtr = O.forward(sd, O.cfg_dict(cfg), return_trace=True, **kw)
    m = LatentMDGenModel(cfg)
    m.load_state_dict(sd)
    out, tr, ran = _profiled_forward(m, dkw)
    rep = {k: rel_l2(tr[k].cpu(), rtr[k]) for k in ["ipa_out"] + [f"h{i}" for i in range(cfg.num_layers + 1)]}
    rep["out"] = rel_l2(out.cpu(), ref)
    print(case, {k: f"{v:.2e}" for k, v in rep.items()}, ran)
    assert torch.isfinite(out).all()
    for k, v in rep.items():
        assert v < TOL_FWD, (k, v)
    nl = cfg.num_layers
    assert ran.get("proj_mlp@p4") == nl and "mlp" not in ran and "proj_mlp@p8" not in ran, ran   # k_mlp<3, true>, not k_mlp8 / k_mlp_rows
    if L > 8:
        assert ran.get("projL_qkvT") == nl, ran                                                   # k_ln_qkv<false, true>
    else:
        assert ran.get("ln_qkv_T") == nl and "ln_qkv_T@p8" not in ran, ran                        # k_ln_qkv<false>, not k_ln_qkv8
    # the worst element, not only the norm: nothing in the window is off by more than a few bf16 roundings of the output scale
    assert (out.cpu() - ref).abs().max() < 0.1 * ref.abs().max()


def test_two_stream_views_in_the_panel_window_match_one_stream():
    """sample_euler B 10 x T 1000 x L 4 with `streams` 2 = two views of B 5 (313 panels each: k_mlp<3, true>, k_ln_qkv<false>)
    against one stream -- (a) with the same kernels forced (`mlp_path` 0, `fuse_proj` 2: bit-identical, a panel's arithmetic does
    not depend on the launch it is in) and (b) with the defaults (one view of 625 panels takes the row-owner MLP: rounding level)."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.model import LatentMDGenModel
    from mdgen_amd.synthetic import synth_state_dict
    dev = _cuda()
    B, T, L, S = 10, 1000, 4, 3
    cfg = ModelConfig.forward_sim(num_frames=T, crop=4)
    sd = synth_state_dict(cfg, 3)
    gen = torch.Generator().manual_seed(16)
    zs = torch.randn(B, T, L, 21, generator=gen).to(dev)
    mask = torch.ones(B, T, L, device=dev)
    R = torch.eye(3, device=dev).expand(B, L, 3, 3).contiguous()
    tr_ = torch.randn(B, L, 3, generator=gen).to(dev)
    cm = torch.zeros(B, T, L, dtype=torch.long, device=dev)
    cm[:, 0] = 1
    xc = torch.where(cm.unsqueeze(-1).bool(), torch.randn(B, T, L, 21, generator=gen).to(dev), torch.zeros((), device=dev))
    aat = torch.randint(0, 20, (B, L), generator=gen).to(dev)
    kw = dict(mask=mask, start_frames=(R, tr_), x_cond=xc, x_cond_mask=cm, aatype=aat)
    outs = {}
    # (`flash_proj_form` 4 in the pair that must agree bit for bit: by shape a 5-sample view takes the 64-query fused attention kernel
    # and the 10-sample launch the 128-query one, which rotates its key-tile walk by 128-query chunks -- another order of the same sum)
    for key, opts in (("two streams", {"streams": 2, "flash_proj_form": 4}),
                      ("one stream, same kernels", {"streams": 1, "mlp_path": 0, "fuse_proj": 2, "flash_proj_form": 4}),
                      ("one stream, defaults", {"streams": 1})):
        m = LatentMDGenModel(cfg)
        m.load_state_dict(sd)
        for k, v in opts.items():
            m.set_option(k, v)
        outs[key] = [m.sample_euler(zs, S, use_graph=g, **kw) for g in (False, True)]
        torch.cuda.synchronize()
        assert torch.isfinite(outs[key][0]).all() and torch.equal(outs[key][0], outs[key][1]), key   # graph == eager
        del m
    a = outs["two streams"][0]
    assert torch.equal(a, outs["one stream, same kernels"][0])
    e = rel_l2(a, outs["one stream, defaults"][0])
    print(f"two 313-panel views vs one 625-panel view (row-owner MLP): {e:.2e}")
    assert e < 3e-3


# ---- round 6: the gate-folded row-owner MLP, the headline's own kernel mix against the oracle ------------------------------------
def _euler_kw(dkw, tps=False):
    return {k: v for k, v in dkw.items() if k not in (("x", "t") if tps else ("x", "t", "end_frames"))}


def _oracle_two_euler_steps(sd, cfg, kw, v0=None, cd=None):
    """x2 of the fixed-grid Euler rollout with S = 2 (t = 0, 0.5; integrators.py:95-114) from the CPU oracle; v0: the oracle's velocity
    at (x0, t = 0) when the caller has it already."""
    from oracle import mdgen_oracle as O
    B = kw["x"].shape[0]
    cd = cd if cd is not None else O.cfg_dict(cfg)
    if v0 is None:
        v0 = O.forward(sd, cd, **dict(kw, t=torch.zeros(B)))
    x1 = kw["x"] + 0.5 * v0
    v1 = O.forward(sd, cd, **dict(kw, x=x1, t=torch.full((B,), 0.5)))
    return x1 + 0.5 * v1


@pytest.mark.parametrize("shape", [(1, 250, 64, 3), (1, 1000, 4, 0), (1, 70, 9, 1)], ids=["T250_L64_pad", "T1000_L4", "T70_L9_pad"])
def test_mlp_gate_fold_vs_unfolded_kernel_and_oracle(shape):
    """Option `mlp_fold` (round 6, default on): where a call shares t across the batch (sampling, integrators.py:99; a forward of
    B = 1) the MLP block's gate (latent_model.py:481) is ONE vector per (step, layer): k_pack_fold folds it into per-(step, layer)
    fc2 fragments (rounded to bf16 once, from the fp32 weight) and into b2' = gate * b2; k_mlp_rows<., FOLD> starts its fc2
    accumulators from the residual rows + b2' and its epilogue only stores.  Against the CPU oracle (every trace, 0xFF workspace),
    against the unfolded kernel (the product gate * w is rounded instead of w: same order of error), the same bits on a second
    call, and the report must say which form ran.  `mlp_path` 2 forces the row-owner kernel at these sizes (partial last tiles
    and padded residues included)."""
    from oracle import mdgen_oracle as O
    from mdgen_amd.model import LatentMDGenModel
    B, T, L, n_pad = shape
    cfg, sd, kw, dkw = _fwd_case(B, T, L, n_pad, 1600 + T + L)
    ref, rtr = O.forward(sd, O.cfg_dict(cfg), return_trace=True, **kw)
    nl = cfg.num_layers
    outs = {}
    for key, fold in (("fold", 1), ("unfolded", 0)):
        m = LatentMDGenModel(cfg)
        m.load_state_dict(sd)
        for k, v in (("mlp_path", 2), ("fuse_proj", 0), ("mlp_fold", fold)):
            m.set_option(k, v)
        out, tr, ran = _profiled_forward(m, dkw)
        rep = {k: rel_l2(tr[k].cpu(), rtr[k]) for k in ["ipa_out"] + [f"h{i}" for i in range(nl + 1)]}
        rep["out"] = rel_l2(out.cpu(), ref)
        print(shape, key, {k: f"{v:.2e}" for k, v in rep.items()}, {k: v for k, v in ran.items() if "mlp" in k or "fold" in k})
        assert torch.isfinite(out).all()
        for k, v in rep.items():
            assert v < TOL_FWD, (key, k, v)
        assert (ran.get("mlp@fold") == nl and "mlp" not in ran and ran.get("fold_pack") == 1) if fold else \
            (ran.get("mlp") == nl and "mlp@fold" not in ran and "fold_pack" not in ran), ran
        # without a residual-stream trace the folded form also runs the FinalLayer as the last MLP launch's tail (option mlp_tail):
        # the same bits call to call, the traced call's values to fp32 summation order (+ a bf16 rounding flip now and then)
        m.profile(True)
        o2 = m.forward(**dkw)
        ran2 = {k: v["count"] for k, v in m.profile_report().items()}
        m.profile(False)
        assert torch.equal(m.forward(**dkw), o2), key
        assert (ran2.get("mlp@fold+final") == 1 and ran2.get("mlp@fold") == nl - 1 and "final_euler" not in ran2) if fold else \
            (ran2.get("final_euler") == 1 and ran2.get("mlp") == nl), ran2
        e2 = rel_l2(o2.cpu(), out.cpu())
        print(shape, key, f"untraced call (FinalLayer as the MLP's tail: {bool(fold)}) vs traced call: {e2:.2e}; vs oracle {rel_l2(o2.cpu(), ref):.2e}")
        assert e2 < 2e-3 and rel_l2(o2.cpu(), ref) < TOL_FWD
        if fold:
            m.set_option("mlp_tail", 0)
            o3 = m.forward(**dkw)
            assert torch.equal(o3, out), "mlp_tail 0 is the separate k_final launch"
        outs[key] = (out.cpu(), tr[f"h{nl}"].cpu())
        del m
    e_out, e_h = rel_l2(outs["fold"][0], outs["unfolded"][0]), rel_l2(outs["fold"][1], outs["unfolded"][1])
    print(shape, f"folded vs unfolded kernel: out {e_out:.2e} h {e_h:.2e}")
    assert e_out < 3e-3 and e_h < 3e-3


def test_mlp_gate_fold_in_the_euler_rollout():
    """The folded form through `sample_euler` (S = 3 distinct gates per layer; B 3 x T 300 x L 4, `mlp_path` 2): graph replay ==
    eager launches bit for bit, two sub-batch streams == one stream (a panel's arithmetic does not depend on its launch), and
    the trajectory matches the unfolded kernel's to rounding."""
    from mdgen_amd.model import LatentMDGenModel
    cfg, sd, kw, dkw = _fwd_case(3, 300, 4, 0, 1777)
    ekw = _euler_kw(dkw)
    res = {}
    for key, opts in (("fold", {}), ("fold, one stream", {"streams": 1}), ("fold, separate final layer", {"mlp_tail": 0}),
                      ("fold, separate embedding", {"mlp_tail": 1}), ("unfolded", {"mlp_fold": 0})):
        m = LatentMDGenModel(cfg)
        m.load_state_dict(sd)
        for k, v in dict({"mlp_path": 2, "fuse_proj": 0, "streams": 2}, **opts).items():
            m.set_option(k, v)
        a = m.sample_euler(dkw["x"], 3, use_graph=False, **ekw)
        b = m.sample_euler(dkw["x"], 3, use_graph=True, **ekw)
        c = m.sample_euler(dkw["x"], 3, use_graph=True, **ekw)
        assert torch.isfinite(a).all() and torch.equal(a, b) and torch.equal(b, c), key
        m.profile(True)
        m.sample_euler(dkw["x"], 3, use_graph=False, **ekw)
        ran = {k: v["count"] for k, v in m.profile_report().items()}
        m.profile(False)
        nl = cfg.num_layers
        want = {"unfolded": {"mlp": 3 * nl, "final_euler": 3, "embed": 3},
                "fold, separate final layer": {"mlp@fold": 3 * nl, "final_euler": 3, "embed": 3},
                "fold, separate embedding": {"mlp@fold": 3 * (nl - 1), "mlp@fold+final": 3, "embed": 3}}.get(
            key, {"mlp@fold": 3 * (nl - 1), "mlp@fold+final": 1, "mlp@fold+final+embed": 2, "embed": 1, "embed_base": 1})
        for k in ("mlp", "mlp@fold", "mlp@fold+final", "mlp@fold+final+embed", "final_euler", "embed", "embed_base"):
            assert ran.get(k, 0) == want.get(k, 0), (key, k, ran)
        res[key] = a.cpu()
        del m
    assert torch.equal(res["fold"], res["fold, one stream"])
    e = rel_l2(res["fold, separate embedding"], res["fold, separate final layer"])
    print(f"3 Euler steps, FinalLayer as the last MLP launch's tail vs k_final: {e:.2e}")
    assert e < 1e-3
    e = rel_l2(res["fold"], res["fold, separate embedding"])   # (fp32-level products either way; measured 1.9e-4 after three steps: the
    print(f"3 Euler steps, next step's token embedding as that launch's tail vs k_embed: {e:.2e}")   # last bits of h0 flip bf16 roundings downstream)
    assert e < 1e-3
    e = rel_l2(res["fold"], res["unfolded"])
    print(f"3 Euler steps, folded vs unfolded MLP kernel: {e:.2e}")
    assert e < 3e-3


def test_headline_kernel_mix_at_B8_T1000_vs_oracle():
    """One view of the headline (BASELINE.json configs[1]: B 16 = two sub-batch views of B 8 x T 1000 x L 4: 500 panels, 512 fused
    attention jobs) with DEFAULT options, every element and every trace against the CPU oracle (latent_model.py:446-483), asserting
    from the profile report that the launches took the headline's own kernels: k_flash_proj8 (`flash_proj_T@q128`), k_mlp_rows
    (`mlp` / `mlp@fold`), k_ln_qkv<false, false> (`ln_qkv_T`), k_ln_qkv_attn4<true> (`attn_L_fused`).  (a) `forward` (per-sample t
    path: the unfolded row-owner MLP) at t = 0; (b) ONE Euler step of size 1 through `sample_euler` -- x1 - x0 = v(x0, t = 0), the
    shared-t path: the gate-folded MLP, the last layer's launch with the FinalLayer + Euler update as its tail (`mlp@fold+final`) --
    against the same oracle evaluation."""
    from oracle import mdgen_oracle as O
    from mdgen_amd.model import LatentMDGenModel
    B, T, L = 8, 1000, 4
    cfg, sd, kw, dkw = _fwd_case(B, T, L, 0, 4242, weights_seed=0)
    kw["t"] = torch.zeros(B)
    dkw["t"] = kw["t"].to(dkw["x"].device)
    ref, rtr = O.forward(sd, O.cfg_dict(cfg), return_trace=True, **kw)
    nl = cfg.num_layers
    m = LatentMDGenModel(cfg)
    m.load_state_dict(sd)
    out, tr, ran = _profiled_forward(m, dkw)
    rep = {k: rel_l2(tr[k].cpu(), rtr[k]) for k in ["ipa_out"] + [f"h{i}" for i in range(nl + 1)]}
    rep["out"] = rel_l2(out.cpu(), ref)
    print("B8 T1000 L4 forward, default dispatch:", {k: f"{v:.2e}" for k, v in rep.items()}, ran)
    assert torch.isfinite(out).all()
    for k, v in rep.items():
        assert v < TOL_FWD, (k, v)
    want = {"flash_proj_T@q128": nl, "mlp": nl, "ln_qkv_T": nl, "attn_L_fused": nl, "embed": 1, "final_euler": 1}
    for k, n in want.items():
        assert ran.get(k) == n, (k, ran)
    assert not any(k.split("@")[0] in ("flash_T", "proj_T", "proj_mlp", "projL_qkvT") for k in ran), ran
    assert (out.cpu() - ref).abs().max() < 0.1 * ref.abs().max()
    # (b) the sampler's path (t shared): eager with the profile on to read the classes, then the product's graph path
    ekw = _euler_kw(dkw)
    m.profile(True)
    x1 = m.sample_euler(dkw["x"], 1, use_graph=False, **ekw)
    ran = {k: v["count"] for k, v in m.profile_report().items()}
    m.profile(False)
    want = {"flash_proj_T@q128": nl, "mlp@fold": nl - 1, "mlp@fold+final": 1, "ln_qkv_T": nl, "attn_L_fused": nl, "fold_pack": 1}
    for k, n in want.items():
        assert ran.get(k) == n, (k, ran)
    assert "mlp" not in ran and "final_euler" not in ran, ran
    xg = m.sample_euler(dkw["x"], 1, use_graph=True, **ekw)
    assert torch.equal(x1, xg)
    v = (xg - dkw["x"]).cpu()
    e = rel_l2(v, ref)
    per_t = ((v - ref).double().pow(2).sum((0, 2, 3)) / ref.double().pow(2).sum((0, 2, 3))).sqrt()
    print(f"B8 T1000 L4 one Euler step (folded MLP): velocity vs oracle {e:.2e}, worst frame {float(per_t.max()):.2e}")
    assert e < TOL_FWD and float(per_t.max()) < 3 * TOL_FWD
    # (c) TWO Euler steps: the first step's last MLP launch also computes the second step's token embedding (`mlp@fold+final+embed`,
    # no k_embed for step 1); x2 - x0 against the oracle's two steps
    m.profile(True)
    x2 = m.sample_euler(dkw["x"], 2, use_graph=False, **ekw)
    ran = {k: v["count"] for k, v in m.profile_report().items()}
    m.profile(False)
    for k, n in {"mlp@fold": 2 * (nl - 1), "mlp@fold+final+embed": 1, "mlp@fold+final": 1, "embed": 1, "embed_base": 1}.items():
        assert ran.get(k) == n, (k, ran)
    assert torch.equal(m.sample_euler(dkw["x"], 2, use_graph=True, **ekw), x2)
    x2r = _oracle_two_euler_steps(sd, cfg, kw, v0=ref)
    e2 = rel_l2((x2 - dkw["x"]).cpu(), x2r - kw["x"])
    print(f"B8 T1000 L4 two Euler steps (second step's embedding from the first step's MLP tail): x2 - x0 vs oracle {e2:.2e}")
    assert e2 < TOL_FWD


def test_sample_euler_B16_graph_eager_and_stream_counts_agree():
    """The bench line's own call shape: `sample_euler` B 16 x T 1000 x L 4, 3 steps, `streams` 2 (two views of B 8: the kernels of
    test_headline_kernel_mix_at_B8_T1000_vs_oracle) -- graph replay == eager launches == one stream of B 16 bit for bit (both
    launch sizes select the same kernel forms and a workgroup's arithmetic does not depend on the launch it is in)."""
    from mdgen_amd.model import LatentMDGenModel
    cfg, sd, kw, dkw = _fwd_case(16, 1000, 4, 0, 99, weights_seed=0)
    ekw = _euler_kw(dkw)
    m = LatentMDGenModel(cfg)
    m.load_state_dict(sd)
    res = {}
    for ns in (2, 1):
        m.set_option("streams", ns)
        a = m.sample_euler(dkw["x"], 3, use_graph=False, **ekw)
        b = m.sample_euler(dkw["x"], 3, use_graph=True, **ekw)
        c = m.sample_euler(dkw["x"], 3, use_graph=True, **ekw)
        assert torch.isfinite(a).all() and torch.equal(a, b) and torch.equal(b, c), ns
        res[ns] = a
    assert torch.equal(res[1], res[2])


def _registry_cases():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dispatch_registry import CASES
    return [c for c in CASES if c["covered_by"] is None]


@pytest.mark.parametrize("case", _registry_cases(), ids=lambda c: c["name"])
def test_dispatch_registry_case_vs_oracle(case):
    """tests/dispatch_registry.py: every combination of kernel forms the dispatch logic can produce for the shapes the entry points
    are used with must be compared with the CPU oracle somewhere.  This test runs the registry's own cases (the combinations no
    older test reaches), DEFAULT options, 0xFF-filled workspace: (1) the kernel classes that ran == `mdgen_debug_dispatch_plan`'s
    prediction for the same call (the plan is the library's own orchestration code in a mode that skips every HIP call: here it is
    held against the real thing); (2) forward: every trace and the velocity against the oracle; euler: TWO Euler steps, x2 - x0
    against the oracle's two steps (the shared-t path: gate-folded MLP, FinalLayer as its tail, the second step's token
    embedding computed by the first step's last MLP launch)."""
    from oracle import mdgen_oracle as O
    from mdgen_amd._lib import dispatch_plan
    from mdgen_amd.model import LatentMDGenModel
    B, T, L, n_pad, tps = case["B"], case["T"], case["L"], case["n_pad"], case.get("tps", False)
    cfg, sd, kw, dkw = _fwd_case(B, T, L, n_pad, 3100 + 7 * T + L, weights_seed=11, tps=tps)
    euler = case["mode"] == "euler"
    if euler:
        kw["t"] = torch.zeros(B)
        dkw["t"] = kw["t"].to(dkw["x"].device)
    # (two-sided model: the library's relative-frame quaternions carry w >= 0; the oracle is told to do the same)
    cd = dict(O.cfg_dict(cfg), quat_sign="w_nonneg") if tps else O.cfg_dict(cfg)
    ref, rtr = O.forward(sd, cd, return_trace=True, **kw)
    m = LatentMDGenModel(cfg)
    m.load_state_dict(sd)
    copts = case.get("options") or {}
    for k, v in copts.items():
        m.set_option(k, v)
    m.forward(**dkw)   # (workspace of the shape)
    nl = cfg.num_layers
    info = None
    if not euler:
        out, tr, ran = _profiled_forward(m, dkw)
        info = m.context_info
        rep = {k: rel_l2(tr[k].cpu(), rtr[k]) for k in ["ipa_out"] + [f"h{i}" for i in range(nl + 1)]}
        rep["out"] = rel_l2(out.cpu(), ref)
        want = dispatch_plan(B, T, L, mode=3, tps=tps, ncu=info["ncu"], xcd_round_robin=bool(info["xcd_round_robin"]), options=copts)
    else:
        ekw = _euler_kw(dkw, tps)
        m.sample_euler(dkw["x"], 2, use_graph=False, **ekw)
        for ws in m._ws.values():
            ws.view(torch.uint8).fill_(0xFF)
        m.profile(True)
        x2 = m.sample_euler(dkw["x"], 2, use_graph=False, **ekw)
        ran = {k: v["count"] for k, v in m.profile_report().items()}
        info = m.context_info
        m.profile(False)
        xg = m.sample_euler(dkw["x"], 2, use_graph=True, **ekw)   # the product's path: graph, sub-batch streams
        d_ref = _oracle_two_euler_steps(sd, cfg, kw, v0=ref, cd=cd) - kw["x"]
        rep = {"x2 - x0": rel_l2((xg - dkw["x"]).cpu(), d_ref), "eager": rel_l2((x2 - dkw["x"]).cpu(), d_ref)}
        want = dispatch_plan(B, T, L, n_steps=2, mode=2, tps=tps, ncu=info["ncu"], xcd_round_robin=bool(info["xcd_round_robin"]), options=copts)
    planned = dict(want["prepare"])
    for vw in want["views"]:
        for k, n in vw["classes"].items():
            planned[k] = planned.get(k, 0) + n
    print(case["name"], {k: f"{e:.2e}" for k, e in rep.items()}, ran)
    assert ran == planned, (case["name"], ran, planned)
    for k, e in rep.items():
        assert e < TOL_FWD, (case["name"], k, e)


@pytest.mark.parametrize("shape", [(1, 256, 16, 49), (32, 4, 0, 49)], ids=["ATLAS_L256_S49", "shard_B32_L4_S49"])
def test_ipa_table_of_all_steps_vs_oracle(shape):
    """The IPA stack runs ONCE per call for all S steps (S * B * L rows per launch: ATLAS 12 544 rows = 196 panels, cfg-3's shard 6 272
    = 98 -- the eight-wave, unsplit forms `ipa.mlp@p8` / `ipa.ln_qkv@p8` that no single forward reaches).  Its output does not depend
    on T or x (latent_model.py:175-210: aatype, frames, t), so the table a T = 2 rollout leaves in the workspace is compared, step
    by step, with the oracle's `ipa_out` trace of a forward at that step's t; the report must name the forms the plan predicts."""
    from oracle import mdgen_oracle as O
    from mdgen_amd._lib import dispatch_plan
    from mdgen_amd.model import LatentMDGenModel
    B, L, n_pad, S = shape
    T = 2
    cfg, sd, kw, dkw = _fwd_case(B, T, L, n_pad, 5100 + L, weights_seed=13)
    m = LatentMDGenModel(cfg)
    m.load_state_dict(sd)
    ekw = _euler_kw(dkw)
    m.profile(True)
    x = m.sample_euler(dkw["x"], S, use_graph=False, **ekw)
    ran = {k: v["count"] for k, v in m.profile_report().items() if k.startswith("ipa.")}
    info = m.context_info
    m.profile(False)
    assert torch.isfinite(x).all()
    want = dispatch_plan(B, T, L, n_steps=S, mode=2, ncu=info["ncu"], xcd_round_robin=bool(info["xcd_round_robin"]))
    assert ran == {k: v for k, v in want["prepare"].items() if k.startswith("ipa.")}, (ran, want["prepare"])
    # ... and the bench shapes' IPA launches have the same forms (the signature depends on S * B * L only)
    full = dispatch_plan(B, 250 if L == 256 else 100, L, n_steps=S, mode=0, ncu=info["ncu"], xcd_round_robin=bool(info["xcd_round_robin"]))
    assert set(ran) == {k for k in full["prepare"] if k.startswith("ipa.")}, (ran, full["prepare"])
    lay = m.workspace_layout(B, T, L, S, True)
    ws = m._ws[(B, T, L, S, 1)]
    C_ = cfg.embed_dim
    tab = ws[lay.ipa_out:lay.ipa_out + S * B * L * C_ * 4].view(torch.float32).view(S, B, L, C_).cpu()
    tg = torch.linspace(0, 1, S + 1)[:S]
    worst = 0.0
    for s in range(0, S, 6):   # every sixth step (9 oracle evaluations)
        _, rtr = O.forward(sd, O.cfg_dict(cfg), return_trace=True, **dict(kw, t=torch.full((B,), float(tg[s]))))
        valid = kw["mask"][:, 0].bool()                       # padded residues' rows are not defined
        worst = max(worst, rel_l2(tab[s][valid], rtr["ipa_out"][valid]))
    print(shape, f"IPA table of {S} steps, forms {sorted(ran)}: worst step rel-L2 vs the oracle {worst:.2e}")
    assert worst < TOL_FWD


def test_rccl_paths_execute_with_one_rank(tmp_path):
    """The N > 1 code has never run on this pool's one-GPU boxes (round-5 verdict, item 8).  With ONE rank: (a) `bench.py
    --dist-selftest` -- RCCL process-group initialisation bound to the device, barrier, `max_over_ranks` / `sum_over_ranks` /
    `gather_over_ranks` on device tensors, teardown; (b) the data-parallel training step's worker with backend "nccl" in a group of one
    (reference: Lightning DDP = NCCL, train.py:46-68): the construction-time broadcasts and every gradient bucket's all-reduce through
    `GradBucketer.launch_on_events` are issued (identities) -- and the parameters after two steps equal the run without a process
    group bit for bit."""
    import json
    import socket
    import subprocess
    _cuda()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dist-selftest"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    print("bench.py --dist-selftest:", res)
    assert res["dist_selftest"] == "ok" and res["backend"] == "nccl" and res["world"] == 1
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    worker = os.path.join(ROOT, "tests", "ddp_worker.py")
    env = dict(env, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    one, rc = str(tmp_path / "plain.pt"), str(tmp_path / "rccl.pt")
    subprocess.run([sys.executable, worker, one], check=True, timeout=900, env=env)
    subprocess.run([sys.executable, worker, rc], check=True, timeout=900, env=dict(env, DDP_BACKEND="nccl", DDP_SELFTEST="1"))
    a, b = torch.load(one), torch.load(rc)
    assert b["backend"] == "nccl" and b["reduces"] and not a["reduces"]
    assert torch.equal(a["params"], b["params"]) and torch.equal(a["ema"], b["ema"])


def test_small_split_hand_over_stress_200_graph_replays():
    """The cross-workgroup hand-over of k_mlp8<., 3> (round 5: a panel's hidden chunks over three workgroups, the last arriver sums the
    partials; round 6: the arrival is a RELEASE at agent scope, the read side an acquire -- correct wherever the workgroups run) under
    load: the B = 1 rollout at S = 49 (what `sim_inference.py` runs: 490 split launches per call) replayed 200 times from its hipGraph,
    every result bit-equal to the first; the report's "@context" entry says how many split launches a profiled call makes and what
    the placement probe found."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict
    from mdgen_amd.wrapper import NewMDGenWrapper
    import bench
    dev = _cuda()
    B, T, L, abs_pos, n_pad = bench.WORKLOADS["tetrapeptide_fwdsim_crop4_T1000_B1"]
    cfg = ModelConfig(crop=L, num_frames=T, abs_pos_emb=abs_pos, sim_condition=True)
    w = NewMDGenWrapper(cfg, device=dev)
    w.model.load_state_dict(synth_state_dict(cfg, 0))
    batch = bench.synth_batch(B, T, L, n_pad, dev, seed=100)
    zs = torch.randn(1, 1000, 4, 21, generator=torch.Generator().manual_seed(137)).to(dev)
    first, _ = w.inference(batch, zs=zs, num_steps=49, use_graph=True)
    assert torch.isfinite(first).all()
    for i in range(200):
        again, _ = w.inference(batch, zs=zs, num_steps=49, use_graph=True)
        assert torch.equal(again, first), i
    w.model.profile_report()   # (resets the split-launch counter: it also counted the launches the graph capture enqueued)
    w.model.profile(True)
    w.inference(batch, zs=zs, num_steps=49, use_graph=False)
    ran = {k: v["count"] for k, v in w.model.profile_report().items()}
    info = w.model.context_info
    w.model.profile(False)
    print("B = 1, S = 49: split launches", info, {k: v for k, v in ran.items() if "x3" in k or "x2" in k})
    assert info["xcd_round_robin"] in (0, 1) and info["ncu"] > 0
    if info["xcd_round_robin"]:   # the split form is only picked where the placement rule holds
        assert info["count"] == ran.get("proj_mlp@p8x3", 0) + ran.get("mlp@p8x3", 0) + ran.get("ipa.mlp@p8x3", 0) == 49 * 5 + 5
    else:
        assert info["count"] == 0 and not any("x3" in k for k in ran)
