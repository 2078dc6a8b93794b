"""Generate golden vectors by RUNNING THE REFERENCE (this container only).

    PYTHONPATH=oracle/shims:/root/reference:. MODEL_DIR=/tmp/mdl python oracle/gen_golden.py

The reference (pure Python, /root/reference) is imported on CPU with the stand-in modules of
`oracle/shims/`; its outputs on seeded inputs are written as small .npz fixtures under
`tests/golden/`.  Fixtures hold DATA only (inputs + expected outputs + the weight seed and a
weight checksum); weights are regenerated from the seed by `mdgen_amd.synthetic` on both sides.
Nothing here runs on the GPU box.
"""
import argparse
import os
import sys
from functools import partial

import numpy as np
import torch

from mdgen.wrapper import NewMDGenWrapper
from mdgen.rigid_utils import Rigid, Rotation
from mdgen import geometry as G
from mdgen.utils import get_offsets

from mdgen_amd.config import ModelConfig
from mdgen_amd.synthetic import synth_state_dict, synth_forward_inputs, tensor_checksum

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_grad_enabled(False)
torch.set_num_threads(8)


def ref_args(cfg: ModelConfig):
    a = dict(
        ckpt=None, validate=False, num_workers=0, epochs=1, overfit=False, overfit_peptide=None,
        overfit_frame=False, train_batches=None, val_batches=None, val_repeat=1, inference_batches=0,
        batch_size=1, val_freq=None, val_epoch_freq=1, no_validate=False, designability_freq=1,
        print_freq=100, ckpt_freq=1, wandb=False, run_name="x", accumulate_grad=1, grad_clip=1.0,
        check_grad=False, grad_checkpointing=False, adamW=False, ema=False, ema_decay=0.999, lr=1e-4,
        precision="32-true", train_split="", val_split="", data_dir="", num_frames=cfg.num_frames,
        crop=cfg.crop, suffix="", atlas=False, copy_frames=False, no_pad=False, short_md=False,
        design_key_frames=False, no_aa_emb=False, no_torsion=False, no_design_torsion=False,
        supervise_no_torsions=False, supervise_all_torsions=False, no_offsets=False, no_frames=False,
        hyena=False, no_rope=False, dropout=0.0, scale_factor=1.0, interleave_ipa=False,
        prepend_ipa=True, oracle=False, num_layers=cfg.num_layers, embed_dim=cfg.embed_dim,
        mha_heads=cfg.mha_heads, ipa_heads=cfg.ipa_heads, ipa_head_dim=cfg.ipa_head_dim,
        ipa_qk=cfg.ipa_qk, ipa_v=cfg.ipa_v, time_multiplier=cfg.time_multiplier,
        abs_pos_emb=cfg.abs_pos_emb, abs_time_emb=False, path_type="GVP", prediction="velocity",
        sampling_method="euler", alpha_max=8, discrete_loss_weight=0.5, dirichlet_flow_temp=1.0,
        allow_nan_cfactor=False, tps_condition=cfg.tps_condition, design=False, design_from_traj=False,
        sim_condition=cfg.sim_condition, inpainting=False, dynamic_mpnn=False, mpnn=False,
        frame_interval=None, cond_interval=None)
    return argparse.Namespace(**a)


def build(cfg, seed, **overrides):
    a = ref_args(cfg)
    for k, v in overrides.items():
        setattr(a, k, v)
    m = NewMDGenWrapper(a).eval()
    sd = synth_state_dict(cfg, seed)
    m.model.load_state_dict(sd)
    chk = np.array([sum(float(v.double().sum()) for v in sd.values()),
                    sum(float(v.double().abs().sum()) for v in sd.values())])
    return m, chk


def rand_rot(g, *shape):
    q = torch.randn(*shape, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    return Rotation(quats=q).get_rot_mats()


def synth_structure(g, B, T, L, seqres):
    """Self-consistent random trajectory: frames+torsions -> atom14 -> (sim_inference.get_batch)."""
    rots0 = rand_rot(g, B, 1, L)
    trans0 = torch.cumsum(2.2 * torch.randn(B, 1, L, 3, generator=g), dim=2)
    # per-frame perturbation so that offsets are non-trivial rotations
    drot = rand_rot(g, B, T, L)
    ident = torch.eye(3).expand(B, T, L, 3, 3)
    w = torch.linspace(0, 1, T)[None, :, None, None, None]
    mix = ident * (1 - 0.6 * w) + drot * 0.6 * w
    u, _, vh = torch.linalg.svd(mix)
    dR = u @ vh
    dR = dR * torch.sign(torch.linalg.det(dR))[..., None, None]
    rots = torch.einsum("btlij,btljk->btlik", rots0.expand(B, T, L, 3, 3), dR)
    trans = trans0 + 0.8 * torch.randn(B, T, L, 3, generator=g) * w[..., 0]
    ang = 2 * np.pi * torch.rand(B, T, L, 7, generator=g)
    tors = torch.stack([torch.sin(ang), torch.cos(ang)], -1)
    frames = Rigid(trans=trans, rots=Rotation(rot_mats=rots))
    atom14 = G.frames_torsions_to_atom14(frames, tors, seqres[:, None].expand(B, T, L))
    return atom14


def get_batch_like_sim_inference(atom14, seqres):
    """sim_inference.get_batch (sim_inference.py:32-59) on torch tensors, per batch element, collated."""
    items = []
    for b in range(atom14.shape[0]):
        arr = atom14[b]                                  # [F,L,14,3]
        frames = G.atom14_to_frames(arr)
        atom37 = G.atom14_to_atom37(arr, seqres[b][None]).float()
        tors, tmask = G.atom37_to_torsions(atom37, seqres[b][None])
        items.append(dict(torsions=tors, torsion_mask=tmask[0], trans=frames._trans,
                          rots=frames._rots._rot_mats, seqres=seqres[b], mask=torch.ones(len(seqres[b]))))
    return {k: torch.stack([it[k] for it in items]) for k in items[0]}


def save(name, **kw):
    arrs = {}
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        arrs[k] = np.asarray(v)
    p = os.path.join(OUT, name + ".npz")
    np.savez_compressed(p, **arrs)
    print(f"{name:28s} {os.path.getsize(p) / 1024:8.1f} KB")


def trace_forward(model, **kw):
    """Run LatentMDGenModel.forward capturing the IPA-stack output and each trunk layer's output."""
    tr = {}
    hooks = []
    lm = model.model
    orig = lm.run_ipa

    def run_ipa(*a, **k):
        o = orig(*a, **k)
        tr["ipa_out"] = o
        return o
    lm.run_ipa = run_ipa
    for i, layer in enumerate(lm.layers):
        hooks.append(layer.register_forward_hook(lambda m, inp, out, i=i: tr.__setitem__(f"h{i + 1}", out)))
    hooks.append(lm.layers[0].register_forward_pre_hook(lambda m, inp: tr.__setitem__("h0", inp[0])))
    out = lm.forward(**kw)
    for h in hooks:
        h.remove()
    lm.run_ipa = orig
    return out, tr


def gen_forward(name, cfg, seed, B, T, L, n_pad, data_seed, keep=("ipa_out", "h0")):
    g = torch.Generator().manual_seed(data_seed)
    m, chk = build(cfg, seed)
    D = cfg.latent_dim
    x = torch.randn(B, T, L, D, generator=g)
    t = torch.rand(B, generator=g)
    mask = torch.ones(B, L)
    if n_pad:
        mask[-1, L - n_pad:] = 0
    mask_btl = mask[:, None].expand(B, T, L).contiguous()
    aatype = torch.randint(0, 20, (B, L), generator=g)
    sR, eR = rand_rot(g, B, L), rand_rot(g, B, L)
    st = torch.cumsum(2.2 * torch.randn(B, L, 3, generator=g), 1)
    et = st + torch.randn(B, L, 3, generator=g)
    cm = torch.zeros(B, T, L, dtype=torch.long)
    cm[:, 0] = 1
    if cfg.tps_condition:
        cm[:, -1] = 1
    lat = torch.randn(B, T, L, D, generator=g)
    x_cond = torch.where(cm.unsqueeze(-1).bool(), lat, 0.0)
    out, tr = trace_forward(
        m, x=x, t=t, mask=mask_btl,
        start_frames=Rigid(trans=st, rots=Rotation(rot_mats=sR)),
        end_frames=Rigid(trans=et, rots=Rotation(rot_mats=eR)),
        x_cond=x_cond, x_cond_mask=cm, aatype=aatype)
    nl = cfg.num_layers
    extra = {k: tr[k] for k in keep}
    extra[f"h{nl}"] = tr[f"h{nl}"]
    if cfg.tps_condition:
        # the relative-frame inputs exactly as the reference's forward computes them (latent_model.py:194-195): the quaternion
        # SIGN is whatever torch.linalg.eigh returned here (rigid_utils.py:191-210); a caller hands them to the library as `rel7`
        sf = Rigid(trans=st, rots=Rotation(rot_mats=sR))
        ef = Rigid(trans=et, rots=Rotation(rot_mats=eR))
        extra["rel7"] = torch.stack([sf.invert().compose(ef).to_tensor_7(), ef.invert().compose(sf).to_tensor_7()])
    save(name, cfg=str(cfg.to_dict()), seed=seed, weight_checksum=chk, x=x, t=t, mask=mask_btl,
         start_rot=sR, start_trans=st, end_rot=eR, end_trans=et, x_cond=x_cond, x_cond_mask=cm,
         aatype=aatype, out=out, **extra)


def gen_forward_big(name, cfg, seed, B, T, L, n_pad, data_seed, sub):
    """Full-size configuration (e.g. cfg-4: B1 T250 L256): inputs are NOT stored (they are regenerated from
    `data_seed` by mdgen_amd.synthetic.synth_forward_inputs on both sides, with a checksum); outputs and traces
    of the reference are stored SUB-SAMPLED: `out` at frames [::sub[0]], residues [::sub[1]]; the 384-wide traces
    (h0, h_last) at a 5x / 4x coarser stride, ipa_out at residues [::sub[1]]."""
    m, chk = build(cfg, seed)
    inp = synth_forward_inputs(cfg, B, T, L, n_pad, data_seed)
    out, tr = trace_forward(
        m, x=inp["x"], t=inp["t"], mask=inp["mask"],
        start_frames=Rigid(trans=inp["start_trans"], rots=Rotation(rot_mats=inp["start_rot"])),
        end_frames=Rigid(trans=inp["end_trans"], rots=Rotation(rot_mats=inp["end_rot"])),
        x_cond=inp["x_cond"], x_cond_mask=inp["x_cond_mask"], aatype=inp["aatype"])
    nl = cfg.num_layers
    st, sl = sub
    ht, hl = 5 * st, 4 * sl
    save(name, cfg=str(cfg.to_dict()), seed=seed, weight_checksum=chk, shape=np.array([B, T, L, n_pad]),
         data_seed=data_seed, input_checksum=tensor_checksum(inp), sub=np.array(sub), sub_h=np.array([ht, hl]),
         out=out[:, ::st, ::sl], ipa_out=tr["ipa_out"][:, ::sl], h0=tr["h0"][:, ::ht, ::hl],
         **{f"h{nl}": tr[f"h{nl}"][:, ::ht, ::hl]},
         norms=np.array([float(out.double().norm()), float(tr["ipa_out"].double().norm()),
                         float(tr["h0"].double().norm()), float(tr[f"h{nl}"].double().norm())]))


def gen_prep(name, cfg, B, T, L, data_seed, cond_interval=None):
    g = torch.Generator().manual_seed(data_seed)
    m, _ = build(ModelConfig(embed_dim=48, mha_heads=2, num_layers=1, crop=L, num_frames=T,
                             abs_pos_emb=cfg.abs_pos_emb, sim_condition=cfg.sim_condition,
                             tps_condition=cfg.tps_condition), 0, cond_interval=cond_interval)
    seqres = torch.randint(0, 20, (B, L), generator=g)
    atom14 = synth_structure(g, B, T, L, seqres)
    batch = get_batch_like_sim_inference(atom14, seqres)
    batch["mask"][-1, -1] = 0
    prep = m.prep_batch(batch)
    kw = prep["model_kwargs"]
    save(name, cfg=str(cfg.to_dict()), atom14=atom14, cond_interval=np.array(cond_interval or 0),
         **{"in_" + k: v for k, v in batch.items()},
         latents=prep["latents"], loss_mask=prep["loss_mask"].contiguous(), x_cond=kw["x_cond"],
         x_cond_mask=kw["x_cond_mask"], mask=kw["mask"].contiguous(), aatype=kw["aatype"],
         start_rot=kw["start_frames"]._rots.get_rot_mats(), start_trans=kw["start_frames"]._trans,
         end_rot=kw["end_frames"]._rots.get_rot_mats(), end_trans=kw["end_frames"]._trans)


def gen_inference(name, cfg, seed, B, T, L, steps, data_seed, n_blocks=1, slim=False):
    """End-to-end inference() (wrapper.py:405-484) with explicit zs, S steps, plus the rollout glue
    (sim_inference.py:61-98) chaining `n_blocks` blocks."""
    g = torch.Generator().manual_seed(data_seed)
    m, chk = build(cfg, seed)
    seqres = torch.randint(0, 20, (B, L), generator=g)
    atom14_0 = synth_structure(g, B, 1, L, seqres)
    batch = get_batch_like_sim_inference(atom14_0, seqres)
    out = dict(cfg=str(cfg.to_dict()), seed=seed, weight_checksum=chk, steps=np.array(steps),
               atom14_init=atom14_0, **{"in_" + k: v for k, v in batch.items()})
    for S in steps:
        cur = dict(batch)
        for blk in range(n_blocks):
            ex = dict(cur)
            ex["torsions"] = cur["torsions"].expand(-1, T, -1, -1, -1)
            ex["trans"] = cur["trans"].expand(-1, T, -1, -1)
            ex["rots"] = cur["rots"].expand(-1, T, -1, -1, -1)
            zs = torch.randn(B, T, L, cfg.latent_dim, generator=torch.Generator().manual_seed(137 + blk))
            prep = m.prep_batch(ex)
            sample_fn = m.transport_sampler.sample_ode(sampling_method="euler", num_steps=S + 1)
            samples = sample_fn(zs, partial(m.model.forward_inference, **prep["model_kwargs"]))[-1]
            # the reference's own inference() (wrapper.py:405-484) with its device randn replaced by
            # the explicit zs and its hard-coded 50-point grid replaced by S+1 points
            orig_sample_ode = m.transport_sampler.sample_ode
            m.transport_sampler.sample_ode = lambda **k: orig_sample_ode(sampling_method="euler", num_steps=S + 1)
            orig_randn = torch.randn
            torch.randn = lambda *a, **k: zs.clone()
            try:
                atom14, aa = m.inference(ex)
            finally:
                torch.randn = orig_randn
                del m.transport_sampler.sample_ode
            out[f"S{S}_b{blk}_zs"] = zs
            if not slim:
                out[f"S{S}_b{blk}_samples"] = samples
            out[f"S{S}_b{blk}_atom14"] = atom14
            # rollout glue (sim_inference.py:91-96)
            fr = G.atom14_to_frames(atom14[:, -1])
            a37 = G.atom14_to_atom37(atom14[:, -1], seqres)
            tors, _ = G.atom37_to_torsions(a37, seqres)
            cur = dict(cur)
            cur["trans"] = fr._trans[:, None]
            cur["rots"] = fr._rots._rot_mats[:, None]
            cur["torsions"] = tors[:, None]
            if not slim:
                out[f"S{S}_b{blk}_next_trans"] = cur["trans"]
                out[f"S{S}_b{blk}_next_rots"] = cur["rots"]
                out[f"S{S}_b{blk}_next_torsions"] = cur["torsions"]
    save(name, **out)


def gen_inference_big(name, cfg, seed, B, T, L, S, data_seed, sub_t):
    """One block of the reference's own inference() (wrapper.py:405-484) at a BASELINE size (e.g. cfg-2's regime: 1000
    frames, S = 49 Euler steps).  zs is NOT stored: torch.randn(B, T, L, D, generator=manual_seed(137)), checksummed;
    atom14 is stored at frames [::sub_t]."""
    g = torch.Generator().manual_seed(data_seed)
    m, chk = build(cfg, seed)
    seqres = torch.randint(0, 20, (B, L), generator=g)
    atom14_0 = synth_structure(g, B, 1, L, seqres)
    batch = get_batch_like_sim_inference(atom14_0, seqres)
    ex = dict(batch)
    ex["torsions"] = batch["torsions"].expand(-1, T, -1, -1, -1)
    ex["trans"] = batch["trans"].expand(-1, T, -1, -1)
    ex["rots"] = batch["rots"].expand(-1, T, -1, -1, -1)
    zs = torch.randn(B, T, L, cfg.latent_dim, generator=torch.Generator().manual_seed(137))
    orig_sample_ode = m.transport_sampler.sample_ode
    m.transport_sampler.sample_ode = lambda **k: orig_sample_ode(sampling_method="euler", num_steps=S + 1)
    orig_randn = torch.randn
    torch.randn = lambda *a, **k: zs.clone()
    try:
        atom14, aa = m.inference(ex)
    finally:
        torch.randn = orig_randn
        del m.transport_sampler.sample_ode
    save(name, cfg=str(cfg.to_dict()), seed=seed, weight_checksum=chk, S=np.array(S), sub_t=np.array(sub_t),
         shape=np.array([B, T, L]), zs_checksum=tensor_checksum({"zs": zs}), atom14_init=atom14_0,
         **{"in_" + k: v for k, v in batch.items()}, atom14=atom14[:, ::sub_t])


def gen_rigid(name, data_seed):
    g = torch.Generator().manual_seed(data_seed)
    n = 64
    R1, R2 = rand_rot(g, n), rand_rot(g, n)
    t1, t2 = torch.randn(n, 3, generator=g) * 5, torch.randn(n, 3, generator=g) * 5
    p = torch.randn(n, 3, generator=g) * 3
    A = Rigid(trans=t1, rots=Rotation(rot_mats=R1))
    Bq = Rigid(trans=t2, rots=Rotation(rot_mats=R2))
    comp = A.compose(Bq)
    inv = A.invert()
    q = A.to_tensor_7()
    q7 = torch.randn(n, 7, generator=g)
    f7 = Rigid.from_tensor_7(q7, normalize_quats=True)
    off = get_offsets(A[None, :1, None], Bq[None, :, None])
    p3a, p3b, p3c = (torch.randn(n, 3, generator=g) * 2 for _ in range(3))
    f3 = Rigid.from_3_points(p3a, p3b, p3c)
    save(name, R1=R1, t1=t1, R2=R2, t2=t2, p=p,
         comp_R=comp.get_rots().get_rot_mats(), comp_t=comp.get_trans(),
         inv_R=inv.get_rots().get_rot_mats(), inv_t=inv.get_trans(),
         apply=A.apply(p), invert_apply=A.invert_apply(p), tensor7=q,
         q7=q7, from7_R=f7.get_rots().get_rot_mats(), from7_t=f7.get_trans(),
         offsets=off, p3a=p3a, p3b=p3b, p3c=p3c, f3_R=f3.get_rots().get_rot_mats(), f3_t=f3.get_trans())


def gen_rigid_views(name, data_seed):
    """SURVEY rows r-7 / r-8 (view-level `Rigid` / `Rotation` operations, no arithmetic beyond a mask product) evaluated by the
    REFERENCE's classes (rigid_utils.py:820-862, 892-942, 1122-1141, 1220-1261) on a (2, 5) batch of frames."""
    g = torch.Generator().manual_seed(data_seed)
    R = rand_rot(g, 2, 5)
    t = torch.randn(2, 5, 3, generator=g)
    q = torch.randn(2, 5, 4, generator=g)
    mk = torch.tensor([[1., 0, 1, 0, 1], [0, 1, 0, 1, 0]])
    r = Rigid(Rotation(rot_mats=R), t)
    rm = lambda x: x.get_rots().get_rot_mats()
    out = dict(R=R, t=t, q=q, mk=mk)
    a, b = Rigid(Rotation(rot_mats=R), None), Rigid(None, t)
    out.update(fill_t=a.get_trans(), fill_R=rm(b))
    i = Rigid.identity((4, 3), fmt="rot_mat")
    out.update(ident_R=rm(i), ident_t=i.get_trans())
    for key, v in (("idx_col", r[:, 0:1]), ("idx_row", r[1]), ("idx_ell", r[..., 3]), ("idx_none", r[..., None]),
                   ("unsq_last", r.unsqueeze(-1)), ("unsq_first", r.unsqueeze(0)),
                   ("cat1", Rigid.cat([r, r[:, :2]], dim=1)), ("cat_last", Rigid.cat([r, r], dim=-1)), ("mul", r * mk),
                   ("map_sum", (Rigid(Rotation(rot_mats=R), t) * mk).map_tensor_fn(lambda x: torch.sum(x, dim=-1)))):
        out[key + "_R"], out[key + "_t"] = rm(v), v.get_trans()
    out["rotcat_R"] = Rotation.cat([Rotation(rot_mats=R), Rotation(rot_mats=R)], dim=0).get_rot_mats()
    rq = Rotation(quats=q, normalize_quats=True)
    out.update(quat_norm=rq.get_quats(), quat_idx=rq[0].get_quats(), quat_unsq=rq.unsqueeze(1).get_quats(),
               quat_raw=Rotation(quats=q, normalize_quats=False).get_quats())
    T4 = torch.zeros(2, 5, 4, 4)
    T4[..., :3, :3], T4[..., :3, 3], T4[..., 3, 3] = R, t, 1.0
    f = Rigid.from_tensor_4x4(T4)
    out.update(T4=T4, f4_R=rm(f), f4_t=f.get_trans())
    f7 = Rigid.from_tensor_7(torch.cat([q, t], -1), normalize_quats=True)
    out.update(f7_q=f7.get_rots().get_quats(), f7_t=f7.get_trans())
    save(name, **out)


def gen_geometry(name, data_seed):
    g = torch.Generator().manual_seed(data_seed)
    B, T, L = 2, 3, 20
    seqres = torch.arange(20)[None].expand(B, L).contiguous()      # every residue type once
    atom14 = synth_structure(g, B, T, L, seqres)
    aat = seqres[:, None].expand(B, T, L)
    fr = G.atom14_to_frames(atom14.reshape(B * T, L, 14, 3))
    a37 = G.atom14_to_atom37(atom14, aat)
    tors, tmask = G.atom37_to_torsions(a37, aat)
    back = G.frames_torsions_to_atom14(
        Rigid(trans=fr._trans.reshape(B, T, L, 3), rots=Rotation(rot_mats=fr._rots._rot_mats.reshape(B, T, L, 3, 3))),
        tors, aat)
    save(name, seqres=seqres, atom14=atom14, frames_R=fr._rots._rot_mats.reshape(B, T, L, 3, 3),
         frames_t=fr._trans.reshape(B, T, L, 3), atom37=a37, torsions=tors, torsion_mask=tmask, atom14_back=back)


if __name__ == "__main__":
    tiny = dict(embed_dim=48, mha_heads=2, num_layers=2)
    JOBS = {
        "rigid_ops": lambda: gen_rigid("rigid_ops", 11),
        "geometry": lambda: gen_geometry("geometry", 12),
        "rigid_views": lambda: gen_rigid_views("rigid_views", 13),
        "fwd_tiny_sim": lambda: gen_forward("fwd_tiny_sim", ModelConfig(crop=5, num_frames=6, **tiny), 3, B=2, T=6, L=5,
                                            n_pad=2, data_seed=21, keep=("ipa_out", "h0", "h1")),
        "fwd_tiny_tps": lambda: gen_forward("fwd_tiny_tps", ModelConfig(crop=5, num_frames=6, sim_condition=False,
                                                                        tps_condition=True, **tiny),
                                            4, B=2, T=6, L=5, n_pad=1, data_seed=22, keep=("ipa_out", "h0", "h1")),
        "fwd_full_sim": lambda: gen_forward("fwd_full_sim", ModelConfig.forward_sim(num_frames=6, crop=5), 5, B=2, T=6,
                                            L=5, n_pad=2, data_seed=23),
        "fwd_full_pep": lambda: gen_forward("fwd_full_pep", ModelConfig.forward_sim(num_frames=40, crop=4), 5, B=1, T=40,
                                            L=4, n_pad=0, data_seed=26),
        "fwd_full_atlas": lambda: gen_forward("fwd_full_atlas", ModelConfig.atlas(num_frames=8, crop=40), 6, B=1, T=8,
                                              L=40, n_pad=6, data_seed=24),
        "fwd_full_tps": lambda: gen_forward("fwd_full_tps", ModelConfig.tps(num_frames=6, crop=4), 7, B=2, T=6, L=4,
                                            n_pad=0, data_seed=25),
        # BASELINE.json configs[3] at its full size: ATLAS crop 256 x 250 frames, B 1, 16 padded residues
        "fwd_cfg4_atlas_full": lambda: gen_forward_big("fwd_cfg4_atlas_full", ModelConfig.atlas(num_frames=250, crop=256),
                                                       6, B=1, T=250, L=256, n_pad=16, data_seed=27, sub=(10, 8)),
        # BASELINE.json configs[1]'s regime: tetrapeptide, 1000 frames (1001 temporal keys = 32 key tiles, RoPE positions up
        # to 999), B 2 with distinct t; configs[0]'s exact forward shape (B1 T100 L4) as well
        "fwd_cfg2_T1000": lambda: gen_forward_big("fwd_cfg2_T1000", ModelConfig.forward_sim(num_frames=1000, crop=4),
                                                  5, B=2, T=1000, L=4, n_pad=0, data_seed=28, sub=(8, 1)),
        "fwd_cfg1_T100": lambda: gen_forward_big("fwd_cfg1_T100", ModelConfig.forward_sim(num_frames=100, crop=4),
                                                 5, B=1, T=100, L=4, n_pad=0, data_seed=29, sub=(1, 1)),
        "prep_sim": lambda: gen_prep("prep_sim", ModelConfig.forward_sim(num_frames=6, crop=5), B=2, T=6, L=5, data_seed=31),
        # --cond_interval 3 (wrapper.py:343-344; the upsampling models are sim_condition networks with every k-th frame given)
        "prep_sim_interval": lambda: gen_prep("prep_sim_interval", ModelConfig.forward_sim(num_frames=8, crop=5), B=2, T=8, L=5,
                                              data_seed=33, cond_interval=3),
        "prep_tps": lambda: gen_prep("prep_tps", ModelConfig.tps(num_frames=6, crop=5), B=2, T=6, L=5, data_seed=32),
        # S = 49 is the reference's hard-coded step count (wrapper.py:441-442) and the product default
        "inference_sim": lambda: gen_inference("inference_sim", ModelConfig.forward_sim(num_frames=12, crop=4), 8, B=2,
                                               T=12, L=4, steps=[1, 10, 49], data_seed=41, n_blocks=2),
        # the README run chains 10 blocks (sim_inference.py:110-113, README.md:72 --num_rollouts 10): error growth per block
        "rollout10_sim": lambda: gen_inference("rollout10_sim", ModelConfig.forward_sim(num_frames=8, crop=4), 8, B=1,
                                               T=8, L=4, steps=[10], data_seed=43, n_blocks=10, slim=True),
        # BASELINE.json configs[1]'s regime end to end (B 1 of the 16): 1000 frames, the reference's 49 Euler steps
        "inference_cfg2_T1000": lambda: gen_inference_big("inference_cfg2_T1000", ModelConfig.forward_sim(num_frames=1000, crop=4),
                                                          8, B=1, T=1000, L=4, S=49, data_seed=45, sub_t=4),
        # BASELINE.json configs[0]'s exact shape: single tetrapeptide, 100 frames, 10 Euler steps
        "inference_cfg1": lambda: gen_inference_big("inference_cfg1", ModelConfig.forward_sim(num_frames=100, crop=4),
                                                    8, B=1, T=100, L=4, S=10, data_seed=46, sub_t=1),
        "inference_tiny": lambda: gen_inference("inference_tiny", ModelConfig(crop=4, num_frames=10, **tiny), 9, B=1,
                                                T=10, L=4, steps=[10, 49], data_seed=42, n_blocks=1),
    }
    for n in (sys.argv[1:] or list(JOBS)):
        JOBS[n]()
