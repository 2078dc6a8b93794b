"""CPU-side checks of the host code: the C-ABI library loads and exports every symbol declared in
include/mdgen_amd.h, config mapping, seeded weights, loud failure without a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "mdgen_amd.h")).read()
    return sorted(set(re.findall(r"\b(mdgen_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    so = os.path.join(ROOT, "mdgen_amd", "libmdgen_amd.so")
    if not os.path.exists(so):
        from mdgen_amd.build import build
        build(verbose=False)
    import mdgen_amd._lib as L
    lib = ctypes.CDLL(so)
    names = _declared_symbols()
    assert len(names) >= 23
    for n in names:
        assert hasattr(lib, n), n
    assert set(L.EXPORTS) == set(names)
    assert L.lib.mdgen_abi_version() == L.ABI_VERSION == 7   # include/mdgen_amd.h MDGEN_ABI_VERSION; _lib refuses a mismatch


def test_public_struct_layouts_agree_between_header_python_mirror_and_the_integration_stub():
    """mdgen_ws_layout is written by the library into caller memory: the header's field list, `_lib.WsLayout` and the ctypes stub a
    maintainer would copy from INTEGRATION.md must have the same number of size_t fields (a shorter buffer is a heap overrun: the
    round-5 `split` field crashed the stub's GPU test before this check existed), and the stub must pin the ABI version."""
    import re
    import mdgen_amd._lib as L
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "mdgen_amd.h")).read()
    body = re.search(r"typedef struct mdgen_ws_layout \{(.*?)\} mdgen_ws_layout;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [n.strip() for decl in re.findall(r"size_t\s+([^;]+);", body) for n in decl.split(",")]
    assert fields == [n for n, _ in L.WsLayout._fields_], (fields, L.WsLayout._fields_)
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    n_stub = int(re.search(r"lay = \(C\.c_size_t \* (\d+)\)\(\)", doc).group(1))
    assert n_stub == len(fields)
    v = int(re.search(r"#define MDGEN_ABI_VERSION (\d+)", hdr).group(1))
    assert v == L.ABI_VERSION and f"assert lib.mdgen_abi_version() == {v}" in doc


def test_argument_validation_without_gpu():
    import mdgen_amd._lib as L
    assert L.lib.mdgen_rigid_compose(4, None, None, None, None, None, None, None) == -1
    assert b"null" in L.lib.mdgen_last_error()
    d = L.ModelDesc(128, 16, 5, 21, 4, 32, 8, 8, 1, 4, 0, 100.0)
    ctx = ctypes.c_void_p()
    assert L.lib.mdgen_ctx_create(ctypes.byref(ctx), ctypes.byref(d)) == -2       # embed_dim must be 384


def test_no_cpu_fallback():
    """The product path must fail loudly, not fall back, when there is no GPU / for CPU tensors."""
    from mdgen_amd._lib import MdgenError
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.rigid_utils import Rigid, Rotation
    r = Rigid(Rotation(rot_mats=torch.eye(3).expand(2, 3, 3)), torch.zeros(2, 3))
    with pytest.raises(MdgenError):
        r.compose(r)
    if not torch.cuda.is_available():
        from mdgen_amd.model import LatentMDGenModel
        with pytest.raises(MdgenError):
            LatentMDGenModel(ModelConfig.forward_sim())


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under mdgen_amd/ may import or execute it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mdgen_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, re.M), f
                assert "mdgen_oracle" not in src, f


def test_library_never_aborts_the_host_process_or_reads_the_environment():
    """A shared library that a Python trainer has loaded must report through its error returns: no `abort()` / `exit()` in the
    native sources (a launcher that refuses a shape leaves a message that the entry point's LAUNCHCHK turns into `fail(-7)`),
    no environment variables (behaviour is set through `mdgen_ctx_set_option` only), and the per-call operand mode of the
    training kernels is thread-local rather than a process-wide toggle."""
    csrc = os.path.join(ROOT, "mdgen_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h", ".inc")):
            continue
        src = re.sub(r"//[^\n]*", "", open(os.path.join(csrc, f)).read())
        assert not re.search(r"\b(std::)?(abort|exit|_Exit|quick_exit)\s*\(", src), f
        assert "getenv" not in src, f
    k = open(os.path.join(csrc, "k_fp32.hip")).read()
    assert "thread_local int g_k32_bf16_operands" in k and "thread_local const char* g_k32_launch_error" in k
    assert "k32_take_launch_error()" in open(os.path.join(csrc, "api.hip")).read()


def test_config_from_reference_namespace():
    import argparse
    from mdgen_amd.config import ModelConfig
    ns = argparse.Namespace(embed_dim=384, num_layers=5, mha_heads=16, crop=256, num_frames=250, abs_pos_emb=False,
                            sim_condition=True, tps_condition=False, prepend_ipa=True, lr=1e-4, batch_size=8)
    c = ModelConfig.from_args(ns)
    assert c.crop == 256 and c.latent_dim == 21 and not c.abs_pos_emb
    assert ModelConfig.tps().latent_dim == 28


def test_synthetic_weights_are_deterministic_and_nonzero():
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.synthetic import synth_state_dict, state_shapes
    cfg = ModelConfig(num_layers=1)
    a, b = synth_state_dict(cfg, 3), synth_state_dict(cfg, 3)
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    assert set(a) == set(state_shapes(cfg))
    for k in ("layers.0.adaLN_modulation.1.weight", "emb_to_latent.linear.weight", "ipa_layers.0.ipa.linear_out.weight"):
        assert a[k].abs().max() > 0      # the reference zero-inits these; parity would be vacuous


def test_sampler_surface():
    from mdgen_amd.transport import Sampler, create_transport
    s = Sampler(create_transport(None, "GVP", "velocity"))
    with pytest.raises(NotImplementedError):
        s.sample_ode(sampling_method="dopri5")
    fn = s.sample_ode(sampling_method="euler", num_steps=11)
    # generic drift: dx/dt = 1 from x0 = 0 over [0,1] -> 1 (host loop, any callable)
    out = fn(torch.zeros(2, 3), lambda x, t: torch.ones_like(x))
    assert out.shape == (1, 2, 3) and torch.allclose(out[-1], torch.ones(2, 3), atol=1e-6)


def test_pdb_writer_matches_reference_text():
    """mdgen_amd.pdb.frames_to_pdb_string reproduces the reference's `atom14_to_pdb` output byte for byte
    (fixture tests/golden/pdb_small.npz made by oracle/gen_golden_pdb.py from the reference itself: 3 frames,
    residues FLRHGW, one atom exactly at the origin -> dropped)."""
    import numpy as np
    from mdgen_amd.pdb import frames_to_pdb_string, atom14_to_pdb
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pdb_small.npz"))
    ref = bytes(g["text"]).decode()
    assert frames_to_pdb_string(g["atom14"], g["aatype"]) == ref
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        pth = os.path.join(td, "x.pdb")
        atom14_to_pdb(g["atom14"], g["aatype"], pth)
        assert open(pth).read() == ref
    with pytest.raises(ValueError):
        frames_to_pdb_string(g["atom14"], np.full(6, 25))
    with pytest.raises(ValueError):
        frames_to_pdb_string(g["atom14"][0], g["aatype"])


def test_sincos_pos_embed_matches_reference():
    """`synthetic.sincos_pos_embed` (used when a state dict has no `pos_embed`) equals the
    reference's `get_1d_sincos_pos_embed_from_grid` table (tests/golden/pos_embed.npz, oracle/gen_golden_posembed.py)."""
    import numpy as np
    from mdgen_amd.synthetic import sincos_pos_embed
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pos_embed.npz"))
    for C, key in ((384, "c384"), (48, "c48")):
        ours = sincos_pos_embed(C, 9)[0].numpy()
        assert ours.shape == g[key].shape
        assert np.abs(ours - g[key]).max() < 1e-6


def test_view_plan_respects_the_32bit_offset_limit():
    """panel prologues/epilogues address h as base + 32-bit byte offset token*1536 -> at most 2 796 202
    token rows per launch.  `mdgen_debug_view_plan` (the function `mdgen_sample_euler` / `mdgen_denoiser_forward`
    cut batches with) must keep every view under the limit, cover the batch, and reject a single over-size sample."""
    import ctypes as C
    import mdgen_amd._lib as L
    LIMIT = 0xFFFFFFFF // 1536
    nv, per = C.c_int32(), C.c_int32()
    for (B, T, Lr, streams) in [(16, 1000, 4, 2), (1, 250, 256, 2), (44, 250, 256, 1), (45, 250, 256, 1), (45, 250, 256, 2),
                                (1000, 250, 256, 2), (5, 300, 4, 3), (3, 8000, 300, 8), (7, 1, 1, 8)]:
        assert L.lib.mdgen_debug_view_plan(C.byref(L.Shape(B, T, Lr)), streams, C.byref(nv), C.byref(per)) == 0
        assert per.value * T * Lr <= LIMIT, (B, T, Lr)
        assert nv.value >= min(streams, B) and nv.value <= B
        assert per.value == -(-B // nv.value)                    # views differ by at most one sample
        assert (nv.value - 1) * per.value < B <= nv.value * per.value
    assert nv.value == 7
    L.lib.mdgen_debug_view_plan(C.byref(L.Shape(43, 250, 256)), 1, C.byref(nv), C.byref(per))
    assert nv.value == 1                                          # 43 * 64 000 = 2 752 000 tokens fit one launch
    L.lib.mdgen_debug_view_plan(C.byref(L.Shape(44, 250, 256)), 1, C.byref(nv), C.byref(per))
    assert nv.value == 2 and per.value == 22                      # 44 * 64 000 = 2 816 000 do not
    assert L.lib.mdgen_debug_view_plan(C.byref(L.Shape(1, 8000, 400)), 1, C.byref(nv), C.byref(per)) == -2
    assert b"per-launch limit" in L.lib.mdgen_last_error()


def test_cli_work_selection():
    """`sim_inference.select_names` / `group_batches`: the reference's --chunk_idx/--n_chunks split
    (tps_inference.py:160-161, np.array_split), rank sharding under torch.distributed.run, --pdb_id filter; batches
    group peptides of equal length."""
    from mdgen_amd.sim_inference import select_names, group_batches
    names = [f"p{i}" for i in range(11)]
    assert select_names(names) == names
    chunks = [select_names(names, chunk_idx=i, n_chunks=3) for i in range(3)]
    assert [len(c) for c in chunks] == [4, 4, 3] and sum(chunks, []) == names          # np.array_split sizes
    per_rank = [select_names(names, chunk_idx=1, n_chunks=3, rank=r, world=2) for r in range(2)]
    assert sum(per_rank, []) == chunks[1] and abs(len(per_rank[0]) - len(per_rank[1])) <= 1
    assert select_names(names, pdb_id=["p2", "p9"], chunk_idx=0, n_chunks=3) == ["p2"]
    with pytest.raises(ValueError):
        select_names(names, chunk_idx=3, n_chunks=3)
    seq = {"a": "FLRH", "b": "IMRY", "c": "AAAAAAAA", "d": "GSTV", "e": "WWWWWWWW", "f": "KKKK"}
    g = group_batches(list(seq), seq, 3)
    assert g == [["a", "b", "d"], ["f"], ["c", "e"]]
    assert group_batches(list(seq), seq, 1) == [["a"], ["b"], ["d"], ["f"], ["c"], ["e"]]


def test_rigid_view_level_ops_cpu():
    """`Rigid.__init__` identity fill, `Rigid.identity`, `__getitem__`, `unsqueeze`,
    `cat`, `Rotation.cat`, `from_tensor_4x4`, `from_tensor_7` field handling -- view-level glue that runs no kernel,
    checked on CPU tensors against the semantics of rigid_utils.py:820-862, 892-921, 1122-1141, 1220-1261."""
    from mdgen_amd.rigid_utils import Rigid, Rotation
    g = torch.Generator().manual_seed(0)
    R = torch.linalg.qr(torch.randn(2, 5, 3, 3, generator=g))[0]
    t = torch.randn(2, 5, 3, generator=g)
    # r-7: missing half filled with identity (rigid_utils.py:838-853); shape / device checks
    a = Rigid(Rotation(rot_mats=R), None)
    assert a.shape == (2, 5) and torch.equal(a.get_trans(), torch.zeros(2, 5, 3)) and a.get_trans().dtype == torch.float32
    b = Rigid(None, t)
    assert torch.equal(b.get_rots().get_rot_mats(), torch.eye(3).expand(2, 5, 3, 3))
    with pytest.raises(ValueError):
        Rigid(None, None)
    with pytest.raises(ValueError):
        Rigid(Rotation(rot_mats=R), t[:1])
    with pytest.raises(ValueError):
        Rotation(rot_mats=R, quats=torch.zeros(2, 5, 4))
    with pytest.raises(ValueError):
        Rotation(rot_mats=R[..., :2])
    i = Rigid.identity((4, 3), fmt="rot_mat")       # dataset.py:82-85 padding frames
    assert i.shape == (4, 3) and torch.equal(i.get_rots().get_rot_mats(), torch.eye(3).expand(4, 3, 3, 3))
    assert torch.equal(i.get_trans(), torch.zeros(4, 3, 3))
    # fp32 is forced (rigid_utils.py:318-322, 859)
    assert Rigid(Rotation(rot_mats=R.double()), t.double()).get_trans().dtype == torch.float32
    assert Rotation(rot_mats=R.double()).get_rot_mats().dtype == torch.float32
    # r-8: indexing / unsqueeze / cat act on the virtual batch shape
    r = Rigid(Rotation(rot_mats=R), t)
    assert r[:, 0:1].shape == (2, 1) and torch.equal(r[:, 0:1].get_trans(), t[:, 0:1])
    assert torch.equal(r[1].get_rots().get_rot_mats(), R[1]) and r[1, 2].shape == ()
    assert torch.equal(r[..., 3].get_trans(), t[:, 3])
    assert r[..., None].shape == (2, 5, 1) and r.unsqueeze(-1).shape == (2, 5, 1) and r.unsqueeze(0).shape == (1, 2, 5)
    assert torch.equal(r.unsqueeze(-1).get_rots().get_rot_mats(), R[:, :, None])
    with pytest.raises(ValueError):
        r.unsqueeze(2)
    c = Rigid.cat([r, r[:, :2]], dim=1)
    assert c.shape == (2, 7) and torch.equal(c.get_trans(), torch.cat([t, t[:, :2]], 1))
    assert torch.equal(c.get_rots().get_rot_mats(), torch.cat([R, R[:, :2]], 1))
    c2 = Rigid.cat([r, r], dim=-1)
    assert c2.shape == (2, 10) and torch.equal(c2.get_rots().get_rot_mats()[:, 5:], R)
    assert Rotation.cat([Rotation(rot_mats=R), Rotation(rot_mats=R)], dim=0).shape == (4, 5)
    # quaternion-backed rotations keep their format and normalisation flag through views
    q = torch.randn(2, 5, 4, generator=g)
    rq = Rotation(quats=q, normalize_quats=True)
    assert rq[0].shape == (5,) and rq[0]._quats is not None and rq[0]._normalize and rq.unsqueeze(1).shape == (2, 1, 5)
    assert torch.allclose(rq.get_quats().norm(dim=-1), torch.ones(2, 5), atol=1e-6)
    assert torch.equal(Rotation(quats=q, normalize_quats=False).get_quats(), q)
    # `* mask` multiplies all 9 + 3 entries (rigid_utils.py:923-942)
    mk = torch.tensor([[1., 0, 1, 0, 1], [0, 1, 0, 1, 0]])
    m = r * mk
    assert torch.equal(m.get_trans(), t * mk[..., None]) and torch.equal(m.get_rots().get_rot_mats(), R * mk[..., None, None])
    # r-9: from_tensor_4x4 / from_tensor_7 split the fields; wrong shapes raise
    T4 = torch.zeros(2, 5, 4, 4)
    T4[..., :3, :3], T4[..., :3, 3], T4[..., 3, 3] = R, t, 1.0
    f = Rigid.from_tensor_4x4(T4)
    assert torch.equal(f.get_rots().get_rot_mats(), R) and torch.equal(f.get_trans(), t)
    with pytest.raises(ValueError):
        Rigid.from_tensor_4x4(T4[..., :3, :])
    x7 = torch.cat([q, t], -1)
    f7 = Rigid.from_tensor_7(x7, normalize_quats=True)
    assert torch.equal(f7.get_trans(), t) and f7.get_rots()._normalize and torch.equal(f7.get_rots()._quats, q)
    with pytest.raises(ValueError):
        Rigid.from_tensor_7(x7[..., :6])


def test_rigid_view_level_ops_vs_reference_fixture():
    """SURVEY rows r-7 / r-8 against the REFERENCE's own classes (tests/golden/rigid_views.npz, written by
    oracle/gen_golden.py gen_rigid_views running rigid_utils.py:820-862, 892-942, 1122-1141, 1220-1261): identity fill of a
    missing half, `Rigid.identity`, `__getitem__` (slice / int / ellipsis / None), `unsqueeze`, `cat`, `Rotation.cat`,
    `* mask`, `map_tensor_fn`, quaternion-backed views, `from_tensor_4x4`, `from_tensor_7`.  View-level glue: no kernel runs,
    so the comparison is on CPU tensors and exact."""
    from conftest import load_golden
    from mdgen_amd.rigid_utils import Rigid, Rotation
    g = load_golden("rigid_views")
    R, t, q, mk = g["R"], g["t"], g["q"], g["mk"]
    rm = lambda x: x.get_rots().get_rot_mats()
    r = Rigid(Rotation(rot_mats=R), t)
    assert torch.equal(Rigid(Rotation(rot_mats=R), None).get_trans(), g["fill_t"])
    assert torch.equal(rm(Rigid(None, t)), g["fill_R"])
    i = Rigid.identity((4, 3), fmt="rot_mat")
    assert torch.equal(rm(i), g["ident_R"]) and torch.equal(i.get_trans(), g["ident_t"])
    cases = {"idx_col": r[:, 0:1], "idx_row": r[1], "idx_ell": r[..., 3], "idx_none": r[..., None],
             "unsq_last": r.unsqueeze(-1), "unsq_first": r.unsqueeze(0), "cat1": Rigid.cat([r, r[:, :2]], dim=1),
             "cat_last": Rigid.cat([r, r], dim=-1), "mul": r * mk,
             "map_sum": (r * mk).map_tensor_fn(lambda x: torch.sum(x, dim=-1))}
    for key, v in cases.items():
        assert tuple(rm(v).shape) == tuple(g[key + "_R"].shape), key
        assert torch.equal(rm(v), g[key + "_R"]) and torch.equal(v.get_trans(), g[key + "_t"]), key
    assert torch.equal(Rotation.cat([Rotation(rot_mats=R), Rotation(rot_mats=R)], dim=0).get_rot_mats(), g["rotcat_R"])
    rq = Rotation(quats=q, normalize_quats=True)
    assert torch.allclose(rq.get_quats(), g["quat_norm"], atol=1e-7)
    assert torch.allclose(rq[0].get_quats(), g["quat_idx"], atol=1e-7)
    assert torch.allclose(rq.unsqueeze(1).get_quats(), g["quat_unsq"], atol=1e-7)
    assert torch.equal(Rotation(quats=q, normalize_quats=False).get_quats(), g["quat_raw"])
    f = Rigid.from_tensor_4x4(g["T4"])
    assert torch.equal(rm(f), g["f4_R"]) and torch.equal(f.get_trans(), g["f4_t"])
    f7 = Rigid.from_tensor_7(torch.cat([q, t], -1), normalize_quats=True)
    assert torch.allclose(f7.get_rots().get_quats(), g["f7_q"], atol=1e-7) and torch.equal(f7.get_trans(), g["f7_t"])


def test_epoch_shards_have_the_same_step_count_on_every_rank():
    """`train.shard_epoch`: DistributedSampler(drop_last)-style sharding -- every rank runs the SAME number of steps (a rank
    with one step more would sit in a bucketed all-reduce the others never enter), shards are disjoint."""
    from mdgen_amd.train import shard_epoch
    for n_items in (31, 32, 33, 7, 100, 101):
        for world in (1, 2, 3, 8):
            for bs in (1, 4, 8):
                order = list(range(n_items))[::-1]
                shards = [shard_epoch(order, r, world, bs) for r in range(world)]
                ns = {n for _, n in shards}
                assert len(ns) == 1, (n_items, world, bs, ns)
                n = ns.pop()
                assert n == (n_items // world) // bs
                seen = [j for s, _ in shards for j in s[:n * bs]]
                assert len(seen) == len(set(seen)) == n * bs * world
                assert all(len(s) >= n * bs for s, _ in shards)


def test_adam_state_torch_layout_round_trip():
    """`optim.adam_state_to_torch` / `adam_state_from_torch`: the optimiser state travels in torch.optim.Adam's own
    state_dict layout (what a Lightning checkpoint of the reference holds: state[i] in `model.parameters()` order + one param
    group), so a reference checkpoint can be resumed here and vice versa; cross-checked against a real torch.optim.Adam."""
    from collections import OrderedDict
    from mdgen_amd.optim import adam_state_from_torch, adam_state_to_torch
    shapes = OrderedDict([("a.weight", (3, 4)), ("a.bias", (3,)), ("b.weight", (2, 3))])
    ps = [torch.nn.Parameter(torch.randn(*s)) for s in shapes.values()]
    opt = torch.optim.Adam(ps, lr=3e-4)
    for _ in range(2):
        for p in ps:
            p.grad = torch.randn_like(p)
        opt.step()
    tsd = opt.state_dict()
    named = adam_state_from_torch(tsd, list(shapes))
    assert named["step"] == 2 and abs(named["lr"] - 3e-4) < 1e-12
    for i, k in enumerate(shapes):
        assert torch.equal(named["exp_avg"][k], tsd["state"][i]["exp_avg"])
        assert torch.equal(named["exp_avg_sq"][k], tsd["state"][i]["exp_avg_sq"])
    named.update(adamw=False)
    back = adam_state_to_torch(named, list(shapes))
    opt2 = torch.optim.Adam([torch.nn.Parameter(torch.zeros(*s)) for s in shapes.values()], lr=1.0)
    opt2.load_state_dict(back)          # torch accepts it
    assert opt2.state_dict()["param_groups"][0]["lr"] == 3e-4
    for i in range(3):
        assert torch.equal(opt2.state_dict()["state"][i]["exp_avg"], tsd["state"][i]["exp_avg"])
        assert float(opt2.state_dict()["state"][i]["step"]) == 2.0
    # a parameter that was never stepped has no state entry: zero moments
    tsd2 = {"state": {0: tsd["state"][0]}, "param_groups": tsd["param_groups"]}
    n2 = adam_state_from_torch(tsd2, list(shapes))
    assert n2["exp_avg"]["a.bias"] is None and n2["step"] == 2
