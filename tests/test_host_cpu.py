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
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), n
    assert set(L.EXPORTS) == set(names)
    assert L.lib.mdgen_abi_version() == 1


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
