"""Drop-in for the geometry helpers on the sampler path (mdgen/geometry.py), GPU-only:
`frames_torsions_to_atom14` (geometry.py:61-79) is part of `mdgen_samples_to_atom14`; the rollout glue
`atom14_to_frames` + `atom14_to_atom37` + `atom37_to_torsions` (sim_inference.py:91-96) is one kernel,
`mdgen_atom14_to_cond`, so that consecutive T-frame blocks chain on the device without a host
round trip."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L
from ._lib import lib, launch, ptr, require_cuda

_TABLES = {}
_NPZ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "residue_tables.npz")
RESTYPES = "ARNDCQEGHILKMFPSTWYV"
restype_order = {c: i for i, c in enumerate(RESTYPES)}


def residue_tables(device):
    """AlphaFold residue constant tables (data dumped from mdgen/residue_constants.py:1124-1216,
    1367-1480 by oracle/gen_residue_tables.py) on `device`."""
    key = str(device)
    if key not in _TABLES:
        d = np.load(_NPZ)
        _TABLES[key] = {k: torch.from_numpy(d[k]).to(device).contiguous() for k in d.files
                        if d[k].dtype.kind in "fi"}
    return _TABLES[key]


def samples_to_atom14(samples, rot0, trans0, seqres, tps: bool):
    """wrapper.py:456-478 + geometry.py:61-79: latents (B,T,L,D) + first-frame rigids + seqres -> atom14."""
    require_cuda(samples, rot0, trans0, seqres)
    B, T, L_, D = samples.shape
    tb = residue_tables(samples.device)
    s = samples.to(torch.float32).contiguous()
    r0, t0 = rot0.to(torch.float32).contiguous(), trans0.to(torch.float32).contiguous()
    sq = seqres.to(torch.int64).contiguous()
    out = torch.empty(B, T, L_, 14, 3, dtype=torch.float32, device=s.device)
    sh = L.Shape(B, T, L_)
    launch(lib.mdgen_samples_to_atom14, s, C.byref(sh), D, int(tps), ptr(s), ptr(r0), ptr(t0), ptr(sq),
           ptr(tb["default_frames"]), ptr(tb["lit_positions"]), ptr(tb["atom14_group"]), ptr(tb["atom14_mask"]), ptr(out))
    return out


def atom14_to_cond(atom14, seqres):
    """sim_inference.py:91-96: atom14 (B,L,14,3) -> dict(rots (B,L,3,3), trans (B,L,3), torsions (B,L,7,2),
    torsion_mask (B,L,7)) -- `atom14_to_frames` (geometry.py:218-231) and `atom37_to_torsions`
    (geometry.py:82-202) of the same frame."""
    require_cuda(atom14, seqres)
    B, L_ = atom14.shape[:2]
    tb = residue_tables(atom14.device)
    a = atom14.to(torch.float32).contiguous()
    sq = seqres.to(torch.int64).contiguous()
    dev = a.device
    rots = torch.empty(B, L_, 3, 3, device=dev)
    trans = torch.empty(B, L_, 3, device=dev)
    tors = torch.empty(B, L_, 7, 2, device=dev)
    tmask = torch.empty(B, L_, 7, device=dev)
    launch(lib.mdgen_atom14_to_cond, a, B, L_, ptr(a), ptr(sq), ptr(tb["atom37_to_atom14"]), ptr(tb["atom37_mask"]),
           ptr(tb["chi_atom_indices"]), ptr(tb["chi_angles_mask"]), ptr(rots), ptr(trans), ptr(tors), ptr(tmask))
    return {"rots": rots, "trans": trans, "torsions": tors, "torsion_mask": tmask}


def atom14_to_frames(atom14):
    """geometry.py:218-231 for atom14 (B,L,14,3); returns a Rigid."""
    from .rigid_utils import Rigid, Rotation
    B, L_ = atom14.shape[:2]
    c = atom14_to_cond(atom14, torch.zeros(B, L_, dtype=torch.int64, device=atom14.device))
    return Rigid(Rotation(rot_mats=c["rots"]), c["trans"])
