"""Transition-path (two-sided conditioning) batch layout, mirroring the reference's
`tps_inference.get_sample` (tps_inference.py:43-80): the start frame's torsions / translations / rotations are
expanded over `num_frames` and frame -1 is replaced by the end frame's; `NewMDGenWrapper.prep_batch` then marks
frames 0 and -1 as conditioning (wrapper.py:341-342) and the model runs its IPA stack on both frame sets.

What is NOT here: the reference chooses the start/end MD frames from a Markov-state model built with
pyemma/mdtraj (tps_inference.py:82-118); that analysis is outside the sampler path.  `get_sample` therefore takes
the two frames explicitly.  The geometry (atom14 -> frames, torsions) runs on the device through the same glue
kernel the forward-simulation rollout uses (`mdgen_atom14_to_cond`)."""
from __future__ import annotations

import numpy as np
import torch


def get_sample(start_arr, end_arr, seqres_str: str, num_frames: int, device="cuda"):
    """start_arr / end_arr: atom14 coordinates [1, L, 14, 3] (or [L, 14, 3]) of the two end states, Angstrom.
    Returns the batch dict of tps_inference.py:68-79 with a leading batch dimension of 1 (what the reference's
    DataLoader collation adds): torsions (1,T,L,7,2), torsion_mask (1,L,7), trans (1,T,L,3), rots (1,T,L,3,3),
    seqres (1,L), mask (1,L)."""
    from .geometry import atom14_to_cond, restype_order

    def frame(a):
        a = np.asarray(a, dtype=np.float32)
        if a.ndim == 3:
            a = a[None]
        return torch.from_numpy(np.copy(a[0:1])).to(device)

    seqres = torch.tensor([restype_order[c] for c in seqres_str], device=device)[None]
    s = atom14_to_cond(frame(start_arr), seqres)
    e = atom14_to_cond(frame(end_arr), seqres)
    T = int(num_frames)

    def traj(key):
        x = s[key][:, None].expand(-1, T, *s[key].shape[1:]).clone()
        x[:, -1] = e[key]
        return x

    L_ = seqres.shape[1]
    return {"torsions": traj("torsions"), "torsion_mask": s["torsion_mask"], "trans": traj("trans"),
            "rots": traj("rots"), "seqres": seqres, "mask": torch.ones(1, L_, device=device)}


def collate(samples):
    """Stack `get_sample` dicts along the batch dimension (the reference uses a DataLoader for this, :127)."""
    return {k: torch.cat([s[k] for s in samples], 0) for k in samples[0]}
