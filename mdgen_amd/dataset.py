"""`MDGenDataset` -- drop-in for `mdgen.dataset.MDGenDataset` (dataset.py:8-100): one training / validation item =
a `num_frames` window of an MD trajectory stored as `{data_dir}/{name}{suffix}.npy` (fp16/fp32 `[frames, L, 14, 3]`,
Angstrom), turned into the tensors `NewMDGenWrapper.prep_batch` consumes (torsions, torsion_mask, trans, rots, seqres,
mask), cropped / padded to `crop` residues for ATLAS proteins (dataset.py:70-89).

The window, replica (`_R1.._R3`) and crop offsets are drawn from numpy's global RNG in exactly the reference's order
(`np.random.randint(1, 4)` -> `np.random.choice(arange(n - num_frames))` -> `np.random.randint(0, L - crop + 1)`), so a
seeded run selects the same data.  The geometry (atom14 -> backbone frames, atom37 torsions; geometry.py:82-231) runs
on the GPU in `mdgen_atom14_to_cond` with the frames as the batch dimension; items come back as device tensors
(use `DataLoader(num_workers=0)`).  Not implemented: `--no_frames` (another model family), `--overfit_peptide`."""
from __future__ import annotations

import numpy as np
import pandas as pd
import torch

from .geometry import atom14_to_cond, restype_order


class MDGenDataset(torch.utils.data.Dataset):
    def __init__(self, args, split, repeat=1, device="cuda"):
        super().__init__()
        self.df = pd.read_csv(split, index_col="name")
        self.args = args
        self.repeat = repeat
        self.device = torch.device(device)
        if getattr(args, "no_frames", False) or getattr(args, "overfit_peptide", None):
            raise NotImplementedError("no_frames / overfit_peptide are outside the accelerated path")

    def __len__(self):
        return self.repeat * len(self.df)

    def __getitem__(self, idx):
        a = self.args
        idx = idx % len(self.df)
        if getattr(a, "overfit", False):
            idx = 0
        name = self.df.index[idx]
        seqres_str = self.df.seqres[name]
        if getattr(a, "atlas", False):
            i = np.random.randint(1, 4)                                   # dataset.py:31-33
            full_name = f"{name}_R{i}"
        else:
            full_name = name
        arr = np.lib.format.open_memmap(f"{a.data_dir}/{full_name}{getattr(a, 'suffix', '')}.npy", "r")
        if getattr(a, "frame_interval", None):
            arr = arr[::a.frame_interval]
        frame_start = np.random.choice(np.arange(arr.shape[0] - a.num_frames))   # dataset.py:40
        if getattr(a, "overfit_frame", False):
            frame_start = 0
        arr = np.copy(arr[frame_start:frame_start + a.num_frames]).astype(np.float32)
        if getattr(a, "copy_frames", False):
            arr[1:] = arr[0]
        T, L = arr.shape[:2]
        seqres = np.array([restype_order[c] for c in seqres_str])
        atom14 = torch.from_numpy(arr).to(self.device)                     # frames are the batch dimension
        aat = torch.from_numpy(seqres).to(self.device)[None].expand(T, -1).contiguous()
        c = atom14_to_cond(atom14, aat)                                    # rots (T,L,3,3) trans (T,L,3) torsions (T,L,7,2)
        torsions, rots, trans = c["torsions"], c["rots"], c["trans"]
        torsion_mask = c["torsion_mask"][0]
        mask = np.ones(L, dtype=np.float32)
        if getattr(a, "atlas", False):
            if L > a.crop:                                                 # dataset.py:71-77
                start = np.random.randint(0, L - a.crop + 1)
                sl = slice(start, start + a.crop)
                torsions, rots, trans = torsions[:, sl], rots[:, sl], trans[:, sl]
                seqres, mask, torsion_mask = seqres[sl], mask[sl], torsion_mask[sl]
            elif L < a.crop:                                               # dataset.py:80-89: identity frames, zeros
                pad = a.crop - L
                dev = self.device
                eye = torch.eye(3, device=dev).expand(T, pad, 3, 3)
                rots = torch.cat([rots, eye], 1)
                trans = torch.cat([trans, torch.zeros(T, pad, 3, device=dev)], 1)
                torsions = torch.cat([torsions, torch.zeros(T, pad, 7, 2, device=dev)], 1)
                torsion_mask = torch.cat([torsion_mask, torch.zeros(pad, 7, device=dev)])
                mask = np.concatenate([mask, np.zeros(pad, dtype=np.float32)])
                seqres = np.concatenate([seqres, np.zeros(pad, dtype=int)])
        return {
            "name": full_name,
            "frame_start": frame_start,
            "torsions": torsions.contiguous(),
            "torsion_mask": torsion_mask.contiguous(),
            "trans": trans.contiguous(),
            "rots": rots.contiguous(),
            "seqres": torch.from_numpy(np.ascontiguousarray(seqres)).to(self.device),
            "mask": torch.from_numpy(mask).to(self.device),
        }
