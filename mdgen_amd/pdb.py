"""Multi-model PDB writer for sampled trajectories, byte-compatible with the reference's
`mdgen.utils.atom14_to_pdb` (utils.py:58-64 -> `create_full_prot` :67-92 -> `prots_to_pdb` :95-102 ->
`protein.to_pdb` protein.py:321-443), without mdtraj / Biopython.

Format (one block per frame):  `MODEL <i>` (0-based, unpadded), then one ATOM record per present atom in atom37
order (an atom is present when |x|+|y|+|z| > 1e-7, utils.py:77), a `TER` record, `ENDMDL`.  Records are padded
to 80 columns; residue numbers are 0-based, chain `A`, occupancy 1.00, B-factor 0.00, element = first letter of
the atom name (protein.py:388-413).  Host-side IO only: nothing here is on the sampling path."""
from __future__ import annotations

import os

import numpy as np

_NPZ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "residue_tables.npz")
_T = None


def residue_tables():
    """Residue constant tables as numpy arrays (oracle/gen_residue_tables.py); no torch, no device library."""
    global _T
    if _T is None:
        d = np.load(_NPZ)
        _T = {k: d[k] for k in d.files}
    return _T


def atom14_to_atom37(atom14: np.ndarray, aatype: np.ndarray) -> np.ndarray:
    """geometry.py atom14_to_atom37: gather per residue type + mask.  atom14 [..., L, 14, 3], aatype [L]."""
    t = residue_tables()
    idx = np.asarray(t["atom37_to_atom14"])[aatype]                       # [L, 37]
    m = np.asarray(t["atom37_mask"])[aatype]                              # [L, 37]
    a37 = np.take_along_axis(atom14, np.broadcast_to(idx[..., None], atom14.shape[:-3] + idx.shape + (3,)), axis=-2)
    return a37 * m[..., None]


def frames_to_pdb_string(atom14: np.ndarray, aatype: np.ndarray) -> str:
    """atom14 [M, L, 14, 3] (Angstrom), aatype [L] -> the text `atom14_to_pdb` writes."""
    t = residue_tables()
    atom_types = [str(a) for a in t["atom_types"]]
    res3 = [str(r) for r in t["restype_3"]]
    atom14 = np.asarray(atom14, dtype=np.float32)
    aatype = np.asarray(aatype).astype(np.int64)
    if atom14.ndim != 4 or atom14.shape[-2:] != (14, 3) or atom14.shape[1] != aatype.shape[0]:
        raise ValueError(f"atom14 {atom14.shape} / aatype {aatype.shape}: expected [M,L,14,3] and [L]")
    if np.any(aatype > 20) or np.any(aatype < 0):
        raise ValueError("Invalid aatypes.")
    a37 = atom14_to_atom37(atom14, aatype)
    present = np.abs(a37).sum(-1) > 1e-7
    names = [a if len(a) == 4 else f" {a}" for a in atom_types]
    out = []
    L = aatype.shape[0]
    for m in range(a37.shape[0]):
        out.append(f"MODEL {m}")
        serial = 1
        for i in range(L):
            r3 = res3[aatype[i]]
            for k in range(37):
                if not present[m, i, k]:
                    continue
                x, y, z = a37[m, i, k]
                line = (f"{'ATOM':<6}{serial:>5} {names[k]:<4}{'':>1}{r3:>3} {'A':>1}{i:>4}{'':>1}   "
                        f"{x:>8.3f}{y:>8.3f}{z:>8.3f}{1.0:>6.2f}{0.0:>6.2f}          {atom_types[k][0]:>2}{'':>2}")
                out.append(line.ljust(80))
                serial += 1
        out.append(f"{'TER':<6}{serial:>5}      {res3[aatype[L - 1]]:>3} {'A':>1}{L - 1:>4}".ljust(80))
        out.append("ENDMDL")
    return "\n".join(out) + "\n"


def atom14_to_pdb(atom14, aatype, path) -> None:
    """Drop-in for `mdgen.utils.atom14_to_pdb(atom14, aatype, path)`."""
    with open(path, "w") as f:
        f.write(frames_to_pdb_string(np.asarray(atom14), np.asarray(aatype)))
