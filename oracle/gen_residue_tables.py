"""Dump the AlphaFold residue constant tables the hot path needs (THIS CONTAINER ONLY).

Run:  PYTHONPATH=oracle/shims:/root/reference MODEL_DIR=/tmp/mdl python oracle/gen_residue_tables.py

Source of the numbers: the reference's `mdgen/residue_constants.py` (tables built at import time,
`:1124-1216` rigid-group frames/positions, `:1367-1480` atom14<->atom37 maps) and
`mdgen/geometry.py:337-358` (chi atom indices). Only numeric DATA is written (a ~20 KB .npz);
no reference source travels. The product loads `mdgen_amd/data/residue_tables.npz`.
"""
import os
import numpy as np

import mdgen.residue_constants as rc
from mdgen.geometry import get_chi_atom_indices

out = os.path.join(os.path.dirname(__file__), "..", "mdgen_amd", "data", "residue_tables.npz")
chi_mask = list(rc.chi_angles_mask) + [[0.0, 0.0, 0.0, 0.0]]
np.savez_compressed(
    out,
    default_frames=rc.restype_rigid_group_default_frame.astype(np.float32),        # [21,8,4,4]
    lit_positions=rc.restype_atom14_rigid_group_positions.astype(np.float32),      # [21,14,3]
    atom14_group=rc.restype_atom14_to_rigid_group.astype(np.int64),                # [21,14]
    atom14_mask=rc.restype_atom14_mask.astype(np.float32),                         # [21,14]
    atom37_to_atom14=np.asarray(rc.RESTYPE_ATOM37_TO_ATOM14).astype(np.int64),     # [21,37]
    atom37_mask=np.asarray(rc.RESTYPE_ATOM37_MASK).astype(np.float32),             # [21,37]
    atom14_to_atom37=np.asarray(rc.RESTYPE_ATOM14_TO_ATOM37).astype(np.int64),     # [21,14]
    atom14_mask_b=np.asarray(rc.RESTYPE_ATOM14_MASK).astype(np.float32),           # [21,14]
    chi_atom_indices=np.asarray(get_chi_atom_indices()).astype(np.int64),          # [21,4,4]
    chi_angles_mask=np.asarray(chi_mask).astype(np.float32),                       # [21,4]
    restypes=np.asarray(list(rc.restypes)),                                        # 20 one-letter codes
    atom_types=np.asarray(list(rc.atom_types)),                                    # 37 atom names
    restype_3=np.asarray([rc.restype_1to3[r] for r in rc.restypes] + ["UNK"]),      # 21 three-letter codes
)
print("wrote", os.path.abspath(out), os.path.getsize(out), "bytes")
