"""Golden vector for the multi-model PDB writer (THIS CONTAINER ONLY; imports the reference).

Run:  PYTHONPATH=oracle/shims:/root/reference:. MODEL_DIR=/tmp/mdl python oracle/gen_golden_pdb.py

Follows mdgen/utils.py:58-64 `atom14_to_pdb` (atom14 -> atom37 -> Protein -> `prots_to_pdb`), passing torch
tensors to `atom14_to_atom37` (the numpy path of `tensor_utils.batched_gather` does not run on numpy 2).
Writes tests/golden/pdb_small.npz: inputs (atom14 [3,6,14,3] float32, aatype [6]) and the exact text."""
import os
import numpy as np
import torch

from mdgen.geometry import atom14_to_atom37
from mdgen.utils import create_full_prot, prots_to_pdb
import mdgen.residue_constants as rc

g = torch.Generator().manual_seed(11)
aatype = torch.tensor([rc.restype_order[c] for c in "FLRHGW"])
mask = torch.from_numpy(np.asarray(rc.RESTYPE_ATOM14_MASK)[aatype.numpy()]).float()
atom14 = (torch.randn(3, 6, 14, 3, generator=g) * 12.0).float() * mask[None, :, :, None]
atom14[1, 2, 1] = 0.0   # an atom sitting exactly at the origin is dropped by the writer (|x|+|y|+|z| <= 1e-7)
prots = []
for pos in atom14:
    a37 = atom14_to_atom37(pos, aatype)
    prots.append(create_full_prot(a37.numpy(), aatype=aatype.numpy()))
text = prots_to_pdb(prots)
out = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "pdb_small.npz")
np.savez_compressed(out, atom14=atom14.numpy(), aatype=aatype.numpy(), text=np.frombuffer(text.encode(), dtype=np.uint8))
print("wrote", os.path.abspath(out), len(text), "chars;", text.count("\n"), "lines")
print(text[:400])
