"""Golden vectors of `mdgen.dataset.MDGenDataset.__getitem__` (dataset.py:19-100), made by RUNNING THE REFERENCE here:

    PYTHONPATH=oracle/shims:/root/reference:. MODEL_DIR=/tmp/mdl python oracle/gen_golden_dataset.py

A small synthetic data directory (self-consistent random structures, fp16 like scripts/prep_sims.py writes them) is
built in a temp dir; the reference dataset is asked for items under fixed numpy seeds in three regimes -- ATLAS with
L > crop (random crop), ATLAS with L < crop (identity-frame padding), tetrapeptide (no crop) -- and the inputs (the
.npy arrays) plus the returned tensors are stored in tests/golden/dataset.npz (data only).
numpy-2 note: the reference calls atom14_to_atom37 on a numpy array, which numpy 2 rejects
inside tensor_utils.batched_gather; the call is routed through torch tensors here, arithmetic unchanged."""
import argparse
import os
import sys
import tempfile

import numpy as np
import pandas as pd
import torch

import mdgen.dataset as D
from mdgen import geometry as G
from mdgen.residue_constants import restype_order

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import synth_structure, save   # noqa: E402

_orig = G.atom14_to_atom37
D.atom14_to_atom37 = lambda arr, aat: _orig(torch.from_numpy(np.asarray(arr)), aat).numpy()

if __name__ == "__main__":
    g = torch.Generator().manual_seed(77)
    seqs = {"pLong": "MKTAYIAKQRQISF", "pShort": "GSHMKV", "FLRH": "FLRH"}
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, sq in seqs.items():
            seqres = torch.tensor([[restype_order[c] for c in sq]])
            reps = ["_R1", "_R2", "_R3"] if name.startswith("p") else [""]
            for r in reps:
                a14 = synth_structure(g, 1, 9, len(sq), seqres)[0].numpy().astype(np.float16)   # [frames, L, 14, 3]
                np.save(os.path.join(td, f"{name}{r}.npy"), a14)
                out[f"arr_{name}{r}"] = a14
        cases = []
        for tag, names, atlas, crop in (("atlas", ["pLong", "pShort"], True, 8), ("pep", ["FLRH"], False, 4)):
            split = os.path.join(td, f"{tag}.csv")
            pd.DataFrame({"name": names, "seqres": [seqs[n] for n in names]}).to_csv(split, index=False)
            args = argparse.Namespace(data_dir=td, suffix="", atlas=atlas, crop=crop, num_frames=4, overfit=False,
                                      overfit_peptide=None, overfit_frame=False, frame_interval=None, copy_frames=False,
                                      no_frames=False)
            ds = D.MDGenDataset(args, split, repeat=2)
            assert len(ds) == 2 * len(names)
            for seed in (0, 1, 2, 3):
                for idx in range(len(ds)):
                    np.random.seed(100 * seed + idx)
                    it = ds[idx]
                    key = f"{tag}_s{seed}_i{idx}"
                    cases.append(key)
                    out[key + "_name"] = np.array(it["name"])
                    out[key + "_frame_start"] = np.array(it["frame_start"])
                    for k in ("torsions", "torsion_mask", "trans", "rots"):
                        out[key + "_" + k] = it[k].numpy()
                    out[key + "_seqres"] = np.asarray(it["seqres"]).astype(np.int64)
                    out[key + "_mask"] = np.asarray(it["mask"]).astype(np.float32)
        out["cases"] = np.array(cases)
        out["seq_names"] = np.array(list(seqs))
        out["seq_strings"] = np.array(list(seqs.values()))
    save("dataset", **out)
