"""Forward-simulation driver, command-line compatible with the reference's `sim_inference.py:1-14`
(`--sim_ckpt --data_dir --suffix --pdb_id --num_frames --num_rollouts --out_dir --split`), plus
`--num_steps` (Euler steps; reference hard-codes 49) and `--synthetic` (seeded weights, no checkpoint).

The rollout (sim_inference.py:61-98) stays on the device: each block's last frame is turned into the
next block's conditioning frame by `mdgen_atom14_to_cond` (no D->H->D round trip).  Output, as the reference
(sim_inference.py:117-119): `{out_dir}/{name}.pdb`, a multi-model PDB of all sampled frames written by
`mdgen_amd.pdb.atom14_to_pdb` (byte-compatible with `mdgen.utils.atom14_to_pdb`; no mdtraj / Biopython);
`--npy` additionally saves the float32 array [num_rollouts*num_frames, L, 14, 3].  `--xtc` needs mdtraj and
`--no_frames` / `--tps` select other models: accepted and rejected loudly.
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import torch


def get_batch(arr, seqres_str, device):
    """sim_inference.get_batch (:32-59): first MD frame -> conditioning batch (B=1)."""
    from .geometry import atom14_to_cond, restype_order
    a = torch.from_numpy(np.copy(arr[0:1]).astype(np.float32)).to(device)          # [1,L,14,3]
    seqres = torch.tensor([restype_order[c] for c in seqres_str], device=device)[None]
    c = atom14_to_cond(a, seqres)
    L_ = seqres.shape[1]
    return {"torsions": c["torsions"][:, None], "torsion_mask": c["torsion_mask"], "trans": c["trans"][:, None],
            "rots": c["rots"][:, None], "seqres": seqres, "mask": torch.ones(1, L_, device=device)}


def rollout(model, batch, num_frames, num_steps, zs=None):
    """sim_inference.rollout (:61-98) with the glue on the device."""
    from .geometry import atom14_to_cond
    ex = dict(batch)
    ex["torsions"] = batch["torsions"].expand(-1, num_frames, -1, -1, -1)
    ex["trans"] = batch["trans"].expand(-1, num_frames, -1, -1)
    ex["rots"] = batch["rots"].expand(-1, num_frames, -1, -1, -1)
    atom14, _ = model.inference(ex, zs=zs, num_steps=num_steps)
    c = atom14_to_cond(atom14[:, -1], batch["seqres"])
    new = dict(batch)
    new["trans"], new["rots"], new["torsions"] = c["trans"][:, None], c["rots"][:, None], c["torsions"][:, None]
    return atom14, new


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--sim_ckpt", type=str, default=None)
    p.add_argument("--data_dir", type=str, required=True)
    p.add_argument("--suffix", type=str, default="")
    p.add_argument("--pdb_id", nargs="*", default=[])
    p.add_argument("--num_frames", type=int, default=1000)
    p.add_argument("--num_rollouts", type=int, default=100)
    p.add_argument("--no_frames", action="store_true")
    p.add_argument("--tps", action="store_true")
    p.add_argument("--xtc", action="store_true")
    p.add_argument("--out_dir", type=str, default=".")
    p.add_argument("--split", type=str, default="splits/4AA_test.csv")
    p.add_argument("--num_steps", type=int, default=49)
    p.add_argument("--synthetic", action="store_true", help="seeded synthetic weights instead of --sim_ckpt")
    p.add_argument("--npy", action="store_true", help="also save the sampled atom14 array as .npy")
    args = p.parse_args(argv)
    if args.no_frames or args.tps or args.xtc:
        raise SystemExit("--no_frames / --tps / --xtc are outside this build's scope (see DESIGN.md)")
    import pandas as pd
    from .config import ModelConfig
    from .synthetic import synth_state_dict
    from .wrapper import NewMDGenWrapper
    os.makedirs(args.out_dir, exist_ok=True)
    if args.synthetic:
        cfg = ModelConfig.forward_sim(num_frames=args.num_frames)
        model = NewMDGenWrapper(cfg)
        model.model.load_state_dict(synth_state_dict(cfg, 0))
    else:
        if not args.sim_ckpt:
            raise SystemExit("--sim_ckpt is required (or --synthetic)")
        model = NewMDGenWrapper.load_from_checkpoint(args.sim_ckpt)
    df = pd.read_csv(args.split, index_col="name")
    for name in df.index:
        if args.pdb_id and name not in args.pdb_id:
            continue
        arr = np.lib.format.open_memmap(f"{args.data_dir}/{name}{args.suffix}.npy", "r")
        batch = get_batch(arr, df.seqres[name], model.device)
        out = []
        torch.cuda.synchronize()
        start = time.time()
        for _ in range(args.num_rollouts):
            atom14, batch = rollout(model, batch, args.num_frames, args.num_steps)
            out.append(atom14)
        torch.cuda.synchronize()
        dur = time.time() - start
        print(f"{name}: {args.num_rollouts * args.num_frames / dur:.1f} frames/s ({dur:.3f} s)")
        all_atom14 = torch.cat(out, 1)[0].cpu().numpy()
        from .pdb import atom14_to_pdb
        atom14_to_pdb(all_atom14, batch["seqres"][0].cpu().numpy(), os.path.join(args.out_dir, f"{name}.pdb"))
        if args.npy:
            np.save(os.path.join(args.out_dir, f"{name}.npy"), all_atom14)


if __name__ == "__main__":
    with torch.no_grad():
        main()
