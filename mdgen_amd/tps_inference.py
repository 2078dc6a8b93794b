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


def build_parser():
    """Arguments of the reference's driver (tps_inference.py:6-18) minus the MSM analysis inputs (`--mddir`), plus
    the two MD frame indices the MSM would have chosen, `--num_steps` and `--synthetic`."""
    import argparse
    p = argparse.ArgumentParser()
    p.add_argument("--sim_ckpt", type=str, default=None)
    p.add_argument("--data_dir", type=str, default="share/4AA_data")
    p.add_argument("--suffix", type=str, default="")
    p.add_argument("--pdb_id", nargs="*", default=[])
    p.add_argument("--num_frames", type=int, default=1000)
    p.add_argument("--num_batches", type=int, default=100)
    p.add_argument("--batch_size", type=int, default=10)
    p.add_argument("--out_dir", type=str, default=".")
    p.add_argument("--split", type=str, default="splits/4AA_test.csv")
    p.add_argument("--chunk_idx", type=int, default=0)
    p.add_argument("--n_chunks", type=int, default=1)
    p.add_argument("--start_frame", type=int, default=0, help="MD frame of {name}.npy used as the start state")
    p.add_argument("--end_frame", type=int, default=-1, help="MD frame of {name}.npy used as the end state")
    p.add_argument("--num_steps", type=int, default=None)
    p.add_argument("--synthetic", action="store_true")
    return p


def run(args, model, device, names_seqres, rank=0, world=1):
    """tps_inference.py:118-168 (`do` + `main`): for every peptide of this process's chunk / rank shard,
    `num_batches` batches of `batch_size` transition paths between the two end states -> `{name}_{idx}.pdb`."""
    import os
    from .geometry import restype_order
    from .pdb import atom14_to_pdb
    from .sim_inference import select_names
    names = select_names(list(names_seqres), args.pdb_id, args.chunk_idx, args.n_chunks, rank, world)
    os.makedirs(args.out_dir, exist_ok=True)
    done = []
    for name in names:
        arr = np.lib.format.open_memmap(f"{args.data_dir}/{name}{args.suffix}.npy", "r")
        seq = names_seqres[name]
        one = get_sample(arr[args.start_frame], arr[args.end_frame], seq, args.num_frames, device)
        batch = collate([one] * args.batch_size)
        aat = np.array([restype_order[c] for c in seq])
        for i in range(args.num_batches):
            atom14s, _ = model.inference(batch, num_steps=args.num_steps)
            host = atom14s.cpu().numpy()
            for j in range(args.batch_size):
                idx = i * args.batch_size + j
                atom14_to_pdb(host[j], aat, os.path.join(args.out_dir, f"{name}_{idx}.pdb"))
        done.append(name)
    return {"names": done, "paths": len(done) * args.num_batches * args.batch_size}


def main(argv=None):
    import pandas as pd
    from .config import ModelConfig
    from .sim_inference import dist_env
    from .synthetic import synth_state_dict
    from .wrapper import NewMDGenWrapper
    args = build_parser().parse_args(argv)
    rank, world, local_rank = dist_env()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if args.synthetic:
        cfg = ModelConfig.tps(num_frames=args.num_frames)
        model = NewMDGenWrapper(cfg, device=device)
        model.model.load_state_dict(synth_state_dict(cfg, 0))
    else:
        if not args.sim_ckpt:
            raise SystemExit("--sim_ckpt is required (or --synthetic)")
        model = NewMDGenWrapper.load_from_checkpoint(args.sim_ckpt, device=device)
    df = pd.read_csv(args.split, index_col="name")
    return run(args, model, device, {str(n): df.seqres[n] for n in df.index}, rank, world)


if __name__ == "__main__":
    with torch.no_grad():
        main()
