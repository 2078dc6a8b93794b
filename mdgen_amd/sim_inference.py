"""Forward-simulation driver, command-line compatible with the reference's `sim_inference.py:1-14`
(`--sim_ckpt --data_dir --suffix --pdb_id --num_frames --num_rollouts --out_dir --split`), plus

  --num_steps S      Euler steps per block (the reference samples with its checkpoint's `sampling_method`; this
                     build has fixed-grid Euler only and refuses a non-Euler checkpoint unless S is given);
  --batch N          peptides of equal length sampled together in one `inference()` call (the reference runs
                     B = 1, sim_inference.py:101-102; B = 16 is the regime BASELINE.json's metric is quoted on);
  --chunk_idx/--n_chunks   the reference's own sharding switches (tps_inference.py:17-18,160-161): this process
                     handles chunk `chunk_idx` of `n_chunks` of the split.  Under `torch.distributed.run`
                     (RANK / WORLD_SIZE in the environment) the chunk is further sharded over the ranks -- one
                     process per GPU, no data-path collective (mdgen_amd/sharding.py);
  --synthetic        seeded weights instead of --sim_ckpt;  --npy  also save the sampled array.

The rollout (sim_inference.py:61-98) stays on the device: each block's last frame becomes the next block's
conditioning frame through `mdgen_atom14_to_cond` (no D->H->D round trip).  Output, as the reference
(sim_inference.py:117-119): `{out_dir}/{name}.pdb`, a multi-model PDB of all sampled frames written by
`mdgen_amd.pdb.atom14_to_pdb` (byte-compatible with `mdgen.utils.atom14_to_pdb`).  `--xtc` (sim_inference.py:121-125, and part of
the README's own forward-simulation command, README.md:72) is accepted: the PDB is always written; where `mdtraj` imports, the
reference's four lines run as they stand (superposed trajectory -> `{name}.xtc`, the PDB cut to its first frame), where it does not
(this image) a warning says so and the multi-model PDB stays.  `--no_frames` / `--tps` select other models: rejected loudly.
"""
from __future__ import annotations

import argparse
import sys
import os
import time
from typing import Dict, List, Sequence

import numpy as np
import torch


# ---- work selection (pure host logic; covered by the CPU / gloo tests) -------------------------------------------
def select_names(all_names: Sequence[str], pdb_id: Sequence[str] = (), chunk_idx: int = 0, n_chunks: int = 1,
                 rank: int = 0, world: int = 1) -> List[str]:
    """Names this process samples: chunk `chunk_idx` of `np.array_split(names, n_chunks)` (tps_inference.py:160-161),
    then the contiguous shard of `rank` among `world` processes, then the `--pdb_id` filter (applied last, as the
    reference does inside its loop, :166-167)."""
    from .sharding import shard_list
    if not (0 <= chunk_idx < n_chunks):
        raise ValueError(f"chunk_idx {chunk_idx} outside 0..{n_chunks - 1}")
    chunk = [str(n) for n in np.array_split(np.array(list(all_names), dtype=object), n_chunks)[chunk_idx]]
    mine = shard_list(chunk, rank, world)
    return [n for n in mine if not pdb_id or n in pdb_id]


def group_batches(names: Sequence[str], seqres: Dict[str, str], batch: int) -> List[List[str]]:
    """Consecutive groups of at most `batch` names with EQUAL sequence length (one `inference()` call each);
    order of first appearance is kept so that outputs are reproducible."""
    by_len: Dict[int, List[str]] = {}
    for n in names:
        by_len.setdefault(len(seqres[n]), []).append(n)
    out = []
    for _, ns in by_len.items():
        out += [ns[i:i + batch] for i in range(0, len(ns), max(1, batch))]
    return out


def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


# ---- batches ---------------------------------------------------------------------------------------------------
def get_batch(arr, seqres_str, device):
    """sim_inference.get_batch (:32-59): first MD frame -> conditioning batch (B=1)."""
    from .geometry import atom14_to_cond, restype_order
    a = torch.from_numpy(np.copy(arr[0:1]).astype(np.float32)).to(device)          # [1,L,14,3]
    seqres = torch.tensor([restype_order[c] for c in seqres_str], device=device)[None]
    c = atom14_to_cond(a, seqres)
    L_ = seqres.shape[1]
    return {"torsions": c["torsions"][:, None], "torsion_mask": c["torsion_mask"], "trans": c["trans"][:, None],
            "rots": c["rots"][:, None], "seqres": seqres, "mask": torch.ones(1, L_, device=device)}


def collate(batches):
    """Stack B = 1 batches of equal L along the batch dimension (what the reference's DataLoader would do)."""
    return {k: torch.cat([b[k] for b in batches], 0) for k in batches[0]}


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


def make_group_batch(names, arrs, seqres, device):
    """One conditioning batch (B = len(names)) for a group of equal-length peptides."""
    return collate([get_batch(arrs[n], seqres[n], device) for n in names])


def sample_group(model, batch, args):
    """`num_rollouts` chained blocks for one batch; returns atom14 [B, R*T, L, 14, 3]."""
    if hasattr(model, "rollout") and not getattr(args, "per_block", False):
        return model.rollout(batch, args.num_frames, args.num_rollouts, num_steps=args.num_steps)
    out = []
    for _ in range(args.num_rollouts):
        atom14, batch = rollout(model, batch, args.num_frames, args.num_steps)
        out.append(atom14)
    return torch.cat(out, 1)


def require_xtc_writer():
    """`--xtc` is the reference's mdtraj post-processing (sim_inference.py:121-125).  A pipeline that asks for `{name}.xtc` must not
    find out after hours of sampling that none was written: without mdtraj the command is refused BEFORE anything is sampled
    (exit status 2); drop `--xtc` to get the multi-model PDB, which holds every frame."""
    try:
        import mdtraj  # noqa: F401
    except ImportError:
        print("error: --xtc needs mdtraj (the reference's dependency for the XTC file), which is not installed; "
              "run without --xtc to get the multi-model PDB with all frames", file=sys.stderr)
        raise SystemExit(2)


def write_xtc(pdb_path, xtc_path):
    """`--xtc` (sim_inference.py:121-125): superpose the sampled trajectory on its first frame, save it as XTC and cut the PDB to
    that frame (mdtraj: checked by require_xtc_writer before sampling starts)."""
    import mdtraj
    traj = mdtraj.load(pdb_path)
    traj.superpose(traj)
    traj.save(xtc_path)
    traj[0].save(pdb_path)
    return True


def run(args, model, device, names_seqres, rank=0, world=1, batch_fn=make_group_batch, sync=None, dist=None):
    """The driver proper (sim_inference.py:100-128), given a loaded model: select this process's peptides
    (chunk, rank shard, --pdb_id), group them into batches of equal length, sample every group, write
    `{out_dir}/{name}.pdb`.  `names_seqres`: ordered {name: sequence} of the whole split.  With `dist` (an
    initialised torch.distributed module) the ranks meet at a barrier before and after the timed region and the
    reported rate is whole-job frames / slowest rank's time -- no collective touches sample data."""
    from .geometry import restype_order
    from .pdb import atom14_to_pdb
    from .sharding import max_over_ranks, sum_over_ranks
    sync = sync or (lambda: None)
    if getattr(args, "xtc", False):
        require_xtc_writer()
    names = select_names(list(names_seqres), args.pdb_id, args.chunk_idx, args.n_chunks, rank, world)
    seqres = {n: names_seqres[n] for n in names}
    os.makedirs(args.out_dir, exist_ok=True)
    total_frames, total_s = 0, 0.0
    if dist is not None:
        dist.barrier()
    for group in group_batches(names, seqres, args.batch):
        arrs = {n: np.lib.format.open_memmap(f"{args.data_dir}/{n}{args.suffix}.npy", "r") for n in group}
        batch = batch_fn(group, arrs, seqres, device)
        sync()
        start = time.time()
        all_atom14 = sample_group(model, batch, args)
        sync()
        dur = time.time() - start
        nfr = len(group) * args.num_rollouts * args.num_frames
        total_frames, total_s = total_frames + nfr, total_s + dur
        print(f"{','.join(group)}: {nfr / dur:.1f} frames/s ({dur:.3f} s, batch {len(group)})")
        host = all_atom14.cpu().numpy()
        for i, n in enumerate(group):
            aat = np.array([restype_order[c] for c in seqres[n]])
            atom14_to_pdb(host[i], aat, os.path.join(args.out_dir, f"{n}.pdb"))
            if getattr(args, "xtc", False):
                write_xtc(os.path.join(args.out_dir, f"{n}.pdb"), os.path.join(args.out_dir, f"{n}.xtc"))
            if args.npy:
                np.save(os.path.join(args.out_dir, f"{n}.npy"), host[i])
    job_s = max_over_ranks(total_s, dist)
    job_frames = sum_over_ranks(total_frames, dist)
    if dist is not None:
        dist.barrier()
    if job_s > 0:
        print(f"rank {rank}/{world}: {len(names)} peptides, {total_frames} frames in {total_s:.3f} s; "
              f"job: {job_frames / job_s:.1f} frames/s")
    return {"names": names, "frames": total_frames, "seconds": total_s, "job_frames": int(job_frames),
            "job_seconds": job_s}


def build_parser():
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
    p.add_argument("--num_steps", type=int, default=None,
                   help="Euler steps per block (default: 49 = the reference's 50-point grid, Euler checkpoints only)")
    p.add_argument("--batch", type=int, default=1, help="peptides of equal length per inference() call")
    p.add_argument("--chunk_idx", type=int, default=0)
    p.add_argument("--n_chunks", type=int, default=1)
    p.add_argument("--per_block", action="store_true",
                   help="one inference() call per block from Python instead of the multi-block device rollout")
    p.add_argument("--precision", choices=["bf16", "fp32"], default="bf16",
                   help="bf16 MFMA operands (default) or the fp32-operand tolerance mode (~10x slower)")
    p.add_argument("--synthetic", action="store_true", help="seeded synthetic weights instead of --sim_ckpt")
    p.add_argument("--npy", action="store_true", help="also save the sampled atom14 array as .npy")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.no_frames or args.tps:
        raise SystemExit("--no_frames / --tps are outside this build's scope (see DESIGN.md)")
    import pandas as pd
    from .config import ModelConfig
    from .synthetic import synth_state_dict
    from .wrapper import NewMDGenWrapper
    rank, world, local_rank = dist_env()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:   # one process per GPU (torch.distributed.run); RCCL is used for the barrier and the timing only
        import torch.distributed as dist_
        if not dist_.is_initialized():
            dist_.init_process_group("nccl", device_id=device)
        dist = dist_
    if args.synthetic:
        cfg = ModelConfig.forward_sim(num_frames=args.num_frames)
        model = NewMDGenWrapper(cfg, device=device, precision=args.precision)
        model.model.load_state_dict(synth_state_dict(cfg, 0))
    else:
        if not args.sim_ckpt:
            raise SystemExit("--sim_ckpt is required (or --synthetic)")
        model = NewMDGenWrapper.load_from_checkpoint(args.sim_ckpt, device=device, precision=args.precision)
    df = pd.read_csv(args.split, index_col="name")
    names_seqres = {str(n): df.seqres[n] for n in df.index}
    return run(args, model, device, names_seqres, rank, world, sync=torch.cuda.synchronize, dist=dist)


if __name__ == "__main__":
    with torch.no_grad():
        main()
