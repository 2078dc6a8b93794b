"""Worker of test_ddp_two_processes_match_one (not a test): one rank of a 2-process data-parallel training step on ONE GPU.

Launched twice (RANK 0 / 1, WORLD_SIZE 2, gloo backend -- RCCL refuses two ranks on one device; gloo all-reduces CUDA
tensors through the host, which is all the gradient exchange needs here).  Each rank builds a `Trainer(dist=...)` -- rank 1
deliberately from DIFFERENT initial weights, so the construction-time broadcast is what makes the ranks agree -- runs
`Trainer.training_step` (the `launch_on_events` path: bucketed all-reduce on a communication stream behind the library's
gradient milestones) on ITS item of a 2-item batch, and rank 0 writes the resulting flat parameters to argv[1]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist

from conftest import load_golden
from mdgen_amd.config import ModelConfig
from mdgen_amd.synthetic import synth_state_dict
from mdgen_amd.train import Trainer
from mdgen_amd.wrapper import NewMDGenWrapper


def run(out_path, steps=2):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("gloo")
    B, T, L = 2, 6, 5
    cfg = ModelConfig(crop=L, num_frames=T, num_layers=1, abs_pos_emb=True, sim_condition=True)
    w = NewMDGenWrapper(cfg, device=dev)
    w.load_model_state_dict(synth_state_dict(cfg, 23 if rank == 0 else 999))   # rank 1 starts elsewhere: broadcast must fix it
    tr = Trainer(w, lr=1e-3, adamw=False, grad_clip=1.0, ema_decay=0.9, dist=dist if world > 1 else None)
    g0 = load_golden("prep_sim")
    batch = {k[3:]: v.to(dev) for k, v in g0.items() if k.startswith("in_")}
    gen = torch.Generator().manual_seed(3)
    launched = []
    tr.on_bucket = lambda i, view: launched.append(i)
    for _ in range(steps):
        t = torch.rand(B, generator=gen).to(dev)
        x0 = torch.randn(B, T, L, cfg.latent_dim, generator=gen).to(dev)
        if world > 1:   # this rank's item of the global batch
            sl = slice(rank, rank + 1)
            loss = tr.training_step({k: v[sl].contiguous() for k, v in batch.items()}, t=t[sl].contiguous(), x0=x0[sl].contiguous())
        else:
            loss = tr.training_step(batch, t=t, x0=x0)
    torch.cuda.synchronize()
    assert len(launched) == steps * len(tr.buckets.buckets)
    if rank == 0:
        torch.save({"params": tr.tm.params.data.cpu(), "ema": tr.ema.data.cpu(), "loss": float(loss), "world": world}, out_path)
    tr.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    run(sys.argv[1])
