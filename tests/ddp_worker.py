"""Worker of test_ddp_two_processes_match_one / test_ddp_over_rccl_two_gpus (not a test): one rank of a 2-process
data-parallel training step.

Launched twice (RANK 0 / 1, WORLD_SIZE 2).  DDP_BACKEND=gloo (default): both ranks on cuda:0 -- RCCL refuses two ranks on
one device; gloo all-reduces CUDA tensors through the host, which is all the gradient exchange needs on a one-GPU box.
DDP_BACKEND=nccl (= RCCL; boxes with >= 2 GPUs): rank r on cuda:r, the production path.  DDP_TIMING=1 additionally runs
steps at BASELINE.json configs[4]'s per-GPU shape (B 1, T 250, L 256) and records the step time and the exposed all-reduce
wait.  Each rank builds a `Trainer(dist=...)` -- rank 1
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
    backend = os.environ.get("DDP_BACKEND", "gloo")
    dev = torch.device("cuda", rank if backend == "nccl" and world > 1 else 0)
    torch.cuda.set_device(dev)
    selftest = os.environ.get("DDP_SELFTEST") == "1"   # one rank, backend nccl: RCCL initialisation + every collective of the step as identities
    if world > 1 or selftest:
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    B, T, L = 2, 6, 5
    cfg = ModelConfig(crop=L, num_frames=T, num_layers=1, abs_pos_emb=True, sim_condition=True)
    w = NewMDGenWrapper(cfg, device=dev)
    w.load_model_state_dict(synth_state_dict(cfg, 23 if rank == 0 else 999))   # rank 1 starts elsewhere: broadcast must fix it
    tr = Trainer(w, lr=1e-3, adamw=False, grad_clip=1.0, ema_decay=0.9, dist=dist if (world > 1 or selftest) else None,
                 single_rank_collectives=selftest)
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
    res = {"params": tr.tm.params.data.cpu(), "ema": tr.ema.data.cpu(), "loss": float(loss), "world": world, "backend": backend,
           "collectives": len(tr.buckets._handles) if hasattr(tr.buckets, "_handles") else 0, "reduces": tr.buckets._reduces()}
    tr.close()
    if os.environ.get("DDP_TIMING") == "1":   # cfg-5's per-GPU shape: how much of the all-reduce is NOT hidden behind the backward pass
        import time
        from mdgen_amd.synthetic import synth_batch
        del tr, w
        cfg5 = ModelConfig.atlas(num_frames=250, crop=256)
        w5 = NewMDGenWrapper(cfg5, device=dev)
        w5.load_model_state_dict(synth_state_dict(cfg5, 5))
        tr5 = Trainer(w5, lr=1e-4, grad_clip=1.0, ema_decay=0.999, dist=dist if world > 1 else None)
        tr5.tm.model.set_option("train_precision", 16)
        b5 = synth_batch(1, 250, 256, 16, dev, seed=100 + rank)
        comm, n = 0.0, 0
        for i in range(5):
            if i == 2:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            tr5.training_step(b5)
            if i >= 2:
                comm += tr5.exposed_comm_ms
                n += 1
        torch.cuda.synchronize()
        res["step_ms"] = (time.perf_counter() - t0) * 1e3 / n
        res["exposed_comm_ms"] = comm / n
        tr5.close()
    if rank == 0:
        torch.save(res, out_path)
    if world > 1 or selftest:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    run(sys.argv[1])
