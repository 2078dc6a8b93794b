"""world_size-2 gloo test of the N>1 path: batch sharding is a disjoint cover, every rank's work is counted
once, the job time is the max over ranks, and no collective touches sample data."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    from mdgen_amd.sharding import gather_over_ranks, shard_list, max_over_ranks, sum_over_ranks
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    names = [f"pep{i}" for i in range(n_items)]
    mine = shard_list(names, rank, world)
    frames = 1000 * len(mine)                      # every peptide: one 1000-frame block
    dist.barrier()
    dt = max_over_ranks(0.1 * (rank + 1), dist)    # pretend rank r took 0.1*(r+1) s
    total = sum_over_ranks(frames, dist)
    per_rank = gather_over_ranks(0.1 * (rank + 1), dist)    # bench.py's per-rank rates (stragglers)
    dist.barrier()
    q.put((rank, mine, dt, total, per_rank))
    dist.destroy_process_group()


def test_two_rank_batch_sharding():
    world, n_items = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shards = [r[1] for r in res]
    assert sorted(sum(shards, [])) == sorted(f"pep{i}" for i in range(n_items))     # disjoint cover
    assert abs(len(shards[0]) - len(shards[1])) <= 1
    assert all(abs(r[2] - 0.2) < 1e-9 for r in res)                                   # max over ranks
    assert all(len(r[4]) == 2 and abs(r[4][0] - 0.1) < 1e-9 and abs(r[4][1] - 0.2) < 1e-9 for r in res)   # every rank sees all
    assert all(r[3] == 1000 * n_items for r in res)                                   # whole-job frames


class _FakeSampler:
    """Stands in for NewMDGenWrapper on CPU: `rollout` returns a deterministic trajectory that encodes which
    peptide it belongs to (so the test can see who sampled what)."""

    def rollout(self, batch, num_frames, num_rollouts, num_steps=None):
        B, L = batch["seqres"].shape
        base = batch["tag"].view(B, 1, 1, 1, 1).float()
        return base + torch.zeros(B, num_rollouts * num_frames, L, 14, 3) + torch.arange(14).view(1, 1, 1, 14, 1) * 0.1


def _cli_worker(rank, world, port, data_dir, out_dir, q):
    """The REAL driver (`mdgen_amd.sim_inference.run`: chunking, rank sharding, batching by length, PDB writing,
    barrier + max-over-ranks timing) on a fake sampler, gloo backend."""
    import argparse
    from mdgen_amd import sim_inference as cli
    from mdgen_amd.geometry import restype_order
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seqs = {"pA": "FLRH", "pB": "IMRY", "pC": "AWKDGG", "pD": "GSTV", "pE": "KKKK", "pF": "WWWWWW", "pG": "AAAA"}
    args = cli.build_parser().parse_args(["--data_dir", data_dir, "--out_dir", out_dir, "--num_frames", "3",
                                          "--num_rollouts", "2", "--batch", "2", "--npy"])

    def batch_fn(names, arrs, seqres, device):      # CPU stand-in for get_batch + collate (those need the GPU glue)
        for n in names:
            assert arrs[n].shape[1] == len(seqres[n])
        return {"seqres": torch.tensor([[restype_order[c] for c in seqres[n]] for n in names]),
                "tag": torch.tensor([float(ord(n[1])) for n in names])}
    res = cli.run(args, _FakeSampler(), "cpu", seqs, rank, world, batch_fn=batch_fn, dist=dist)
    q.put((rank, res))
    dist.destroy_process_group()


def test_two_rank_cli_driver_on_fake_sampler(tmp_path):
    import numpy as np
    seqs = {"pA": 4, "pB": 4, "pC": 6, "pD": 4, "pE": 4, "pF": 6, "pG": 4}
    data = tmp_path / "data"
    data.mkdir()
    for n, L in seqs.items():
        np.save(data / f"{n}.npy", np.zeros((2, L, 14, 3), np.float16))
    out = tmp_path / "out"
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cli_worker, args=(r, world, port, str(data), str(out), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0]["names"] == ["pA", "pB", "pC", "pD"] and res[1]["names"] == ["pE", "pF", "pG"]      # contiguous shards
    assert res[0]["frames"] == 4 * 6 and res[1]["frames"] == 3 * 6
    assert res[0]["job_frames"] == res[1]["job_frames"] == 7 * 6                                    # whole job, both ranks
    assert res[0]["job_seconds"] == res[1]["job_seconds"] == max(res[0]["seconds"], res[1]["seconds"])
    for n, L in seqs.items():                                                                        # every peptide once
        a = np.load(out / f"{n}.npy")
        assert a.shape == (6, L, 14, 3) and abs(a[0, 0, 0, 0] - ord(n[1])) < 1e-6
        assert open(out / f"{n}.pdb").read().count("MODEL") == 6


def test_shard_range_properties():
    from mdgen_amd.sharding import shard_range
    for n in (0, 1, 5, 16, 255, 256):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                cover += list(range(lo, hi))
            assert cover == list(range(n))


def _bucket_worker(rank, world, port, q):
    """World-2 gloo run of the DDP gradient path: the bucket schedule of `GradBucketer` over the REAL model's parameter
    list (reverse order = backward order), asynchronous SUM all-reduce per bucket as soon as its last gradient is
    marked ready, 1 / world averaging returned for the optimiser."""
    from collections import OrderedDict
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.optim import FlatParams, GradBucketer
    from mdgen_amd.synthetic import state_shapes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = OrderedDict((k, v) for k, v in state_shapes(ModelConfig(num_layers=1, crop=4)).items()
                         if not k.endswith("inv_freq") and k != "pos_embed")
    fp = FlatParams(shapes, device="cpu")
    grads = fp.like()
    gen = torch.Generator().manual_seed(1234)                 # same base on both ranks, scaled by (rank + 1)
    base = torch.randn(fp.numel, generator=gen)
    grads.copy_(base * (rank + 1))
    gb = GradBucketer(fp, grads, dist=dist, bucket_bytes=4 << 20)
    for name in list(shapes)[::-1]:                           # the order a backward pass produces gradients in
        gb.mark_ready(name)
    scale = gb.finish()
    expect = base * sum(r + 1 for r in range(world))
    q.put((rank, len(gb.buckets), gb.launch_order, scale, float((grads - expect).abs().max()),
           [(b["lo"], b["hi"]) for b in gb.buckets], fp.numel))
    dist.destroy_process_group()


def test_two_rank_bucketed_gradient_allreduce():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, nb, order, scale, err, spans, numel in res:
        assert nb >= 2 and order == list(range(nb))            # buckets complete in backward order
        assert scale == 0.5 and err < 1e-5                     # SUM over ranks; averaging left to the optimiser
        cover = sorted(spans)                                  # buckets tile the flat buffer exactly
        assert cover[0][0] == 0 and cover[-1][1] == numel
        assert all(a[1] == b[0] for a, b in zip(cover, cover[1:]))


def test_bucket_plan_of_the_full_model():
    """The full 34 M-parameter model in 16 MiB buckets: 136.6 MB of fp32 gradients -> 8 buckets."""
    from collections import OrderedDict
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.optim import FlatParams, GradBucketer
    from mdgen_amd.synthetic import state_shapes
    shapes = OrderedDict((k, v) for k, v in state_shapes(ModelConfig.atlas()).items() if not k.endswith("inv_freq"))
    fp = FlatParams(shapes, device="cpu")
    assert fp.numel == 34152521                                   # forward-sim model parameters
    gb = GradBucketer(fp, fp.like(), dist=None)
    sizes = [(b["hi"] - b["lo"]) * 4 for b in gb.buckets]
    assert len(gb.buckets) == 8 and sum(sizes) == fp.numel * 4 and max(sizes) < 24 << 20
    import pytest
    from mdgen_amd._lib import MdgenError
    gb.mark_ready("layers.4.fc2.weight")
    with pytest.raises(MdgenError):
        gb.finish()                                            # a gradient that was never produced is an error


def test_flat_order_follows_the_backward_pass():
    """The flat parameter order used for training (`train.flat_order`): walking it from the end meets the parameter
    groups in the order the library's backward pass completes them (milestones 0, 1, ... of mdgen_amd.h), so that with
    16 MiB buckets 7 of the 8 all-reduces of the 34 M-parameter model can start before the backward pass has finished
    -- in the reference's registration order the FIRST bucket would already contain `t_embedder`, final only at the
    very end."""
    from mdgen_amd.config import ModelConfig
    from mdgen_amd.optim import FlatParams, GradBucketer
    from mdgen_amd.train import flat_order, grad_milestone, trainable_shapes
    cfg = ModelConfig.atlas()
    nl = cfg.num_layers
    order = flat_order(cfg)
    assert sorted(order) == sorted(trainable_shapes(cfg)) and list(order) != list(trainable_shapes(cfg))
    ms = [grad_milestone(k, nl) for k in order]
    assert ms == sorted(ms, reverse=True) and ms[-1] == 0 and ms[0] == 2 * nl + 2
    fp = FlatParams(order, device="cpu")
    gb = GradBucketer(fp, fp.like(), dist=None)
    bm = [max(grad_milestone(n, nl) for n in b["names"]) for b in gb.buckets]
    assert bm == sorted(bm) and len(bm) == 8
    assert sum(1 for m in bm if m < 2 * nl + 2) >= 7, bm
    # the reference's own order, for contrast: its first bucket waits for the last milestone
    fr = FlatParams(trainable_shapes(cfg), device="cpu")
    gr = GradBucketer(fr, fr.like(), dist=None)
    assert max(grad_milestone(n, nl) for n in gr.buckets[0]["names"]) == 2 * nl + 2


def test_bench_self_launches_its_ranks_when_run_without_a_launcher():
    """`python bench.py --gpus 2 ...` with WORLD_SIZE unset (the shape of the driver's 1-GPU command) must start its own two
    ranks (torch.distributed.run --standalone, 127.0.0.1) and print ONE JSON line with n_gpus 2, per-rank rates and the
    max-over-ranks time -- driven here on CPU through bench.py's test hook `--fake-sampler` (step = sleep of 10 ms x (rank + 1),
    gloo in place of RCCL): launcher, rendezvous, barrier-bracketed timed region, reductions and JSON assembly are the real
    code (train.py:46-68 / tps_inference.py:160-161 are the reference's multi-process entry points)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--fake-sampler"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert len(d["per_rank_frames_per_s"]) == 2 and d["rank_max_over_min"] > 1.5          # rank 1 sleeps twice as long
    # whole-job value = frames of BOTH ranks / the slowest rank's time (4 steps x 20 ms)
    assert d["ms_per_step"] >= 20.0 and abs(d["value"] - 2 * 16 * 1000 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-3 * d["value"]
    # a launcher that disagrees with --gpus is refused, not silently accepted
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--fake-sampler"], capture_output=True, text=True,
                        timeout=120, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=root)
    assert r2.returncode != 0 and "WORLD_SIZE" in (r2.stderr + r2.stdout)


def test_cli_accepts_the_readme_flags_incl_xtc(tmp_path, capsys):
    """The reference README's forward-simulation command line (README.md:72: `--num_rollouts 10 --num_frames 1000 --xtc`) must
    run in the drop-in CLI: `--xtc` is accepted and, where mdtraj imports, the XTC is written and the PDB cut to one frame
    (sim_inference.py:121-125).  Where it does not (this image), the command is refused with exit status 2 BEFORE anything is sampled
    -- a pipeline that consumes `{name}.xtc` must fail at the command, not later; without `--xtc` the multi-model PDB is written."""
    import numpy as np
    from mdgen_amd import sim_inference as cli
    from mdgen_amd.geometry import restype_order
    data = tmp_path / "data"
    data.mkdir()
    np.save(data / "pA.npy", np.zeros((2, 4, 14, 3), np.float16))
    out = tmp_path / "out"
    args = cli.build_parser().parse_args(["--data_dir", str(data), "--out_dir", str(out), "--num_frames", "3", "--num_rollouts", "2",
                                          "--xtc", "--suffix", ""])
    assert args.xtc

    def batch_fn(names, arrs, seqres, device):
        return {"seqres": torch.tensor([[restype_order[c] for c in seqres[n]] for n in names]),
                "tag": torch.tensor([float(ord(n[1])) for n in names])}
    try:
        import mdtraj  # noqa: F401
        have = True
    except ImportError:
        have = False
    if have:
        res = cli.run(args, _FakeSampler(), "cpu", {"pA": "FLRH"}, batch_fn=batch_fn)
        assert res["frames"] == 6
        assert (out / "pA.xtc").exists() and open(out / "pA.pdb").read().count("MODEL") <= 1
    else:
        with pytest.raises(SystemExit) as ei:
            cli.run(args, _FakeSampler(), "cpu", {"pA": "FLRH"}, batch_fn=batch_fn)
        assert ei.value.code == 2 and "mdtraj" in capsys.readouterr().err
        assert not out.exists() or not list(out.iterdir())            # nothing was sampled or written
        args.xtc = False
        res = cli.run(args, _FakeSampler(), "cpu", {"pA": "FLRH"}, batch_fn=batch_fn)
        assert res["frames"] == 6 and open(out / "pA.pdb").read().count("MODEL") == 6
