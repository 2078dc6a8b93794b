#!/bin/bash
# round 6, call Q: the L2 warmer (option l2_warm 1 against 0) at B = 1, B = 2 and at cfg-3's shard: same bits, per-launch and end-to-end timing
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06q; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python - > $O/equal.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
from mdgen_amd.config import ModelConfig
from mdgen_amd.synthetic import synth_state_dict
from mdgen_amd.wrapper import NewMDGenWrapper
dev = torch.device("cuda")
for wl in ("tetrapeptide_fwdsim_crop4_T1000_B1", "tetrapeptide_tps_crop4_T100_B32"):
    B, T, L, abs_pos, n_pad = bench.WORKLOADS[wl]
    cfg = ModelConfig(crop=L, num_frames=T, abs_pos_emb=abs_pos, sim_condition=True)
    outs = {}
    for warm in (0, 1):
        w = NewMDGenWrapper(cfg, device=dev); w.model.load_state_dict(synth_state_dict(cfg, 0))
        w.model.set_option("l2_warm", warm)
        batch = bench.synth_batch(B, T, L, n_pad, dev, seed=100)
        zs = torch.randn(B, T, L, 21, generator=torch.Generator().manual_seed(137)).to(dev)
        a = w.inference(batch, zs=zs, num_steps=5, use_graph=False)[0]
        b = w.inference(batch, zs=zs, num_steps=5, use_graph=True)[0]
        c = w.inference(batch, zs=zs, num_steps=5, use_graph=True)[0]
        assert torch.isfinite(a).all() and torch.equal(a, b) and torch.equal(b, c), (wl, warm)
        outs[warm] = a.cpu()
        del w
    print(wl, "l2_warm 1 == 0 bit for bit:", torch.equal(outs[0], outs[1]), flush=True)
PY
cat $O/equal.txt | grep -v amdgpu | tail -4
B1=tetrapeptide_fwdsim_crop4_T1000_B1; TP=tetrapeptide_tps_crop4_T100_B32
run_k() { echo "== $1 $2" | tee -a $O/kbench.txt; timeout 300 python scripts/kbench.py $1 3 $2 2>&1 | grep -v parity | grep -v amdgpu | head -7 | tee -a $O/kbench.txt; }
run_b() { timeout 300 python bench.py --workload $1 --steps 8 --warmup 3 --no-extra --no-cpu-baseline --no-roofline $2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
for rep in 1 2; do
  run_k $B1 l2_warm=0; run_k $B1 l2_warm=1; run_k $TP l2_warm=0; run_k $TP l2_warm=1
done
for rep in 1 2 3; do
  run_b $B1 "--option l2_warm=0"; run_b $B1 "--option l2_warm=1"; run_b $TP "--option l2_warm=0"; run_b $TP "--option l2_warm=1"
done
