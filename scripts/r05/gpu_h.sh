#!/bin/bash
# round 5, call H: de-phasing the two co-resident workgroups of the panel kernels (k_ln_qkv<false>, k_ln_qkv_attn4<true>): one of
# every two workgroups starts `stagger` shader cycles late.  Per-kernel times (kbench, single stream, 1000 workgroups) and end to end.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r05h; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for pat in 0 1; do for st in 0 3000 6000 10000 15000 25000; do
  [ $pat = 1 ] && [ $st = 0 ] && continue
  timeout 300 python scripts/kbench.py tetrapeptide_fwdsim_crop4_T1000_B16 3 stagger=$st stagger_pattern=$pat 2>&1 | grep "S=3\|attn_L\|ln_qkv_T" | tee -a $O/kbench.txt
done; done
run() { timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
run
for pat in 0 1; do for st in 6000 10000 15000; do run --option stagger=$st --option stagger_pattern=$pat; done; done
run
