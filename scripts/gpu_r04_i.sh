#!/bin/bash
# round 4: residue out-projection fused into the temporal q/k/v kernel (fuse_proj_qkv): parity on L > 8 shapes + ATLAS A/B
mkdir -p gpurun_out/r04i
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forward_vs_reference_golden or forward_vs_oracle_shapes or cfg4_full_size_vs_reference or row_owner or residue_axis_paths" > gpurun_out/r04i/pytest.log 2>&1
tail -3 gpurun_out/r04i/pytest.log
for fp in 0 1 0 1; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --workload atlas_crop256_T250_B1 --option fuse_proj_qkv=$fp > gpurun_out/r04i/atlas$fp.json 2> gpurun_out/r04i/atlas$fp.err
  python -c "
import json
d=json.load(open('gpurun_out/r04i/atlas$fp.json')); print('fuse_proj_qkv=$fp', d['value'], d['ms_per_step'], {k:v for k,v in list(d['roofline']['by_kernel_ms_per_call'].items())[:8]})"
done
