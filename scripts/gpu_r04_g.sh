#!/bin/bash
# round 4: fuse_proj A/B with the LDS-modulation MLP kernel; ATLAS and small-N lines
mkdir -p gpurun_out/r04g
run() {  n=$1; shift
  timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline "$@" > gpurun_out/r04g/$n.json 2> gpurun_out/r04g/$n.err
  python -c "
import json
d=json.load(open('gpurun_out/r04g/$n.json')); r=d['roofline'] or {}; print('$n', d['value'], d['ms_per_step'], {k:v for k,v in list((r.get('by_kernel_ms_per_call') or {}).items())[:7]})"
}
run fp0 --option fuse_proj=0
run fp1 --option fuse_proj=1
run fp0b --option fuse_proj=0
run fp1b --option fuse_proj=1
run atlas_fp0 --workload atlas_crop256_T250_B1 --option fuse_proj=0
run atlas_fp1 --workload atlas_crop256_T250_B1 --option fuse_proj=1
