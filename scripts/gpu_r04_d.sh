#!/bin/bash
# round 4, call D: experiments -- MLP workgroup de-phasing (mlp_stagger), small-N stream count
mkdir -p gpurun_out/r04d
run() {  # name, args...
  n=$1; shift
  timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline "$@" > gpurun_out/r04d/$n.json 2> gpurun_out/r04d/$n.err
  python -c "
import json
d=json.load(open('gpurun_out/r04d/$n.json')); r=d['roofline'] or {}; print('$n', d['value'], d['ms_per_step'], {k:v for k,v in list((r.get('by_kernel_ms_per_call') or {}).items())[:6]})"
}
for sg in 0 1 2 3 4; do run stag$sg --option chain_path=0 --option mlp_stagger=$sg; done
for sg in 0 2; do run stag${sg}_s1 --option chain_path=0 --option mlp_stagger=$sg --streams 1; done
run tps_s2 --workload tetrapeptide_tps_crop4_T100_B32 --streams 2
run tps_s1 --workload tetrapeptide_tps_crop4_T100_B32 --streams 1
run b1_s1 --workload tetrapeptide_fwdsim_crop4_T1000_B1 --streams 1
