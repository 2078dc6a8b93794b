#!/bin/bash
# round 4, call E: k_mlp_rows with LDS-resident modulation vectors: parity + bench
mkdir -p gpurun_out/r04e
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_owner or forward_vs_reference_golden or dual_stream or full_size_properties_cfg2 or tps_cfg3" > gpurun_out/r04e/pytest.log 2>&1
tail -3 gpurun_out/r04e/pytest.log
for i in 1 2; do
timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --option chain_path=0 > gpurun_out/r04e/bench$i.json 2> gpurun_out/r04e/bench$i.err
python -c "
import json
d=json.load(open('gpurun_out/r04e/bench$i.json')); print(d['value'], d['ms_per_step'], d['roofline']['by_kernel_ms_per_call'])"
done
timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --workload tetrapeptide_tps_crop4_T100_B32 > gpurun_out/r04e/tps.json 2>/dev/null; python -c "
import json
d=json.load(open('gpurun_out/r04e/tps.json')); print('tps auto streams', d['value'])"
