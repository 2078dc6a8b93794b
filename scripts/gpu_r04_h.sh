#!/bin/bash
# round 4: transposed k_embed (16-byte stores): parity (h0 traces) + bench
mkdir -p gpurun_out/r04h
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forward_vs_reference_golden or forward_vs_oracle_shapes or tps_vs_oracle or fp32_mode_forward or training_step_gradients or headline" > gpurun_out/r04h/pytest.log 2>&1
tail -3 gpurun_out/r04h/pytest.log
for i in 1 2; do
timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/r04h/bench$i.json 2> gpurun_out/r04h/bench$i.err
python -c "
import json
d=json.load(open('gpurun_out/r04h/bench$i.json')); print(d['value'], d['ms_per_step'], d['roofline']['by_kernel_ms_per_call'])"
done
timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --workload atlas_crop256_T250_B1 > gpurun_out/r04h/atlas.json 2>/dev/null; python -c "
import json
d=json.load(open('gpurun_out/r04h/atlas.json')); print('atlas', d['value'], d['roofline']['by_kernel_ms_per_call'])"
