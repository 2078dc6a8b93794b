#!/bin/bash
# round 6, call V: the sequence-resident attention backward of the training step (k16_attn_bwd_seq, option train_attn_form):
# unit test (every length, both forms), training gradients vs the reference fixtures, step time A/B, rocprofv3 kernel averages
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06v; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "training_attention_kernels_unit or training_gradients or training_bf16_operand" 2>&1 | tail -8 | tee $O/pytest.log
for rep in 1 2; do
  for f in 1 0; do
    echo "train_attn_form=$f $(timeout 300 python scripts/train_bench.py 1 250 256 10 16 train_attn_form=$f 2>&1 | tail -1)" | tee -a $O/train_ab.txt
  done
done
for f in 1 0; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$f -o kt -- python $R/scripts/train_bench.py 1 250 256 3 16 train_attn_form=$f > $O/rocprof$f.log 2>&1 < /dev/null)
  s=$(find $O/prof$f -name "*kernel_stats.csv" | head -1)
  [ -n "$s" ] && cp $s $O/kernel_stats_train_form$f.csv && echo "== form $f" && grep -i "attn" $O/kernel_stats_train_form$f.csv | cut -c1-160
  rm -rf $O/prof$f
done
