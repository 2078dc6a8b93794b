#!/bin/bash
# round 6, call H: the split MLP form's hand-over through agent-scope atomic stores / loads (no L2 write-back): parity, stress, B = 1 timing
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06h; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -s -k "small_ or stress or split or row_owner or inference_headline or B1_T" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit" | cut -c1-250 | tail -8
B1=tetrapeptide_fwdsim_crop4_T1000_B1
for rep in 1 2; do timeout 300 python scripts/kbench.py $B1 3 2>&1 | grep -v parity | grep -v amdgpu | head -6 | tee -a $O/kbench.txt; done
for rep in 1 2 3; do timeout 300 python bench.py --workload $B1 --steps 8 --warmup 3 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('B1', d['value'], d['ms_per_step'])" | tee -a $O/bench.txt; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ktrace -- python $R/bench.py --workload $B1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extra --no-graph --streams 1 > $O/rocprof.log 2>&1)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cp {} '$O'/kernel_stats_B1_T1000.csv; head -6 {} | cut -c1-130'
rm -rf $O/prof
