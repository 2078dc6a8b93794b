#!/bin/bash
# Round 5, last measurement call on the final build: the whole -m gpu suite, smoke(), the default bench line (cfg-2 with roofline,
# cpu_baseline and the extra legs) and the rocprofv3 kernel stats of B = 1 (the one workload whose kernels changed after
# scripts/r05/gpu_final.sh ran: small_split).  Outputs in gpurun_out/r05final2; summaries are copied to profiles/r05_*.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r05final2; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -s > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|FAILED\|exit" | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.log | tail -4
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_cfg2.json
cut -c1-700 $O/bench_cfg2.json; echo
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_B1 -o ktrace -- python $R/bench.py --workload tetrapeptide_fwdsim_crop4_T1000_B1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extra --no-graph --streams 1 > $O/rocprof_B1.log 2>&1)
find $O/prof_B1 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cp {} '$O'/kernel_stats_B1_T1000.csv; head -9 {} | cut -c1-150'
rm -rf $O/prof_B1
