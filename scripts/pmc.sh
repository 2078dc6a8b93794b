#!/bin/bash
# PMC pass(es) over kbench (S=1): usage scripts/pmc.sh "<regex>" "CTR1 CTR2 ..." ["CTR ..." more passes]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
RE="$1"; shift
i=0
for ctrs in "$@"; do
  i=$((i+1))
  (cd /tmp && timeout 400 rocprofv3 --pmc $ctrs --kernel-include-regex "$RE" --output-format csv -d $R/gpurun_out/pmc$i -o pmc -- python $R/scripts/kbench.py ${WL:-tetrapeptide_fwdsim_crop4_T1000_B16} 1 > $R/gpurun_out/pmc$i.log 2>&1)
  python - "$R/gpurun_out/pmc$i" <<'PY'
import sys, glob, csv, collections
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in fs:
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
PY
done
