#!/bin/bash
# HBM traffic (rocprofv3 PMC, FETCH_SIZE and WRITE_SIZE in separate passes) of the training step's big kernels at ATLAS
# 256 x 250 per GPU, train_precision 16, grouped by (kernel, grid).  FETCH_SIZE is doubled (gfx950 wide-read correction).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
RE="${1:-k16_linear_fast|k16_dw|k16_attn}"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_train_$c
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-include-regex "$RE" --output-format csv -d $O/pmc_train_$c -o pmc -- python $R/scripts/train_bench.py 1 250 256 1 16 > $O/pmc_train_$c.log 2>&1 < /dev/null)
done
python - "$O" <<'PY'
import sys, glob, csv, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{O}/pmc_train_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[(r["Kernel_Name"][:48], int(r["Grid_Size"]))][c].append(float(r["Counter_Value"]))
for (k, g), d in sorted(acc.items(), key=lambda kv: -kv[0][1]):
    f = d.get("FETCH_SIZE", [0]); w = d.get("WRITE_SIZE", [0])
    print(f"{k:48s} grid {g:8d} n={len(f):3d} read {2 * sum(f) / len(f) / 1024:9.1f} MiB  write {sum(w) / len(w) / 1024:9.1f} MiB")
PY
