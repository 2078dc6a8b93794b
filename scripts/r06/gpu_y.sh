#!/bin/bash
# round 6, call Y: training step A/B of library options given as arguments ("name=value" each; "base" = defaults), three rounds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06y; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for rep in 1 2 3; do
  for f in "$@"; do
    a=$f; [ $f = base ] && a=""
    echo "$f $(timeout 300 python scripts/train_bench.py 1 250 256 10 16 $a 2>&1 | tail -1)" | tee -a $O/train_ab.txt
  done
done
