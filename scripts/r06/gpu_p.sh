#!/bin/bash
# round 6, call P: the rest of the parity subset with the 32-row form of the L = 4 sub-layer kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06p; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 2000 python -m pytest tests -q -m gpu -s -k "small_ or split or registry or forward_vs or golden or fwd or inference or cfg1 or tps or ipa_table or headline or rollout or graph or row_owner or fold or stress" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit" | cut -c1-300 | tail -12
