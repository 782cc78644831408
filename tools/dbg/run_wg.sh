#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3wg; mkdir -p $O
SKIP_TESTS=0 bash tools/dbg/run_ab_stats.sh default dw1 dw2 dw8 ft64 ft128 ft512 2>&1 | tee $O/ab.txt
bash tools/prof_occ.sh r3occ > $O/occ.log 2>&1; cat $O/occ.log; python3 tools/pmc_summary.py r3occ > $O/occ_summary.txt 2>&1; grep -c mean $O/occ_summary.txt
