#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2j; mkdir -p $O
ORBX_BATCH_TRACE=1 python3 - 2>&1 <<'PY' | tail -4
import sys, os, time
sys.path.insert(0, "tests"); import conftest
import numpy as np
import my_slam_amd as M, my_slam_amd.synth as synth
fr = synth.stream(4, 640, 480, 64)
ex = M.ORBextractor(1000, max_width=640, max_height=480, max_batch=64)
for _ in range(4): ex.extract_batch_raw(fr)
PY
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc -o mf_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > /dev/null 2> $O/$name.err || echo "pass $name failed"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_I8
run grbm GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py mf $O/pmc | grep -A40 "k_best2_mfma" | head -45
