#!/bin/bash
# rocprofv3 counter passes for bench.py (run on the GPU box through gpurun).  Each pass is its own
# run with --kernel-trace only (never combined with sys/runtime traces).  Output: gpurun_out/pmc/<tag>_*.csv
# usage: tools/prof_pmc.sh <tag> [bench args...]
set -u
tag=${1:-pmc}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
run() { # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc -o ${tag}_${name} -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs "${BARGS[@]}" > gpurun_out/pmc/${tag}_${name}.log 2>&1 || echo "pass $name failed"
}
BARGS=("$@")
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq3 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
ls gpurun_out/pmc | head -40
