#!/bin/bash
# rocprofv3 occupancy / dispatch counter passes for bench.py: waves in flight (SQ_LEVEL_WAVES over SQ_BUSY_CYCLES), VALU issue cycles,
# and what the workgroup dispatcher (SPI) waited for.  Each pass is its own run with --kernel-trace only.
# usage: tools/prof_occ.sh <tag> [bench args...]
set -u
tag=${1:-occ}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
BARGS=("$@")
run() { local name=$1; shift
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc -o ${tag}_${name} -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs "${BARGS[@]}" > gpurun_out/pmc/${tag}_${name}.log 2>&1 || { echo "pass $name failed"; return 1; }
}
run occ1 SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE || exit 1
run occ2 SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY || exit 1
run spi1 SPI_CSN_BUSY SPI_CSN_WAVE SPI_CSN_NUM_THREADGROUPS SPI_RA_REQ_NO_ALLOC_CSN || exit 1
run spi2 SPI_RA_RES_STALL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN || exit 1
run spi3 SPI_RA_SGPR_SIMD_FULL_CSN SPI_RA_BAR_CU_FULL_CSN SPI_RA_TGLIM_CU_FULL_CSN SPI_RA_WVLIM_STALL_CSN || exit 1
