#!/bin/bash
# rocprofv3 memory-system counter passes for bench.py: vector / scalar memory latency as the SQ sees it (LEVEL / INSTS), L1 (TCP) and
# L2 (TCC) requests, hits and request latency, address-translation misses, scalar and instruction cache hit rates.
# (No TA_* pass: TA_TA_BUSY / TA_*_STALLED_BY_* left a dispatch incomplete and hung the profiler on this pool.)
# Each pass is its own run with --kernel-trace only.  Output: gpurun_out/pmc/<tag>_*.csv ; usage: tools/prof_mem.sh <tag> [bench args...]
set -u
tag=${1:-mem}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
BARGS=("$@")
run() { # name counters...
  local name=$1; shift
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc -o ${tag}_${name} -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs "${BARGS[@]}" > gpurun_out/pmc/${tag}_${name}.log 2>&1 || echo "pass $name failed"
}
run lvl SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVES
run ifetch SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
run tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tcp3 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run tcc1 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
run tcc2 TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_RDREQ_LEVEL_sum
run sqc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_TC_STALL
ls gpurun_out/pmc | grep -c counter_collection
