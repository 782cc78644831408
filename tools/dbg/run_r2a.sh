#!/bin/bash
# GPU call: VALU issue-rate microbenchmark + A/B of the FAST queue order (stage times, phase clocks, LDS counters)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2a; mkdir -p $O
V=my-slam_amd/lib/variants
tools/ubench/valu_issue 2000 > $O/valu_issue.json 2> $O/valu_issue.err; echo "ubench rc=$?"
for v in base rowmajor; do
  ORBX_LIB=$PWD/$V/liborbx_$v.so python3 tools/dbg/ab_run.py 640 480 1000 64 30 2>&1 | tail -1 | tee -a $O/ab.jsonl
done
for v in base rowmajor; do
  echo "== $v" >> $O/phase.txt
  ORBX_LIB=$PWD/$V/liborbx_${v}_trace.so python3 tools/dbg/phase_trace.py 640 480 1000 64 >> $O/phase.txt 2>&1
done
for v in base rowmajor; do
  ORBX_LIB=$PWD/$V/liborbx_$v.so rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc -o ${v}_sq2 -- python3 tools/dbg/ab_run.py 640 480 1000 64 3 > $O/pmc_$v.log 2>&1 || echo "pmc $v failed"
done
python3 tools/pmc_summary.py base $O/pmc > $O/pmc_base.txt 2>&1
python3 tools/pmc_summary.py rowmajor $O/pmc > $O/pmc_rowmajor.txt 2>&1
cat $O/phase.txt
grep -A20 "k_fast" $O/pmc_base.txt | head -12; grep -A20 "k_fast" $O/pmc_rowmajor.txt | head -12
