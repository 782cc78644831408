#!/bin/bash
# Evidence of one round for profiles/: bench line, rocprofv3 per-kernel stats of the same command (and of the 8 x 1080p shape),
# PMC passes (each its own run, --kernel-trace only), HBM traffic passes.  usage (on the GPU box): tools/prof_round.sh r03
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag; mkdir -p $O
timeout -k 10 500 python3 bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > $O/bench_prof.json 2> $O/bench_prof.err; echo "stats rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b8_1080p -- python3 bench.py --width 1920 --height 1080 --nfeatures 4000 --batch 8 --steps 30 --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > $O/bench_1080p.json 2> $O/bench_1080p.err; echo "stats1080 rc=$?"
bash tools/prof_pmc.sh $tag > $O/pmc.log 2>&1; python3 tools/pmc_summary.py $tag > $O/pmc_summary.txt 2>&1
bash tools/prof_traffic.sh > $O/traffic.log 2>&1; cp gpurun_out/traffic/traffic.json $O/traffic.json 2>/dev/null
find $O/stats -name "*kernel_stats.csv" | head
tail -c 1500 $O/bench_line.json
