#!/bin/bash
# Round-2 evidence for profiles/: bench line, rocprofv3 per-kernel stats of the same command, PMC passes, HBM traffic.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 400 python3 bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > $O/bench_prof.json 2> $O/bench_prof.err; echo "stats rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b8_1080p -- python3 bench.py --width 1920 --height 1080 --nfeatures 4000 --batch 8 --steps 30 --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > $O/bench_1080p.json 2> $O/bench_1080p.err; echo "stats1080 rc=$?"
bash tools/prof_pmc.sh r02 > $O/pmc.log 2>&1; python3 tools/pmc_summary.py r02 > $O/pmc_summary.txt 2>&1
bash tools/prof_traffic.sh > $O/traffic.log 2>&1; cp gpurun_out/traffic/traffic.json $O/traffic.json 2>/dev/null
tail -c 2500 $O/bench_line.json
