set -e
timeout -k 10 300 bash tools/track/run.sh > gpurun_out/track_cxx.log 2>&1 || { tail -20 gpurun_out/track_cxx.log; exit 1; }
cat gpurun_out/track_cxx.log
timeout -k 10 300 python tools/bench_track.py > gpurun_out/bench_track.json 2>gpurun_out/bench_track.err || { tail -20 gpurun_out/bench_track.err; exit 1; }
cat gpurun_out/bench_track.json
