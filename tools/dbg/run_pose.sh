set -e
timeout -k 10 300 bash tools/track/run.sh > gpurun_out/track_cxx.log 2>&1 || { tail -20 gpurun_out/track_cxx.log; exit 1; }
cat gpurun_out/track_cxx.log
