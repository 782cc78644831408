set -e
timeout -k 10 600 python -m pytest tests/test_frame_grid.py tests/test_track_harness_gpu.py -x -q > gpurun_out/sfi.log 2>&1 || { tail -40 gpurun_out/sfi.log; exit 1; }
tail -1 gpurun_out/sfi.log
timeout -k 10 300 bash tools/track/run.sh > gpurun_out/track_cxx.log 2>&1 || { tail -20 gpurun_out/track_cxx.log; exit 1; }
cat gpurun_out/track_cxx.log
