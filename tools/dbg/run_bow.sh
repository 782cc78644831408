set -e
timeout -k 10 400 python -m pytest tests/test_vocabulary.py tests/test_track_harness_gpu.py -x -q > gpurun_out/bow_tests.log 2>&1 || { tail -30 gpurun_out/bow_tests.log; exit 1; }
tail -2 gpurun_out/bow_tests.log
timeout -k 10 300 bash tools/track/run.sh > gpurun_out/track_cxx.log 2>&1 || { tail -20 gpurun_out/track_cxx.log; exit 1; }
cat gpurun_out/track_cxx.log
