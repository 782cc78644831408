set -e
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
timeout -k 10 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || { tail -20 gpurun_out/bench_final.err; exit 1; }
cat gpurun_out/bench_final.json
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
