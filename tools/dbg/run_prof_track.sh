set -e
R=$PWD
timeout -k 10 600 python -m pytest tests/test_vocabulary.py -x -q > gpurun_out/bow_tests.log 2>&1 || { tail -30 gpurun_out/bow_tests.log; exit 1; }
tail -1 gpurun_out/bow_tests.log
g++ -O2 -std=c++17 -I include tools/track/track_harness.cc -L my-slam_amd/lib -lorbx -Wl,-rpath,"$PWD/my-slam_amd/lib" -o /tmp/track_harness
python3 tools/track/prep_inputs.py /tmp/track_in 1241 376 40 > /dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/proftrack
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/proftrack -o trk -- /tmp/track_harness /tmp/track_in/frames_layers.raw 1241 376 40 /tmp/track_in/voc.txt 2000 2 /tmp/track_in/layer.raw 0.5 2 4 6 > $R/gpurun_out/proftrack.log 2>&1
cd $R
python3 - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/proftrack/trk_kernel_stats.csv")):
    if "bow" in r["Name"]:
        print("  %-44s calls %4s avg %8.1f us  min %8.1f max %8.1f" % (r["Name"][:44], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
timeout -k 10 300 bash tools/track/run.sh > gpurun_out/track_cxx.log 2>&1 || { tail -20 gpurun_out/track_cxx.log; exit 1; }
cat gpurun_out/track_cxx.log
