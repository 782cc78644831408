set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof1080 -o s1080 -- python3 $R/tools/prof_shape.py 1920 1080 4000 8 20 > $R/gpurun_out/prof1080.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof1241 -o s1241 -- python3 $R/tools/prof_shape.py 1241 376 2000 32 20 > $R/gpurun_out/prof1241.log 2>&1
cd $R
find gpurun_out/prof1080 gpurun_out/prof1241 -name "*kernel_stats.csv" | head
for f in $(find gpurun_out/prof1080 gpurun_out/prof1241 -name "*kernel_stats.csv"); do echo $f; cut -d, -f1-6 $f | head -16; done
