set -e
cd "$GRAFT_REPO_ROOT/my-slam_amd"
rm -f build/orbx_octree.o
make -s HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DOCT_TRACE" > /dev/null 2>&1
cd ..
timeout -k 10 200 python tools/dbg/oct_trace.py 1920 1080 4000 > gpurun_out/oct_trace_1080.txt 2>&1
cat gpurun_out/oct_trace_1080.txt
timeout -k 10 200 python tools/dbg/oct_trace.py 640 480 1000 > gpurun_out/oct_trace_480.txt 2>&1
cat gpurun_out/oct_trace_480.txt
