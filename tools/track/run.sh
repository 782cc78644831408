#!/bin/bash
# builds and runs the C++ tracking harness on the GPU box; prints its JSON line
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
g++ -O2 -std=c++17 -I include tools/track/track_harness.cc -L my-slam_amd/lib -lorbx -Wl,-rpath,"$PWD/my-slam_amd/lib" -o /tmp/track_harness
python3 tools/track/prep_inputs.py /tmp/track_in 1241 376 40 > /dev/null
/tmp/track_harness /tmp/track_in/frames.raw 1241 376 40 /tmp/track_in/voc.txt 2000 2
