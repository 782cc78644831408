#!/bin/bash
# builds and runs the C++ tracking harness on the GPU box; prints its JSON line
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
g++ -O2 -std=c++17 -I include tools/track/track_harness.cc -L my-slam_amd/lib -lorbx -Wl,-rpath,"$PWD/my-slam_amd/lib" -o /tmp/track_harness
python3 tools/track/prep_inputs.py /tmp/track_in 1241 376 40 > /dev/null
/tmp/track_harness /tmp/track_in/frames.raw 1241 376 40 /tmp/track_in/voc.txt 2000 2
# the same loop with the pose stages on the three-depth scene (baseline 0.5 m, layer shifts 2/4/6 px per frame)
/tmp/track_harness /tmp/track_in/frames_layers.raw 1241 376 40 /tmp/track_in/voc.txt 2000 2 /tmp/track_in/layer.raw 0.5 2 4 6
