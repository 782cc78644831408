#!/bin/bash
cd "$GRAFT_REPO_ROOT"
T=/tmp/trk; mkdir -p $T
python tools/track/prep_inputs.py $T 1241 376 40 > /dev/null
g++ -O2 -std=c++17 -I include tools/track/track_harness.cc -L my-slam_amd/lib -lorbx -Wl,-rpath,$PWD/my-slam_amd/lib -o $T/h
for m in 0 8 16 24; do echo "== margin $m"; ORBX_TRACK_EDGE_MARGIN=$m $T/h $T/frames_layers.raw 1241 376 40 $T/voc.txt 2000 2 $T/layer.raw 0.5 2 4 6 | head -1 | cut -c1-700; done
