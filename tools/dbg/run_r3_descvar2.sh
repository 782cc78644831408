#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, "tests")
import conftest, oracle_lib as O
orbx = conftest._load_pkg()
import my_slam_amd.synth as synth
img = synth.texture(1, 640, 480)
k, d = orbx.ORBextractor(1000, max_width=640, max_height=480)(img)
ok, od, _ = O.Extractor(1000).extract(img)
print("n", len(k), len(ok), "kps equal", k.tobytes() == ok.tobytes())
x = np.unpackbits(d ^ od, axis=1)
print("rows differing", int(x.any(1).sum()), "bits differing per row (mean)", float(x.sum(1).mean()), "by byte position", x.reshape(len(d), 32, 8).sum((0, 2))[:32])
PY
cp tools/dbg/tmp/desc_head.hip.txt my-slam_amd/csrc/orbx_describe.hip
cd my-slam_amd && rm -f build/orbx_describe.o && make -s > /dev/null 2>&1 && cd ..
python -m pytest tests/test_extractor_gpu.py -m gpu -x -q -k "matches_oracle" 2>&1 | tail -1
