#!/usr/bin/env python3
"""Inputs of tools/track/track_harness: the synthetic 1241x376 stream (my_slam_amd.synth, seed 5) as a raw file and a
synthetic 10^4-word vocabulary in DBoW2's text format; plus the layered (three-depth) stream and its layer map for the
pose stages.  usage: prep_inputs.py outdir [W H nframes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa
import my_slam_amd.synth as synth
from test_vocabulary import make_vocabulary
out = sys.argv[1]
W, H, K = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (1241, 376, 40)
os.makedirs(out, exist_ok=True)
synth.stream(5, W, H, K).tofile(os.path.join(out, "frames.raw"))
lf, layer = synth.stream_layers(5, W, H, K, shifts=(2, 4, 6))        # the pose stages' scene: three depths
lf.tofile(os.path.join(out, "frames_layers.raw"))
layer.tofile(os.path.join(out, "layer.raw"))
make_vocabulary(os.path.join(out, "voc.txt"), 10, 4, seed=1)
print(W, H, K)
