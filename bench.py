#!/usr/bin/env python3
"""bench.py -- ORB extract(+match) throughput on MI355X, BASELINE.json's metric.

One step = one pass of the hot path over one batch of synthetic frames that are already resident in
HBM: 8-level pyramid -> per-cell FAST -> quadtree -> orientation + blur + rBRIEF for every frame of
the batch, then best/second-best Hamming matching of every frame against its predecessor, all
through the C ABI of my-slam_amd/lib/liborbx.so.

--gpus N: one process per GPU.  Started by a launcher (torch.distributed.run sets WORLD_SIZE) the
process is one rank; started plainly with --gpus N > 1 it starts the N ranks itself (before anything
touches the GPU) and relays rank 0's line.  The ranks cut ONE synthetic stream into contiguous blocks
(my-slam_amd/shard.py): --scaling weak = --batch frames per GPU (default), strong = --batch frames in
total (BASELINE configs[3]: 64 frames over 1/2/4/8 GPUs).  No data-path collective: per step one
boundary frame (<= 62 KB) goes to the next rank for the pair that straddles two blocks, and one
gather brings the keypoint/descriptor/match block back to rank 0.

Prints ONE JSON line on rank 0 (contract in the task prompt): value = keypoints/s over all ranks.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
METRIC = "ORB keypoints/s + frames/s, 1000 feats/frame @640x480, 1/2/4/8 GPU"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step (weak) / frames in total (strong)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--subbatches", type=int, default=0, help="0 = library default")
    ap.add_argument("--step-graph", type=int, default=-1, help="1: one HIP graph launch per step (N = 1); 0: kernel by kernel; -1: default (kernel by kernel; measured equal)")
    ap.add_argument("--no-match", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the extra two-stream / HIP-graph throughput pass")
    ap.add_argument("--no-host-api", action="store_true", help="skip the host-buffer API (PCIe inclusive) measurement")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the other BASELINE configs (one 640x480 frame, 8x1920x1080 n=4000 extract+match, the 1241x376 tracking loop) and the realistic-density stream")
    ap.add_argument("--pipelined-streams", type=int, default=2)
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    return ap.parse_args(argv)


def launch_ranks(args):
    """--gpus N > 1 without a launcher: start N ranks with torch.distributed.run.  Nothing in this process has touched
    the GPU (torch is not even imported), the ranks are ordinary children, rank 0's JSON line is relayed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in p.stdout.splitlines():
        s = ln.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        elif s:
            print(ln, file=sys.stderr)
    if p.returncode != 0 or line is None:
        print("bench.py: the %d-rank run failed (exit code %d%s)" % (args.gpus, p.returncode, "" if line else ", no result line"), file=sys.stderr)
        return p.returncode or 1
    print(line)
    return 0


def load_pkg():
    if "my_slam_amd" in sys.modules:
        return sys.modules["my_slam_amd"]
    spec = importlib.util.spec_from_file_location(
        "my_slam_amd", os.path.join(ROOT, "my-slam_amd", "__init__.py"),
        submodule_search_locations=[os.path.join(ROOT, "my-slam_amd")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["my_slam_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def level_geometry(W, H, nlevels=8, scale=1.2):
    import numpy as np
    sf = np.float32(1.0)
    out = []
    for l in range(nlevels):
        inv = np.float32(1.0) / sf
        out.append((int(np.rint(np.float32(W) * inv)), int(np.rint(np.float32(H) * inv))))
        sf = np.float32(sf * np.float32(scale))
    return out


def algorithmic_bytes(W, H, nlevels, n_kp, n_cand, n_q, n_t):
    """SURVEY.md 8(d) B_extract = 5P - px_last + 1321 N, split per stage (DESIGN.md 'Kernels')."""
    lv = level_geometry(W, H, nlevels)
    px = [w * h for (w, h) in lv]
    P = sum(px)
    return {
        "pyramid": W * H + (P - px[-1]) + (P - px[0]),
        "fast": P,
        "quadtree": 12 * n_cand + 8 * n_kp,            # not in the survey's formula: candidates in, keypoints out
        "describe": 2 * P + n_kp * (749 + 512 + 32 + 28),
        "match": (n_q + n_t) * 32 + n_q * 12,
    }


def cpu_baseline(frames, nfeatures, budget_s=12.0, all_cores_s=8.0):
    """Oracle (CPU restatement, -O3 -march=native, 1 thread) on a bounded sample of the same frames."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    ex = O.Extractor(nfeatures, native=True)
    t0 = time.perf_counter()
    nk = 0
    nf = 0
    prev = None
    while True:
        img = frames[nf % len(frames)]
        kps, desc, _ = ex.extract(img)
        if prev is not None:
            O.match_dense(desc, kps["angle"], prev[1], prev[0]["angle"], 50, 0.9, True)
        prev = (kps, desc)
        nk += len(kps)
        nf += 1
        el = time.perf_counter() - t0
        if el > budget_s or nf >= 4096:
            break
    out = {"value": nk / el, "unit": "keypoints/s", "cores": 1, "kind": "port",
           "sample": "%d frames (extract + match vs previous frame) in %.1f s, 1 thread, oracle -O3 -march=native -ffp-contract=off; %.2f frames/s"
                     % (nf, el, nf / el),
           "host_cpus": os.cpu_count()}
    # the same work frame-parallel over the host cores this process may use (one extractor per thread;
    # the oracle calls release the GIL).  Reported beside the 1-thread figure, never instead of it.
    try:
        ncore = len(os.sched_getaffinity(0))
    except Exception:
        ncore = os.cpu_count() or 1
    try:   # a container's CPU quota (cgroup v2 cpu.max = "<quota> <period>") is the real core count
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            ncore = max(1, min(ncore, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    ncore = min(ncore, 64)
    if ncore > 1 and all_cores_s > 0:
        import threading
        tot = [0, 0]
        lock = threading.Lock()
        t1 = time.perf_counter()

        def worker(tid):
            e = O.Extractor(nfeatures, native=True)
            i, k_loc, f_loc, prv = tid, 0, 0, None
            while time.perf_counter() - t1 < all_cores_s:
                kp, de, _ = e.extract(frames[i % len(frames)])
                if prv is not None:
                    O.match_dense(de, kp["angle"], prv[1], prv[0]["angle"], 50, 0.9, True)
                prv = (kp, de)
                k_loc += len(kp); f_loc += 1; i += ncore
            with lock:
                tot[0] += k_loc; tot[1] += f_loc
        th = [threading.Thread(target=worker, args=(t,)) for t in range(ncore)]
        for t in th: t.start()
        for t in th: t.join()
        el2 = time.perf_counter() - t1
        out["all_cores"] = {"value": tot[0] / el2, "unit": "keypoints/s", "cores": ncore,
                            "sample": "%d frames in %.1f s on %d threads; %.1f frames/s" % (tot[1], el2, ncore, tot[1] / el2)}
    return out


class DevicePath:
    """The hot path on one GPU for `B` resident frames: extractor + matcher handles and the flat result blocks of
    my-slam_amd/shard.py.  step() = shard.run_step with the HIP library behind both callbacks."""

    def __init__(self, pkg, torch, shard, frames, W, H, B, NF, device, do_match, rank=0, world=1, nslot=1, stream=None):
        self.pkg, self.torch, self.shard = pkg, torch, shard
        self.frames, self.W, self.H, self.B, self.rank, self.world = frames, W, H, B, rank, world
        self.ex = pkg.ORBextractor(NF, 1.2, 8, 20, 7, device=device, max_width=W, max_height=H, max_batch=B)
        self.cap = self.ex.cap
        self.layout = shard.FlatLayout(B, self.cap)
        self.bufs = [self.layout.alloc("cuda") for _ in range(nslot)]
        self.status = torch.zeros(B, dtype=torch.int32, device="cuda")
        self.do_match = do_match
        self.mt = pkg.ORBmatcher(0.9, True, device=device, max_queries=self.cap, max_train=self.cap, max_pairs=1) if do_match else None
        self.stream = stream
        self.nown = B

    def extract_into(self, buf, s):
        kps, desc, _, counts, _ = self.layout.views(buf)
        fr = self.frames
        self.ex.extract_batch_device(fr.data_ptr(), self.nown, self.W, self.H, fr.stride(1), fr.stride(0),
                                     kps[1].data_ptr(), desc[1].data_ptr(), counts[1:].data_ptr(), self.status.data_ptr(), s)

    def match_slots(self, buf, first, npairs, s):
        if not self.do_match or npairs < 1:
            return
        kps, desc, m12, counts, nmatch = self.layout.views(buf)
        self.mt.match_batch_device(desc[first].data_ptr(), kps[first].data_ptr(), counts[first:].data_ptr(),
                                   desc[first - 1].data_ptr(), kps[first - 1].data_ptr(), counts[first - 1:].data_ptr(),
                                   self.cap, npairs, m12[first].data_ptr(), nmatch[first:].data_ptr(), stream=s)

    def step(self, k=0, group=None, comm=None):
        buf, s = self.bufs[k], self.stream.cuda_stream
        return self.shard.run_step(self.layout, buf, self.rank, self.world, self.nown,
                                   lambda *_: self.extract_into(buf, s), lambda first, npairs: self.match_slots(buf, first, npairs, s),
                                   group=group, **(comm or {}))


def pipelined_pass(pkg, torch, shard, frames, W, H, B, NF, device, do_match, steps, nk_per_step, nstreams=2):
    """Throughput of the same steps issued round-robin on `nstreams` streams, each with its own handles and buffers and one
    captured HIP graph per step (13 kernels, one host call)."""
    pipes = []
    for _ in range(nstreams):
        p = DevicePath(pkg, torch, shard, frames, W, H, B, NF, device, do_match, stream=torch.cuda.Stream())
        pipes.append(p)
    for p in pipes:                     # size the workspaces, then capture
        with torch.cuda.stream(p.stream):
            p.step(); p.step()
    torch.cuda.synchronize()
    for p in pipes:
        p.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(p.graph, stream=p.stream):
            p.step()
    ends = [torch.cuda.Event() for _ in range(steps + 2 * nstreams)]

    def run(n, base):
        for i in range(n):
            p = pipes[i % nstreams]
            if i > nstreams:            # the hardware queues do not arbitrate fairly: keep the streams within a step of each other
                p.stream.wait_event(ends[base + i - nstreams - 1])
            with torch.cuda.stream(p.stream):
                p.graph.replay()
            ends[base + i].record(p.stream)
    run(2 * nstreams, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps, 2 * nstreams)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    for p in pipes:
        counts = p.layout.views(p.bufs[0])[3]
        if int(p.status.abs().sum().item()) != 0 or int(counts[1:].sum().item()) != nk_per_step:
            raise RuntimeError("pipelined pass produced different results")
    return {"streams": nstreams, "hip_graph_per_step": True, "steps": steps, "ms_per_step": round(el * 1e3, 4),
            "frames_per_s": round(B / el, 1), "keypoints_per_s": round(nk_per_step / el, 1)}


def host_api_pass(pkg, frames_np, W, H, NF, device, reps=12):
    """The reference's own call shape: host images in, host keypoints/descriptors out (orbx_extract_batch), PCIe
    inclusive.  Reported beside the contract line, never as `value`."""
    import numpy as np
    B = len(frames_np)
    ex = pkg.ORBextractor(NF, 1.2, 8, 20, 7, device=device, max_width=W, max_height=H, max_batch=B)
    if os.environ.get("ORBX_BENCH_CHUNK"):   # A/B switch: frames per pipeline chunk (library default 16)
        ex.set_batch_chunk(int(os.environ["ORBX_BENCH_CHUNK"]))
    for _ in range(3):
        ex.extract_batch_raw(frames_np)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        counts = ex.extract_batch_raw(frames_np)[2]
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    out = {"entry": "orbx_extract_batch (host buffers in and out, blocking)", "frames": B, "ms_per_batch": round(med * 1e3, 4),
           "ms_per_64_frames": round(med * 1e3 * 64 / B, 4), "frames_per_s": round(B / med, 1),
           "keypoints_per_s": round(float(counts.sum()) / med, 1), "pcie_bytes_per_batch": int(frames_np.nbytes + counts.sum() * 60 + 8 * B)}
    # the same call with the caller's frames in page-locked memory (a capture pipeline's buffers): the library uploads them where
    # they lie instead of repacking them into its own staging block first
    try:                                   # a side measurement must never cost the contract line
        import torch
        pinned = torch.from_numpy(frames_np).pin_memory().numpy()
        for _ in range(3):
            ex.extract_batch_raw(pinned)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            counts2 = ex.extract_batch_raw(pinned)[2]
            ts.append(time.perf_counter() - t0)
        if not (counts2 == counts).all():
            raise RuntimeError("page-locked input produced different keypoint counts")
        medp = float(np.median(ts))
        out["page_locked_input"] = {"ms_per_batch": round(medp * 1e3, 4), "frames_per_s": round(B / medp, 1)}
    except Exception as e:                 # noqa: BLE001
        out["page_locked_input"] = {"error": str(e)[:200]}
    return out


def extra_config3(pkg, torch, shard, synth, device, steps=20, warmup=3):
    """BASELINE configs[2]: 1920x1080, nFeatures=4000, extract + dense match against the previous frame; 8 frames per step."""
    W, H, NF, B = 1920, 1080, 4000, 8
    frames = torch.from_numpy(synth.stream(2, W, H, B)).cuda()
    st = torch.cuda.Stream()
    p = DevicePath(pkg, torch, shard, frames, W, H, B, NF, device, True, stream=st)
    with torch.cuda.stream(st):
        for _ in range(warmup):
            p.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            p.step()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
    _, _, _, counts, nmatch = p.layout.views(p.bufs[0])
    if int(p.status.abs().sum().item()) != 0:
        raise RuntimeError("device status nonzero")
    nk = int(counts[1:].sum().item())
    p.ex.set_profiling(True)
    import numpy as np
    acc = np.zeros(4)
    for _ in range(3):
        p.extract_into(p.bufs[0], st.cuda_stream)
        acc += p.ex.stage_ms()
    p.ex.set_profiling(False)
    acc /= 3
    ab = algorithmic_bytes(W, H, 8, nk / B, 0, 0, 0)
    whole = (ab["pyramid"] + ab["fast"] + ab["describe"]) * B / el / 1e9
    return {"workload": "8 x 1920x1080 nFeatures=4000, extract + dense match vs previous frame (BASELINE configs[2]), resident in HBM",
            "steps": steps, "ms_per_step": round(el * 1e3, 4), "frames_per_s": round(B / el, 1), "keypoints_per_s": round(nk / el, 1),
            "keypoints_per_frame": round(nk / B, 1), "matches_per_step": int(nmatch[2:].sum().item()),
            "stage_ms": {k: round(float(v), 4) for k, v in zip(("pyramid", "fast", "quadtree", "describe"), acc)},
            "whole_path_algorithmic_GBps": round(whole, 1), "frac_of_hbm_peak": round(whole / 8000.0, 4)}


def extra_survey_stream(pkg, torch, shard, synth, device, steps=20, warmup=3):
    """The headline step (64 x 640x480, n = 1000, extract + match) on SURVEY.md 8(d)'s texture AS WRITTEN (16-px blocks, 0.002*W*H
    rectangles of 8..64 px, one 3x3 box blur): about half the corner density of the headline stream, whose texture is corner-dense
    on purpose.  FAST's second stage and the quadtree scale with that density."""
    import numpy as np
    W, H, NF, B = 640, 480, 1000, 64
    frames = torch.from_numpy(synth.stream_survey(4, W, H, B)).cuda()
    st = torch.cuda.Stream()
    p = DevicePath(pkg, torch, shard, frames, W, H, B, NF, device, True, stream=st)
    with torch.cuda.stream(st):
        for _ in range(warmup):
            p.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            p.step()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
    _, _, _, counts, nmatch = p.layout.views(p.bufs[0])
    if int(p.status.abs().sum().item()) != 0:
        raise RuntimeError("device status nonzero")
    nk = int(counts[1:].sum().item())
    p.ex.set_profiling(True)
    acc = np.zeros(4)
    for _ in range(3):
        p.extract_into(p.bufs[0], st.cuda_stream)
        acc += p.ex.stage_ms()
    p.ex.set_profiling(False)
    acc /= 3
    ncand = sum(len(p.ex.candidates(0, l)) for l in range(8))
    return {"workload": "64 x 640x480 nFeatures=1000, extract + dense match vs previous frame on SURVEY.md 8(d)'s texture as written "
                        "(realistic corner density), resident in HBM",
            "steps": steps, "ms_per_step": round(el * 1e3, 4), "frames_per_s": round(B / el, 1), "keypoints_per_s": round(nk / el, 1),
            "keypoints_per_frame": round(nk / B, 1), "fast_candidates_per_frame": ncand, "matches_per_step": int(nmatch[2:].sum().item()),
            "stage_ms": {k: round(float(v), 4) for k, v in zip(("pyramid", "fast", "quadtree", "describe"), acc)}}


def extra_single_frame(pkg, torch, synth, device, reps=200):
    """BASELINE configs[1] as written: ONE 640x480 frame, nFeatures = 1000.  Latency (p50 / p90) of (a) the device-resident call
    (frame already in HBM, results stay there: one synchronisation per call) and (b) the reference's own call shape, host image in,
    host keypoints + descriptors out (orbx_extract on a one-frame handle: upload, ten kernels and download are one HIP graph), PCIe
    inclusive."""
    import numpy as np
    W, H, NF = 640, 480, 1000
    img = synth.stream(4, W, H, 1)[0]
    ex = pkg.ORBextractor(NF, 1.2, 8, 20, 7, device=device, max_width=W, max_height=H, max_batch=1)
    cap = ex.cap
    fr = torch.from_numpy(img[None]).cuda()
    k = torch.zeros((1, cap, 7), device="cuda"); d = torch.zeros((1, cap, 32), dtype=torch.uint8, device="cuda")
    c = torch.zeros(1, dtype=torch.int32, device="cuda"); s = torch.zeros(1, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()

    def dev_call():
        ex.extract_batch_device(fr.data_ptr(), 1, W, H, fr.stride(1), fr.stride(0), k.data_ptr(), d.data_ptr(), c.data_ptr(), s.data_ptr(), st.cuda_stream)
        st.synchronize()
    for _ in range(20):
        dev_call()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); dev_call(); ts.append(time.perf_counter() - t0)
    if int(s.item()) != 0:
        raise RuntimeError("device status nonzero")
    nk = int(c.item())
    dev = np.sort(np.array(ts)) * 1e3
    ex.set_profiling(True)
    acc = np.zeros(4)
    for _ in range(5):
        dev_call(); acc += ex.stage_ms()
    ex.set_profiling(False)
    for _ in range(20):
        ex(img)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); kk, dd = ex(img); ts.append(time.perf_counter() - t0)
    host = np.sort(np.array(ts)) * 1e3
    q = lambda a, f: round(float(a[min(len(a) - 1, int(f * len(a)))]), 4)
    return {"workload": "1 x 640x480 nFeatures=1000, extract (BASELINE configs[1] as written: a single frame)",
            "keypoints": nk, "device_resident_ms": {"p50": q(dev, 0.5), "p90": q(dev, 0.9)},
            "host_api_ms_pcie_inclusive": {"p50": q(host, 0.5), "p90": q(host, 0.9), "entry": "orbx_extract (one-frame handle: one HIP graph per call)"},
            "frames_per_s_device_resident": round(1e3 / q(dev, 0.5), 1), "keypoints_per_s_device_resident": round(nk * 1e3 / q(dev, 0.5), 1),
            "stage_ms": {n: round(float(v), 4) for n, v in zip(("pyramid", "fast", "quadtree", "describe"), acc / 5)}}


def extra_tracking_loop(pkg, synth, nframes=40):
    """BASELINE configs[4]: the Tracking-shaped per-frame loop (extract -> grid -> BoW -> SearchByBoW -> [PoseOptimization, EPnP RANSAC
    relocalisation every 8th frame] -> SearchByProjection) on 1241x376 frames, nFeatures = 2000, host side in C++ through the C ABI
    (tools/track/track_harness.cc, compiled here with g++; an ordinary child process).  Two runs: without and with the pose stages."""
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_vocabulary import make_vocabulary
    W, H, NF = 1241, 376, 2000
    tmp = tempfile.mkdtemp(prefix="orbx_track_")
    try:
        exe = os.path.join(tmp, "track_harness")
        libdir = os.path.dirname(pkg.LIB_PATH)
        subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "track", "track_harness.cc"),
                        "-L", libdir, "-lorbx", "-Wl,-rpath," + libdir, "-o", exe], check=True, capture_output=True, timeout=300)
        # the inputs of tools/track/prep_inputs.py: the plain stream for the loop without pose, the three-depth scene for the pose
        # stages, a synthetic 10^4-word vocabulary (k = 10, L = 4; FeatureVector nodes two levels up: 100 nodes)
        synth.stream(5, W, H, nframes).tofile(os.path.join(tmp, "frames.raw"))
        frames, layer = synth.stream_layers(5, W, H, nframes, shifts=(2, 4, 6))
        frames.tofile(os.path.join(tmp, "frames_layers.raw")); layer.tofile(os.path.join(tmp, "layer.raw"))
        make_vocabulary(os.path.join(tmp, "voc.txt"), 10, 4, seed=1)
        tail = [str(W), str(H), str(nframes), os.path.join(tmp, "voc.txt"), str(NF), "2"]
        base = [exe, os.path.join(tmp, "frames.raw")] + tail
        base_pose = [exe, os.path.join(tmp, "frames_layers.raw")] + tail
        out = {"workload": "%d x 1241x376 nFeatures=2000, one frame at a time: extract -> grid -> BoW -> SearchByBoW -> SearchByProjection, "
                           "host side in C++ through the C ABI (BASELINE configs[4]); PCIe inclusive" % nframes}
        r = subprocess.run(base, check=True, capture_output=True, text=True, timeout=300).stdout.strip().splitlines()
        out["without_pose"] = json.loads(r[-1])
        r = subprocess.run(base_pose + [os.path.join(tmp, "layer.raw"), "0.5", "2", "4", "6"], check=True, capture_output=True, text=True,
                           timeout=300).stdout.strip().splitlines()
        out["with_pose"] = json.loads(r[-1])
        out["with_pose"]["pose"] = json.loads(r[-2])["pose"]
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: liborbx has no CPU path")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("ORBX_BENCH_BACKEND", "nccl")   # "gloo": rehearsal of the N-rank path on fewer GPUs than ranks
    if backend == "nccl" and world > ndev:
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible (set ORBX_BENCH_BACKEND=gloo to rehearse on fewer)" % (world, ndev))
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    dist = None
    # ORBX_BENCH_FORCE_COMM=1: take the communication branch with ONE rank too (process group of size 1, the per-step gather and a
    # self-addressed boundary exchange on the same device-resident block): warms the RCCL path on a one-GPU box
    force_comm = world == 1 and os.environ.get("ORBX_BENCH_FORCE_COMM", "0") == "1"
    comm_on = world > 1 or force_comm
    if comm_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_comm and "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    pkg = load_pkg()
    import my_slam_amd.shard as shard
    import my_slam_amd.synth as synth
    W, H, NF = args.width, args.height, args.nfeatures
    # ONE synthetic TUM-mono-like stream; every rank takes its contiguous block of it
    total = args.batch * world if args.scaling == "weak" else args.batch
    if total < world:
        raise SystemExit("bench.py: %d frames cannot be split over %d ranks" % (total, world))
    ranges = [shard.shard_range(total, world, r) for r in range(world)]
    lo, hi = ranges[rank]
    B = shard.max_shard(total, world)                 # block capacity (identical on every rank: one gather size)
    nown = hi - lo
    do_match = not args.no_match and total > 1
    frames_np = synth.stream(4, W, H, total, first=lo, count=nown)
    frames = torch.from_numpy(frames_np).cuda()

    # one explicit (non-default) stream carries the whole path, so extract -> exchange -> match -> gather are
    # ordered by the stream itself (a NULL stream would select each handle's private stream)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    nslot = 2 if comm_on else 1                       # two result blocks: the gather of step i overlaps step i+1
    path = DevicePath(pkg, torch, shard, frames, W, H, B, NF, local, do_match, rank, world, nslot, stream)
    path.nown = nown
    ex, cap, layout = path.ex, path.cap, path.layout
    if args.subbatches:
        ex.set_subbatches(args.subbatches)
    on_gpu = backend == "nccl"
    gather_bufs = None
    if comm_on and rank == 0:
        gather_bufs = [[torch.empty(layout.nbytes, dtype=torch.uint8, device="cuda" if on_gpu else "cpu") for _ in range(world)]
                       for _ in range(nslot)]
    # gloo rehearsal: communication goes through a host mirror of the block
    mirror = [torch.empty(layout.nbytes, dtype=torch.uint8) for _ in range(nslot)] if (comm_on and not on_gpu) else None
    pending = [None] * nslot
    sends = [[] for _ in range(nslot)]
    step_no = [0]
    self_xchg = [None]

    def step():
        k = step_no[0] % nslot
        step_no[0] += 1
        if pending[k] is not None:         # the gather that last read this block must be done before it is rewritten
            pending[k].wait()
            pending[k] = None
        for w in sends[k]:
            w.wait()
        comm = None
        if mirror is not None:
            def stage_out(k=k):
                stream.synchronize(); mirror[k].copy_(path.bufs[k])
            def stage_in(k=k):
                kv, dv, _, cv, _ = layout.views(path.bufs[k]); km, dm, _, cm, _ = layout.views(mirror[k])
                kv[0].copy_(km[0]); dv[0].copy_(dm[0]); cv[0:1].copy_(cm[0:1])
            comm = {"comm_buf": mirror[k], "stage_out": stage_out, "stage_in": stage_in}
        sends[k] = path.step(k, comm=comm)
        if force_comm and on_gpu and self_xchg[0] is not False:   # one rank: the boundary frame goes to itself (slot nown -> slot 0, which rank 0 never matches)
            kv, dv, _, cv, _ = layout.views(path.bufs[k])
            try:
                sends[k] = dist.batch_isend_irecv([dist.P2POp(dist.isend, kv[nown], 0), dist.P2POp(dist.isend, dv[nown], 0),
                                                   dist.P2POp(dist.isend, cv[nown:nown + 1], 0), dist.P2POp(dist.irecv, kv[0], 0),
                                                   dist.P2POp(dist.irecv, dv[0], 0), dist.P2POp(dist.irecv, cv[0:1], 0)])
                self_xchg[0] = True
            except Exception as e:         # noqa: BLE001 -- RCCL may refuse a send to the sending rank: the gather below is still exercised
                self_xchg[0] = False; self_xchg.append(str(e)[:200])
        if comm_on:                        # RCCL over xGMI: the block goes back to rank 0, nothing else crosses GPUs
            if on_gpu:
                pending[k] = shard.gather_flat(path.bufs[k], gather_bufs[k] if rank == 0 else None, async_op=True)
            else:
                stream.synchronize(); mirror[k].copy_(path.bufs[k])
                shard.gather_flat(mirror[k], gather_bufs[k] if rank == 0 else None)

    def drain():
        for k in range(nslot):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
            for w in sends[k]:
                w.wait()
            sends[k] = []

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if int(path.status.abs().sum().item()) != 0:
        raise SystemExit("device status nonzero: %s" % path.status.tolist())
    # One HIP graph launch per step (N = 1): the step's launches -- with --subbatches > 1 two or more concurrent branches, whose
    # latency-bound kernels (resize chain, quadtree, acceptance) run beside the VALU-bound ones of the other branch -- are captured
    # once and replayed; issued kernel by kernel the host needs ~5 us per launch.
    step_graph = None
    use_graph = args.step_graph == 1 or (args.step_graph < 0 and world == 1 and os.environ.get("ORBX_BENCH_STEP_GRAPH", "0") == "1")
    if use_graph and world == 1:
        step_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(step_graph, stream=stream):
            path.step(0)
        torch.cuda.set_stream(stream)
        plain_step = step

        def step():
            step_graph.replay()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        if int(path.status.abs().sum().item()) != 0:
            raise SystemExit("device status nonzero after graph replay: %s" % path.status.tolist())
    if comm_on:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-step GPU times for the p50
    t0 = time.perf_counter()
    evs[0].record(stream)
    for i in range(args.steps):
        step()
        evs[i + 1].record(stream)
    drain()                      # every gather has landed on rank 0 inside the timed region
    torch.cuda.synchronize()
    if comm_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if comm_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    pct = lambda q: step_ms[min(len(step_ms) - 1, int(q * len(step_ms)))]
    kps_v, desc_v, m12_v, counts_v, nmatch_v = layout.views(path.bufs[0])
    nk_local = int(counts_v[1:nown + 1].sum().item())
    nm_local = int(nmatch_v[(2 if rank == 0 else 1):nown + 1].sum().item()) if do_match else 0
    tot = torch.tensor([nk_local, nm_local], dtype=torch.int64, device="cuda" if (world == 1 or on_gpu) else "cpu")
    if comm_on:
        dist.all_reduce(tot)
    nk_all, nm_all = int(tot[0].item()), int(tot[1].item())
    gathered_ok = None
    if comm_on and rank == 0:     # what rank 0 holds after the last gather is the whole stream's result
        last = (step_no[0] - 1) % nslot
        gk = sum(int(layout.views(gather_bufs[last][r])[3][1:ranges[r][1] - ranges[r][0] + 1].sum().item()) for r in range(world))
        gathered_ok = gk == nk_all
        if not gathered_ok:
            raise SystemExit("gathered block holds %d keypoints, the ranks produced %d" % (gk, nk_all))

    # ---- per-stage GPU time (HIP events on the launch stream, inside the library) ----
    roof = None
    stage = {}
    s = stream.cuda_stream
    if rank == 0:
        ex.set_profiling(True)
        acc = np.zeros(4)
        tm = 0.0
        reps = max(3, min(10, args.steps))
        npairs = nown - 1
        for _ in range(reps):
            path.extract_into(path.bufs[0], s)
            acc += ex.stage_ms()
            if do_match and npairs > 0:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                path.match_slots(path.bufs[0], 2, npairs, s)
                e1.record()
                torch.cuda.synchronize()
                tm += e0.elapsed_time(e1)
        ex.set_profiling(False)
        acc /= reps
        stage = {"pyramid": acc[0], "fast": acc[1], "quadtree": acc[2], "describe": acc[3]}
        if do_match and npairs > 0:
            stage["match"] = tm / reps
        ncand = sum(len(ex.candidates(0, l)) for l in range(8))   # frame 0's candidates, representative
        nkf = nk_local / nown
        ab = algorithmic_bytes(W, H, 8, nkf, ncand, nkf, nkf)
        dom = max(stage, key=lambda k: stage[k])
        nunits = npairs if dom == "match" else nown
        achieved = ab[dom] * nunits / (stage[dom] * 1e-3) / 1e9
        traffic = None
        traffic_all = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf) and (W, H, NF, nown) == (640, 480, 1000, 64):   # the PMC passes were taken on this workload
            try:
                tj = json.load(open(tf))
                traffic = tj.get(dom)
                if all(k in tj for k in ("pyramid", "fast", "quadtree", "describe", "match")):
                    traffic_all = sum(int(tj[k]) for k in ("pyramid", "fast", "quadtree", "describe") + (("match",) if do_match else ()))
            except Exception:
                traffic = None
        # the other ceiling SURVEY.md 8(d) asks for: a device-to-device copy measured in this run (bytes read + written)
        cbuf = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
        cdst = torch.empty_like(cbuf)
        cdst.copy_(cbuf)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(5):
            cdst.copy_(cbuf)
        c1.record()
        torch.cuda.synchronize()
        copy_gbps = 5 * 2 * cbuf.numel() / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del cbuf, cdst
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(ab[dom] * nunits),
                "avg_launch_ms": round(stage[dom], 4),
                "measured_copy_GBps": round(copy_gbps, 1), "frac_of_measured_copy": round(achieved / copy_gbps, 5),
                "stage_ms": {k: round(float(v), 4) for k, v in stage.items()},
                "whole_path_algorithmic_GBps": round(sum(ab[k] for k in ("pyramid", "fast", "describe")) * total
                                                     / (elapsed / args.steps) / 1e9, 2),
                # measured: the counter bytes of every kernel of a step (profiles/traffic.json: FETCH_SIZE / WRITE_SIZE passes of this
                # command, corrected as MI355X_MICROARCH.md prescribes) over the step time of THIS run
                "whole_path_measured_bytes_per_step": traffic_all,
                "whole_path_measured_GBps": round(traffic_all / (elapsed / args.steps) / 1e9, 2) if traffic_all else None}
        # The ceiling this path actually runs into is vector-instruction issue, not HBM: the dominant kernel's
        # SQ_INSTS_VALU per launch (profiles/valu_insts.json, a PMC pass of this command) over its launch time measured in THIS run,
        # against 256 CUs x 4 SIMDs x 2.4 GHz / 4 clocks per wave instruction (the rate of most of the instructions it is made of).
        vf = os.path.join(ROOT, "profiles", "valu_insts.json")
        if os.path.exists(vf) and (W, H, NF, nown) == (640, 480, 1000, 64):
            try:
                vi = json.load(open(vf)).get(dom)
                if vi:
                    peak_issue = 256 * 4 * 2.4e9 / 4
                    roof["valu_issue"] = {"kernel": dom, "wave_insts_per_launch": int(vi), "achieved_Ginst_per_s": round(vi / (stage[dom] * 1e-3) / 1e9, 1),
                                          "peak_Ginst_per_s": round(peak_issue / 1e9, 1), "frac": round(vi / (stage[dom] * 1e-3) / peak_issue, 4)}
            except Exception:
                pass

    # ---- extras (N = 1 only; reported beside the contract's line, never as `value`) ----
    pipelined = host_api = extra = cpu = None
    if rank == 0 and world == 1:
        # the same steps pipelined over two HIP streams, one captured HIP graph per step: consecutive steps are independent, so
        # the short latency-bound kernels of one step run underneath the VALU-bound ones of the next.  Kernels of overlapping
        # steps time-share the GPU, so per-kernel durations (and `roofline`) are not defined in that mode.
        if not args.no_pipelined:
            try:
                pipelined = pipelined_pass(pkg, torch, shard, frames, W, H, B, NF, local, do_match, args.steps, nk_local, args.pipelined_streams)
            except Exception as e:
                pipelined = {"error": str(e)[:200]}
        if not args.no_host_api:
            try:
                host_api = host_api_pass(pkg, frames_np, W, H, NF, local)
            except Exception as e:
                host_api = {"error": str(e)[:200]}
        if not args.no_extra_configs:     # the other BASELINE configs, so that the driver's run observes them (never `value`)
            extra = []
            for fn in (lambda: extra_single_frame(pkg, torch, synth, local),                  # configs[1] as written
                       lambda: extra_config3(pkg, torch, shard, synth, local),                # configs[2]
                       lambda: extra_tracking_loop(pkg, synth),                               # configs[4]
                       lambda: extra_survey_stream(pkg, torch, shard, synth, local)):         # the headline step at a realistic corner density
                try:
                    extra.append(fn())
                except Exception as e:             # noqa: BLE001 -- a side measurement must never cost the contract line
                    extra.append({"error": str(e)[:300]})
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(frames_np, NF, args.cpu_budget)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        if world == 1:
            par = "1 GPU, no communication"
        else:
            par = ("one stream of %d frames cut into %d contiguous blocks, one rank per GPU; per step one boundary frame to the next rank "
                   "(send/recv) and one gather of the keypoint/descriptor/match block to rank 0; backend %s%s"
                   % (total, world, backend, " = RCCL over xGMI" if on_gpu else " (CPU rehearsal of the plumbing, host mirrors)"))
        out = {
            "metric": METRIC,
            "value": round(nk_all / (elapsed / args.steps), 1),
            "unit": "keypoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": args.scaling,
            "launch": ("one HIP graph per step" if step_graph is not None else "kernel by kernel") + (", %d concurrent sub-batches" % args.subbatches if args.subbatches > 1 else ""),
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "frames_per_s": round(total / (elapsed / args.steps), 1),
            "timed_region_s": round(elapsed, 5),
            "ms_per_step_gpu": {"p10": round(pct(0.1), 4), "p50": round(pct(0.5), 4), "p90": round(pct(0.9), 4), "rank": 0},
            "matches_per_step": nm_all,
            "keypoints_per_frame": round(nk_all / total, 2),
            "config": {"workload": "%dx%d nFeatures=%d 8 levels scale 1.2 (BASELINE configs[1] shape), %d frames per step%s "
                                   "(configs[3] stream), extract%s; inputs resident in HBM"
                                   % (W, H, NF, total, " (%d per GPU)" % B if world > 1 else "",
                                      " + dense match vs previous frame (ratio 0.9, TH_LOW, rotation filter)" if do_match else ""),
                       "frames_per_step": total, "frames_per_gpu": B, "width": W, "height": H, "nfeatures": NF,
                       "parallelism": par},
            "roofline": roof,
            "cpu_baseline": cpu,
            "host_api": host_api,
            "extra_configs": extra,
            "pipelined": pipelined,
        }
        if gathered_ok is not None:
            out["gathered_on_rank0"] = gathered_ok
        if force_comm:
            kv, dv, _, cv, _ = layout.views(path.bufs[(step_no[0] - 1) % nslot])
            out["forced_comm"] = {"backend": backend, "world_size": 1,
                                  "self_exchange": "done" if self_xchg[0] else ("refused: " + (self_xchg[1] if len(self_xchg) > 1 else "not attempted")),
                                  "self_exchange_ok": bool(self_xchg[0] and on_gpu and int(cv[0].item()) == int(cv[nown].item()) and bool((dv[0] == dv[nown]).all().item())
                                                           and bool((kv[0].view(torch.int32) == kv[nown].view(torch.int32)).all().item())),   # bit patterns: class_id = -1 reads as NaN through the float view
                                  "note": "process group of one rank: per step one batch_isend_irecv of the boundary frame to itself and one "
                                          "asynchronous gather of the flat block, both on the device-resident block the N-rank run uses"}
        print(json.dumps(out))
    if comm_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
