"""bench.py end to end on the GPU box: the N-rank line equals the 1-rank line on the same stream.
The box has one GPU, so the two ranks share it and talk gloo (ORBX_BENCH_BACKEND=gloo: host mirrors of the same
flat blocks; the nccl path differs only in where the tensors live)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
COMMON = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-pipelined", "--no-host-api", "--no-extra-configs"]


def _run(extra, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env)
    p = subprocess.run([sys.executable, BENCH] + extra + COMMON, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_two_ranks_equal_one_rank():
    one = _run(["--gpus", "1", "--batch", "8"])
    weak = _run(["--gpus", "2", "--batch", "4", "--scaling", "weak"], ORBX_BENCH_BACKEND="gloo")
    strong = _run(["--gpus", "2", "--batch", "8", "--scaling", "strong"], ORBX_BENCH_BACKEND="gloo")
    assert one["n_gpus"] == 1 and weak["n_gpus"] == 2 and strong["n_gpus"] == 2
    assert weak["scaling"] == "weak" and strong["scaling"] == "strong"
    assert "gloo" in weak["config"]["parallelism"] and "RCCL" not in weak["config"]["parallelism"]
    for r in (weak, strong):
        assert r["gathered_on_rank0"] is True
        assert r["config"]["frames_per_step"] == 8
        assert r["keypoints_per_frame"] == one["keypoints_per_frame"]        # the same 8 frames of the same stream
        assert r["matches_per_step"] == one["matches_per_step"]              # including the pair that straddles the ranks
    assert one["roofline"]["frac"] > 0 and one["value"] > 0


def test_rccl_path_on_one_gpu():
    """The communication branch with ONE rank on the box's GPU, backend "nccl" (= RCCL), in a fresh process: the process group
    initialises with device_id, dist.gather takes the uint8 flat block asynchronously every step and rank 0 ends up holding the
    stream's result; the boundary frame is sent to the rank itself through batch_isend_irecv on the same views the N-rank run
    uses.  What an 8-GPU run adds is more ranks, not other calls."""
    r = _run(["--gpus", "1", "--batch", "8"], ORBX_BENCH_FORCE_COMM="1")
    one = _run(["--gpus", "1", "--batch", "8"])
    assert r["gathered_on_rank0"] is True
    fc = r["forced_comm"]
    assert fc["backend"] == "nccl" and fc["world_size"] == 1
    assert fc["self_exchange_ok"], fc            # RCCL 2.26 accepts a send to the sending rank inside one group call
    assert r["keypoints_per_frame"] == one["keypoints_per_frame"] and r["matches_per_step"] == one["matches_per_step"]
