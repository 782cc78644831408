"""bench.py's launch contract, without a GPU: --gpus N > 1 starts N ranks itself and fails loudly when they fail;
a launcher that started a different number of ranks than --gpus says is refused."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(kw)
    return e


def test_world_size_must_match_gpus():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 2 and "WORLD_SIZE=4" in p.stderr and p.stdout.strip() == ""


def test_gpus_flag_starts_the_ranks_and_relays_failure():
    """No GPU here, so both ranks stop with "bench.py needs a GPU": the parent must exit non-zero and print no result line
    (with GPUs the same path relays rank 0's JSON line; tests/test_bench_gpu.py)."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("the failure relay is exercised on GPU-less hosts")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    assert "2-rank run failed" in p.stderr and "needs a GPU" in p.stderr
