"""Sanitizer runs on the CPU (GPU sanitizers are not available on the pool): AddressSanitizer + UBSan on the oracle through every
entry point, ThreadSanitizer on the staging thread pool of orbx_extract_batch."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "asan_oracle")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-I" + os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "asan_oracle_main.c"), os.path.join(ROOT, "oracle", "orb_oracle.c"), os.path.join(ROOT, "oracle", "orb_oracle_kf.c"),
                           "-lm", "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([exe], env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "rc=0" in out.stdout and "ERROR" not in out.stderr


def test_stage_pool_under_tsan(tmp_path):
    """my-slam_amd/csrc/stage_pool.h (host-only C++): 4000 back-to-back jobs of alternating sizes, every item exactly once, no job
    returning before its items, no data race."""
    exe = str(tmp_path / "stage_pool_tsan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-I" + os.path.join(ROOT, "my-slam_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cxx", "stage_pool_tsan.cc"), "-lpthread", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
    assert "bad 0" in out.stdout and "WARNING: ThreadSanitizer" not in out.stderr
