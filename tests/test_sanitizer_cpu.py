"""AddressSanitizer + UBSan run of the oracle (the CPU restatement) through every entry point."""
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
