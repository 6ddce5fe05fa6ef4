"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the sanitizer build of the CPU restatement).
oracle/sanitize_check.c drives every entry point of oracle/oracle.c on ragged inputs with exactly sized heap buffers; the same driver
built with -DSANITIZE_SELFTEST holds one out-of-bounds CSR index and must be stopped, which shows the harness can fail."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O1", "-g", "-std=c99", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]


def build(tmp_path, name, extra):
    exe = str(tmp_path / name)
    cmd = ["gcc"] + FLAGS + extra + [os.path.join(ROOT, "oracle", "sanitize_check.c"), "-o", exe, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and ("asan" in r.stderr or "ubsan" in r.stderr or "sanitize" in r.stderr):
        pytest.skip("no sanitizer runtime for gcc here: " + r.stderr.strip().splitlines()[-1])
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc")
@pytest.mark.parametrize("openmp", [False, True])
def test_oracle_is_clean_under_asan_and_ubsan(tmp_path, openmp):
    exe = build(tmp_path, "sanitize_check", ["-fopenmp"] if openmp else [])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=%d" % (0 if openmp else 1), OMP_NUM_THREADS="4")  # libgomp keeps its pool: not a leak of ours
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.startswith("sanitize_check ok")


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc")
def test_the_sanitizer_harness_can_fail(tmp_path):
    exe = build(tmp_path, "sanitize_selftest", ["-DSANITIZE_SELFTEST"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "AddressSanitizer" in r.stderr
