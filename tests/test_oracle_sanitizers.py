"""CPU-suite sanitizer job: the oracle's C file is rebuilt with AddressSanitizer + UBSan and the
oracle test files are re-run against that build in a child process.  GPU sanitizers are not available on the pool; this covers the
one piece of native code that runs on the CPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_tests_under_asan_and_ubsan():
    if os.environ.get("SMOQY_ORACLE_LIB"):
        pytest.skip("already the sanitizer child run")
    odir = os.path.join(ROOT, "oracle")
    r = subprocess.run(["make", "-C", odir, "-B", "libsmoqy_oracle_asan.so"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.isabs(libasan) and os.path.exists(libasan), "libasan.so not found next to gcc"
    env = dict(os.environ, SMOQY_ORACLE_LIB=os.path.join(odir, "libsmoqy_oracle_asan.so"), LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    files = ["test_oracle_dense.py", "test_oracle_complex_T.py", "test_oracle_force.py", "test_oracle_irregular.py", "test_oracle_phonon_fields.py", "test_golden.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + [os.path.join(ROOT, "tests", f) for f in files], capture_output=True, text=True,
                       env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert " passed" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
