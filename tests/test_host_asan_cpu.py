"""The library's host code under AddressSanitizer + UBSan (the kernels are validated by parity, the host
side by an instrumented build).  `make asan` compiles the sources with the sanitizers on the host side only and links
tests/host/abi_asan.c, which walks the C ABI without a GPU: size / layout queries over a grid and 200 000 random shapes
(valid, degenerate, huge), NULL arguments to every entry point, and -- when no device is visible -- the whole host-side
launch preparation with fake device pointers."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc to build")
def test_host_code_is_clean_under_asan_and_ubsan():
    b = subprocess.run(["make", "-j8", "asan"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(ROOT, "build", "abi_asan")], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "abi_asan: OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
