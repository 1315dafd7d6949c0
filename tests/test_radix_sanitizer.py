"""CPU: the native radix tree (csrc/radix_tree.hip is plain C++ host code) built with AddressSanitizer + UndefinedBehaviorSanitizer
by g++ and driven through the product's RadixCache over the reference's own traces: no heap error, no UB, same slots.
(GPU sanitizers are not available on the pool; this is the host half of the path under a sanitizer.)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_radix_tree_under_asan_ubsan(tmp_path, pkg):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    so = str(tmp_path / "libradix_asan.so")
    src = os.path.join(ROOT, "ltp-sglang_amd", "csrc", "radix_tree.hip")
    build = subprocess.run([gxx, "-x", "c++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            "-fno-omit-frame-pointer", "-shared", "-fPIC", src, "-o", so], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_radix_asan_driver.py"), ROOT, so], env=env,
                         capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "mismatches 0" in run.stdout
