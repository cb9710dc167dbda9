"""CPU suite: the boundary's host-only code (key remapping, name mapping, safetensors parsing, file resolution, presets,
PNG / GIF writers) built with AddressSanitizer + UndefinedBehaviorSanitizer and driven through the C ABI on well-formed
and malformed inputs (tests/host/asan_main.cpp).  GPU sanitizers are not available on this pool; the device code has its
own parity suite.  (First run of this build found a real out-of-bounds read: the GIF quantiser's pixel sampler on frames
smaller than its sampling stride.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "candle-video_amd")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    b = subprocess.run(["make", "-C", PKG, "asan"], capture_output=True, text=True)
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(PKG, "build", "asan", "ltx_host_asan"), str(tmp_path / "work")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "host sanitizer driver: clean" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr
