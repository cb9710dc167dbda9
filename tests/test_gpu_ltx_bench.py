"""tools/ltx_bench: the stand-alone C++ driver (no Python / torch in its process) builds both models from named weights and runs
LtxPipeline::call through the C ABI alone - evidence that nothing above include/ltxhip*.h is needed by a host."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "candle-video_amd", "build", "ltx_bench")


def test_ltx_bench_is_built_and_links_only_the_library():
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.join(ROOT, "candle-video_amd"), "ltx_bench"], check=True)
    out = subprocess.run(["ldd", EXE], capture_output=True, text=True, check=True).stdout
    assert "libltxhip.so" in out and "libamdhip64" in out
    assert "torch" not in out and "python" not in out


@pytest.mark.gpu
def test_ltx_bench_runs_config_c1_end_to_end():
    r = subprocess.run([EXE, "--config", "c1", "--steps", "2", "--warmup", "0"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["video_ok"] == 1 and line["num_frames"] == 25 and line["denoise_steps"] == 7
    assert line["frames_per_s"] > 50            # an order of magnitude below the measured rate: a liveness bar, not a perf claim
