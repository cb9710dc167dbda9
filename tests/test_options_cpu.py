"""CPU suite: the run-time option interface of the C ABI (include/ltxhip.h "run-time options"): one parsed structure, read once
from LTX_OPTIONS, changed with ltx_set_option - no launch path reads the environment."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_set_reset_and_bad_entries():
    import ltxhip
    ltxhip.set_option("gemm_tune", 0)
    ltxhip.set_option("gemm_plan", "asm16:160x256")
    ltxhip.set_option("gemm_off", "asm16+ring+halo_out")
    ltxhip.set_option("attn_off", "q64+cross")
    ltxhip.set_option("q2_fold", 2)
    ltxhip.set_option("gemm_plan", None)                       # back to the default
    ltxhip.set_option("x_some_experiment_knob", 3)             # accepted by every build, read by experiment builds only
    for key, val in (("no_such_option", "1"), ("gemm_off", "asm16+nope"), ("gemm_tune", "yes"), ("gemm_plan", "x" * 64)):
        with pytest.raises(ltxhip.LtxError, match="unknown option or bad value"):
            ltxhip.set_option(key, val)
    with ltxhip.options(norm_lean=0, xattn_compact=0):
        pass
    ltxhip.reset_options()
    assert ltxhip.has_experiments() is False                    # the shipped library: measured-negative variants compiled out


def test_get_option_reads_back_and_the_with_block_restores_previous_values():
    import ltxhip
    ltxhip.reset_options()
    assert ltxhip.get_option("ff2_defer") == "1" and ltxhip.get_option("gemm_off") == "" and ltxhip.get_option("gemm_plan") == ""
    ltxhip.set_option("gemm_off", "ring+big"); ltxhip.set_option("norm_presum", 2); ltxhip.set_option("gemm_plan", "asm16")
    assert ltxhip.get_option("gemm_off") == "ring+big"
    with ltxhip.options(gemm_off="asm16", norm_presum=0, gemm_plan="ring", x_knob=5):
        assert ltxhip.get_option("gemm_off") == "asm16" and ltxhip.get_option("norm_presum") == "0" and ltxhip.get_option("x_knob") == "5"
        with ltxhip.options(gemm_off=None):
            assert ltxhip.get_option("gemm_off") == ""
        assert ltxhip.get_option("gemm_off") == "asm16"
    # the values from before the block, not the built-in defaults
    assert ltxhip.get_option("gemm_off") == "ring+big" and ltxhip.get_option("norm_presum") == "2" and ltxhip.get_option("gemm_plan") == "asm16"
    assert ltxhip.get_option("x_knob") is None
    with pytest.raises(ltxhip.LtxError, match="unknown option"):
        ltxhip.get_option("no_such_option")
    ltxhip.reset_options()


def test_no_environment_reads_outside_the_option_parser():
    """The product sources read the environment in exactly two places: options.cpp (LTX_OPTIONS, once) and the RCCL loader's
    library path."""
    hits = []
    for sub in ("csrc", "host"):
        d = os.path.join(ROOT, "candle-video_amd", sub)
        for f in sorted(os.listdir(d)):
            if not f.endswith((".hip", ".cpp", ".h")): continue
            for i, line in enumerate(open(os.path.join(d, f)), 1):
                if "getenv(" in line and not line.lstrip().startswith("//"): hits.append(f"{sub}/{f}:{i}")
    assert sorted(h.split(":")[0] for h in hits) == ["csrc/options.cpp", "host/team.cpp"], hits


def test_options_from_the_environment_in_a_fresh_process():
    code = ("import sys; sys.path.insert(0, %r); import ltxhip; ltxhip.set_option('gemm_tune', None); print('loaded')" % os.path.join(ROOT, "candle-video_amd"))
    ok = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LTX_OPTIONS="gemm_tune=0,gemm_off=ring+p8,x_knob=2"), capture_output=True, text=True)
    assert ok.returncode == 0 and "loaded" in ok.stdout and "not understood" not in ok.stderr, ok.stderr
    bad = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LTX_OPTIONS="gemm_tune=0,bogus=1"), capture_output=True, text=True)
    assert bad.returncode == 0 and "entry 'bogus=1' not understood" in bad.stderr, bad.stderr
