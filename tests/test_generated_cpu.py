"""CPU suite: the generated kernel bodies under candle-video_amd/csrc/*.inc are exactly what their generators emit with default
options (tools/gen_attn_q64_asm.py, tools/gen_attn_q128_asm.py, tools/gen_gemm_asm.py): a hand edit of an .inc, or a generator
change without regenerating, fails here instead of shipping a kernel nobody can reproduce."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gen,inc", [("gen_attn_q64_asm.py", "attn_q64_loop.inc"), ("gen_attn_q128_asm.py", "attn_q128_loop.inc"),
                                     ("gen_gemm_asm.py", "gemm_asm_loop.inc")])
def test_committed_inc_matches_its_generator(tmp_path, gen, inc):
    out = str(tmp_path / inc)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", gen), "--out", out], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-1000:]
    want = open(os.path.join(ROOT, "candle-video_amd", "csrc", inc)).read()
    got = open(out).read()
    assert got == want, f"{inc} differs from the output of tools/{gen}: regenerate it (python3 tools/{gen})"
