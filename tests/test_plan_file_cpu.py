"""CPU suite: GEMM plan files (include/ltxhip.h ltx_plan_load / ltx_plan_save).  Loading parses and validates only - no GPU
call - so the shape-level predicates a plan must pass are checked here, including the round-3 plan family of the
one-wave-per-SIMD kernel ("asm16:*": linear layers only, K a multiple of 64, N >= 512)."""
import pytest


@pytest.fixture(scope="module")
def hip():
    import ltxhip
    return ltxhip


def write(tmp_path, text):
    p = tmp_path / "plans.txt"
    p.write_text("# ltxhip GEMM plans: M N K conv ntaps T H W plan\n" + text)
    return str(p)


def test_valid_plans_load_and_round_trip(hip, tmp_path):
    src = write(tmp_path, "4992 6144 2048 0 0 0 0 0 asm16:256x256\n4992 8192 2048 0 0 0 0 0 asm16:320x256\n"
                          "4992 2048 8192 0 0 0 0 0 asm16:160x256\n4992 2048 2048 0 0 0 0 0 160x256w16\n"
                          "2383872 128 128 1 27 97 128 192 halo:128\n")
    hip.plan_load(src)
    assert hip.ops.gemm_plan(4992, 8192, 2048) == "asm16:320x256"
    assert hip.ops.gemm_plan(4992, 2048, 2048) == "160x256w16"
    out = str(tmp_path / "saved.txt")
    hip.plan_save(out)
    saved = open(out).read()
    for line in ("4992 6144 2048 0 0 0 0 0 asm16:256x256", "2383872 128 128 1 27 97 128 192 halo:128"):
        assert line in saved


@pytest.mark.parametrize("line,msg", [
    ("4992 6144 2048 0 0 0 0 0 asm16:512x512", "unknown plan name"),
    ("2383872 256 256 1 27 97 128 192 asm16:256x256", "plan not valid"),      # the asm16 family serves linear layers only
    ("4992 6144 2080 0 0 0 0 0 asm16:256x256", "plan not valid"),             # K not a multiple of 64
    ("4992 256 2048 0 0 0 0 0 asm16:256x256", "plan not valid"),              # N below the family's floor
    ("4992 6144 2048 0 0 0 0 0 halo:128", "plan not valid"),                  # a conv plan on a linear shape
    ("4992 6144 2048 0 0 0", "malformed line"),
    ("0 6144 2048 0 0 0 0 0 256x256", "plan not valid"),
])
def test_invalid_plan_lines_are_refused(hip, tmp_path, line, msg):
    with pytest.raises(hip.LtxError, match=msg):
        hip.plan_load(write(tmp_path, line + "\n"))
