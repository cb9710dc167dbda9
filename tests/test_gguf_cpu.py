"""CPU suite: the GGUF side of the reference's default text encoder (quantized_t5_encoder.rs:558-679).

  * the oracle's dequantisers against hand-computed known answers (ggml block formats: every field of every format is
    exercised with values whose result can be read off the block: nibble order, the fifth / sixth bit planes, the 6-bit
    packed scales of the K-quants incl. their high-bit path, signed scales, half-precision d);
  * the host-side GGUF parser behind the C ABI against files written by the oracle's writer (names, types, shapes reported
    outermost-first as candle does, payload bytes, alignment, metadata incl. string arrays skipped), and its refusal of
    truncated / lying / malformed files."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gguf_oracle as G


def h16(x):
    return np.asarray([x], np.float16).view(np.uint8).tobytes()


def test_q8_0_q4_0_q5_0_known_answers():
    qs = np.arange(-16, 16, dtype=np.int8)
    np.testing.assert_array_equal(G.dequantize(G.Q8_0, h16(0.5) + qs.tobytes(), 32), 0.5 * qs.astype(np.float32))
    b = bytes([(j & 0xF) | ((15 - j) << 4) for j in range(16)])
    want = np.concatenate([(np.arange(16) - 8) * 2.0, ((15 - np.arange(16)) - 8) * 2.0]).astype(np.float32)
    np.testing.assert_array_equal(G.dequantize(G.Q4_0, h16(2.0) + b, 32), want)
    # Q5_0: qh = 0xFFFF0000 -> fifth bit clear for elements 0..15, set for 16..31; nibbles 0 -> -16 | 0
    np.testing.assert_array_equal(G.dequantize(G.Q5_0, h16(1.0) + struct.pack("<I", 0xFFFF0000) + bytes(16), 32),
                                  np.concatenate([np.full(16, -16.0), np.zeros(16)]).astype(np.float32))
    # and the other way round, nibbles 0xF/0x1: (15 | 16) - 16 = 15 for the first half, 1 - 16 = -15 for the second
    np.testing.assert_array_equal(G.dequantize(G.Q5_0, h16(1.0) + struct.pack("<I", 0x0000FFFF) + bytes([0x1F] * 16), 32),
                                  np.concatenate([np.full(16, 15.0), np.full(16, -15.0)]).astype(np.float32))


def k_scales(sc, mn):
    """pack eight 6-bit (scale, min) pairs the way get_scale_min_k4 unpacks them"""
    s = [0] * 12
    for j in range(4):
        s[j] = (sc[j] & 63) | ((sc[j + 4] >> 4) << 6)
        s[j + 4] = (mn[j] & 63) | ((mn[j + 4] >> 4) << 6)
        s[j + 8] = (sc[j + 4] & 0xF) | ((mn[j + 4] & 0xF) << 4)
    return bytes(s)


def test_q4_k_q5_k_known_answers():
    sc = [1, 2, 3, 4, 5, 22, 39, 63]            # the last three need the high-bit path (>= 16)
    mn = [0, 1, 2, 3, 36, 5, 6, 47]
    head = h16(1.0) + h16(0.5) + k_scales(sc, mn)
    got = G.dequantize(G.Q4_K, head + bytes([0x21] * 128), 256).reshape(4, 2, 32)        # low nibble 1, high nibble 2
    for j64 in range(4):
        for half in range(2):
            i = 2 * j64 + half
            assert (got[j64, half] == np.float32(sc[i] * (1 + half) - 0.5 * mn[i])).all(), (j64, half)
    # Q5_K: qh = 0b01010101 sets the fifth bit of every LOW-nibble element (bits 0, 2, 4, 6), never of a high-nibble one
    got = G.dequantize(G.Q5_K, head + bytes([0x55] * 32) + bytes([0x21] * 128), 256).reshape(4, 2, 32)
    for j64 in range(4):
        for half in range(2):
            i = 2 * j64 + half
            v = (1 + 16) if half == 0 else 2
            assert (got[j64, half] == np.float32(sc[i] * v - 0.5 * mn[i])).all(), (j64, half)
    # a single bit of qh belongs to ONE element: byte l = 5, bit 3 -> chunk 1, high nibble, element 5
    qh = bytearray(32); qh[5] = 1 << 3
    a = G.dequantize(G.Q5_K, head + bytes(qh) + bytes(128), 256)
    b = G.dequantize(G.Q5_K, head + bytes(32) + bytes(128), 256)
    diff = np.nonzero(a != b)[0]
    assert list(diff) == [64 + 32 + 5] and a[101] - b[101] == np.float32(16 * sc[3])


def test_q6_k_known_answers():
    scales = np.arange(-8, 8, dtype=np.int8)
    blk = bytes([0x31] * 128) + bytes([0b11100100] * 64) + scales.tobytes() + h16(0.25)
    got = G.dequantize(G.Q6_K, blk, 256).reshape(2, 4, 32)
    qv = [1 | (0 << 4), 1 | (1 << 4), 3 | (2 << 4), 3 | (3 << 4)]
    for n in range(2):
        for quarter in range(4):
            for l in range(32):
                s = int(scales[n * 8 + l // 16 + 2 * quarter])
                assert got[n, quarter, l] == np.float32(0.25 * s * (qv[quarter] - 32)), (n, quarter, l)


def test_f16_bf16_f32_passthrough():
    x = np.array([1.5, -2.25, 65504.0, 6e-8], np.float16)
    np.testing.assert_array_equal(G.dequantize(G.F16, x.tobytes(), 4), x.astype(np.float32))
    y = np.array([1.0, -3.5, 1e30, 1e-30], np.float32)
    np.testing.assert_array_equal(G.dequantize(G.BF16, (y.view(np.uint32) >> 16).astype(np.uint16).tobytes(), 4), ((y.view(np.uint32) >> 16) << 16).view(np.float32))
    np.testing.assert_array_equal(G.dequantize(G.F32, y.tobytes(), 4), y)


@pytest.fixture(scope="module")
def hip():
    import ltxhip
    return ltxhip


def sample_tensors(rng):
    t = {}
    for name, ty, shape in [("token_embd.weight", G.Q8_0, (8, 64)), ("enc.blk.0.attn_q.weight", G.Q5_K, (4, 256)), ("enc.blk.0.ffn_up.weight", G.Q6_K, (2, 512)),
                            ("enc.blk.0.ffn_gate.weight", G.Q4_K, (3, 256)), ("enc.blk.0.attn_norm.weight", G.F32, (64,)), ("half", G.F16, (5, 7)),
                            ("q40", G.Q4_0, (2, 3, 32)), ("q50", G.Q5_0, (1, 64)), ("bf", G.BF16, (3,))]:
        t[name] = (ty, shape, G.random_blocks(ty, int(np.prod(shape)), rng))
    return t


@pytest.mark.parametrize("alignment", [32, 64, 256])
def test_parser_reads_what_the_writer_wrote(hip, tmp_path, alignment):
    t = sample_tensors(np.random.default_rng(1))
    p = str(tmp_path / "m.gguf")
    G.write_gguf(p, t, alignment=alignment)
    got = hip.gguf_tensors(p)
    assert [g[0] for g in got] == list(t)
    for name, ty, shape, raw in got:
        assert (ty, tuple(shape)) == (t[name][0], tuple(t[name][1])), name
        assert raw == t[name][2], name
    assert G.read_gguf(p).keys() == t.keys()             # the oracle's own reader agrees with its writer
    for name, (ty, shape, raw) in G.read_gguf(p).items():
        assert raw == t[name][2]
    assert hip.gguf_type_info(G.Q5_K) == (256, 176) and hip.gguf_type_info(G.Q8_0) == (32, 34)


def test_parser_refuses_malformed_files(hip, tmp_path):
    t = sample_tensors(np.random.default_rng(2))
    p = str(tmp_path / "m.gguf")
    G.write_gguf(p, t)
    good = open(p, "rb").read()

    def expect_error(data, what):
        q = str(tmp_path / "bad.gguf")
        open(q, "wb").write(data)
        with pytest.raises(hip.LtxError):
            hip.gguf_tensors(q)

    expect_error(b"GGML" + good[4:], "magic")
    expect_error(good[:4] + struct.pack("<I", 1) + good[8:], "version 1")
    expect_error(good[:8] + struct.pack("<Q", 1 << 40) + good[16:], "tensor count beyond the file")
    expect_error(good[:16] + struct.pack("<Q", 1 << 50) + good[24:], "kv count beyond the file")
    expect_error(good[:100], "cut inside the metadata")
    expect_error(good[:len(good) - 40], "last tensor's data cut")
    expect_error(b"", "empty")
    # a string length that runs past the end of the file
    i = good.index(b"general.architecture") - 8
    expect_error(good[:i] + struct.pack("<Q", 1 << 62) + good[i + 8:], "string length")
    # an unknown quantisation is listed but cannot be read
    t2 = {"x": (10, (256,), bytes(84))}                  # Q2_K
    G.BLOCK[10] = (256, 84)
    try:
        G.write_gguf(p, t2)
    finally:
        del G.BLOCK[10]
    got = hip.gguf_tensors(p)
    assert got[0][0] == "x" and got[0][1] == 10 and got[0][3] is None
    # a row that is not a multiple of the block size
    G.write_gguf(p, {"x": (G.Q8_0, (2, 48), bytes(3 * 34))})
    with pytest.raises(hip.LtxError):
        hip.gguf_tensors(p)
