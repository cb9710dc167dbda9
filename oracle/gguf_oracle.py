"""CPU restatement of the GGUF side of the reference's default text encoder - TEST INFRASTRUCTURE ONLY (tests/, never the
product path).

What it follows: `QuantizedT5EncoderModel` (src/models/ltx_video/quantized_t5_encoder.rs:558-679) reads its weights through
candle's `VarBuilder::from_gguf` and `QTensor::dequantize` (:53-72, 83-86, 624-634).  candle-core ^0.9.2 is NOT in the
checkout (Cargo.toml:16-19), so the container layout and the block formats are restated from the published GGUF v3 / ggml
definitions (ggml-common.h block_q8_0 / q4_0 / q5_0 / q4_K / q5_K / q6_K, ggml-quants.c dequantize_row_*), element by element
in f32 with every product and difference rounded separately.  PARITY UNPINNED against candle itself; pinned here by
hand-computed known answers (tests/test_gguf_cpu.py) - the formats are simple enough for that - and by the reference's own
statement of the weight names and shapes (:127-150, 415-430, 471-482, 582-592).

  read_gguf(path)             -> {name: (ggml_type, shape outermost-first, raw block bytes)}
  dequantize(type, raw, n)    -> float32 numpy [n]
  write_gguf(path, tensors)   -> a GGUF v3 file from {name: (ggml_type, shape, raw)}  (test fixtures: the reference ships none)
  random_blocks(type, n, rng) -> raw bytes of n elements with well-scaled random blocks
"""
import struct

import numpy as np

F32, F16, Q4_0, Q5_0, Q8_0, Q4_K, Q5_K, Q6_K, BF16 = 0, 1, 2, 6, 8, 12, 13, 14, 30
BLOCK = {F32: (1, 4), F16: (1, 2), BF16: (1, 2), Q4_0: (32, 18), Q5_0: (32, 22), Q8_0: (32, 34), Q4_K: (256, 144), Q5_K: (256, 176), Q6_K: (256, 210)}


def _half(b: np.ndarray) -> np.ndarray:                 # [..., 2] uint8 little-endian -> float32
    return np.ascontiguousarray(b).view(np.float16).astype(np.float32)[..., 0]


def _scale_min_k4(sc: np.ndarray):
    """get_scale_min_k4 for j = 0..7: scales[12] uint8 -> (sc [.., 8], m [.., 8]) as float32"""
    sc = sc.astype(np.int32)
    d = np.empty(sc.shape[:-1] + (8,), np.int32); m = np.empty_like(d)
    for j in range(8):
        if j < 4:
            d[..., j] = sc[..., j] & 63; m[..., j] = sc[..., j + 4] & 63
        else:
            d[..., j] = (sc[..., j + 4] & 0xF) | ((sc[..., j - 4] >> 6) << 4)
            m[..., j] = (sc[..., j + 4] >> 4) | ((sc[..., j] >> 6) << 4)
    return d.astype(np.float32), m.astype(np.float32)


def dequantize(ggml_type: int, raw: bytes, numel: int) -> np.ndarray:
    be, bb = BLOCK[ggml_type]
    assert numel % be == 0 and len(raw) == numel // be * bb, (ggml_type, numel, len(raw))
    a = np.frombuffer(raw, np.uint8).reshape(numel // be, bb)
    f = np.float32
    if ggml_type == F32: return np.frombuffer(raw, np.float32).copy()
    if ggml_type == F16: return np.frombuffer(raw, np.float16).astype(f)
    if ggml_type == BF16: return (np.frombuffer(raw, np.uint16).astype(np.uint32) << 16).view(np.float32).copy()
    if ggml_type == Q8_0:
        return (a[:, 2:].view(np.int8).astype(f) * _half(a[:, None, 0:2])).reshape(-1)
    if ggml_type == Q4_0:
        q = np.concatenate([a[:, 2:] & 0xF, a[:, 2:] >> 4], 1).astype(np.int32) - 8
        return (q.astype(f) * _half(a[:, None, 0:2])).reshape(-1)
    if ggml_type == Q5_0:
        qh = a[:, 2:6].copy().view(np.uint32)[:, 0].astype(np.int64)
        j = np.arange(16)
        lo = (a[:, 6:] & 0xF).astype(np.int64) | (((qh[:, None] >> j) << 4) & 0x10)
        hi = (a[:, 6:] >> 4).astype(np.int64) | ((qh[:, None] >> (j + 12)) & 0x10)
        q = np.concatenate([lo, hi], 1) - 16
        return (q.astype(f) * _half(a[:, None, 0:2])).reshape(-1)
    if ggml_type in (Q4_K, Q5_K):
        d, dmin = _half(a[:, 0:2]), _half(a[:, 2:4])
        sc, mn = _scale_min_k4(a[:, 4:16])
        d1 = (d[:, None] * sc).astype(f); m1 = (dmin[:, None] * mn).astype(f)           # [nb, 8]
        if ggml_type == Q4_K:
            qs = a[:, 16:144].reshape(-1, 4, 32)
            v = np.stack([qs & 0xF, qs >> 4], 2).astype(f)                                 # [nb, 4, 2, 32]
        else:
            qh = a[:, 16:48]; qs = a[:, 48:176].reshape(-1, 4, 32)
            bits = np.stack([(qh >> (2 * j64 + half)) & 1 for j64 in range(4) for half in range(2)], 1).reshape(-1, 4, 2, 32)
            v = (np.stack([qs & 0xF, qs >> 4], 2).astype(np.int32) + 16 * bits.astype(np.int32)).astype(f)
        y = (d1.reshape(-1, 4, 2, 1) * v).astype(f) - m1.reshape(-1, 4, 2, 1)
        return y.astype(f).reshape(-1)
    if ggml_type == Q6_K:
        ql = a[:, 0:128].reshape(-1, 2, 64); qh = a[:, 128:192].reshape(-1, 2, 32)
        sc = a[:, 192:208].view(np.int8).reshape(-1, 2, 8).astype(f); d = _half(a[:, 208:210])
        out = np.empty((a.shape[0], 2, 4, 32), f)
        l = np.arange(32)
        for quarter in range(4):
            lo = ql[:, :, 32:64] if quarter & 1 else ql[:, :, 0:32]
            q = ((lo >> 4) if quarter & 2 else (lo & 0xF)).astype(np.int32) | (((qh >> (2 * quarter)) & 3).astype(np.int32) << 4)
            s = sc[:, :, (l >> 4) + 2 * quarter]                                           # [nb, 2, 32]
            out[:, :, quarter, :] = ((d[:, None, None] * s).astype(f) * (q - 32).astype(f)).astype(f)
        return out.reshape(-1)
    raise ValueError(f"ggml type {ggml_type} not read")


def random_blocks(ggml_type: int, numel: int, rng: np.random.Generator, scale: float = 0.02) -> bytes:
    """valid random blocks whose dequantised values are O(scale)"""
    be, bb = BLOCK[ggml_type]
    nb = numel // be
    if ggml_type == F32: return (rng.standard_normal(numel) * scale).astype(np.float32).tobytes()
    if ggml_type == F16: return (rng.standard_normal(numel) * scale).astype(np.float16).tobytes()
    if ggml_type == BF16: return ((rng.standard_normal(numel) * scale).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16).tobytes()
    a = rng.integers(0, 256, (nb, bb), dtype=np.uint8)
    half = lambda x: np.asarray(x, np.float16).reshape(nb, 1).view(np.uint8)
    if ggml_type == Q8_0: a[:, 0:2] = half(scale / 64 * (0.5 + rng.random(nb)))
    elif ggml_type in (Q4_0, Q5_0): a[:, 0:2] = half(scale / (4 if ggml_type == Q4_0 else 8) * (0.5 + rng.random(nb)))
    elif ggml_type in (Q4_K, Q5_K):
        a[:, 0:2] = half(scale / (32 * (8 if ggml_type == Q4_K else 16)) * (0.5 + rng.random(nb)))
        a[:, 2:4] = half(scale / 64 * (0.5 + rng.random(nb)))
    elif ggml_type == Q6_K: a[:, 208:210] = half(scale / (64 * 16) * (0.5 + rng.random(nb)))
    return a.tobytes()


def _wstr(s: str) -> bytes:
    b = s.encode(); return struct.pack("<Q", len(b)) + b


def write_gguf(path: str, tensors, alignment: int = 32, version: int = 3, extra_kv: bytes = b"", n_extra_kv: int = 0):
    """tensors: {name: (ggml_type, shape outermost-first, raw bytes)} in insertion order"""
    kv = _wstr("general.architecture") + struct.pack("<I", 8) + _wstr("t5encoder")
    kv += _wstr("general.alignment") + struct.pack("<II", 4, alignment)
    kv += _wstr("tokenizer.ggml.tokens") + struct.pack("<IIQ", 9, 8, 3) + _wstr("<pad>") + _wstr("</s>") + _wstr("<unk>")   # an array of strings to skip
    kv += _wstr("t5encoder.attention.layer_norm_epsilon") + struct.pack("<If", 6, 1e-6)
    head = struct.pack("<IIQQ", 0x46554747, version, len(tensors), 4 + n_extra_kv) + kv + extra_kv
    infos, blobs, off = b"", [], 0
    for name, (ty, shape, raw) in tensors.items():
        ne = list(reversed(shape))
        infos += _wstr(name) + struct.pack("<I", len(ne)) + b"".join(struct.pack("<Q", int(x)) for x in ne) + struct.pack("<IQ", ty, off)
        pad = (-len(raw)) % alignment
        blobs.append(raw + b"\0" * pad); off += len(raw) + pad
    body = head + infos
    body += b"\0" * ((-len(body)) % alignment)
    with open(path, "wb") as f:
        f.write(body + b"".join(blobs))


def read_gguf(path: str):
    b = open(path, "rb").read()
    at = [0]
    def g(fmt):
        v = struct.unpack_from("<" + fmt, b, at[0]); at[0] += struct.calcsize("<" + fmt); return v if len(v) > 1 else v[0]
    def s():
        n = g("Q"); v = b[at[0]:at[0] + n].decode(); at[0] += n; return v
    size = {0: 1, 1: 1, 2: 2, 3: 2, 4: 4, 5: 4, 6: 4, 7: 1, 10: 8, 11: 8, 12: 8}
    magic, version, nt, nkv = g("I"), g("I"), g("Q"), g("Q")
    assert magic == 0x46554747 and version in (2, 3)
    align = 32
    for _ in range(nkv):
        key, t = s(), g("I")
        if t == 8: s()
        elif t == 9:
            et, cnt = g("I"), g("Q")
            for _ in range(cnt):
                if et == 8: s()
                else: at[0] += size[et]
        elif key == "general.alignment" and t == 4: align = g("I")
        else: at[0] += size[t]
    infos = []
    for _ in range(nt):
        name, nd = s(), g("I")
        ne = [g("Q") for _ in range(nd)]
        infos.append((name, tuple(reversed(ne)), g("I"), g("Q")))
    start = (at[0] + align - 1) // align * align
    out = {}
    for name, shape, ty, off in infos:
        be, bb = BLOCK[ty]
        n = int(np.prod(shape))
        out[name] = (ty, shape, b[start + off:start + off + n // be * bb])
    return out


def t5_gguf_names(num_layers: int):
    """GGUF tensor name -> Hugging Face name of the same weight (quantized_t5_encoder.rs:127-150, 415-430, 471-482, 582-592)"""
    m = {"token_embd.weight": "shared.weight", "enc.output_norm.weight": "encoder.final_layer_norm.weight",
         "enc.blk.0.attn_rel_b.weight": "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"}
    for i in range(num_layers):
        g, h = f"enc.blk.{i}.", f"encoder.block.{i}.layer."
        m.update({g + "attn_q.weight": h + "0.SelfAttention.q.weight", g + "attn_k.weight": h + "0.SelfAttention.k.weight",
                  g + "attn_v.weight": h + "0.SelfAttention.v.weight", g + "attn_o.weight": h + "0.SelfAttention.o.weight",
                  g + "attn_norm.weight": h + "0.layer_norm.weight", g + "ffn_gate.weight": h + "1.DenseReluDense.wi_0.weight",
                  g + "ffn_up.weight": h + "1.DenseReluDense.wi_1.weight", g + "ffn_down.weight": h + "1.DenseReluDense.wo.weight",
                  g + "ffn_norm.weight": h + "1.layer_norm.weight"})
    return m
