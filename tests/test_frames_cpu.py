"""CPU suite, part 8: the host half of frame output (include/ltxhip_frames.h): ltx_write_png produces files that an
independent decoder (zlib + the PNG chunk grammar, below) reads back bit-exactly."""
import struct
import zlib

import torch

import ltxhip


def read_png(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(b):
        n, typ = struct.unpack(">I4s", b[pos:pos + 8]); data = b[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + data)      # chunk CRC
        if typ == b"IHDR": hdr = struct.unpack(">IIBBBBB", data)
        if typ == b"IDAT": idat += data
        pos += 12 + n
    w, h, depth, ctype, _, _, _ = hdr
    assert (depth, ctype) == (8, 2)
    raw = zlib.decompress(idat)
    rows = [raw[y * (1 + 3 * w):(y + 1) * (1 + 3 * w)] for y in range(h)]
    assert all(r[0] == 0 for r in rows)                                                            # filter type 0
    return torch.tensor([list(r[1:]) for r in rows], dtype=torch.uint8).reshape(h, w, 3)


def test_write_png_round_trip(tmp_path):
    g = torch.Generator().manual_seed(0)
    for (h, w) in [(1, 1), (7, 5), (64, 96)]:
        img = torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8)
        p = str(tmp_path / f"f_{h}x{w}.png")
        ltxhip.write_png(p, img)
        assert torch.equal(read_png(p), img)
