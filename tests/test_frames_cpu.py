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


# ---- GIF (the reference's default output, examples/ltx-video/main.rs:683-707) ------------------------------------------
def read_gif(path):
    """Independent GIF89a reader: block grammar + variable-width LZW.  Returns (width, height, loop_count, frames) with
    frames = [(delay_cs, [H,W,3] u8 tensor)]."""
    b = open(path, "rb").read()
    assert b[:6] == b"GIF89a"
    w, h, flags, _bg, _aspect = struct.unpack("<HHBBB", b[6:13])
    assert flags & 0x80 == 0                                   # no global colour table (Encoder::new(.., &[]))
    pos, loop, frames, delay = 13, None, [], None
    while True:
        tag = b[pos]; pos += 1
        if tag == 0x3B:
            break
        if tag == 0x21:
            label = b[pos]; pos += 1
            blocks = []
            while b[pos]:
                n = b[pos]; blocks.append(b[pos + 1:pos + 1 + n]); pos += 1 + n
            pos += 1
            if label == 0xFF and blocks[0] == b"NETSCAPE2.0":
                assert blocks[1][0] == 1
                loop = struct.unpack("<H", blocks[1][1:3])[0]
            if label == 0xF9:
                delay = struct.unpack("<H", blocks[0][1:3])[0]
            continue
        assert tag == 0x2C
        x0, y0, fw, fh, fl = struct.unpack("<HHHHB", b[pos:pos + 9]); pos += 9
        assert (x0, y0, fw, fh) == (0, 0, w, h) and fl & 0x80 and (fl & 7) == 7     # local table of 256, full frame
        pal = torch.tensor(list(b[pos:pos + 768]), dtype=torch.uint8).reshape(256, 3); pos += 768
        mcs = b[pos]; pos += 1
        assert mcs == 8
        data = bytearray()
        while b[pos]:
            n = b[pos]; data += b[pos + 1:pos + 1 + n]; pos += 1 + n
        pos += 1
        # LZW decode
        clear, eoi = 256, 257
        out = []
        table = None; width = 9; nxt = 258; prev = None
        acc = 0; nbits = 0; i = 0
        while True:
            while nbits < width:
                acc |= data[i] << nbits; nbits += 8; i += 1
            code = acc & ((1 << width) - 1); acc >>= width; nbits -= width
            if code == clear:
                table = {k: bytes([k]) for k in range(256)}; width = 9; nxt = 258; prev = None
                continue
            if code == eoi:
                break
            if prev is None:
                entry = table[code]
            else:
                entry = table[code] if code in table else table[prev] + table[prev][:1]
                if nxt < 4096:
                    table[nxt] = table[prev] + entry[:1]; nxt += 1
                    if nxt == (1 << width) and width < 12:
                        width += 1
            out.append(entry); prev = code
        idx = torch.tensor(list(b"".join(out)), dtype=torch.long)
        assert idx.numel() == w * h
        frames.append((delay, pal[idx].reshape(h, w, 3)))
    return w, h, loop, frames


def psnr_u8(a, b):
    import math
    mse = float(((a.float() - b.float()) ** 2).mean())
    return 10 * math.log10(255.0 ** 2 / max(mse, 1e-9))


def test_write_gif_structure_and_decode_back(tmp_path):
    """delay 4, infinite loop, no global palette, one local 256-colour table per frame, LZW min code size 8 (main.rs:692-699);
    an independent decoder recovers the frames: a shaded three-surface scene with a moving object (few hues, smooth shading -
    what 256 colours per frame are meant for) > 40 dB per frame at the reference's sampling factor 30; a full-gamut gradient
    (a 2-D sheet through the colour cube: the 256-colour limit itself is ~30 dB) > 20 dB; and the LZW stream survives
    table resets (a noisy 200 x 300 frame needs ~20 of them)."""
    H, W, N = 96, 144, 5
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    sky, ground, obj = torch.tensor([90.0, 150.0, 230.0]), torch.tensor([60.0, 140.0, 50.0]), torch.tensor([220.0, 70.0, 40.0])
    frames = []
    for t in range(N):
        shade = (0.35 + 0.65 * (0.5 + 0.5 * torch.sin(xx / W * 3.1 + yy / H * 2.0 + 0.3 * t)))[..., None]
        img = torch.where((yy < H * 0.55)[..., None], sky, ground) * shade
        blob = (((xx - W * (0.2 + 0.1 * t)) ** 2 + (yy - H * 0.5) ** 2) < (H * 0.18) ** 2)[..., None]
        frames.append(torch.where(blob, obj * shade, img).clamp(0, 255).to(torch.uint8))
    vid = torch.stack(frames)
    p = str(tmp_path / "video.gif")
    ltxhip.write_gif(p, vid)                                       # reference settings: delay 4, speed 30
    w, h, loop, dec = read_gif(p)
    assert (w, h, loop, len(dec)) == (W, H, 0, N)
    for (delay, img), src in zip(dec, vid):
        assert delay == 4
        assert psnr_u8(img, src) > 40.0, psnr_u8(img, src)
    # speed 1 (every pixel sampled) is at least as good as speed 30 on the same frame
    p1 = str(tmp_path / "q1.gif"); ltxhip.write_gif(p1, vid[:1], speed=1)
    assert psnr_u8(read_gif(p1)[3][0][1], vid[0]) >= psnr_u8(dec[0][1], vid[0]) - 0.5
    gam = torch.stack([255 * xx / W, 255 * yy / H, 128 + 127 * torch.sin((xx + yy) / 17)], -1).clamp(0, 255).to(torch.uint8)[None]
    pg = str(tmp_path / "gamut.gif"); ltxhip.write_gif(pg, gam)
    assert psnr_u8(read_gif(pg)[3][0][1], gam[0]) > 20.0
    # noise: many LZW table resets, odd size, single frame, other delay
    g = torch.Generator().manual_seed(1)
    noise = torch.randint(0, 256, (1, 200, 300, 3), generator=g, dtype=torch.uint8)
    p2 = str(tmp_path / "noise.gif"); ltxhip.write_gif(p2, noise, delay_cs=7)
    w, h, loop, dec = read_gif(p2)
    assert (w, h, loop, dec[0][0]) == (300, 200, 0, 7)
    assert psnr_u8(dec[0][1], noise[0]) > 15.0                      # 256 colours for uniform noise: coarse but not garbage
