// Frame files (include/ltxhip_frames.h), host side only: the PNG writer (zlib deflate, CRC per chunk) and the GIF writer.
// No device code here: this file is also part of the host sanitizer build (make asan).
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ltxhip_frames.h"
#include "../csrc/errors.h"

namespace {
void put32(std::vector<unsigned char>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void chunk(std::vector<unsigned char>& png, const char* type, const unsigned char* data, size_t n) {
    put32(png, (uint32_t)n);
    const size_t start = png.size();
    png.insert(png.end(), type, type + 4);
    if (n) png.insert(png.end(), data, data + n);
    put32(png, (uint32_t)crc32(0L, png.data() + start, (uInt)(n + 4)));
}
}  // namespace

extern "C" int ltx_write_png(const char* path, const uint8_t* rgb, int width, int height) {
    if (!path || !rgb || width < 1 || height < 1) LTX_FAIL(LTX_ERR_ARG, "ltx_write_png: bad argument");
    const size_t row = (size_t)width * 3;
    std::vector<unsigned char> raw((row + 1) * height);
    for (int y = 0; y < height; ++y) { raw[y * (row + 1)] = 0; memcpy(&raw[y * (row + 1) + 1], rgb + y * row, row); }   // filter 0
    uLongf zn = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) LTX_FAIL(LTX_ERR_ARG, "ltx_write_png: deflate failed");
    std::vector<unsigned char> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<unsigned char> ihdr;
    put32(ihdr, (uint32_t)width); put32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);    // 8-bit, colour type 2 (RGB)
    chunk(png, "IHDR", ihdr.data(), ihdr.size());
    chunk(png, "IDAT", z.data(), zn);
    chunk(png, "IEND", nullptr, 0);
    FILE* f = fopen(path, "wb");
    if (!f) LTX_FAIL(LTX_ERR_ARG, std::string("ltx_write_png: cannot open '") + path + "'");
    const size_t w = fwrite(png.data(), 1, png.size(), f);
    fclose(f);
    if (w != png.size()) LTX_FAIL(LTX_ERR_ARG, std::string("ltx_write_png: short write to '") + path + "'");
    return LTX_OK;
}

// ---- GIF output (examples/ltx-video/main.rs:683-707: the reference's DEFAULT output) ---------------------------------
// main.rs builds every frame with gif::Frame::from_rgb_speed(w, h, rgb, 30) - the `gif` crate (crates.io ^0.13, absent from
// the checkout) quantises each frame to a LOCAL 256-colour palette with NeuQuant at sampling factor `speed` and LZW-encodes
// the indices - sets delay = 4 (centiseconds, ~25 fps), Repeat::Infinite, no global palette, and writes the frames in order.
// This is a restatement of the published algorithms (NeuQuant: Dekker 1994, the integer form of the reference C code;
// GIF89a + variable-width LZW: the CompuServe specification), not of the crate's source: the file structure, delay, loop
// extension, palette size and sampling factor are the reference's, the palette entries themselves may differ from the
// crate's (its NeuQuant port trains on RGBA with floating-point neurons).
namespace {
struct NeuQuant {
    static constexpr int netsize = 256, ncycles = 100, netbiasshift = 4, intbiasshift = 16, intbias = 1 << intbiasshift, gammashift = 10,
                         betashift = 10, beta = intbias >> betashift, betagamma = intbias << (gammashift - betashift), initrad = netsize >> 3,
                         radiusbiasshift = 6, radiusbias = 1 << radiusbiasshift, initradius = initrad * radiusbias, radiusdec = 30, alphabiasshift = 10,
                         initalpha = 1 << alphabiasshift, radbiasshift = 8, radbias = 1 << radbiasshift, alpharadbshift = alphabiasshift + radbiasshift,
                         alpharadbias = 1 << alpharadbshift;
    int net[netsize][4];           // b, g, r, original index
    int netindex[256], bias[netsize], freq[netsize], radpower[initrad];

    void init() {
        for (int i = 0; i < netsize; ++i) {
            net[i][0] = net[i][1] = net[i][2] = (i << (netbiasshift + 8)) / netsize;
            freq[i] = intbias / netsize; bias[i] = 0;
        }
    }
    int contest(int b, int g, int r) {          // nearest neuron with the conscience bias; updates freq / bias
        int bestd = 0x7fffffff, bestbiasd = bestd, bestpos = -1, bestbiaspos = -1;
        for (int i = 0; i < netsize; ++i) {
            const int* n = net[i];
            int dist = std::abs(n[0] - b) + std::abs(n[1] - g) + std::abs(n[2] - r);
            if (dist < bestd) { bestd = dist; bestpos = i; }
            const int biasdist = dist - (bias[i] >> (intbiasshift - netbiasshift));
            if (biasdist < bestbiasd) { bestbiasd = biasdist; bestbiaspos = i; }
            const int betafreq = freq[i] >> betashift;
            freq[i] -= betafreq; bias[i] += betafreq << gammashift;
        }
        freq[bestpos] += beta; bias[bestpos] -= betagamma;
        return bestbiaspos;
    }
    void altersingle(int alpha, int i, int b, int g, int r) {
        int* n = net[i];
        n[0] -= (alpha * (n[0] - b)) / initalpha; n[1] -= (alpha * (n[1] - g)) / initalpha; n[2] -= (alpha * (n[2] - r)) / initalpha;
    }
    void alterneigh(int rad, int i, int b, int g, int r) {
        int lo = i - rad; if (lo < -1) lo = -1;
        int hi = i + rad; if (hi > netsize) hi = netsize;
        int j = i + 1, k = i - 1, m = 1;
        while (j < hi || k > lo) {
            const int a = radpower[m++];
            if (j < hi) { int* p = net[j++]; p[0] -= (a * (p[0] - b)) / alpharadbias; p[1] -= (a * (p[1] - g)) / alpharadbias; p[2] -= (a * (p[2] - r)) / alpharadbias; }
            if (k > lo) { int* p = net[k--]; p[0] -= (a * (p[0] - b)) / alpharadbias; p[1] -= (a * (p[1] - g)) / alpharadbias; p[2] -= (a * (p[2] - r)) / alpharadbias; }
        }
    }
    void learn(const uint8_t* rgb, int npix, int samplefac) {
        const int lengthcount = npix * 3;
        if (lengthcount < 3 * 503) samplefac = 1;         // pictures smaller than the largest sampling prime: every pixel (Dekker's minpicturebytes rule)
        const int alphadec = 30 + (samplefac - 1) / 3;
        int samplepixels = npix / samplefac; if (samplepixels < 1) samplepixels = 1;
        int delta = samplepixels / ncycles; if (delta < 1) delta = 1;
        int alpha = initalpha, radius = initradius, rad = radius >> radiusbiasshift;
        if (rad <= 1) rad = 0;
        for (int i = 0; i < rad; ++i) radpower[i] = alpha * (((rad * rad - i * i) * radbias) / (rad * rad));
        int step = 3 * 503;
        if (lengthcount % (3 * 499) != 0) step = 3 * 499; else if (lengthcount % (3 * 491) != 0) step = 3 * 491; else if (lengthcount % (3 * 487) != 0) step = 3 * 487;
        int pix = 0;
        for (int i = 0; i < samplepixels;) {
            const int r = rgb[pix] << netbiasshift, g = rgb[pix + 1] << netbiasshift, b = rgb[pix + 2] << netbiasshift;
            const int j = contest(b, g, r);
            altersingle(alpha, j, b, g, r);
            if (rad) alterneigh(rad, j, b, g, r);
            pix += step; while (pix >= lengthcount) pix -= lengthcount;
            if (++i % delta == 0) {
                alpha -= alpha / alphadec; radius -= radius / radiusdec;
                rad = radius >> radiusbiasshift; if (rad <= 1) rad = 0;
                for (int q = 0; q < rad; ++q) radpower[q] = alpha * (((rad * rad - q * q) * radbias) / (rad * rad));
            }
        }
    }
    void finish() {                 // unbias, sort by green, build the green index
        for (int i = 0; i < netsize; ++i) { for (int c = 0; c < 3; ++c) { int v = net[i][c] >> netbiasshift; net[i][c] = v < 0 ? 0 : (v > 255 ? 255 : v); } net[i][3] = i; }
        int previouscol = 0, startpos = 0;
        for (int i = 0; i < netsize; ++i) {
            int smallpos = i, smallval = net[i][1];
            for (int j = i + 1; j < netsize; ++j) if (net[j][1] < smallval) { smallpos = j; smallval = net[j][1]; }
            if (i != smallpos) for (int c = 0; c < 4; ++c) std::swap(net[i][c], net[smallpos][c]);
            if (smallval != previouscol) {
                netindex[previouscol] = (startpos + i) >> 1;
                for (int j = previouscol + 1; j < smallval; ++j) netindex[j] = i;
                previouscol = smallval; startpos = i;
            }
        }
        netindex[previouscol] = (startpos + netsize - 1) >> 1;
        for (int j = previouscol + 1; j < 256; ++j) netindex[j] = netsize - 1;
    }
    int search(int b, int g, int r) const {      // palette position (in green-sorted order) nearest to the colour
        int bestd = 1000, best = 0, i = netindex[g], j = i - 1;
        while (i < netsize || j >= 0) {
            if (i < netsize) {
                const int* p = net[i]; int dist = p[1] - g;
                if (dist >= bestd) i = netsize;
                else { ++i; if (dist < 0) dist = -dist; int a = p[0] - b; dist += a < 0 ? -a : a; if (dist < bestd) { a = p[2] - r; dist += a < 0 ? -a : a; if (dist < bestd) { bestd = dist; best = (int)(p - net[0]) / 4; } } }
            }
            if (j >= 0) {
                const int* p = net[j]; int dist = g - p[1];
                if (dist >= bestd) j = -1;
                else { --j; if (dist < 0) dist = -dist; int a = p[0] - b; dist += a < 0 ? -a : a; if (dist < bestd) { a = p[2] - r; dist += a < 0 ? -a : a; if (dist < bestd) { bestd = dist; best = (int)(p - net[0]) / 4; } } }
            }
        }
        return best;
    }
};

// variable-width LZW of 8-bit indices (GIF89a appendix F): codes 0..255 literals, 256 clear, 257 end; table reset at 4096
void lzw_encode(const uint8_t* idx, size_t n, std::vector<uint8_t>& out) {
    const int clear = 256, eoi = 257;
    std::vector<int> table(4096 * 256, -1);     // (prefix code, byte) -> code
    int next = eoi + 1, width = 9;
    uint32_t acc = 0; int nbits = 0;
    std::vector<uint8_t> bytes;
    auto emit = [&](int code) { acc |= (uint32_t)code << nbits; nbits += width; while (nbits >= 8) { bytes.push_back(acc & 0xff); acc >>= 8; nbits -= 8; } };
    std::vector<int> used;                       // table slots to clear on reset (avoids a 4 MB memset per reset)
    emit(clear);
    int prefix = n ? idx[0] : -1;
    for (size_t i = 1; i < n; ++i) {
        const int c = idx[i];
        const int slot = prefix * 256 + c;
        if (table[slot] >= 0) { prefix = table[slot]; continue; }
        emit(prefix);
        if (next < 4096) {
            table[slot] = next++; used.push_back(slot);
            if (next > (1 << width) && width < 12) ++width;
        } else {
            emit(clear);
            for (int sl : used) table[sl] = -1;
            used.clear(); next = eoi + 1; width = 9;
        }
        prefix = c;
    }
    if (prefix >= 0) emit(prefix);
    emit(eoi);
    if (nbits > 0) bytes.push_back(acc & 0xff);
    for (size_t i = 0; i < bytes.size(); i += 255) {          // data sub-blocks
        const size_t m = std::min<size_t>(255, bytes.size() - i);
        out.push_back((uint8_t)m); out.insert(out.end(), bytes.begin() + i, bytes.begin() + i + m);
    }
    out.push_back(0);
}
void put16(std::vector<uint8_t>& v, int x) { v.push_back(x & 0xff); v.push_back((x >> 8) & 0xff); }
}  // namespace

extern "C" int ltx_write_gif(const char* path, const uint8_t* rgb_frames, int n_frames, int width, int height, int delay_cs, int speed) {
    if (!path || !rgb_frames || n_frames < 1 || width < 1 || height < 1 || width > 65535 || height > 65535) LTX_FAIL(LTX_ERR_ARG, "ltx_write_gif: bad argument");
    speed = speed < 1 ? 1 : (speed > 30 ? 30 : speed);
    const size_t npix = (size_t)width * height;
    std::vector<std::vector<uint8_t>> blocks(n_frames);
    std::atomic<int> next{0};
    auto work = [&]() {
        std::vector<uint8_t> idx(npix);
        std::vector<NeuQuant> nqv(1); NeuQuant& nq = nqv[0];
        for (int f; (f = next.fetch_add(1)) < n_frames;) {
            const uint8_t* rgb = rgb_frames + (size_t)f * npix * 3;
            nq.init(); nq.learn(rgb, (int)npix, speed); nq.finish();
            for (size_t p = 0; p < npix; ++p) idx[p] = (uint8_t)nq.search(rgb[3 * p + 2], rgb[3 * p + 1], rgb[3 * p]);
            std::vector<uint8_t>& b = blocks[f];
            b.push_back(0x21); b.push_back(0xF9); b.push_back(4); b.push_back(0); put16(b, delay_cs); b.push_back(0); b.push_back(0);   // graphic control: no disposal, no transparency
            b.push_back(0x2C); put16(b, 0); put16(b, 0); put16(b, width); put16(b, height); b.push_back(0x80 | 7);                  // image descriptor, local table of 256
            for (int i = 0; i < 256; ++i) { b.push_back((uint8_t)nq.net[i][2]); b.push_back((uint8_t)nq.net[i][1]); b.push_back((uint8_t)nq.net[i][0]); }
            b.push_back(8);                                                                                                           // LZW minimum code size
            lzw_encode(idx.data(), npix, b);
        }
    };
    unsigned nth = std::thread::hardware_concurrency(); if (nth < 1) nth = 1; if (nth > 16) nth = 16; if ((int)nth > n_frames) nth = n_frames;
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nth; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    FILE* f = fopen(path, "wb");
    if (!f) LTX_FAIL(LTX_ERR_ARG, std::string("ltx_write_gif: cannot open '") + path + "'");
    std::vector<uint8_t> head = {'G', 'I', 'F', '8', '9', 'a'};
    put16(head, width); put16(head, height); head.push_back(0x00); head.push_back(0); head.push_back(0);        // no global colour table (Encoder::new(.., &[]))
    const uint8_t loop[] = {0x21, 0xFF, 11, 'N', 'E', 'T', 'S', 'C', 'A', 'P', 'E', '2', '.', '0', 3, 1, 0, 0, 0};   // Repeat::Infinite
    head.insert(head.end(), loop, loop + sizeof(loop));
    bool ok = fwrite(head.data(), 1, head.size(), f) == head.size();
    for (int i = 0; ok && i < n_frames; ++i) ok = fwrite(blocks[i].data(), 1, blocks[i].size(), f) == blocks[i].size();
    const uint8_t trailer = 0x3B;
    ok = ok && fwrite(&trailer, 1, 1, f) == 1;
    fclose(f);
    if (!ok) LTX_FAIL(LTX_ERR_ARG, std::string("ltx_write_gif: short write to '") + path + "'");
    return LTX_OK;
}
