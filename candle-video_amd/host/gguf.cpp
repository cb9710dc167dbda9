// GGUF container reader (include/ltxhip_weights.h): what `VarBuilder::from_gguf` gives the reference's default text encoder
// (src/models/ltx_video/quantized_t5_encoder.rs:570-600; candle's gguf_file is not in the checkout).  Host-only code: the
// file is mmap'ed, tensor payloads are never copied; restates the published GGUF v2 / v3 layout
//   magic "GGUF" | u32 version | u64 n_tensors | u64 n_kv | kv[n_kv] | tensor_info[n_tensors] | pad to alignment | data
//   kv          = string key | u32 type | value          (string = u64 length + bytes; array = u32 type | u64 count | items)
//   tensor_info = string name | u32 n_dims | u64 ne[n_dims] (innermost first) | u32 ggml type | u64 offset into data
// Every length is checked against the mapping: a truncated or lying file is an error, never an out-of-range read.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ltxhip_weights.h"
#include "../csrc/errors.h"

struct ltx_gguf {
    struct Entry { std::string name; int type = 0; int ndim = 0; int64_t shape[4] = {0, 0, 0, 0}; uint64_t offset = 0; size_t nbytes = 0; };
    void* map = nullptr; size_t size = 0; size_t data_start = 0;
    uint32_t version = 0; uint32_t alignment = 32;
    std::vector<Entry> entries;
    ~ltx_gguf() { if (map && map != MAP_FAILED) munmap(map, size); }
};

namespace {

struct Cursor {
    const unsigned char* p; size_t n, at = 0; bool ok = true;
    bool need(size_t k) { if (!ok || k > n - at) { ok = false; return false; } return true; }
    template <typename T> T get() { T v{}; if (need(sizeof(T))) { memcpy(&v, p + at, sizeof(T)); at += sizeof(T); } return v; }
    std::string str() {
        const uint64_t len = get<uint64_t>();
        if (!ok || len > (uint64_t)(n - at)) { ok = false; return std::string(); }
        std::string s(reinterpret_cast<const char*>(p + at), (size_t)len); at += (size_t)len; return s;
    }
    void skip(uint64_t k) { if (!ok || k > (uint64_t)(n - at)) ok = false; else at += (size_t)k; }
};

size_t scalar_size(uint32_t t) {
    switch (t) { case 0: case 1: case 7: return 1; case 2: case 3: return 2; case 4: case 5: case 6: return 4; case 10: case 11: case 12: return 8; default: return 0; }
}

// value of type t: skipped, except that a u32 is handed back (general.alignment)
bool skip_value(Cursor& c, uint32_t t, uint32_t* u32_out, int depth) {
    if (t == 8) { (void)c.str(); return c.ok; }
    if (t == 9) {
        if (depth > 0) return false;                                   // arrays of arrays do not occur in model files
        const uint32_t et = c.get<uint32_t>(); const uint64_t cnt = c.get<uint64_t>();
        if (!c.ok) return false;
        if (et == 8) { for (uint64_t i = 0; i < cnt && c.ok; ++i) (void)c.str(); return c.ok; }
        const size_t es = scalar_size(et);
        if (!es || cnt > (uint64_t)-1 / es) return false;
        c.skip(cnt * es); return c.ok;
    }
    const size_t sz = scalar_size(t);
    if (!sz) return false;
    if (t == 4 && u32_out) { *u32_out = c.get<uint32_t>(); return c.ok; }
    c.skip(sz); return c.ok;
}

}  // namespace

// elements per block and bytes per block of a ggml tensor type (0 = a type this library does not read)
extern "C" int ltx_gguf_type_info(int ggml_type, int* block_elems, int* block_bytes) {
    int be = 0, bb = 0;
    switch (ggml_type) {
        case 0: be = 1; bb = 4; break;            // F32
        case 1: be = 1; bb = 2; break;            // F16
        case 30: be = 1; bb = 2; break;           // BF16
        case 2: be = 32; bb = 18; break;          // Q4_0
        case 6: be = 32; bb = 22; break;          // Q5_0
        case 8: be = 32; bb = 34; break;          // Q8_0
        case 12: be = 256; bb = 144; break;       // Q4_K
        case 13: be = 256; bb = 176; break;       // Q5_K
        case 14: be = 256; bb = 210; break;       // Q6_K
        default: break;
    }
    if (block_elems) *block_elems = be;
    if (block_bytes) *block_bytes = bb;
    return be ? LTX_OK : LTX_ERR_UNSUPPORTED;
}

extern "C" int ltx_gguf_open(const char* path, ltx_gguf** out) {
    if (!path || !out) LTX_FAIL(LTX_ERR_ARG, "ltx_gguf_open: null argument");
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) LTX_FAIL(LTX_ERR_ARG, std::string("cannot open '") + path + "'");
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < 24) { close(fd); LTX_FAIL(LTX_ERR_ARG, std::string("'") + path + "': not a GGUF file (too short)"); }
    std::unique_ptr<ltx_gguf> g(new ltx_gguf());
    g->size = (size_t)sb.st_size;
    g->map = mmap(nullptr, g->size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (g->map == MAP_FAILED) { g->map = nullptr; LTX_FAIL(LTX_ERR_ARG, std::string("cannot map '") + path + "'"); }
    Cursor c{static_cast<const unsigned char*>(g->map), g->size};
    if (c.get<uint32_t>() != 0x46554747u) LTX_FAIL(LTX_ERR_ARG, std::string("'") + path + "': bad magic (not GGUF)");
    g->version = c.get<uint32_t>();
    if (g->version < 2 || g->version > 3) LTX_FAIL(LTX_ERR_UNSUPPORTED, std::string("'") + path + "': GGUF version " + std::to_string(g->version) + " (2 and 3 are read)");
    const uint64_t nt = c.get<uint64_t>(), nkv = c.get<uint64_t>();
    if (!c.ok || nt > g->size / 24 || nkv > g->size / 12) LTX_FAIL(LTX_ERR_ARG, std::string("'") + path + "': header counts exceed the file");
    for (uint64_t i = 0; i < nkv; ++i) {
        const std::string key = c.str();
        const uint32_t t = c.get<uint32_t>();
        uint32_t u = 0;
        if (!c.ok || !skip_value(c, t, key == "general.alignment" ? &u : nullptr, 0)) LTX_FAIL(LTX_ERR_ARG, std::string("'") + path + "': malformed metadata");
        if (key == "general.alignment" && t == 4) {
            if (u == 0 || (u & (u - 1)) || u > (1u << 20)) LTX_FAIL(LTX_ERR_ARG, std::string("'") + path + "': bad general.alignment");
            g->alignment = u;
        }
    }
    g->entries.reserve((size_t)nt);
    for (uint64_t i = 0; i < nt; ++i) {
        ltx_gguf::Entry e;
        e.name = c.str();
        const uint32_t nd = c.get<uint32_t>();
        if (!c.ok || nd < 1 || nd > 4) LTX_FAIL(LTX_ERR_ARG, std::string("'") + path + "': malformed tensor table");
        uint64_t ne[4] = {1, 1, 1, 1}; uint64_t numel = 1;
        for (uint32_t d = 0; d < nd; ++d) {
            ne[d] = c.get<uint64_t>();
            if (!c.ok || ne[d] == 0 || ne[d] > ((uint64_t)1 << 40) || numel > ((uint64_t)1 << 44) / ne[d]) LTX_FAIL(LTX_ERR_ARG, "tensor '" + e.name + "': bad dimensions");
            numel *= ne[d];
        }
        e.type = (int)c.get<uint32_t>(); e.offset = c.get<uint64_t>();
        if (!c.ok) LTX_FAIL(LTX_ERR_ARG, std::string("'") + path + "': malformed tensor table");
        e.ndim = (int)nd;
        for (uint32_t d = 0; d < nd; ++d) e.shape[d] = (int64_t)ne[nd - 1 - d];        // outermost first, as candle reports them
        int be = 0, bb = 0;
        if (ltx_gguf_type_info(e.type, &be, &bb) == LTX_OK) {
            if (ne[0] % (uint64_t)be) LTX_FAIL(LTX_ERR_ARG, "tensor '" + e.name + "': row length is not a multiple of its block size");
            e.nbytes = (size_t)(numel / (uint64_t)be) * (size_t)bb;
        } else e.nbytes = 0;                                                              // listed, but not readable (ltx_gguf_tensor says so)
        g->entries.push_back(std::move(e));
    }
    g->data_start = (c.at + g->alignment - 1) / g->alignment * g->alignment;
    if (nt && g->data_start > g->size) LTX_FAIL(LTX_ERR_ARG, std::string("'") + path + "': no data section");
    for (const ltx_gguf::Entry& e : g->entries) {
        if (e.offset % g->alignment) LTX_FAIL(LTX_ERR_ARG, "tensor '" + e.name + "': misaligned offset");
        if (e.offset > g->size - g->data_start || e.nbytes > g->size - g->data_start - e.offset)
            LTX_FAIL(LTX_ERR_ARG, "tensor '" + e.name + "': data range exceeds the file");
    }
    *out = g.release();
    return LTX_OK;
}

extern "C" void ltx_gguf_close(ltx_gguf* g) { delete g; }
extern "C" size_t ltx_gguf_count(const ltx_gguf* g) { return g ? g->entries.size() : 0; }

extern "C" int ltx_gguf_tensor(const ltx_gguf* g, size_t i, const char** name, int* ggml_type, int* ndim, const int64_t** shape,
                               const void** data, size_t* nbytes) {
    if (!g || i >= g->entries.size()) LTX_FAIL(LTX_ERR_ARG, "ltx_gguf_tensor: index out of range");
    const ltx_gguf::Entry& e = g->entries[i];
    if (name) *name = e.name.c_str();
    if (ggml_type) *ggml_type = e.type;
    if (ndim) *ndim = e.ndim;
    if (shape) *shape = e.shape;
    if (data) *data = static_cast<const unsigned char*>(g->map) + g->data_start + e.offset;
    if (nbytes) *nbytes = e.nbytes;
    if (!e.nbytes) LTX_FAIL(LTX_ERR_UNSUPPORTED, "tensor '" + e.name + "': ggml type " + std::to_string(e.type) + " is not read (F32, F16, BF16, Q4_0, Q5_0, Q8_0, Q4_K, Q5_K, Q6_K are)");
    return LTX_OK;
}

extern "C" int ltx_gguf_find(const ltx_gguf* g, const char* name) {
    if (!g || !name) return -1;
    for (size_t i = 0; i < g->entries.size(); ++i) if (g->entries[i].name == name) return (int)i;
    return -1;
}
