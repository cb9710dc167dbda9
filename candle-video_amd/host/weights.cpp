// Weight ingestion (include/ltxhip_weights.h): Official -> Diffusers key remapping, name-mapping rules, safetensors
// reader (mmap, zero host copies) and the from-files model constructors.  Host-only C++; restates
// weight_format.rs / loader.rs / main.rs:455-546 of the reference (cited per function in the header).
#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstring>
#include <memory>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/ltxhip_weights.h"
#include "../csrc/errors.h"

namespace {

// ---- str helpers -------------------------------------------------------------------------------------------------
std::string replace_all(std::string s, const std::string& from, const std::string& to) {
    if (from.empty()) return s;
    size_t pos = 0;
    while ((pos = s.find(from, pos)) != std::string::npos) { s.replace(pos, from.size(), to); pos += to.size(); }
    return s;
}
bool starts_with(const std::string& s, const std::string& p) { return s.size() >= p.size() && s.compare(0, p.size(), p) == 0; }
bool ends_with(const std::string& s, const std::string& p) { return s.size() >= p.size() && s.compare(s.size() - p.size(), p.size(), p) == 0; }
bool contains(const std::string& s, const char* p) { return s.find(p) != std::string::npos; }

// replace every "<prefix><digits>" by table[digits] (or "<fallback_prefix><digits>" beyond the table)
std::string remap_blocks(const std::string& key, const std::string& prefix, const char* const* table, size_t n_table) {
    std::string out; size_t pos = 0;
    while (true) {
        size_t hit = key.find(prefix, pos);
        if (hit == std::string::npos) { out.append(key, pos, std::string::npos); break; }
        size_t d0 = hit + prefix.size(), d1 = d0;
        while (d1 < key.size() && key[d1] >= '0' && key[d1] <= '9') ++d1;
        if (d1 == d0) { out.append(key, pos, d0 - pos); pos = d0; continue; }      // prefix without an index: not a match
        out.append(key, pos, hit - pos);
        unsigned long long idx = 0; bool ovf = false;
        for (size_t i = d0; i < d1; ++i) { idx = idx * 10 + (unsigned)(key[i] - '0'); if (idx > 1000000000ULL) ovf = true; }
        if (ovf) idx = 0;                                                          // parse().unwrap_or(0)
        if (idx < n_table) out += table[idx];
        else out += prefix + std::to_string(idx);
        pos = d1;
    }
    return out;
}

const char* const kEncoderBlocks[] = {                                             // weight_format.rs:96-113
    "encoder.down_blocks.0", "encoder.down_blocks.0.downsamplers.0", "encoder.down_blocks.1", "encoder.down_blocks.1.downsamplers.0",
    "encoder.down_blocks.2", "encoder.down_blocks.2.downsamplers.0", "encoder.down_blocks.3", "encoder.down_blocks.3.downsamplers.0",
    "encoder.mid_block"};
const char* const kDecoderBlocks[] = {                                             // weight_format.rs:124-141
    "decoder.mid_block", "decoder.up_blocks.0.upsamplers.0", "decoder.up_blocks.0", "decoder.up_blocks.1.upsamplers.0",
    "decoder.up_blocks.1", "decoder.up_blocks.2.upsamplers.0", "decoder.up_blocks.2", "decoder.up_blocks.3.upsamplers.0",
    "decoder.up_blocks.3"};

std::string remap_key(const std::string& key) {
    std::string r = key;
    r = replace_all(r, "patchify_proj", "proj_in");                               // 1. transformer
    r = replace_all(r, "adaln_single", "time_embed");
    r = replace_all(r, "q_norm", "norm_q");
    r = replace_all(r, "k_norm", "norm_k");
    r = replace_all(r, "res_blocks", "resnets");                                  // 2. VAE
    r = remap_blocks(r, "encoder.down_blocks.", kEncoderBlocks, 9);               // 3.
    r = remap_blocks(r, "decoder.up_blocks.", kDecoderBlocks, 9);                 // 4.
    r = replace_all(r, "last_time_embedder", "time_embedder");                    // 5.
    r = replace_all(r, "last_scale_shift_table", "scale_shift_table");
    r = replace_all(r, "norm3.norm", "norm3");
    r = replace_all(r, "per_channel_statistics.mean-of-means", "latents_mean");
    r = replace_all(r, "per_channel_statistics.std-of-means", "latents_std");
    return r;
}
bool is_transformer_key(const std::string& k) {
    return starts_with(k, "transformer.") || starts_with(k, "model.diffusion_model.") || contains(k, "transformer_blocks") ||
           contains(k, "patchify_proj") || contains(k, "proj_in") || contains(k, "adaln_single") || contains(k, "time_embed");
}
bool is_vae_key(const std::string& k) {
    return starts_with(k, "vae.") || starts_with(k, "encoder.") || starts_with(k, "decoder.") || contains(k, "per_channel_statistics") ||
           contains(k, "latents_mean") || contains(k, "latents_std");
}
int put(const std::string& s, char* out, size_t cap, const char* what) {
    if (!out || s.size() + 1 > cap) LTX_FAIL(LTX_ERR_ARG, std::string(what) + ": output buffer too small");
    memcpy(out, s.c_str(), s.size() + 1);
    return LTX_OK;
}

// ---- minimal JSON (objects, arrays, strings, numbers, literals): enough for safetensors headers / index.json --------
struct Json {
    enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
    double num = 0; bool b = false; std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;          // keeps file order
    const Json* get(const char* k) const { for (auto& kv : obj) if (kv.first == k) return &kv.second; return nullptr; }
};
struct JsonParser {
    const char* p; const char* e; std::string err;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    bool str(std::string& out) {
        if (p >= e || *p != '"') return fail("expected string");
        ++p;
        while (p < e && *p != '"') {
            if (*p == '\\') {
                if (++p >= e) return fail("bad escape");
                switch (*p) {
                    case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                    case 'u': {
                        if (e - p < 5) return fail("bad \\u escape");
                        unsigned cp = 0;
                        for (int i = 1; i <= 4; ++i) { char c = p[i]; cp <<= 4; cp |= (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : 0; }
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        p += 4; break;
                    }
                    default: out += *p;
                }
                ++p;
            } else out += *p++;
        }
        if (p >= e) return fail("unterminated string");
        ++p; return true;
    }
    bool value(Json& v, int depth = 0) {
        if (depth > 64) return fail("nesting too deep");
        ws();
        if (p >= e) return fail("unexpected end");
        if (*p == '{') {
            v.kind = Json::OBJ; ++p; ws();
            if (p < e && *p == '}') { ++p; return true; }
            while (true) {
                ws(); std::string k; if (!str(k)) return false;
                ws(); if (p >= e || *p != ':') return fail("expected ':'"); ++p;
                Json c; if (!value(c, depth + 1)) return false;
                v.obj.emplace_back(std::move(k), std::move(c));
                ws(); if (p < e && *p == ',') { ++p; continue; }
                if (p < e && *p == '}') { ++p; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (*p == '[') {
            v.kind = Json::ARR; ++p; ws();
            if (p < e && *p == ']') { ++p; return true; }
            while (true) {
                Json c; if (!value(c, depth + 1)) return false;
                v.arr.push_back(std::move(c));
                ws(); if (p < e && *p == ',') { ++p; continue; }
                if (p < e && *p == ']') { ++p; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (*p == '"') { v.kind = Json::STR; return str(v.str); }
        if (!strncmp(p, "true", std::min<size_t>(4, e - p)) && e - p >= 4) { v.kind = Json::BOOL; v.b = true; p += 4; return true; }
        if (!strncmp(p, "false", std::min<size_t>(5, e - p)) && e - p >= 5) { v.kind = Json::BOOL; v.b = false; p += 5; return true; }
        if (!strncmp(p, "null", std::min<size_t>(4, e - p)) && e - p >= 4) { v.kind = Json::NUL; p += 4; return true; }
        char* endp = nullptr; std::string tmp(p, std::min<size_t>(64, e - p));
        double d = strtod(tmp.c_str(), &endp);
        if (endp == tmp.c_str()) return fail("bad value");
        v.kind = Json::NUM; v.num = d; p += endp - tmp.c_str(); return true;
    }
};

struct Mapping { int fd = -1; void* base = nullptr; size_t len = 0;
    ~Mapping() { if (base && base != MAP_FAILED) munmap(base, len); if (fd >= 0) close(fd); } };

}  // namespace

struct ltx_safetensors {
    Mapping map;
    struct Entry { std::string name, dtype; std::vector<int64_t> shape; const void* data; size_t nbytes; };
    std::vector<Entry> entries;
    std::string path;
};
struct ltx_name_mapper {
    struct Rule { int kind; std::string from, to; };
    std::vector<Rule> rules;
    bool apply(const Rule& r, const std::string& name, std::string* out) const {
        if (r.kind == LTX_MAP_EXACT) { if (name != r.from) return false; *out = r.to; return true; }
        if (r.kind == LTX_MAP_PREFIX) { if (!starts_with(name, r.from)) return false; *out = r.to + name.substr(r.from.size()); return true; }
        if (!ends_with(name, r.from)) return false;
        *out = name.substr(0, name.size() - r.from.size()) + r.to; return true;
    }
};

extern "C" int ltx_weights_detect_format(const char* path) {
    struct stat st;
    return (path && stat(path, &st) == 0 && S_ISREG(st.st_mode)) ? 1 : 0;
}
extern "C" int ltx_weights_remap_key(const char* key, char* out, size_t cap) {
    if (!key) LTX_FAIL(LTX_ERR_ARG, "ltx_weights_remap_key: null key");
    return put(remap_key(key), out, cap, "ltx_weights_remap_key");
}
extern "C" int ltx_weights_is_transformer_key(const char* key) { return key && is_transformer_key(key) ? 1 : 0; }
extern "C" int ltx_weights_is_vae_key(const char* key) { return key && is_vae_key(key) ? 1 : 0; }

extern "C" ltx_name_mapper* ltx_name_mapper_create(void) { return new ltx_name_mapper(); }
extern "C" void ltx_name_mapper_destroy(ltx_name_mapper* m) { delete m; }
extern "C" int ltx_name_mapper_add(ltx_name_mapper* m, ltx_map_kind kind, const char* from, const char* to) {
    if (!m || !from || !to || kind < LTX_MAP_EXACT || kind > LTX_MAP_SUFFIX) LTX_FAIL(LTX_ERR_ARG, "ltx_name_mapper_add: bad argument");
    m->rules.push_back({(int)kind, from, to});
    return LTX_OK;
}
extern "C" int ltx_name_mapper_has_mapping(const ltx_name_mapper* m, const char* name) {
    if (!m || !name) return 0;
    std::string tmp;
    for (auto& r : m->rules) if (m->apply(r, name, &tmp)) return 1;
    return 0;
}
extern "C" int ltx_name_mapper_map(const ltx_name_mapper* m, const char* name, char* out, size_t cap) {
    if (!m || !name) LTX_FAIL(LTX_ERR_ARG, "ltx_name_mapper_map: null argument");
    std::string cur = name, next;
    for (auto& r : m->rules) if (m->apply(r, cur, &next)) cur = next;
    return put(cur, out, cap, "ltx_name_mapper_map");
}
extern "C" int ltx_weights_validate_names(const char* const* expected, size_t n_expected, const char* const* actual, size_t n_actual,
                                          size_t* missing_idx, size_t* n_missing) {
    if ((!expected && n_expected) || (!actual && n_actual) || !n_missing) LTX_FAIL(LTX_ERR_ARG, "ltx_weights_validate_names: null argument");
    std::set<std::string> have;
    for (size_t i = 0; i < n_actual; ++i) have.insert(actual[i]);
    size_t n = 0;
    for (size_t i = 0; i < n_expected; ++i)
        if (!have.count(expected[i])) { if (missing_idx) missing_idx[n] = i; ++n; }
    *n_missing = n;
    return LTX_OK;
}

extern "C" int ltx_safetensors_open(const char* path, ltx_safetensors** out) {
    if (!path || !out) LTX_FAIL(LTX_ERR_ARG, "ltx_safetensors_open: null argument");
    *out = nullptr;
    std::unique_ptr<ltx_safetensors> st(new ltx_safetensors());
    st->path = path;
    st->map.fd = open(path, O_RDONLY);
    if (st->map.fd < 0) LTX_FAIL(LTX_ERR_ARG, std::string("cannot open '") + path + "': " + strerror(errno));
    struct stat sb;
    if (fstat(st->map.fd, &sb) != 0 || sb.st_size < 8) LTX_FAIL(LTX_ERR_ARG, std::string("'") + path + "' is not a safetensors file (too short)");
    st->map.len = (size_t)sb.st_size;
    st->map.base = mmap(nullptr, st->map.len, PROT_READ, MAP_PRIVATE, st->map.fd, 0);
    if (st->map.base == MAP_FAILED) { st->map.base = nullptr; LTX_FAIL(LTX_ERR_ARG, std::string("mmap '") + path + "': " + strerror(errno)); }
    const unsigned char* b = static_cast<const unsigned char*>(st->map.base);
    uint64_t hlen = 0;
    for (int i = 7; i >= 0; --i) hlen = (hlen << 8) | b[i];
    if (hlen > st->map.len - 8) LTX_FAIL(LTX_ERR_ARG, std::string("'") + path + "': header length exceeds the file");
    JsonParser jp{reinterpret_cast<const char*>(b + 8), reinterpret_cast<const char*>(b + 8 + hlen), {}};
    Json root;
    if (!jp.value(root) || root.kind != Json::OBJ) LTX_FAIL(LTX_ERR_ARG, std::string("'") + path + "': bad safetensors header: " + (jp.err.empty() ? "not an object" : jp.err));
    const size_t payload = st->map.len - 8 - (size_t)hlen;
    const unsigned char* data0 = b + 8 + hlen;
    for (auto& kv : root.obj) {
        if (kv.first == "__metadata__") continue;
        const Json* dt = kv.second.get("dtype"); const Json* sh = kv.second.get("shape"); const Json* off = kv.second.get("data_offsets");
        if (!dt || dt->kind != Json::STR || !sh || sh->kind != Json::ARR || !off || off->kind != Json::ARR || off->arr.size() != 2)
            LTX_FAIL(LTX_ERR_ARG, std::string("'") + path + "': malformed entry '" + kv.first + "'");
        ltx_safetensors::Entry en; en.name = kv.first; en.dtype = dt->str;
        for (auto& d : sh->arr) en.shape.push_back((int64_t)d.num);
        const double a0 = off->arr[0].num, a1 = off->arr[1].num;
        if (a0 < 0 || a1 < a0 || a1 > (double)payload) LTX_FAIL(LTX_ERR_ARG, std::string("'") + path + "': offsets of '" + kv.first + "' fall outside the file");
        en.data = data0 + (size_t)a0; en.nbytes = (size_t)(a1 - a0);
        st->entries.push_back(std::move(en));
    }
    *out = st.release();
    return LTX_OK;
}
extern "C" void ltx_safetensors_close(ltx_safetensors* st) { delete st; }
extern "C" size_t ltx_safetensors_count(const ltx_safetensors* st) { return st ? st->entries.size() : 0; }
extern "C" int ltx_safetensors_tensor(const ltx_safetensors* st, size_t i, const char** name, const char** dtype_name,
                                      int* ndim, const int64_t** shape, const void** data, size_t* nbytes) {
    if (!st || i >= st->entries.size()) LTX_FAIL(LTX_ERR_ARG, "ltx_safetensors_tensor: index out of range");
    const auto& e = st->entries[i];
    if (name) *name = e.name.c_str();
    if (dtype_name) *dtype_name = e.dtype.c_str();
    if (ndim) *ndim = (int)e.shape.size();
    if (shape) *shape = e.shape.data();
    if (data) *data = e.data;
    if (nbytes) *nbytes = e.nbytes;
    return LTX_OK;
}

namespace {
bool file_exists(const std::string& p) { struct stat sb; return stat(p.c_str(), &sb) == 0 && S_ISREG(sb.st_mode); }
bool dir_exists(const std::string& p) { struct stat sb; return stat(p.c_str(), &sb) == 0 && S_ISDIR(sb.st_mode); }

int resolve(const std::string& path, std::vector<std::string>* files) {
    files->clear();
    if (file_exists(path)) { files->push_back(path); return LTX_OK; }
    if (!dir_exists(path)) LTX_FAIL(LTX_ERR_ARG, "weights path '" + path + "' does not exist");
    const std::string index = path + "/model.safetensors.index.json";
    if (file_exists(index)) {                                                     // loader.rs:349-372
        Mapping m; m.fd = open(index.c_str(), O_RDONLY);
        struct stat sb;
        if (m.fd < 0 || fstat(m.fd, &sb) != 0) LTX_FAIL(LTX_ERR_ARG, "cannot read '" + index + "'");
        std::string text((size_t)sb.st_size, '\0');
        if (sb.st_size && read(m.fd, &text[0], (size_t)sb.st_size) != sb.st_size) LTX_FAIL(LTX_ERR_ARG, "cannot read '" + index + "'");
        JsonParser jp{text.data(), text.data() + text.size(), {}};
        Json root;
        if (!jp.value(root) || root.kind != Json::OBJ) LTX_FAIL(LTX_ERR_ARG, "'" + index + "': invalid JSON");
        const Json* wm = root.get("weight_map");
        if (!wm || wm->kind != Json::OBJ) LTX_FAIL(LTX_ERR_ARG, "'" + index + "': no weight_map");
        std::set<std::string> shards;                                              // SafetensorsIndex::shard_files (:160-166)
        for (auto& kv : wm->obj) if (kv.second.kind == Json::STR) shards.insert(kv.second.str);
        std::string missing;
        for (auto& s : shards) {
            if (file_exists(path + "/" + s)) files->push_back(path + "/" + s);
            else missing += (missing.empty() ? "" : ", ") + s;
        }
        if (!missing.empty()) LTX_FAIL(LTX_ERR_MISSING_WEIGHT, "missing shard files: " + missing);
        return LTX_OK;
    }
    if (file_exists(path + "/model.safetensors")) { files->push_back(path + "/model.safetensors"); return LTX_OK; }   // :375-378
    DIR* d = opendir(path.c_str());                                               // find_sharded_files (:437-456)
    if (!d) LTX_FAIL(LTX_ERR_ARG, "cannot list '" + path + "'");
    while (dirent* de = readdir(d)) { std::string n = de->d_name; if (ends_with(n, ".safetensors")) files->push_back(path + "/" + n); }
    closedir(d);
    std::sort(files->begin(), files->end());
    if (files->empty()) LTX_FAIL(LTX_ERR_MISSING_WEIGHT, "no safetensors files found in '" + path + "'");
    return LTX_OK;
}

#ifndef LTX_HOST_ONLY          /* the sanitizer build (make asan) has no device library to create models in */
struct Loaded {
    std::vector<ltx_safetensors*> files;
    std::vector<std::string> names;          // stable storage for ltx_weight.name
    std::vector<ltx_weight> weights;
    ~Loaded() { for (auto* f : files) ltx_safetensors_close(f); }
};

// component: 0 = transformer, 1 = VAE
int gather(const char* path, int unified, int component, Loaded* L) {
    if (!path) LTX_FAIL(LTX_ERR_ARG, "weights path is null");
    std::vector<std::string> files;
    LTX_TRY(resolve(path, &files));
    std::map<std::string, size_t> seen;
    for (auto& f : files) {
        ltx_safetensors* st = nullptr;
        LTX_TRY(ltx_safetensors_open(f.c_str(), &st));
        L->files.push_back(st);
        for (auto& e : st->entries) {
            std::string name = e.name;
            if (unified) {                                                        // main.rs:480-498
                const std::string remapped = remap_key(e.name);
                if (is_vae_key(e.name)) {
                    if (component != 1) continue;
                    name = starts_with(remapped, "vae.") ? remapped.substr(4) : remapped;
                } else if (is_transformer_key(e.name)) {
                    if (component != 0) continue;
                    if (starts_with(remapped, "model.diffusion_model.")) name = remapped.substr(22);
                    else if (starts_with(remapped, "transformer.")) name = remapped.substr(12);
                    else name = remapped;
                } else continue;
            }
            int dt;
            if (e.dtype == "F32") dt = LTX_F32; else if (e.dtype == "BF16") dt = LTX_BF16;
            else {
                // tensors the path never reads (e.g. an encoder stored in another dtype) must not block loading: skip
                // them here; a needed one surfaces later as "missing weight '<name>'" from the constructor.
                continue;
            }
            if (e.shape.size() > 5) continue;
            int64_t numel = 1; for (auto d : e.shape) numel *= d;
            if ((size_t)numel * (dt == LTX_BF16 ? 2 : 4) != e.nbytes) LTX_FAIL(LTX_ERR_ARG, "'" + f + "': tensor '" + e.name + "' has " + std::to_string(e.nbytes) + " bytes, shape says otherwise");
            if (seen.count(name)) continue;                                       // first shard wins (HashMap insert order is unspecified upstream)
            seen[name] = L->weights.size();
            ltx_weight w; memset(&w, 0, sizeof(w));
            w.data = e.data; w.dtype = (ltx_dtype)dt; w.ndim = (int)e.shape.size();
            for (size_t i = 0; i < e.shape.size(); ++i) w.shape[i] = e.shape[i];
            w.on_device = 0;
            L->names.push_back(name);
            L->weights.push_back(w);
        }
    }
    for (size_t i = 0; i < L->weights.size(); ++i) L->weights[i].name = L->names[i].c_str();
    if (L->weights.empty()) LTX_FAIL(LTX_ERR_MISSING_WEIGHT, std::string("no ") + (component ? "VAE" : "transformer") + " tensors found in '" + path + "'");
    return LTX_OK;
}
#endif
}  // namespace

extern "C" int ltx_weights_resolve(const char* path, char* out, size_t cap, size_t* n_files) {
    if (!path || !out || !n_files) LTX_FAIL(LTX_ERR_ARG, "ltx_weights_resolve: null argument");
    std::vector<std::string> files;
    LTX_TRY(resolve(path, &files));
    size_t need = 1;
    for (auto& f : files) need += f.size() + 1;
    if (need > cap) LTX_FAIL(LTX_ERR_ARG, "ltx_weights_resolve: output buffer too small");
    char* p = out;
    for (auto& f : files) { memcpy(p, f.c_str(), f.size() + 1); p += f.size() + 1; }
    *p = '\0';
    *n_files = files.size();
    return LTX_OK;
}

// AutoencoderKLLtxVideoConfig's serde view of vae/config.json (vae.rs:30-66): field names and aliases, lists in file order
extern "C" int ltx_vae_config_from_json(const char* json_path, ltx_vae_config* cfg) {
    if (!json_path || !cfg) LTX_FAIL(LTX_ERR_ARG, "ltx_vae_config_from_json: null argument");
    std::string text;
    {
        FILE* f = fopen(json_path, "rb");
        if (!f) LTX_FAIL(LTX_ERR_ARG, std::string("cannot open '") + json_path + "'");
        char buf[4096]; size_t n;
        while ((n = fread(buf, 1, sizeof(buf), f)) > 0) { text.append(buf, n); if (text.size() > (1u << 22)) break; }
        fclose(f);
    }
    JsonParser jp{text.data(), text.data() + text.size(), {}};
    Json root;
    if (!jp.value(root) || root.kind != Json::OBJ) LTX_FAIL(LTX_ERR_ARG, std::string("'") + json_path + "': invalid JSON" + (jp.err.empty() ? "" : ": " + jp.err));
    auto field = [&](const char* name, const char* alias) -> const Json* { const Json* j = root.get(name); return j ? j : (alias ? root.get(alias) : nullptr); };
    auto bad = [&](const char* name) { ltx_set_error(std::string("'") + json_path + "': field '" + name + "' has the wrong type or length"); return LTX_ERR_ARG; };
    auto num_i = [&](const char* name, const char* alias, int* dst) -> int {
        const Json* j = field(name, alias); if (!j) return LTX_OK;
        if (j->kind != Json::NUM || j->num < 0 || j->num > 1e9 || j->num != (double)(int64_t)j->num) return bad(name);
        *dst = (int)j->num; return LTX_OK;
    };
    auto num_f = [&](const char* name, const char* alias, float* dst) -> int {
        const Json* j = field(name, alias); if (!j) return LTX_OK;
        if (j->kind != Json::NUM) return bad(name);
        *dst = (float)j->num; return LTX_OK;
    };
    auto flag = [&](const char* name, const char* alias, int* dst) -> int {
        const Json* j = field(name, alias); if (!j) return LTX_OK;
        if (j->kind != Json::BOOL) return bad(name);
        *dst = j->b ? 1 : 0; return LTX_OK;
    };
    // lists: at most `cap` entries; *count (when given) receives the length
    auto list = [&](const char* name, const char* alias, int* dst, int cap, bool booleans, int* count) -> int {
        const Json* j = field(name, alias); if (!j) return LTX_OK;
        if (j->kind != Json::ARR || (int)j->arr.size() > cap) return bad(name);
        for (size_t i = 0; i < j->arr.size(); ++i) {
            const Json& e = j->arr[i];
            if (booleans) { if (e.kind != Json::BOOL) return bad(name); dst[i] = e.b ? 1 : 0; }
            else { if (e.kind != Json::NUM || e.num < 0 || e.num > 1e9 || e.num != (double)(int64_t)e.num) return bad(name); dst[i] = (int)e.num; }
        }
        if (count) *count = (int)j->arr.size();
        return LTX_OK;
    };
    LTX_TRY(num_i("out_channels", nullptr, &cfg->out_channels));
    LTX_TRY(num_i("latent_channels", nullptr, &cfg->latent_channels));
    int nb = cfg->n_blocks;
    LTX_TRY(list("decoder_block_out_channels", nullptr, cfg->decoder_block_out_channels, 4, false, &nb));
    if (nb < 1) return bad("decoder_block_out_channels");
    cfg->n_blocks = nb;
    LTX_TRY(list("decoder_spatiotemporal_scaling", "decoder_spatio_temporal_scaling", cfg->decoder_spatiotemporal_scaling, 4, true, nullptr));
    LTX_TRY(list("decoder_layers_per_block", nullptr, cfg->decoder_layers_per_block, 5, false, nullptr));
    LTX_TRY(num_i("patch_size", nullptr, &cfg->patch_size));
    LTX_TRY(num_i("patch_size_t", nullptr, &cfg->patch_size_t));
    LTX_TRY(num_f("resnet_eps", "resnet_norm_eps", &cfg->resnet_eps));
    LTX_TRY(num_f("scaling_factor", nullptr, &cfg->scaling_factor));
    LTX_TRY(num_i("spatial_compression_ratio", nullptr, &cfg->spatial_compression_ratio));
    LTX_TRY(num_i("temporal_compression_ratio", nullptr, &cfg->temporal_compression_ratio));
    LTX_TRY(list("decoder_inject_noise", nullptr, cfg->decoder_inject_noise, 5, true, nullptr));
    LTX_TRY(list("decoder_upsample_residual", "upsample_residual", cfg->decoder_upsample_residual, 4, true, nullptr));
    LTX_TRY(list("decoder_upsample_factor", "upsample_factor", cfg->decoder_upsample_factor, 4, false, nullptr));
    LTX_TRY(flag("timestep_conditioning", nullptr, &cfg->timestep_conditioning));
    LTX_TRY(flag("decoder_causal", nullptr, &cfg->decoder_causal));
    return LTX_OK;
}

#ifndef LTX_HOST_ONLY
extern "C" int ltx_dit_create_from_files(const ltx_dit_config* cfg, const char* path, int unified,
                                         ltx_dtype model_dtype, int device, ltx_dit** out) {
    Loaded L;
    LTX_TRY(gather(path, unified, 0, &L));
    return ltx_dit_create(cfg, L.weights.data(), L.weights.size(), model_dtype, device, out);
}
extern "C" int ltx_vae_create_from_files(const ltx_vae_config* cfg, const char* path, int unified,
                                         ltx_dtype model_dtype, int device, ltx_vae** out) {
    if (!cfg || !path) LTX_FAIL(LTX_ERR_ARG, "ltx_vae_create_from_files: null argument");
    ltx_vae_config c = *cfg;
    if (!unified) {                                         // main.rs:525-534: vae/config.json beside the weights replaces the preset's config
        struct stat st;
        std::string dir = path;
        if (!(stat(path, &st) == 0 && S_ISDIR(st.st_mode))) { const size_t k = dir.find_last_of('/'); dir = k == std::string::npos ? "." : dir.substr(0, k); }
        const std::string cj = dir + "/config.json";
        if (stat(cj.c_str(), &st) == 0 && S_ISREG(st.st_mode)) {
            ltx_vae_config_default(&c);                     // serde(default): fields the file lacks take Default::default()
            LTX_TRY(ltx_vae_config_from_json(cj.c_str(), &c));
            c.timestep_conditioning = 1;
        }
    }
    Loaded L;
    LTX_TRY(gather(path, unified, 1, &L));
    return ltx_vae_create(&c, L.weights.data(), L.weights.size(), model_dtype, device, out);
}
#endif
