// options.h: parsing of LTX_OPTIONS / ltx_set_option.
#include "options.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include "errors.h"

namespace {
std::mutex g_mu;
LtxOptions g_opt;
std::atomic<bool> g_loaded{false};                    // published with release / read with acquire: the first ltx_opt() of any thread sees a complete g_opt
std::map<std::string, int> g_exp;                       // "x_name" -> value (read only under LTX_EXPERIMENTS)

struct ListBit { const char* name; unsigned bit; };
const ListBit kFam[] = {{"asm16", LTX_FAM_ASM16}, {"ring", LTX_FAM_RING}, {"p8", LTX_FAM_P8}, {"halo", LTX_FAM_HALO}, {"halo_out", LTX_FAM_HALO_OUT}, {"big", LTX_FAM_BIG}};
const ListBit kAttn[] = {{"q64", LTX_ATTN_Q64}, {"q128", LTX_ATTN_Q128}, {"cross", LTX_ATTN_CROSS}, {"pipe", LTX_ATTN_PIPE}};

template <size_t N> bool parse_list(const char* v, const ListBit (&tab)[N], unsigned* out) {
    unsigned bits = 0;
    std::string s(v ? v : "");
    size_t p = 0;
    while (p < s.size()) {
        size_t q = s.find('+', p); if (q == std::string::npos) q = s.size();
        const std::string item = s.substr(p, q - p);
        bool ok = item.empty();
        for (const auto& t : tab) if (item == t.name) { bits |= t.bit; ok = true; }
        if (!ok) return false;
        p = q + 1;
    }
    *out = bits;
    return true;
}

struct IntField { const char* name; int LtxOptions::*field; };
const IntField kInts[] = {
    {"gemm_tune", &LtxOptions::gemm_tune}, {"gemm_wide_epi", &LtxOptions::gemm_wide_epi}, {"gemm_trace", &LtxOptions::gemm_trace},
    {"attn_q64_big", &LtxOptions::attn_q64_big}, {"vae_tile_batch", &LtxOptions::vae_tile_batch}, {"prof_kernel_events", &LtxOptions::prof_kernel_events}, {"ff2_defer", &LtxOptions::ff2_defer},
    {"gemm_splitk", &LtxOptions::gemm_splitk}, {"q2_fold", &LtxOptions::q2_fold}, {"norm_presum", &LtxOptions::norm_presum}, {"norm_lean", &LtxOptions::norm_lean},
    {"xattn_compact", &LtxOptions::xattn_compact}, {"dense_qkv", &LtxOptions::dense_qkv}, {"vae_fuse_norm", &LtxOptions::vae_fuse_norm}, {"t5_attn_mfma", &LtxOptions::t5_attn_mfma}, {"guidance_batch", &LtxOptions::guidance_batch}, {"norm_fold", &LtxOptions::norm_fold}, {"norm_fold_copies", &LtxOptions::norm_fold_copies}, {"attn_q64_stream", &LtxOptions::attn_q64_stream},
};

// value == nullptr: the option's default
bool set_one(LtxOptions& o, const std::string& key, const char* value) {
    static const LtxOptions dflt;
    for (const auto& f : kInts) if (key == f.name) {
        if (!value) { o.*(f.field) = dflt.*(f.field); return true; }
        char* end = nullptr; const long v = strtol(value, &end, 10);
        if (end == value || *end) return false;
        o.*(f.field) = (int)v; return true;
    }
    if (key == "gemm_plan") {
        if (!value) { o.gemm_plan[0] = 0; return true; }
        if (strlen(value) >= sizeof(o.gemm_plan)) return false;
        strcpy(o.gemm_plan, value); return true;
    }
    if (key == "gemm_off") return value ? parse_list(value, kFam, &o.gemm_off) : (o.gemm_off = 0, true);
    if (key == "attn_off") return value ? parse_list(value, kAttn, &o.attn_off) : (o.attn_off = 0, true);
    if (key.rfind("x_", 0) == 0) {                         // experiment knobs: kept in every build, read only by experiment builds
        if (!value) { g_exp.erase(key); return true; }
        char* end = nullptr; const long v = strtol(value, &end, 10);
        if (end == value || *end) return false;
        g_exp[key] = (int)v; return true;
    }
    return false;
}

bool parse_string(LtxOptions& o, const char* text, std::string* bad) {
    std::string s(text ? text : "");
    size_t p = 0;
    while (p < s.size()) {
        size_t q = s.find(',', p); if (q == std::string::npos) q = s.size();
        const std::string item = s.substr(p, q - p);
        p = q + 1;
        if (item.empty()) continue;
        const size_t eq = item.find('=');
        if (eq == std::string::npos || !set_one(o, item.substr(0, eq), item.c_str() + eq + 1)) { if (bad) *bad = item; return false; }
    }
    return true;
}

void load_locked() {
    g_opt = LtxOptions(); g_exp.clear();
    std::string bad;
    if (!parse_string(g_opt, getenv("LTX_OPTIONS"), &bad))
        fprintf(stderr, "[ltx] LTX_OPTIONS: entry '%s' not understood (see include/ltxhip.h); it and what follows it are ignored\n", bad.c_str());
    g_loaded.store(true, std::memory_order_release);
}
}  // namespace

const LtxOptions& ltx_opt() {
    if (!g_loaded.load(std::memory_order_acquire)) { std::lock_guard<std::mutex> lock(g_mu); if (!g_loaded.load(std::memory_order_relaxed)) load_locked(); }
    return g_opt;
}

int ltx_exp_lookup(const char* name, int dflt) {
    (void)ltx_opt();
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_exp.find(std::string("x_") + name);
    return it == g_exp.end() ? dflt : it->second;
}

extern "C" int ltx_set_option(const char* key, const char* value) {
    if (!key) LTX_FAIL(LTX_ERR_ARG, "ltx_set_option: null key");
    (void)ltx_opt();
    std::lock_guard<std::mutex> lock(g_mu);
    if (!set_one(g_opt, key, value)) LTX_FAIL(LTX_ERR_ARG, std::string("ltx_set_option: unknown option or bad value: ") + key + "=" + (value ? value : "(default)"));
    return LTX_OK;
}

// The current value of an option as ltx_set_option would take it back (lists as "a+b", "" = none / default plan).
extern "C" int ltx_get_option(const char* key, char* out, int n) {
    if (!key || !out || n < 2) LTX_FAIL(LTX_ERR_ARG, "ltx_get_option: null key / buffer");
    (void)ltx_opt();
    std::lock_guard<std::mutex> lock(g_mu);
    const std::string k(key);
    std::string v; bool found = false;
    for (const auto& f : kInts) if (k == f.name) { v = std::to_string(g_opt.*(f.field)); found = true; }
    auto list = [&](unsigned bits, const ListBit* tab, size_t cnt) { for (size_t i = 0; i < cnt; ++i) if (bits & tab[i].bit) { if (!v.empty()) v += "+"; v += tab[i].name; } found = true; };
    if (k == "gemm_plan") { v = g_opt.gemm_plan; found = true; }
    if (k == "gemm_off") list(g_opt.gemm_off, kFam, sizeof(kFam) / sizeof(kFam[0]));
    if (k == "attn_off") list(g_opt.attn_off, kAttn, sizeof(kAttn) / sizeof(kAttn[0]));
    if (k.rfind("x_", 0) == 0) { auto it = g_exp.find(k); if (it == g_exp.end()) LTX_FAIL(LTX_ERR_ARG, "ltx_get_option: experiment knob not set: " + k); v = std::to_string(it->second); found = true; }
    if (!found) LTX_FAIL(LTX_ERR_ARG, "ltx_get_option: unknown option: " + k);
    if ((int)v.size() + 1 > n) LTX_FAIL(LTX_ERR_ARG, "ltx_get_option: buffer too small");
    memcpy(out, v.c_str(), v.size() + 1);
    return LTX_OK;
}

extern "C" int ltx_reset_options(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    load_locked();
    return LTX_OK;
}

extern "C" int ltx_has_experiments(void) {
#ifdef LTX_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}
