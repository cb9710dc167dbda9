// Host-side sanitizer driver (sanitizers run on the CPU build only): the string / IO / encoder code of the
// boundary - key remapping, name mapping, safetensors parsing, file resolution, presets, PNG and GIF writers - compiled
// host-only with AddressSanitizer + UndefinedBehaviorSanitizer (make -C candle-video_amd asan) and driven through the C
// ABI on well-formed AND malformed inputs.  No GPU call is made.  Exit code 0 = clean; the sanitizers abort otherwise.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <sys/stat.h>
#include "../../include/ltxhip_weights.h"
#include "../../include/ltxhip_presets.h"
#include "../../include/ltxhip_frames.h"

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (line %d): %s\n", #c, __LINE__, ltx_last_error()); return 1; } } while (0)

static void write_file(const std::string& path, const std::string& bytes) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (f) { std::fwrite(bytes.data(), 1, bytes.size(), f); std::fclose(f); }
}
static std::string safetensors_blob(const std::string& header, size_t payload, uint64_t claim = ~0ull) {
    uint64_t n = claim == ~0ull ? header.size() : claim;
    std::string s((const char*)&n, 8);
    s += header; s += std::string(payload, '\x01');
    return s;
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp/ltx_asan";
    mkdir(dir.c_str(), 0755);
    char buf[512], tiny[8];
    // ---- key remapping (weight_format.rs:55-163), including names longer than the output buffer
    const char* keys[] = {"model.diffusion_model.transformer_blocks.3.attn1.to_q.weight", "vae.decoder.up_blocks.2.res_blocks.1.conv1.conv.weight",
                          "vae.per_channel_statistics.mean-of-means", "patchify_proj.weight", "adaln_single.emb.timestep_embedder.linear_1.bias",
                          "", ".", "decoder.up_blocks.9999999999999999999.x", "q_norm", "k_norm.weight.k_norm.weight"};
    for (const char* k : keys) {
        int rc = ltx_weights_remap_key(k, buf, sizeof buf); CHECK(rc == 0);
        rc = ltx_weights_remap_key(k, tiny, sizeof tiny); (void)rc;                  // too small: must fail cleanly, never overrun
        (void)ltx_weights_is_transformer_key(k); (void)ltx_weights_is_vae_key(k);
    }
    std::string huge(5000, 'a'); huge += ".q_norm.weight";
    CHECK(ltx_weights_remap_key(huge.c_str(), buf, sizeof buf) != 0);
    // ---- name mapper (loader.rs:63-112, 223-317)
    ltx_name_mapper* m = ltx_name_mapper_create(); CHECK(m);
    CHECK(ltx_name_mapper_add(m, LTX_MAP_PREFIX, "model.", "") == 0);
    CHECK(ltx_name_mapper_add(m, LTX_MAP_SUFFIX, ".gamma", ".weight") == 0);
    CHECK(ltx_name_mapper_add(m, LTX_MAP_EXACT, "a", "b") == 0);
    CHECK(ltx_name_mapper_map(m, "model.norm.gamma", buf, sizeof buf) == 0 && std::strcmp(buf, "norm.weight") == 0);
    CHECK(ltx_name_mapper_has_mapping(m, "a") == 1 && ltx_name_mapper_has_mapping(m, "zzz") == 0);
    CHECK(ltx_name_mapper_map(m, huge.c_str(), tiny, sizeof tiny) != 0);
    ltx_name_mapper_destroy(m);
    const char* exp[] = {"x", "y", "z"}; const char* act[] = {"z", "x"};
    size_t miss[3], nmiss = 0;
    CHECK(ltx_weights_validate_names(exp, 3, act, 2, miss, &nmiss) == 0 && nmiss == 1 && miss[0] == 1);
    // ---- safetensors: one good file, then truncated / lying / malformed ones (every open must fail cleanly or succeed)
    const std::string good_hdr = "{\"w\":{\"dtype\":\"F32\",\"shape\":[2,3],\"data_offsets\":[0,24]},\"b\":{\"dtype\":\"BF16\",\"shape\":[4],\"data_offsets\":[24,32]},\"__metadata__\":{\"format\":\"pt\"}}";
    write_file(dir + "/good.safetensors", safetensors_blob(good_hdr, 32));
    ltx_safetensors* st = nullptr;
    CHECK(ltx_safetensors_open((dir + "/good.safetensors").c_str(), &st) == 0 && ltx_safetensors_count(st) == 2);
    for (size_t i = 0; i < 2; ++i) {
        const char* nm; const char* dt; int nd; const int64_t* shp; const void* data; size_t nb;
        CHECK(ltx_safetensors_tensor(st, i, &nm, &dt, &nd, &shp, &data, &nb) == 0);
        unsigned acc = 0; for (size_t j = 0; j < nb; ++j) acc += ((const unsigned char*)data)[j];        // touch every payload byte
        CHECK(acc == nb);
    }
    { const char* nm; const char* dt; int nd; const int64_t* shp; const void* data; size_t nb;
      CHECK(ltx_safetensors_tensor(st, 7, &nm, &dt, &nd, &shp, &data, &nb) != 0); }
    ltx_safetensors_close(st);
    const std::string bad[] = {
        std::string(),                                                     // empty file
        std::string("\x10\0\0", 3),                                         // shorter than the length word
        safetensors_blob(good_hdr, 32, 1ull << 40),                         // header length beyond the file
        safetensors_blob(good_hdr, 8),                                      // payload shorter than data_offsets claim
        safetensors_blob("{\"w\":{\"dtype\":\"F32\",\"shape\":[2,3],\"data_offsets\":[24,0]}}", 32),     // reversed offsets
        safetensors_blob("{\"w\":{\"dtype\":\"F32\",\"shape\":[-2,3],\"data_offsets\":[0,24]}}", 32),    // negative dim
        safetensors_blob("{\"w\":{\"dtype\":\"F32\",\"shape\":[2,3,", 32),                                 // cut JSON
        safetensors_blob("{\"w\":{\"dtype\":\"F32\",\"shape\":[99999999999,99999999999],\"data_offsets\":[0,24]}}", 32),
        safetensors_blob("[1,2,3]", 0), safetensors_blob("{\"w\":\"\\u00", 4), safetensors_blob(std::string(64, '{'), 0)};
    int idx = 0;
    for (const std::string& b : bad) {
        const std::string p = dir + "/bad" + std::to_string(idx++) + ".safetensors";
        write_file(p, b);
        ltx_safetensors* s2 = nullptr;
        if (ltx_safetensors_open(p.c_str(), &s2) == 0) {                     // accepted: then every tensor must be readable in full
            for (size_t i = 0; i < ltx_safetensors_count(s2); ++i) {
                const char* nm; const char* dt; int nd; const int64_t* shp; const void* data; size_t nb;
                if (ltx_safetensors_tensor(s2, i, &nm, &dt, &nd, &shp, &data, &nb) == 0) { volatile unsigned acc = 0; for (size_t j = 0; j < nb; ++j) acc += ((const unsigned char*)data)[j]; }
            }
            ltx_safetensors_close(s2);
        }
    }
    CHECK(ltx_safetensors_open((dir + "/does_not_exist").c_str(), &st) != 0);
    // ---- file resolution (loader.rs:341-397): index file with shards, plain directory, tiny output buffer
    mkdir((dir + "/ckpt").c_str(), 0755);
    write_file(dir + "/ckpt/a.safetensors", safetensors_blob(good_hdr, 32));
    write_file(dir + "/ckpt/b.safetensors", safetensors_blob(good_hdr, 32));
    size_t nfiles = 0;
    CHECK(ltx_weights_resolve((dir + "/ckpt").c_str(), buf, sizeof buf, &nfiles) == 0 && nfiles == 2);
    CHECK(ltx_weights_resolve((dir + "/ckpt").c_str(), tiny, sizeof tiny, &nfiles) != 0);
    write_file(dir + "/ckpt/model.safetensors.index.json", "{\"weight_map\":{\"w\":\"a.safetensors\",\"b\":\"b.safetensors\",\"c\":\"a.safetensors\"}}");
    CHECK(ltx_weights_resolve((dir + "/ckpt").c_str(), buf, sizeof buf, &nfiles) == 0 && nfiles == 2);
    write_file(dir + "/ckpt/model.safetensors.index.json", "{\"weight_map\":{\"w\":\"missing.safetensors\"}}");
    CHECK(ltx_weights_resolve((dir + "/ckpt").c_str(), buf, sizeof buf, &nfiles) != 0);
    write_file(dir + "/ckpt/model.safetensors.index.json", "{\"weight_map\":{\"w\":");
    (void)ltx_weights_resolve((dir + "/ckpt").c_str(), buf, sizeof buf, &nfiles);
    CHECK(ltx_weights_detect_format((dir + "/good.safetensors").c_str()) == 1 && ltx_weights_detect_format((dir + "/ckpt").c_str()) == 0);
    // ---- GGUF container (quantized_t5_encoder.rs:575-603 reads it through candle): a good file, then every kind of damage
    {
        auto u32 = [](uint32_t v) { return std::string((const char*)&v, 4); };
        auto u64 = [](uint64_t v) { return std::string((const char*)&v, 8); };
        auto str = [&](const std::string& x) { return u64(x.size()) + x; };
        std::string kv = str("general.alignment") + u32(4) + u32(32) + str("tokens") + u32(9) + u32(8) + u64(2) + str("a") + str("bc") + str("f") + u32(6) + u32(0x3f800000);
        std::string infos = str("w") + u32(2) + u64(32) + u64(3) + u32(8) + u64(0) + str("n") + u32(1) + u64(4) + u32(0) + u64(128);
        std::string headv = u32(0x46554747) + u32(3) + u64(2) + u64(3) + kv + infos;
        headv += std::string((32 - headv.size() % 32) % 32, '\0');
        const std::string data = std::string(3 * 34, '\x02') + std::string(128 - 3 * 34, '\0') + std::string(16, '\x03');
        const std::string good = headv + data;
        write_file(dir + "/good.gguf", good);
        ltx_gguf* g = nullptr;
        CHECK(ltx_gguf_open((dir + "/good.gguf").c_str(), &g) == 0 && ltx_gguf_count(g) == 2 && ltx_gguf_find(g, "n") == 1 && ltx_gguf_find(g, "zz") == -1);
        for (size_t i = 0; i < 2; ++i) {
            const char* nm; int ty, nd; const int64_t* shp; const void* dat; size_t nb;
            CHECK(ltx_gguf_tensor(g, i, &nm, &ty, &nd, &shp, &dat, &nb) == 0);
            unsigned acc = 0; for (size_t j = 0; j < nb; ++j) acc += ((const unsigned char*)dat)[j];
            CHECK(acc == (i == 0 ? 2u * 102u : 3u * 16u) && shp[0] == (i == 0 ? 3 : 4));
        }
        { const char* nm; int ty, nd; const int64_t* shp; const void* dat; size_t nb; CHECK(ltx_gguf_tensor(g, 9, &nm, &ty, &nd, &shp, &dat, &nb) != 0); }
        ltx_gguf_close(g);
        int n_ok = 0;
        for (size_t cut = 0; cut < good.size(); cut += 3) {              // every truncation must be refused or fully readable
            write_file(dir + "/cut.gguf", good.substr(0, cut));
            ltx_gguf* h = nullptr;
            if (ltx_gguf_open((dir + "/cut.gguf").c_str(), &h) == 0) {
                ++n_ok;
                for (size_t i = 0; i < ltx_gguf_count(h); ++i) {
                    const char* nm; int ty, nd; const int64_t* shp; const void* dat; size_t nb;
                    if (ltx_gguf_tensor(h, i, &nm, &ty, &nd, &shp, &dat, &nb) == 0) { volatile unsigned acc = 0; for (size_t j = 0; j < nb; ++j) acc += ((const unsigned char*)dat)[j]; }
                }
                ltx_gguf_close(h);
            }
        }
        CHECK(n_ok == 0);                                                 // the last tensor ends at the last byte: no prefix is a valid file
        for (size_t pos = 8; pos < headv.size(); pos += 1) {              // every single-byte corruption of the header
            std::string b = good; b[pos] = (char)(b[pos] ^ 0xA5);
            write_file(dir + "/flip.gguf", b);
            ltx_gguf* h = nullptr;
            if (ltx_gguf_open((dir + "/flip.gguf").c_str(), &h) == 0) {
                for (size_t i = 0; i < ltx_gguf_count(h); ++i) {
                    const char* nm; int ty, nd; const int64_t* shp; const void* dat; size_t nb;
                    if (ltx_gguf_tensor(h, i, &nm, &ty, &nd, &shp, &dat, &nb) == 0) { volatile unsigned acc = 0; for (size_t j = 0; j < nb; ++j) acc += ((const unsigned char*)dat)[j]; }
                }
                ltx_gguf_close(h);
            }
        }
        int be, bb; CHECK(ltx_gguf_type_info(13, &be, &bb) == 0 && be == 256 && bb == 176 && ltx_gguf_type_info(11, &be, &bb) != 0);
    }
    // ---- presets (configs.rs:50-283)
    CHECK(ltx_preset_count() == 6);
    for (int i = 0; i < ltx_preset_count(); ++i) {
        ltx_preset p; CHECK(ltx_preset_get(ltx_preset_name(i), &p) == 0 && std::strcmp(p.version, ltx_preset_name(i)) == 0);
        ltx_pipeline_params pp; CHECK(ltx_pipeline_params_from_preset(&p, &pp) == 0);
    }
    CHECK(ltx_preset_name(-1) == nullptr && ltx_preset_name(99) == nullptr);
    { ltx_preset p; CHECK(ltx_preset_get(huge.c_str(), &p) == 0 && std::strcmp(p.version, "0.9.5") == 0); }
    // ---- vae/config.json reader (vae.rs:30-66 serde names and aliases): a full file, every truncation of it, every single-byte
    // corruption of it - the parser may refuse, it must not read out of bounds or leak
    {
        const std::string good = "{\"_class_name\": \"AutoencoderKLLTXVideo\", \"latent_channels\": 128, \"decoder_block_out_channels\": [256, 512, 1024], "
                                 "\"decoder_spatio_temporal_scaling\": [true, true, true], \"decoder_layers_per_block\": [5, 5, 5, 5], \"patch_size\": 4, \"patch_size_t\": 1, "
                                 "\"resnet_norm_eps\": 1e-06, \"scaling_factor\": 1.0, \"decoder_inject_noise\": [false, false, false, false], "
                                 "\"upsample_residual\": [true, false, true], \"upsample_factor\": [2, 2, 2], \"timestep_conditioning\": true, \"decoder_causal\": false, "
                                 "\"nested\": {\"a\": [1, {\"b\": null}], \"s\": \"x\\\"y\\u00e9\"}}";
        const std::string cj = dir + "/config.json";
        auto put = [&](const std::string& text) { FILE* f = std::fopen(cj.c_str(), "wb"); if (!f) return false; std::fwrite(text.data(), 1, text.size(), f); std::fclose(f); return true; };
        CHECK(put(good));
        ltx_vae_config c; ltx_vae_config_default(&c);
        CHECK(ltx_vae_config_from_json(cj.c_str(), &c) == 0);
        CHECK(c.n_blocks == 3 && c.decoder_upsample_residual[1] == 0 && c.decoder_upsample_residual[2] == 1 && c.decoder_block_out_channels[2] == 1024);
        for (size_t n = 0; n < good.size(); ++n) { put(good.substr(0, n)); ltx_vae_config d; ltx_vae_config_default(&d); (void)ltx_vae_config_from_json(cj.c_str(), &d); }
        for (size_t n = 0; n < good.size(); ++n) {
            std::string bad = good; bad[n] = (char)(bad[n] ^ 0x5a);
            put(bad); ltx_vae_config d; ltx_vae_config_default(&d); (void)ltx_vae_config_from_json(cj.c_str(), &d);
        }
        CHECK(ltx_vae_config_from_json((dir + "/absent.json").c_str(), &c) != 0 && ltx_vae_config_from_json(nullptr, &c) != 0);
    }
    // ---- frame files: PNG + GIF (main.rs:653-707) on odd sizes, one pixel, and a refused empty frame
    for (int w : {1, 7, 64}) for (int h : {1, 5, 48}) {
        std::vector<uint8_t> rgb((size_t)3 * w * h * 3);
        for (size_t i = 0; i < rgb.size(); ++i) rgb[i] = (uint8_t)((i * 37 + (i >> 3)) & 255);
        CHECK(ltx_write_png((dir + "/f.png").c_str(), rgb.data(), w, h) == 0);
        CHECK(ltx_write_gif((dir + "/f.gif").c_str(), rgb.data(), 3, w, h, 4, 30) == 0);
        CHECK(ltx_write_gif((dir + "/f1.gif").c_str(), rgb.data(), 1, w, h, 4, 1) == 0);
    }
    CHECK(ltx_write_gif((dir + "/f.gif").c_str(), nullptr, 1, 4, 4, 4, 30) != 0);
    CHECK(ltx_write_png((dir + "/no/such/dir/f.png").c_str(), (const uint8_t*)"abc", 1, 1) != 0);
    std::printf("host sanitizer driver: clean\n");
    return 0;
}
