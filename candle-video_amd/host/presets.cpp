// Version presets as data (include/ltxhip_presets.h; reference: src/models/ltx_video/configs.rs:50-283).
#include <cstring>
#include <string>

#include "../../include/ltxhip_presets.h"
#include "../csrc/errors.h"

namespace {
struct Alias { const char* name; int preset; };
// get_config_by_version (configs.rs:50-70)
const Alias kAliases[] = {
    {"0.9.5", 0}, {"0.9.5-2b", 0},
    {"0.9.6-dev", 1}, {"0.9.6-2b-dev", 1},
    {"0.9.6-distilled", 2}, {"0.9.6-2b-distilled", 2},
    {"0.9.8-2b-distilled", 3}, {"0.9.8-distilled", 3},
    {"0.9.8-13b-dev", 4},
    {"0.9.8-13b-distilled", 5}, {"0.9.8-13b", 5},
};
const char* kNames[] = {"0.9.5", "0.9.6-dev", "0.9.6-distilled", "0.9.8-2b-distilled", "0.9.8-13b-dev", "0.9.8-13b-distilled"};
const float kDistilledSigmas[7] = {1.0000f, 0.9937f, 0.9875f, 0.9812f, 0.9750f, 0.9094f, 0.7250f};   // configs.rs:232, 274

void fill_common(ltx_preset* p, bool is_13b) {
    std::memset(p, 0, sizeof(*p));
    ltx_dit_config_default(&p->transformer);                 // transformer_2b_config / transformer_13b_config (:124-165)
    p->transformer.num_layers = is_13b ? 48 : 28;
    p->transformer.num_attention_heads = 32;
    p->transformer.attention_head_dim = is_13b ? 128 : 64;
    p->transformer.cross_attention_dim = is_13b ? 4096 : 2048;
    p->transformer.caption_channels = 4096;
    ltx_vae_config_default(&p->vae);                         // common_vae_config (:84-92): decoder side keeps the defaults
    p->vae.latent_channels = 128; p->vae.patch_size = 4; p->vae.timestep_conditioning = 1;
    const int enc_ch[5] = {128, 256, 512, 1024, 2048}, enc_layers[5] = {4, 6, 6, 2, 2};
    for (int i = 0; i < 5; ++i) { p->vae_encoder_block_out_channels[i] = enc_ch[i]; p->vae_encoder_layers_per_block[i] = enc_layers[i]; }
    p->num_train_timesteps = 1000; p->shift = 1.0f; p->use_dynamic_shifting = 0;       // common_scheduler_config (:101-121)
    p->base_shift = 0.95f; p->max_shift = 2.05f; p->base_image_seq_len = 1024; p->max_image_seq_len = 4096;
    p->shift_terminal = 0.1f; p->has_shift_terminal = 1; p->time_shift_exponential = 1;
}
void inference(ltx_preset* p, float g, int steps, float stg, float rescale, int stochastic, std::initializer_list<int> skip, bool distilled_sigmas) {
    p->guidance_scale = g; p->num_inference_steps = steps; p->stg_scale = stg; p->rescaling_scale = rescale; p->stochastic_sampling = stochastic;
    for (int b : skip) p->skip_block_list[p->n_skip_blocks++] = b;
    if (distilled_sigmas) {
        for (int i = 0; i < 7; ++i) p->timesteps[i] = kDistilledSigmas[i];
        p->n_timesteps = 7;
        p->decode_timestep = 0.05f; p->has_decode_timestep = 1; p->decode_noise_scale = 0.025f; p->has_decode_noise_scale = 1;
    }
}
void build(int idx, ltx_preset* p) {
    fill_common(p, idx >= 4);
    p->version = kNames[idx];
    switch (idx) {
        case 0: inference(p, 3.0f, 40, 1.0f, 0.7f, 0, {19}, false); break;               // v0_9_5_2b (:167-184)
        case 1: inference(p, 3.0f, 40, 1.0f, 0.7f, 0, {19}, false); break;               // v0_9_6_dev_2b (:186-203)
        case 2: inference(p, 1.0f, 8, 0.0f, 1.0f, 1, {}, false); break;                  // v0_9_6_distilled_2b (:205-222)
        case 3: inference(p, 1.0f, 7, 0.0f, 1.0f, 0, {}, true); break;                   // v0_9_8_distilled_2b (:224-241)
        case 4: inference(p, 8.0f, 30, 4.0f, 0.5f, 0, {11, 25, 35, 39}, false); break;   // v0_9_8_dev_13b (:243-262)
        case 5: inference(p, 1.0f, 7, 0.0f, 1.0f, 0, {42}, true); break;                 // v0_9_8_distilled_13b (:264-282)
    }
}
}  // namespace

extern "C" int ltx_preset_count(void) { return 6; }
extern "C" const char* ltx_preset_name(int index) { return index >= 0 && index < 6 ? kNames[index] : nullptr; }
extern "C" int ltx_preset_get(const char* version, ltx_preset* out) {
    if (!version || !out) LTX_FAIL(LTX_ERR_ARG, "ltx_preset_get: null argument");
    int idx = 0;                                                   // "Default to 0.9.5" (configs.rs:67-68)
    for (const Alias& a : kAliases) if (!std::strcmp(a.name, version)) { idx = a.preset; break; }
    build(idx, out);
    return LTX_OK;
}
extern "C" int ltx_pipeline_params_from_preset(const ltx_preset* preset, ltx_pipeline_params* p) {
    if (!preset || !p) LTX_FAIL(LTX_ERR_ARG, "ltx_pipeline_params_from_preset: null argument");
    ltx_pipeline_params_default(p);
    // pipeline.guidance_rescale = rescaling_scale (main.rs:613); it only acts on the CFG mix (t2v_pipeline.rs:945-955)
    p->guidance_scale = preset->guidance_scale; p->guidance_rescale = preset->rescaling_scale; p->stg_scale = preset->stg_scale;
    p->num_inference_steps = preset->n_timesteps ? preset->n_timesteps : preset->num_inference_steps;
    p->sigmas = preset->n_timesteps ? preset->timesteps : nullptr;
    p->skip_block_list = preset->n_skip_blocks ? preset->skip_block_list : nullptr; p->n_skip_blocks = preset->n_skip_blocks;
    // main.rs:616-621: decode_timestep.unwrap_or(vec![0.0]); a missing noise scale falls back to the decode timestep (t2v_pipeline.rs:1031-1041)
    p->decode_timestep = preset->has_decode_timestep ? preset->decode_timestep : 0.0f;
    p->decode_noise_scale = preset->has_decode_noise_scale ? preset->decode_noise_scale : p->decode_timestep;
    p->shift_terminal = preset->shift_terminal; p->use_shift_terminal = preset->has_shift_terminal;
    p->stochastic_sampling = preset->stochastic_sampling;
    return LTX_OK;
}
