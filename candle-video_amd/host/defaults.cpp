// Default configurations of the boundary structs (include/ltxhip.h) - plain data, host only.
// Reference: LtxVideoTransformer3DModelConfig defaults (ltx_transformer.rs:30-58), AutoencoderKLLtxVideoConfig defaults
// (vae.rs:68-103), tiling parameters (vae.rs:1860-1880), the 0.9.8 distilled call (configs.rs:223-240, main.rs:627).
#include <cstring>
#include "../../include/ltxhip.h"

extern "C" void ltx_dit_config_default(ltx_dit_config* c) {
    c->in_channels = 128; c->out_channels = 128; c->patch_size = 1; c->patch_size_t = 1;
    c->num_attention_heads = 32; c->attention_head_dim = 64; c->cross_attention_dim = 2048;
    c->num_layers = 28; c->norm_eps = 1e-6f; c->caption_channels = 4096;
}

extern "C" void ltx_vae_config_default(ltx_vae_config* c) {
    c->latent_channels = 128; c->out_channels = 3; c->n_blocks = 3;
    int boc[4] = {256, 512, 1024, 0}; int lpb[5] = {5, 5, 5, 5, 0}; int upf[4] = {2, 2, 2, 0};
    for (int i = 0; i < 4; ++i) { c->decoder_block_out_channels[i] = boc[i]; c->decoder_upsample_factor[i] = upf[i]; }
    for (int i = 0; i < 5; ++i) c->decoder_layers_per_block[i] = lpb[i];
    c->patch_size = 4; c->patch_size_t = 1; c->timestep_conditioning = 1; c->decoder_causal = 0;
    c->scaling_factor = 1.0f; c->spatial_compression_ratio = 32; c->temporal_compression_ratio = 8;
    for (int i = 0; i < 5; ++i) c->decoder_inject_noise[i] = 0;                                   // vae.rs:87
    for (int i = 0; i < 4; ++i) { c->decoder_upsample_residual[i] = i < 3; c->decoder_spatiotemporal_scaling[i] = i < 3; }   // vae.rs:78, 88
    c->resnet_eps = 1e-6f;                                                                         // vae.rs:83
}
extern "C" void ltx_tiling_default(ltx_tiling* t) {
    t->use_tiling = 1; t->use_framewise_decoding = 1;
    t->tile_sample_min_height = 512; t->tile_sample_min_width = 512; t->tile_sample_min_num_frames = 16;
    t->tile_sample_stride_height = 384; t->tile_sample_stride_width = 384; t->tile_sample_stride_num_frames = 8;
}

extern "C" void ltx_pipeline_params_default(ltx_pipeline_params* p) {
    std::memset(p, 0, sizeof(*p));
    p->height = 512; p->width = 768; p->num_frames = 97; p->frame_rate = 25;   // main.rs:627
    p->num_inference_steps = 7;                                                  // configs.rs:227
    p->guidance_scale = 1.0f; p->guidance_rescale = 0.0f; p->stg_scale = 0.0f;
    p->decode_timestep = 0.05f; p->decode_noise_scale = 0.025f;                  // configs.rs:233-234
    p->shift_terminal = 0.1f; p->use_shift_terminal = 1;                         // configs.rs:101-121
    p->postprocess = 1;
}

