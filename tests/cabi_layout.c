/* ABI layout guard: compiled as C99 against the headers under include/ by tests/test_cabi_layout_cpu.py (which
 * also proves that the headers ARE plain C), prints sizeof / alignment / offsetof of every struct that crosses the
 * boundary as JSON.  The test compares the output with the ctypes mirrors (candle-video_amd/ltxhip/__init__.py) and
 * with the LAYOUT_* constants asserted at compile time in rust/ltxhip-sys/src/lib.rs. */
#include <stddef.h>
#include <stdio.h>
#include "ltxhip.h"
#include "ltxhip_ops.h"
#include "ltxhip_weights.h"
#include "ltxhip_t5.h"
#include "ltxhip_frames.h"

#define ALIGN_OF(T) offsetof(struct { char c; T x; }, x)
#define BEGIN(T) printf("%s\"%s\": {\"size\": %zu, \"align\": %zu, \"fields\": {", first ? "" : ", ", #T, sizeof(T), ALIGN_OF(T)); first = 0; ff = 1
#define F(T, f) printf("%s\"%s\": %zu", ff ? "" : ", ", #f, offsetof(T, f)); ff = 0
#define END() printf("}}")

int main(void) {
    int first = 1, ff = 1;
    printf("{");
    BEGIN(ltx_weight); F(ltx_weight, name); F(ltx_weight, data); F(ltx_weight, dtype); F(ltx_weight, ndim); F(ltx_weight, shape); F(ltx_weight, on_device); END();
    BEGIN(ltx_dit_config); F(ltx_dit_config, in_channels); F(ltx_dit_config, out_channels); F(ltx_dit_config, patch_size); F(ltx_dit_config, patch_size_t);
        F(ltx_dit_config, num_attention_heads); F(ltx_dit_config, attention_head_dim); F(ltx_dit_config, cross_attention_dim); F(ltx_dit_config, num_layers);
        F(ltx_dit_config, norm_eps); F(ltx_dit_config, caption_channels); END();
    BEGIN(ltx_vae_config); F(ltx_vae_config, latent_channels); F(ltx_vae_config, out_channels); F(ltx_vae_config, n_blocks); F(ltx_vae_config, decoder_block_out_channels);
        F(ltx_vae_config, decoder_layers_per_block); F(ltx_vae_config, decoder_upsample_factor); F(ltx_vae_config, patch_size); F(ltx_vae_config, patch_size_t);
        F(ltx_vae_config, timestep_conditioning); F(ltx_vae_config, decoder_causal); F(ltx_vae_config, scaling_factor); F(ltx_vae_config, spatial_compression_ratio);
        F(ltx_vae_config, temporal_compression_ratio); F(ltx_vae_config, decoder_inject_noise); F(ltx_vae_config, decoder_upsample_residual);
        F(ltx_vae_config, decoder_spatiotemporal_scaling); F(ltx_vae_config, resnet_eps); END();
    BEGIN(ltx_tiling); F(ltx_tiling, use_tiling); F(ltx_tiling, use_framewise_decoding); F(ltx_tiling, tile_sample_min_height); F(ltx_tiling, tile_sample_min_width);
        F(ltx_tiling, tile_sample_min_num_frames); F(ltx_tiling, tile_sample_stride_height); F(ltx_tiling, tile_sample_stride_width); F(ltx_tiling, tile_sample_stride_num_frames); END();
    BEGIN(ltx_pipeline_params); F(ltx_pipeline_params, height); F(ltx_pipeline_params, width); F(ltx_pipeline_params, num_frames); F(ltx_pipeline_params, frame_rate);
        F(ltx_pipeline_params, num_inference_steps); F(ltx_pipeline_params, sigmas); F(ltx_pipeline_params, guidance_scale); F(ltx_pipeline_params, guidance_rescale);
        F(ltx_pipeline_params, stg_scale); F(ltx_pipeline_params, skip_block_list); F(ltx_pipeline_params, n_skip_blocks); F(ltx_pipeline_params, decode_timestep);
        F(ltx_pipeline_params, decode_noise_scale); F(ltx_pipeline_params, output_latent); F(ltx_pipeline_params, postprocess); F(ltx_pipeline_params, tiling);
        F(ltx_pipeline_params, shift_terminal); F(ltx_pipeline_params, use_shift_terminal); F(ltx_pipeline_params, stochastic_sampling); F(ltx_pipeline_params, step_noise);
        F(ltx_pipeline_params, interrupt); F(ltx_pipeline_params, on_step); F(ltx_pipeline_params, on_step_user); END();
    BEGIN(ltx_t5_config); F(ltx_t5_config, vocab_size); F(ltx_t5_config, d_model); F(ltx_t5_config, d_kv); F(ltx_t5_config, d_ff); F(ltx_t5_config, num_layers);
        F(ltx_t5_config, num_heads); F(ltx_t5_config, relative_attention_num_buckets); F(ltx_t5_config, relative_attention_max_distance); F(ltx_t5_config, layer_norm_epsilon); END();
    printf("}\n");
    return 0;
}
