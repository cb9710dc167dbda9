/* ltxhip_presets.h — the reference's version presets as data (src/models/ltx_video/configs.rs:50-283).
 * get_config_by_version(version) -> LTXVFullConfig { inference, transformer, vae, scheduler } for the six presets the
 * reference ships (0.9.5 and later), with its aliases, and its fall-back to 0.9.5 for unknown strings (:67-68). */
#ifndef LTXHIP_PRESETS_H
#define LTXHIP_PRESETS_H
#include "ltxhip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    const char* version;            /* canonical name, e.g. "0.9.8-2b-distilled" */
    /* LTXVInferenceConfig (configs.rs:10-37) */
    float guidance_scale;
    int num_inference_steps;
    float stg_scale, rescaling_scale;
    int stochastic_sampling;
    int skip_block_list[8]; int n_skip_blocks;
    float timesteps[16]; int n_timesteps;           /* custom sigma list (Some(vec)) or n = 0 (None) */
    float decode_timestep; int has_decode_timestep;
    float decode_noise_scale; int has_decode_noise_scale;
    /* transformer (configs.rs:124-165) and decoder-side VAE fields (vae.rs:68-103 defaults; the presets override the
     * ENCODER's block_out_channels / layers_per_block, kept below for completeness, :84-92) */
    ltx_dit_config transformer;
    ltx_vae_config vae;
    int vae_encoder_block_out_channels[5]; int vae_encoder_layers_per_block[5];
    /* FlowMatchEulerDiscreteSchedulerConfig (configs.rs:101-121) */
    int num_train_timesteps;
    float shift; int use_dynamic_shifting;
    float base_shift, max_shift; int base_image_seq_len, max_image_seq_len;
    float shift_terminal; int has_shift_terminal;
    int time_shift_exponential;     /* TimeShiftType::Exponential */
} ltx_preset;

int ltx_preset_count(void);
const char* ltx_preset_name(int index);                       /* canonical names, 0 .. count-1 */
/* get_config_by_version: aliases resolved, unknown strings -> the 0.9.5 preset (configs.rs:67-68); never fails for non-NULL */
int ltx_preset_get(const char* version, ltx_preset* out);
/* fill the call parameters a preset implies (what examples/ltx-video/main.rs:585-646 passes to LtxPipeline::call):
 * guidance / stg / rescale / steps / sigmas (pointing INTO `preset`, which must outlive `p`) / skip blocks / decode
 * timestep (0.0 when the preset has None, main.rs:616-620) + noise scale (the decode timestep when None,
 * t2v_pipeline.rs:1031-1041) / stochastic sampling. */
int ltx_pipeline_params_from_preset(const ltx_preset* preset, ltx_pipeline_params* p);

#ifdef __cplusplus
}
#endif
#endif
