"""CPU suite: the version presets behind include/ltxhip_presets.h replay the reference's own configs.rs unit tests
(configs.rs:285-324) and the rest of the preset table (:167-282); the package-owned weight schema (ltxhip/schema.py)
names exactly the tensors the oracle's restatement of the constructors names."""
import ctypes

import pytest

import ltx_oracle as O


@pytest.fixture(scope="module")
def hip():
    import ltxhip
    return ltxhip


def test_reference_config_unit_tests(hip):
    c = hip.get_config_by_version("0.9.5")                          # test_v0_9_5_2b_config
    assert c.transformer.num_layers == 28 and c.guidance_scale == 3.0 and c.num_inference_steps == 40 and c.skip_block_list == [19]
    c = hip.get_config_by_version("0.9.8-2b-distilled")             # test_v0_9_8_distilled_2b_config
    assert c.transformer.num_layers == 28 and c.guidance_scale == 1.0 and c.stg_scale == 0.0
    c = hip.get_config_by_version("0.9.8-13b-distilled")            # test_v0_9_8_13b_distilled_config
    assert (c.transformer.num_layers, c.transformer.attention_head_dim, c.transformer.cross_attention_dim, c.skip_block_list) == (48, 128, 4096, [42])
    c = hip.get_config_by_version("0.9.5")                          # test_vae_config_5_blocks
    assert c.vae_encoder_block_out_channels == [128, 256, 512, 1024, 2048] and c.vae_encoder_layers_per_block == [4, 6, 6, 2, 2]


def test_preset_table_aliases_and_fallback(hip):
    assert hip.preset_names() == ["0.9.5", "0.9.6-dev", "0.9.6-distilled", "0.9.8-2b-distilled", "0.9.8-13b-dev", "0.9.8-13b-distilled"]
    for alias, canon in [("0.9.5-2b", "0.9.5"), ("0.9.6-2b-dev", "0.9.6-dev"), ("0.9.6-2b-distilled", "0.9.6-distilled"), ("0.9.8-distilled", "0.9.8-2b-distilled"),
                         ("0.9.8-13b", "0.9.8-13b-distilled"), ("no-such-version", "0.9.5"), ("", "0.9.5")]:
        assert hip.get_config_by_version(alias).version == canon, alias
    d = hip.get_config_by_version("0.9.6-distilled")                # configs.rs:205-222
    assert (d.guidance_scale, d.num_inference_steps, d.stg_scale, d.rescaling_scale, d.stochastic_sampling, d.skip_block_list, d.timesteps) == (1.0, 8, 0.0, 1.0, True, [], None)
    d = hip.get_config_by_version("0.9.8-13b-dev")                  # configs.rs:243-262
    assert (d.guidance_scale, d.num_inference_steps, d.stg_scale, d.rescaling_scale, d.skip_block_list) == (8.0, 30, 4.0, 0.5, [11, 25, 35, 39])
    d = hip.get_config_by_version("0.9.8-2b-distilled")             # configs.rs:224-241
    assert [round(x, 4) for x in d.timesteps] == [1.0, 0.9937, 0.9875, 0.9812, 0.975, 0.9094, 0.725]
    assert abs(d.decode_timestep - 0.05) < 1e-7 and abs(d.decode_noise_scale - 0.025) < 1e-7
    s = d.scheduler                                                  # common_scheduler_config, configs.rs:101-121
    assert (s["num_train_timesteps"], s["shift"], s["use_dynamic_shifting"], s["base_image_seq_len"], s["max_image_seq_len"], s["time_shift_type"]) == (1000, 1.0, False, 1024, 4096, "exponential")
    assert abs(s["base_shift"] - 0.95) < 1e-6 and abs(s["max_shift"] - 2.05) < 1e-6 and abs(s["shift_terminal"] - 0.1) < 1e-7
    for name in hip.preset_names():                                  # every preset: 2B or 13B geometry, decoder defaults
        c = hip.get_config_by_version(name)
        assert c.transformer.num_attention_heads == 32 and c.transformer.caption_channels == 4096 and c.transformer.in_channels == 128
        assert c.transformer.cross_attention_dim == 32 * c.transformer.attention_head_dim
        assert c.vae.decoder_block_out_channels == (256, 512, 1024) and c.vae.latent_channels == 128 and c.vae.timestep_conditioning


def test_pipeline_params_from_preset_follow_main_rs(hip):
    """main.rs:585-646: sigmas from the preset's timesteps, decode_timestep.unwrap_or(0.0), noise scale falling back to
    the decode timestep, pipeline.guidance_rescale = rescaling_scale."""
    for name in hip.preset_names():
        pre = hip.PresetC(); assert hip.lib.ltx_preset_get(name.encode(), ctypes.byref(pre)) == 0
        p = hip.PipelineParamsC(); assert hip.lib.ltx_pipeline_params_from_preset(ctypes.byref(pre), ctypes.byref(p)) == 0
        c = hip.get_config_by_version(name)
        call = c.pipeline_call(512, 768, 97)
        assert p.num_inference_steps == call.num_inference_steps == (7 if c.timesteps else c.num_inference_steps)
        assert bool(p.sigmas) == (c.timesteps is not None)
        if c.timesteps:
            assert [round(p.sigmas[i], 4) for i in range(7)] == [round(x, 4) for x in c.timesteps]
        assert (p.guidance_scale, p.stg_scale) == (c.guidance_scale, c.stg_scale) and abs(p.guidance_rescale - c.rescaling_scale) < 1e-7
        assert p.n_skip_blocks == len(c.skip_block_list) and [p.skip_block_list[i] for i in range(p.n_skip_blocks)] == c.skip_block_list
        want_t = c.decode_timestep if c.decode_timestep is not None else 0.0
        want_s = c.decode_noise_scale if c.decode_noise_scale is not None else want_t
        assert abs(p.decode_timestep - want_t) < 1e-7 and abs(p.decode_noise_scale - want_s) < 1e-7
        assert abs(call.decode_timestep - want_t) < 1e-7 and abs(call.decode_noise_scale - want_s) < 1e-7
        assert bool(p.stochastic_sampling) == c.stochastic_sampling and abs(p.shift_terminal - 0.1) < 1e-7 and p.use_shift_terminal == 1


def test_package_schema_names_what_the_constructors_name(hip):
    from ltxhip import schema
    for kw in (dict(), dict(num_layers=48, attention_head_dim=128, cross_attention_dim=4096),
               dict(in_channels=32, out_channels=32, num_attention_heads=4, attention_head_dim=16, cross_attention_dim=64, num_layers=2, caption_channels=32)):
        assert schema.dit_weight_shapes(hip.LtxVideoTransformer3DModelConfig(**kw)) == O.dit_weight_shapes(O.DitConfig(**kw))
    for kw in (dict(), dict(latent_channels=8, decoder_block_out_channels=(32, 64, 128), decoder_layers_per_block=(1, 1, 1, 2)),
               dict(timestep_conditioning=False)):
        assert schema.vae_decoder_weight_shapes(hip.AutoencoderKLLtxVideoConfig(**kw)) == O.vae_decoder_weight_shapes(O.VaeConfig(**kw))
    # 2B parameter count: 16 D^2 L + ... (1.88 B in the blocks)
    n = sum(int(__import__("math").prod(s)) for s in schema.dit_weight_shapes(hip.LtxVideoTransformer3DModelConfig()).values())
    assert 1.9e9 < n < 1.95e9
