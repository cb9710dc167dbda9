"""CPU suite: the VAE config fields of the boundary (include/ltxhip.h `ltx_vae_config`, vae.rs:32-103) - what the engine
does not implement is refused by name instead of being decoded wrongly, and `vae/config.json` is read
with the reference's serde names and aliases (vae.rs:30-66; examples/ltx-video/main.rs:525-534).  No GPU: the refusals
happen before the device is touched, the JSON reader is host code."""
import ctypes
import json

import pytest

import ltxhip


def _default():
    c = ltxhip.VaeConfigC()
    ltxhip.lib.ltx_vae_config_default(ctypes.byref(c))
    return c


def test_defaults_match_vae_rs_68_103():
    c = _default()
    assert list(c.decoder_inject_noise) == [0, 0, 0, 0, 0]                      # vae.rs:87 (4 entries used)
    assert list(c.decoder_upsample_residual)[:3] == [1, 1, 1]                   # vae.rs:88
    assert list(c.decoder_spatiotemporal_scaling)[:3] == [1, 1, 1]              # vae.rs:78
    assert abs(c.resnet_eps - 1e-6) < 1e-12                                     # vae.rs:83


def _create(c):
    w = ltxhip._Weight()                                                        # never read: the refusal comes first
    h = ctypes.c_void_p()
    return ltxhip.lib.ltx_vae_create(ctypes.byref(c), ctypes.byref(w), ctypes.c_size_t(0), 0, 0, ctypes.byref(h))


def test_noise_injection_and_spatial_only_blocks_are_accepted():
    """Round 5: both decoder variants are implemented (vae.rs:676-690 / 741-753 and :1212-1236), so creation goes on to the weights -
    here, without a GPU, that is the device error (on a GPU box: the first missing weight), never the config field."""
    for field, idx in (("decoder_inject_noise", 2), ("decoder_spatiotemporal_scaling", 0)):
        c = _default(); getattr(c, field)[idx] = 1 if field == "decoder_inject_noise" else 0
        rc = _create(c)
        err = ltxhip.lib.ltx_last_error()
        assert rc != 0 and rc != 4 and field.encode() not in err, (rc, err)


def test_config_json_with_serde_names_and_aliases(tmp_path):
    # the 0.9.5 diffusers vae/config.json spelling (aliases of vae.rs:38-61) plus fields the decoder ignores
    cfg = {"_class_name": "AutoencoderKLLTXVideo", "in_channels": 3, "out_channels": 3, "latent_channels": 128,
           "block_out_channels": [128, 256, 512, 1024, 2048], "decoder_block_out_channels": [256, 512, 1024],
           "spatio_temporal_scaling": [True, True, True, True], "decoder_spatio_temporal_scaling": [True, True, False],
           "layers_per_block": [4, 6, 6, 2, 2], "decoder_layers_per_block": [5, 5, 5, 5], "patch_size": 4, "patch_size_t": 1,
           "resnet_norm_eps": 1e-5, "scaling_factor": 0.5, "spatial_compression_ratio": 32, "temporal_compression_ratio": 8,
           "decoder_inject_noise": [False, True, False, False], "upsample_residual": [True, False, True], "upsample_factor": [2, 2, 2],
           "timestep_conditioning": False, "encoder_causal": True, "decoder_causal": True, "downsample_type": ["spatial"]}
    p = tmp_path / "config.json"; p.write_text(json.dumps(cfg))
    c = _default()
    assert ltxhip.lib.ltx_vae_config_from_json(str(p).encode(), ctypes.byref(c)) == 0, ltxhip.lib.ltx_last_error()
    assert c.n_blocks == 3 and list(c.decoder_block_out_channels)[:3] == [256, 512, 1024]
    assert list(c.decoder_spatiotemporal_scaling)[:3] == [1, 1, 0]
    assert list(c.decoder_inject_noise)[:4] == [0, 1, 0, 0]
    assert list(c.decoder_upsample_residual)[:3] == [1, 0, 1]
    assert list(c.decoder_upsample_factor)[:3] == [2, 2, 2] and list(c.decoder_layers_per_block)[:4] == [5, 5, 5, 5]
    assert abs(c.resnet_eps - 1e-5) < 1e-10 and abs(c.scaling_factor - 0.5) < 1e-7
    assert c.timestep_conditioning == 0 and c.decoder_causal == 1
    # the canonical names win over nothing-given; a partial file keeps what *cfg already held (serde(default))
    p.write_text(json.dumps({"decoder_upsample_residual": [False, False, False], "resnet_eps": 1e-4}))
    c = _default()
    assert ltxhip.lib.ltx_vae_config_from_json(str(p).encode(), ctypes.byref(c)) == 0
    assert list(c.decoder_upsample_residual)[:3] == [0, 0, 0] and c.n_blocks == 3 and c.latent_channels == 128


@pytest.mark.parametrize("bad", ['{"decoder_block_out_channels": [1, 2, 3, 4, 5]}', '{"upsample_residual": [1, 0, 1]}',
                                 '{"patch_size": "4"}', '{"decoder_inject_noise": true}', '[1, 2]', '{"latent_channels": 12.5}', '{'])
def test_config_json_rejects_wrong_types(tmp_path, bad):
    p = tmp_path / "config.json"; p.write_text(bad)
    c = _default()
    assert ltxhip.lib.ltx_vae_config_from_json(str(p).encode(), ctypes.byref(c)) == 1      # LTX_ERR_ARG
    assert ltxhip.lib.ltx_last_error()


def test_config_json_missing_file(tmp_path):
    c = _default()
    assert ltxhip.lib.ltx_vae_config_from_json(str(tmp_path / "nope.json").encode(), ctypes.byref(c)) == 1
