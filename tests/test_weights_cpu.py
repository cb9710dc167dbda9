"""CPU suite, part 5: weight ingestion (include/ltxhip_weights.h) through the C ABI.

The expected values of the first block are the reference's OWN unit tests
(src/models/ltx_video/weight_format.rs:171-268 and loader.rs:575-649): these pin the restatement."""
import json
import os
import struct

import numpy as np
import pytest
import torch
from safetensors.torch import save_file

import ltxhip
from ltxhip import weights as W


# ---- weight_format.rs tests (:171-268) -------------------------------------------------------------------------
def test_remap_transformer_key():
    assert W.remap_key("transformer.patchify_proj.weight") == "transformer.proj_in.weight"
    assert W.remap_key("transformer.adaln_single.linear.weight") == "transformer.time_embed.linear.weight"


def test_remap_encoder_blocks_095():
    assert W.remap_key("encoder.down_blocks.0.res_blocks.0.conv1.weight") == "encoder.down_blocks.0.resnets.0.conv1.weight"
    assert W.remap_key("encoder.down_blocks.1.conv.weight") == "encoder.down_blocks.0.downsamplers.0.conv.weight"
    assert W.remap_key("encoder.down_blocks.2.res_blocks.0.conv1.weight") == "encoder.down_blocks.1.resnets.0.conv1.weight"
    assert W.remap_key("encoder.down_blocks.6.res_blocks.0.weight") == "encoder.down_blocks.3.resnets.0.weight"
    assert W.remap_key("encoder.down_blocks.8.res_blocks.0.weight") == "encoder.mid_block.resnets.0.weight"


def test_remap_decoder_blocks_095():
    assert W.remap_key("decoder.up_blocks.0.res_blocks.0.weight") == "decoder.mid_block.resnets.0.weight"
    assert W.remap_key("decoder.up_blocks.1.conv.weight") == "decoder.up_blocks.0.upsamplers.0.conv.weight"
    assert W.remap_key("decoder.up_blocks.2.res_blocks.0.weight") == "decoder.up_blocks.0.resnets.0.weight"
    assert W.remap_key("decoder.up_blocks.8.res_blocks.0.weight") == "decoder.up_blocks.3.resnets.0.weight"


def test_remap_time_embedder_and_latents_stats():
    assert W.remap_key("decoder.last_time_embedder.weight") == "decoder.time_embedder.weight"
    assert W.remap_key("per_channel_statistics.mean-of-means") == "latents_mean"
    assert W.remap_key("per_channel_statistics.std-of-means") == "latents_std"


def test_remap_rules_beyond_the_reference_tests():
    # the remaining rows of remap_key (:60-80) and the out-of-table fallbacks (:110, :138)
    assert W.remap_key("model.diffusion_model.transformer_blocks.3.attn1.q_norm.weight") == "model.diffusion_model.transformer_blocks.3.attn1.norm_q.weight"
    assert W.remap_key("model.diffusion_model.transformer_blocks.3.attn2.k_norm.weight") == "model.diffusion_model.transformer_blocks.3.attn2.norm_k.weight"
    assert W.remap_key("vae.decoder.last_scale_shift_table") == "vae.decoder.scale_shift_table"
    assert W.remap_key("vae.decoder.up_blocks.4.res_blocks.1.norm3.norm.weight") == "vae.decoder.up_blocks.1.resnets.1.norm3.weight"
    assert W.remap_key("decoder.up_blocks.11.x") == "decoder.up_blocks.11.x"
    assert W.remap_key("encoder.down_blocks.x") == "encoder.down_blocks.x"
    assert W.is_transformer_key("model.diffusion_model.scale_shift_table") and W.is_transformer_key("transformer_blocks.0.ff.net.0.proj.weight")
    assert not W.is_transformer_key("vae.decoder.conv_in.conv.weight") and W.is_vae_key("vae.decoder.conv_in.conv.weight")
    assert W.is_vae_key("per_channel_statistics.std-of-means") and not W.is_vae_key("text_encoder.block.0.weight")


def test_detect_format(tmp_path):
    f = tmp_path / "ltx-video-2b-v0.9.5.safetensors"
    f.write_bytes(b"\0" * 16)
    assert W.detect_format(str(f)) == W.OFFICIAL
    assert W.detect_format(str(tmp_path)) == W.DIFFUSERS
    assert W.detect_format(str(tmp_path / "does-not-exist")) == W.DIFFUSERS        # weight_format.rs:26-27


# ---- loader.rs tests (:575-649) ---------------------------------------------------------------------------------
def test_name_mapping_exact_prefix_suffix_chain():
    ld = W.WeightLoader().add_mapping("model.diffusion_model", "diffusion_model")
    assert ld.map_name("model.diffusion_model") == "diffusion_model" and ld.map_name("other.name") == "other.name"
    ld = W.WeightLoader().add_prefix_mapping("model.", "")
    assert ld.map_name("model.transformer.weight") == "transformer.weight" and ld.map_name("other.weight") == "other.weight"
    assert ld.has_mapping("model.x") and not ld.has_mapping("x.model.")
    ld = W.WeightLoader().add_suffix_mapping(".gamma", ".weight")
    assert ld.map_name("layer_norm.gamma") == "layer_norm.weight"
    ld = W.WeightLoader().add_prefix_mapping("model.", "").add_suffix_mapping(".gamma", ".weight")
    assert ld.map_name("model.layer_norm.gamma") == "layer_norm.weight"


def test_validate_tensor_names():
    assert W.validate_tensor_names(["a", "b", "c"], ["a", "b"]) == ["c"]
    assert W.validate_tensor_names([], ["a"]) == [] and W.validate_tensor_names(["x", "y"], []) == ["x", "y"]


def test_safetensors_index_shard_files_and_directory_resolution(tmp_path):
    t = {"a": torch.zeros(2)}
    for n in ("shard1.safetensors", "shard2.safetensors"):
        save_file(t, str(tmp_path / n))
    (tmp_path / "model.safetensors.index.json").write_text(json.dumps(
        {"metadata": {"format": "safetensors"}, "weight_map": {"a": "shard1.safetensors", "b": "shard1.safetensors", "c": "shard2.safetensors"}}))
    files = W.resolve_weight_files(str(tmp_path))
    assert [os.path.basename(f) for f in files] == ["shard1.safetensors", "shard2.safetensors"]       # distinct shards (:160-166)
    os.remove(tmp_path / "shard2.safetensors")
    with pytest.raises(ltxhip.LtxError, match="missing shard files: shard2.safetensors"):           # LoaderError::MissingShards
        W.resolve_weight_files(str(tmp_path))
    os.remove(tmp_path / "model.safetensors.index.json")
    save_file(t, str(tmp_path / "model.safetensors"))
    assert [os.path.basename(f) for f in W.resolve_weight_files(str(tmp_path))] == ["model.safetensors"]   # :375-378
    os.remove(tmp_path / "model.safetensors")
    save_file(t, str(tmp_path / "zz.safetensors"))
    assert [os.path.basename(f) for f in W.resolve_weight_files(str(tmp_path))] == ["shard1.safetensors", "zz.safetensors"]   # sorted scan
    empty = tmp_path / "empty"; empty.mkdir()
    with pytest.raises(ltxhip.LtxError, match="no safetensors files"):
        W.resolve_weight_files(str(empty))
    assert W.resolve_weight_files(str(tmp_path / "zz.safetensors")) == [str(tmp_path / "zz.safetensors")]


# ---- safetensors reader ------------------------------------------------------------------------------------------
def test_safetensors_reader_round_trip(tmp_path):
    g = torch.Generator().manual_seed(0)
    t = {"model.diffusion_model.patchify_proj.weight": torch.randn(6, 4, generator=g),
         "vae.decoder.up_blocks.2.res_blocks.0.conv1.conv.weight": torch.randn(2, 3, 3, 3, 3, generator=g).bfloat16(),
         "vae.per_channel_statistics.mean-of-means": torch.arange(8, dtype=torch.float32),
         "scalar": torch.tensor(3.5), "halfs": torch.ones(4, dtype=torch.float16)}
    p = str(tmp_path / "u.safetensors")
    save_file(t, p, metadata={"note": "quote \" and unicode é"})
    st = W.SafetensorsFile(p)
    got = st.tensors()
    assert set(got) == set(t) and len(st) == len(t)
    for name, ten in t.items():
        dts, shape, raw = got[name]
        assert dts == {torch.float32: "F32", torch.bfloat16: "BF16", torch.float16: "F16"}[ten.dtype]
        assert shape == tuple(ten.shape)
        assert raw == ten.contiguous().view(torch.uint8).numpy().tobytes() if ten.dim() else raw == ten.reshape(1).view(torch.uint8).numpy().tobytes()
    st.close()
    # malformed files fail loudly
    bad = tmp_path / "bad.safetensors"
    bad.write_bytes(struct.pack("<Q", 1 << 40) + b"{}")
    with pytest.raises(ltxhip.LtxError, match="header length"):
        W.SafetensorsFile(str(bad))
    bad.write_bytes(struct.pack("<Q", 9) + b'{"a":[1]}')
    with pytest.raises(ltxhip.LtxError, match="malformed entry"):
        W.SafetensorsFile(str(bad))
    hdr = json.dumps({"a": {"dtype": "F32", "shape": [4], "data_offsets": [0, 16]}}).encode()
    bad.write_bytes(struct.pack("<Q", len(hdr)) + hdr + b"\0" * 8)
    with pytest.raises(ltxhip.LtxError, match="outside the file"):
        W.SafetensorsFile(str(bad))
    with pytest.raises(ltxhip.LtxError, match="cannot open"):
        W.SafetensorsFile(str(tmp_path / "nope.safetensors"))


def test_from_files_reports_missing_paths_without_a_gpu(tmp_path):
    cfg = ltxhip.LtxVideoTransformer3DModelConfig()
    with pytest.raises(ltxhip.LtxError, match="does not exist"):
        ltxhip.LtxVideoTransformer3DModel.from_files(cfg, str(tmp_path / "missing"), unified=False)
    save_file({"text_encoder.w": torch.zeros(2)}, str(tmp_path / "only_t5.safetensors"))
    with pytest.raises(ltxhip.LtxError, match="no transformer tensors"):
        ltxhip.LtxVideoTransformer3DModel.from_files(cfg, str(tmp_path / "only_t5.safetensors"), unified=True)


def test_official_names_of_the_whole_model_remap_to_what_the_constructors_read():
    """Every weight name LtxVideoTransformer3DModel::new / AutoencoderKLLtxVideo::new read (oracle shape tables), written in
    the Official layout, must come back through remap_key + the prefix stripping of main.rs:480-498."""
    import ltx_oracle as O
    from tools_cfg import to_official_names
    dit = list(O.dit_weight_shapes(O.DitConfig(num_layers=2)))
    vae = ["decoder." + k for k in O.vae_decoder_weight_shapes(O.VaeConfig())] + ["latents_mean", "latents_std"]
    off = to_official_names(dit, vae)
    assert len(off) == len(dit) + len(vae)
    for okey, (comp, want) in off.items():
        r = W.remap_key(okey)
        if comp == "dit":
            assert W.is_transformer_key(okey) and not W.is_vae_key(okey)
            assert r.startswith("model.diffusion_model.") and r[len("model.diffusion_model."):] == want, (okey, r, want)
        else:
            assert W.is_vae_key(okey)
            assert r.startswith("vae.") and r[4:] == want, (okey, r, want)
