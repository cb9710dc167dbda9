/* ltxhip_t5.h — T5 v1.1 text encoder (the step before the denoise path), from safetensors (bf16 / f32)
 * or, as the reference does by default, from a GGUF-quantised file (see ltx_t5_create_from_gguf below).
 *
 * Replaces `T5TextEncoderWrapper` as a `VTextEncoder` (reference: src/models/ltx_video/text_encoder.rs:315-345, 597-606),
 * i.e. `candle_transformers::models::t5::T5EncoderModel` built from `T5EncoderConfig::to_candle_t5_config` (:222-249):
 * gated NewGelu feed-forward, bidirectional relative-position bias (block 0's table shared by all layers), T5LayerNorm,
 * unscaled QK^T, NO attention mask (`forward(input_ids)` passes ids only).  Weight names are the Hugging Face ones the
 * reference's VarBuilder reads: shared.weight, encoder.block.N.layer.0.{SelfAttention.{q,k,v,o,relative_attention_bias},
 * layer_norm}.weight, encoder.block.N.layer.1.{DenseReluDense.{wi_0,wi_1,wo},layer_norm}.weight,
 * encoder.final_layer_norm.weight.
 */
#ifndef LTXHIP_T5_H
#define LTXHIP_T5_H
#include "ltxhip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ltx_t5 ltx_t5;
/* T5EncoderConfig (text_encoder.rs:66-113); ltx_t5_config_default = the t5_xxl preset (:169-184). */
typedef struct {
    int vocab_size, d_model, d_kv, d_ff, num_layers, num_heads;
    int relative_attention_num_buckets, relative_attention_max_distance;
    float layer_norm_epsilon;
} ltx_t5_config;
void ltx_t5_config_default(ltx_t5_config* c);

int ltx_t5_create(const ltx_t5_config* cfg, const ltx_weight* weights, size_t n_weights,
                  ltx_dtype model_dtype, int device, ltx_t5** out);
void ltx_t5_destroy(ltx_t5* m);
/* VTextEncoder::forward (text_encoder.rs:600-605): input_ids HOST int32 [B,S] (S <= 512) ->
 * out [B,S,d_model] out_dtype (device).  The caller pads ids with 0 and builds the mask itself (VTokenizer::encode_batch,
 * :612-640); like the reference, the encoder does not mask padded positions. */
int ltx_t5_forward(ltx_t5* m, const int32_t* input_ids, int B, int S, ltx_dtype out_dtype, void* out, ltx_stream stream);

/* The reference's DEFAULT text encoder, `QuantizedT5EncoderModel` (quantized_t5_encoder.rs:558-679; main.rs:441-444): the same
 * network with its weights read from a GGUF file - token_embd.weight, enc.blk.N.{attn_q,attn_k,attn_v,attn_o,attn_norm,
 * ffn_gate (wi_0), ffn_up (wi_1), ffn_down (wo), ffn_norm}.weight, enc.blk.0.attn_rel_b.weight [buckets, heads],
 * enc.output_norm.weight - every tensor dequantised to f32 before use (QLinear::forward, :53-72), and WITH an attention
 * mask: scores + position_bias + (1 - mask) * -1e9 over the keys (:624-634, 218-220).
 *   ltx_t5_create_from_gguf : load_with_config (:575-603); model_dtype F32 = the reference's arithmetic, BF16 rounds the
 *                             dequantised weights once;
 *   ltx_t5_forward_masked   : forward(input_ids, Some(mask)) - attention_mask HOST f32 [B,S] (1 keep, 0 pad) or NULL (no mask,
 *                             then it is ltx_t5_forward). */
int ltx_t5_create_from_gguf(const ltx_t5_config* cfg, const char* gguf_path, ltx_dtype model_dtype, int device, ltx_t5** out);
int ltx_t5_forward_masked(ltx_t5* m, const int32_t* input_ids, const float* attention_mask, int B, int S,
                          ltx_dtype out_dtype, void* out, ltx_stream stream);

#ifdef __cplusplus
}
#endif
#endif
