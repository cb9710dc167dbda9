/* ltxhip_weights.h — weight ingestion for the LTX-Video path: everything between a checkpoint on
 * disk and `ltx_dit_create` / `ltx_vae_create`.  Host-side string/IO logic restating
 *     src/models/ltx_video/weight_format.rs   (format detection, Official -> Diffusers key remapping)
 *     src/models/ltx_video/loader.rs          (name-mapping rules, safetensors index, directory resolution)
 *     examples/ltx-video/main.rs:455-546      (splitting a unified checkpoint into VAE and transformer tensors)
 * Tensor payloads are never copied on the host: a safetensors file is mmap'ed and `ltx_weight.data` points into the map;
 * the model constructors upload + cast on the device (bf16 or f32 sources).
 */
#ifndef LTXHIP_WEIGHTS_H
#define LTXHIP_WEIGHTS_H
#include "ltxhip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* WeightFormat / detect_format (weight_format.rs:13-29): a regular file is the Official single-file checkpoint (1);
 * a directory — or a path that does not exist — is the Diffusers layout (0). */
int ltx_weights_detect_format(const char* path);

/* KeyRemapper::remap_key (weight_format.rs:55-83, block-index tables :96-141): Official (native) -> Diffusers name.
 * Writes the NUL-terminated result into out[cap]; fails with LTX_ERR_ARG if it does not fit. */
int ltx_weights_remap_key(const char* key, char* out, size_t cap);
/* KeyRemapper::is_transformer_key / is_vae_key (weight_format.rs:144-163): 1 / 0. */
int ltx_weights_is_transformer_key(const char* key);
int ltx_weights_is_vae_key(const char* key);

/* WeightLoader name mapping (loader.rs:63-112, 223-317): ordered exact / prefix / suffix rules; every rule that
 * matches rewrites the name and its result feeds the next rule. */
typedef struct ltx_name_mapper ltx_name_mapper;
typedef enum { LTX_MAP_EXACT = 0, LTX_MAP_PREFIX = 1, LTX_MAP_SUFFIX = 2 } ltx_map_kind;
ltx_name_mapper* ltx_name_mapper_create(void);
void ltx_name_mapper_destroy(ltx_name_mapper* m);
int ltx_name_mapper_add(ltx_name_mapper* m, ltx_map_kind kind, const char* from, const char* to);
int ltx_name_mapper_has_mapping(const ltx_name_mapper* m, const char* name);              /* loader.rs:296-300 */
int ltx_name_mapper_map(const ltx_name_mapper* m, const char* name, char* out, size_t cap);   /* loader.rs:306-316 */

/* validate_tensor_names (loader.rs:495-505): writes the indices of `expected` names absent from `actual` into
 * missing_idx[n_expected] (in order) and their count into *n_missing. */
int ltx_weights_validate_names(const char* const* expected, size_t n_expected, const char* const* actual, size_t n_actual,
                               size_t* missing_idx, size_t* n_missing);

/* One mmap'ed safetensors file (8-byte LE header length, JSON header, raw little-endian payload). */
typedef struct ltx_safetensors ltx_safetensors;
int ltx_safetensors_open(const char* path, ltx_safetensors** out);
void ltx_safetensors_close(ltx_safetensors* st);
size_t ltx_safetensors_count(const ltx_safetensors* st);
/* i-th tensor in file (header) order.  dtype_name is the safetensors dtype string ("F32", "BF16", "F16", ...);
 * data points into the mapping and stays valid until close. */
int ltx_safetensors_tensor(const ltx_safetensors* st, size_t i, const char** name, const char** dtype_name,
                           int* ndim, const int64_t** shape, const void** data, size_t* nbytes);

/* WeightLoader::load_from_directory's file resolution (loader.rs:341-397, 437-456) — or a single file path as is:
 *   dir/model.safetensors.index.json -> its distinct shard files (all must exist), else dir/model.safetensors,
 *   else every *.safetensors in dir (sorted).  Writes up to cap NUL-separated absolute-or-relative paths into
 *   out (double-NUL terminated) and their count into *n_files. */
int ltx_weights_resolve(const char* path, char* out, size_t cap, size_t* n_files);

/* Build the models straight from checkpoint files.
 *   unified != 0 : `path` is an Official single-file checkpoint; keys are remapped and split exactly as
 *                  main.rs:461-499 does ("vae." / "model.diffusion_model." / "transformer." prefixes stripped);
 *   unified == 0 : `path` is the component's own safetensors file or directory (Diffusers layout), tensor names taken
 *                  as they are (WeightLoader::load_single / load_from_directory).
 * F32 and BF16 payloads are accepted (anything else: LTX_ERR_UNSUPPORTED naming the tensor). */
/* AutoencoderKLLtxVideoConfig from a diffusers `vae/config.json` (serde names and aliases of vae.rs:30-66); fields the file
 * does not carry keep the value already in *cfg (serde(default) over Default::default() when the caller passed
 * ltx_vae_config_default).  Host only. */
int ltx_vae_config_from_json(const char* json_path, ltx_vae_config* cfg);

/* ltx_vae_create_from_files, unified == 0: when a `config.json` sits beside the weights (in `path` if it is a directory, else
 * in its parent directory) it REPLACES *cfg, exactly as examples/ltx-video/main.rs:525-533 does, and timestep_conditioning is
 * then forced on (main.rs:534).  unified != 0 uses *cfg as given (main.rs:511-512 takes the preset's). */
int ltx_dit_create_from_files(const ltx_dit_config* cfg, const char* path, int unified,
                              ltx_dtype model_dtype, int device, ltx_dit** out);
int ltx_vae_create_from_files(const ltx_vae_config* cfg, const char* path, int unified,
                              ltx_dtype model_dtype, int device, ltx_vae** out);

/* ---- GGUF (the container of the reference's DEFAULT text encoder: examples/ltx-video/main.rs:261-296 picks
 * t5-v1_1-xxl-encoder-Q8_0.gguf / -Q5_K_M.gguf unless --use-bf16-t5; quantized_t5_encoder.rs:570-600 reads it through
 * candle's `VarBuilder::from_gguf`, which is not in the checkout: the published GGUF v2 / v3 layout and ggml block formats are
 * restated).  One mmap'ed file; tensors in file order; shapes reported OUTERMOST FIRST (as candle reports them: a linear
 * weight is [out, in]); `data` points into the mapping. */
typedef struct ltx_gguf ltx_gguf;
int ltx_gguf_open(const char* path, ltx_gguf** out);
void ltx_gguf_close(ltx_gguf* g);
size_t ltx_gguf_count(const ltx_gguf* g);
int ltx_gguf_tensor(const ltx_gguf* g, size_t i, const char** name, int* ggml_type, int* ndim, const int64_t** shape,
                    const void** data, size_t* nbytes);           /* LTX_ERR_UNSUPPORTED (fields still filled) for a type not read */
int ltx_gguf_find(const ltx_gguf* g, const char* name);            /* index or -1 */
/* elements and bytes per block of a ggml type; read: F32 (0), F16 (1), BF16 (30), Q4_0 (2), Q5_0 (6), Q8_0 (8), Q4_K (12),
 * Q5_K (13), Q6_K (14).  Others: LTX_ERR_UNSUPPORTED. */
int ltx_gguf_type_info(int ggml_type, int* block_elems, int* block_bytes);
/* QTensor::dequantize on the device: `numel` elements of ggml blocks (HOST or DEVICE memory) -> dense dst (DEVICE) of dst_dtype,
 * ggml's reference f32 arithmetic (each product / difference rounded separately), then one rounding to bf16 if asked. */
int ltx_gguf_dequantize(int ggml_type, const void* blocks, int on_device, int64_t numel, ltx_dtype dst_dtype, void* dst, ltx_stream stream);

#ifdef __cplusplus
}
#endif
#endif
