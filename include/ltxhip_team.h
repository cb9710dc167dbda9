/* ltxhip_team.h — the exchanges of the multi-GPU forms of the path for hosts that have no
 * torch.distributed: thin entry points over RCCL (xGMI inside a node).  The reference is single-device
 * (examples/ltx-video/main.rs:210-214) and has no counterpart; the two places where `LtxPipeline::call` partitions are
 *     guidance branches   t2v_pipeline.rs:878-939   one all-gather of the f32 noise predictions per denoise step
 *     VAE temporal tiles  vae.rs:2382-2434          one strip (<= blend frames) point to point to the next rank, then one
 *                                                   all-gather of the finished frame ranges
 * (candle-video_amd/ltxhip/sharded.py runs the same pattern through torch.distributed; INTEGRATION.md section 7 shows a
 * host loop over these entry points).
 * librccl.so is opened with dlopen at the first call: libltxhip.so has no link-time dependency on it, and a single-GPU host
 * never loads it.  One process per GPU; pointers are DEVICE pointers; calls are enqueued on `stream`. */
#ifndef LTXHIP_TEAM_H
#define LTXHIP_TEAM_H
#include "ltxhip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ltx_team ltx_team;
#define LTX_TEAM_ID_BYTES 128            /* = NCCL_UNIQUE_ID_BYTES */

/* ncclGetUniqueId: called by ONE rank; the 128 bytes reach the other ranks out of band (file, socket, environment). */
int ltx_team_unique_id(void* id_out_host);
/* ncclCommInitRank on `device` (collective: every rank of the team calls it with the same id and nranks). */
int ltx_team_create(const void* id_host, int nranks, int rank, int device, ltx_team** out);
void ltx_team_destroy(ltx_team* t);
int ltx_team_size(const ltx_team* t);
int ltx_team_rank(const ltx_team* t);

/* ncclAllGather of `count` f32 per rank: recv holds nranks * count values in rank order (send may alias its own slot).
 * Guidance branches: count = B * S * 128 (2.56 MB at C2), then every rank runs ltx_guidance_step on the gathered
 * predictions, so latents stay replicated without a broadcast. */
int ltx_team_allgather_f32(ltx_team* t, const float* send, float* recv, size_t count, ltx_stream stream);
/* One grouped ncclSend + ncclRecv: `send_count` f32 to rank `send_to` and `recv_count` f32 from rank `recv_from`
 * (either peer may be -1 = none).  Temporal tiles: the last min(blend, len) decoded frames of a rank's last tile go to the
 * owner of the next tile before the blend_t of vae.rs:2410-2434. */
int ltx_team_exchange_f32(ltx_team* t, const float* send, size_t send_count, int send_to,
                          float* recv, size_t recv_count, int recv_from, ltx_stream stream);

#ifdef __cplusplus
}
#endif
#endif
