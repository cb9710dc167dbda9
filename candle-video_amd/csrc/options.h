// Run-time options of the library: ONE parsed structure instead of getenv() calls in the launch paths.
//   * read once, at first use, from the environment variable LTX_OPTIONS = "key=value,key=value,...";
//   * changed at run time with ltx_set_option (include/ltxhip.h; tests and the A/B tools);
//   * nothing here is needed to run the engine: every option is an A/B or diagnostic aid.
// The documented list - with the options that select another ALGORITHM (results differ in rounding) marked - is in
// include/ltxhip.h.  Plans (which tile / kernel of a family serves a GEMM shape) never change a bit of the result.
// Measured-negative experiments are compiled only with -DLTX_EXPERIMENTS (tools/ variant builds): their knobs are read
// with ltx_exp("name", default), which is the default itself in the shipped build.
#pragma once

enum { LTX_FAM_ASM16 = 1, LTX_FAM_RING = 2, LTX_FAM_P8 = 4, LTX_FAM_HALO = 8, LTX_FAM_HALO_OUT = 16, LTX_FAM_BIG = 32 };   // gemm_off
enum { LTX_ATTN_Q64 = 1, LTX_ATTN_Q128 = 2, LTX_ATTN_CROSS = 4, LTX_ATTN_PIPE = 8 };                                         // attn_off

struct LtxOptions {
    // ---- speed only (same bits)
    int gemm_tune = 1;            // 0: no plan measurement, the static cost model (ltx_set_autotune(0) does the same per process)
    char gemm_plan[40] = {0};     // force a plan where the shape is eligible: "256x128" (gemm_big tile), "asm16:160x256" / "asm16",
                                  // "ring:96x96" / "ring", "p8:256", "halo:128"
    unsigned gemm_off = 0;        // plan families left out: "asm16+ring+p8+halo+halo_out" ("big" is listed below: another K partition)
    int gemm_wide_epi = 1;        // 0: fragment-wise 8-byte epilogue stores in gemm_big / conv_halo
    int gemm_trace = 0;           // 1: print the bf16 shapes left to gemm.hip's 128 x 128 kernel
    int attn_q64_big = -1;        // attn_q64: number of 256-query blocks per head (-1: the launcher's split)
    int vae_tile_batch = 0;       // leaves per decoder call of the tiled decode (0: sized by free memory)
    int prof_kernel_events = 1;   // 0: stream-level event brackets in the per-class timing
    int ff2_defer = 1;            // 0: small-M ff2 reduces its K ranges in the launch instead of leaving them to the next row norm
    // ---- another algorithm (rounding differs; each is a tested A/B arm)
    int gemm_splitk = 1;          // 0: small outputs keep one K range
    int q2_fold = 1;              // 0: stand-alone cross-attention q-norm pass; 2: fold whatever the shape
    int norm_presum = 1;          // 0: row-reducing RMS norms; 2: the map whatever the shape
    int norm_lean = 1;            // 0: the general presum kernel (same bits as the lean one)
    int xattn_compact = 1;        // 0: cross attention multiplies every text key
    int attn_q64_stream = 0;      // 1: head_dim-64 self-attention as persistent workgroups streaming host-built item lists (attn_q64.hip; round 6: built, tested, 4 - 6 % behind the block grid)
    int norm_fold = 2;            // 0: the DiT's RMS norms between GEMMs as their own (presum) pass; 1: folded into the producer's / consumer's epilogues (dit.hip); 2: the (1 + scale) factor in per-timestep copies of the consumer's weights
    int norm_fold_copies = 10;    // norm_fold=2: scaled-weight copies kept per DiT handle (one per distinct timestep; 1.6 GB each at 2B: a distilled schedule's 7 + the warm-up's)
    int guidance_batch = 1;       // 0: the guidance branches of a step (uncond / text / perturbed) as separate forwards, the reference's call order
    int dense_qkv = 1;            // 0: q | k | v as column slices of one [M, 3D] matrix
    int vae_fuse_norm = 1;        // 0: the resnet's second norm as its own pass; 2: fused on grids below one round of the chip too (tests)
    int t5_attn_mfma = 1;         // 0: the scalar T5 attention kernel
    unsigned attn_off = 0;        // attention kernels left out: "q64+q128+cross+pipe" (the next more general kernel serves)
    // gemm_off bit LTX_FAM_BIG: gemm.hip's 128 x 128 kernel for everything (un-split K)
};

const LtxOptions& ltx_opt();
int ltx_exp_lookup(const char* name, int dflt);      // options.cpp: "x_<name>=<int>" entries of LTX_OPTIONS / ltx_set_option
#ifdef LTX_EXPERIMENTS
#define ltx_exp(name, dflt) ltx_exp_lookup(name, dflt)
#else
#define ltx_exp(name, dflt) (dflt)
#endif
