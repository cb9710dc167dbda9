/* ltxhip_frames.h — frame output (the step after the path; examples/ltx-video/main.rs:648-681).
 * The decoded video [B,3,F,H,W] f32 in [0,255] becomes per-frame interleaved RGB8 (permute (1,2,0), clamp, truncating
 * u8 cast, :659-664) on the device, and `--frames` output files frame_%04d.png (:670-675) on the host. */
#ifndef LTXHIP_FRAMES_H
#define LTXHIP_FRAMES_H
#include "ltxhip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* video DEVICE f32 [B,3,F,H,W] (0..255 scale) -> rgb DEVICE u8 [B,F,H,W,3]; values are clamped to [0,255] and truncated. */
int ltx_video_to_rgb8(const float* video, int B, int F, int H, int W, uint8_t* rgb, ltx_stream stream);
/* Write one 8-bit RGB image (HOST pointer, row-major H x W x 3) as a PNG file (zlib deflate, no filtering). */
int ltx_write_png(const char* path, const uint8_t* rgb, int width, int height);
/* main.rs:653-675 in one call: convert on the device, copy to the host, write dir/frame_%04d.png (index = b*F + f;
 * the directory is created if missing).  Returns the number of files written in *n_written (may be NULL). */
int ltx_save_frames_png(const float* video, int B, int F, int H, int W, const char* dir, int* n_written, ltx_stream stream);

/* The reference's DEFAULT output (main.rs:683-707): an animated GIF - every frame quantised to a local 256-colour palette
 * (NeuQuant, sampling factor `speed` 1..30; the reference uses 30), LZW-coded, `delay_cs` centiseconds per frame (the
 * reference: 4), looping forever, no global palette.  rgb_frames: HOST u8 [n_frames, height, width, 3]. */
int ltx_write_gif(const char* path, const uint8_t* rgb_frames, int n_frames, int width, int height, int delay_cs, int speed);
/* main.rs:653-707 in one call: convert on the device, copy to the host, write the GIF (speed 30, delay 4; frames in b*F+f order). */
int ltx_save_video_gif(const float* video, int B, int F, int H, int W, const char* path, ltx_stream stream);

#ifdef __cplusplus
}
#endif
#endif
