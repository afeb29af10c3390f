/*
 * musica_oracle.h — CPU restatement of the reference's MUSICA path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / the CPU number reported beside the GPU
 * one. libmusica_hip.so never links, loads or calls it.
 *
 * PARITY UNPINNED for the pixel arithmetic: the reference is a Vulkan/SPIR-V
 * program that cannot be built or run here (no Vulkan SDK, no glslc, no GPU
 * ICD), it ships no golden vectors for this path (raw_images/ are
 * missing blobs), and it has no CPU implementation. This file therefore
 * restates shaders/X.comp and the dispatch script of
 * src/vk_processing.cpp:2104-2601 line by line and is pinned only by analytic
 * known-answer tests derived from the shader text (tests/test_oracle_kat.py).
 * The one piece that IS pinned against real reference code is the BMP byte
 * layout: oracle/ref_bmp.c compiles the reference's vendored
 * dependencies/stb/stb_image_write.h into oracle/_ref/ and the tests compare
 * bytes.
 *
 * Two arithmetic orders:
 *   MUSICA_ORDER_REFERENCE  literal 25-tap loops, x-outer / y-inner, exactly
 *                           as the shaders accumulate;
 *   MUSICA_ORDER_FAST       separable (vertical pass then horizontal pass)
 *                           evaluation of the three 5x5 stencils — the order
 *                           the HIP kernels use, so HIP == oracle bit for bit.
 * Everything else (reductions, histograms, curves, remaps) is identical in
 * both orders. All arithmetic is IEEE binary32 without FMA contraction
 * (compile with -ffp-contract=off).
 */
#ifndef MUSICA_ORACLE_H
#define MUSICA_ORACLE_H

#include <stdint.h>
#include "../include/musica.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MUSICA_ORDER_REFERENCE 0
#define MUSICA_ORDER_FAST 1

#define MUSICA_ORACLE_FLAG_CLAHE 0x1u

typedef struct musica_oracle musica_oracle;

/* levels == 0 => ceil(log2 N) as src/vk_processing.cpp:1989. */
musica_oracle* musica_oracle_create(uint32_t image_size, uint32_t levels, int order, uint32_t flags);
/* the same with the parameter formulas' constants given (NULL: the reference's, include/vk_processing.h:16-17, 39-49) */
musica_oracle* musica_oracle_create_ex(uint32_t image_size, uint32_t levels, int order, uint32_t flags, const musica_tunables* tunables);
void musica_oracle_tunables_default(musica_tunables* out);
void musica_oracle_destroy(musica_oracle* o);
/* Number of OpenMP threads used by the oracle's loops (1 = scalar port). */
void musica_oracle_set_threads(int n);
int musica_oracle_get_threads(void);

uint32_t musica_oracle_levels(const musica_oracle* o);
uint32_t musica_oracle_level_size(const musica_oracle* o, uint32_t level);

/* VulkanProcessing::execute restated (src/vk_processing.cpp:2104-2601). */
int musica_oracle_execute(musica_oracle* o, const uint16_t* pixels);
/* Run one stage on the oracle's current state (same stage ids as musica.h). */
int musica_oracle_run_stage(musica_oracle* o, int stage);

/* Every intermediate the reference materialises, dense f32. Kinds 0..11 are
 * musica_image_kind; the extra kinds below exist only in the oracle. */
#define MUSICA_ORACLE_IMG_SMOOTH 100       /* smoothImageStates[level] */
#define MUSICA_ORACLE_IMG_UPSAMPLED 101    /* upsampledImageStates[level] (zero-inserted) */
#define MUSICA_ORACLE_IMG_EXP_UPSAMPLED 102 /* expandUpsampledImageStates[L-1-level] */
#define MUSICA_ORACLE_IMG_EXP_LOWPASS 103  /* expandLowpassImageStates[L-1-level] */
#define MUSICA_ORACLE_IMG_CONTRAST_BAND 104 /* expandBandpassImageStates[L-1-level] (before noise reduction) */
#define MUSICA_ORACLE_IMG_NR_BAND 105      /* expandBandpassNoiseRedImages[2-level], level 0..2 */
const float* musica_oracle_image(const musica_oracle* o, int kind, uint32_t level, uint32_t* side);
int musica_oracle_set_image(musica_oracle* o, int kind, uint32_t level, const float* src);

const uint32_t* musica_oracle_noise_hist(const musica_oracle* o, uint32_t level);
const uint32_t* musica_oracle_grad_hist(const musica_oracle* o);
musica_hist_max_point musica_oracle_noise_hist_max(const musica_oracle* o, uint32_t level);
musica_hist_max_point musica_oracle_grad_hist_max(const musica_oracle* o);
const musica_contrast_curve* musica_oracle_contrast_curve(const musica_oracle* o, uint32_t level);
const musica_grad_curve* musica_oracle_grad_curve(const musica_oracle* o);
musica_contrast_params musica_oracle_contrast_params(const musica_oracle* o, uint32_t level);
musica_nr_params musica_oracle_nr_params(const musica_oracle* o, uint32_t level);
void musica_oracle_minmax(const musica_oracle* o, float* min_sqrt, float* max_sqrt);
void musica_oracle_stats(const musica_oracle* o, musica_stats* dst);
const uint32_t* musica_oracle_clahe_hist(const musica_oracle* o);
const musica_point* musica_oracle_clahe_curves(const musica_oracle* o);

/* saveOutImage restated (src/vk_processing.cpp:2603-2645): cropped 8-bit pixels, side N-20. */
int musica_oracle_out_pixels(const musica_oracle* o, uint8_t* dst);
int musica_oracle_save_out_image(const musica_oracle* o, const char* path);
/* debugProcess (src/vk_processing.cpp:2661-2756): the image dumps norm / red_bandpass_i / red_lowpass_i / sdev / cnr /
 * exp_bandpass_i / exp_lowpass_i / relevant / graded as 8-bit BMPs into `dir` (quantised as src/vk_state.cpp:834). */
int musica_oracle_debug_process(const musica_oracle* o, const char* dir);
/* The two RGBA plots of #define RENDER_HISTS (noise_hist_render.comp on the cnr level, gradation_curve_debug_render.comp):
 * MUSICA_HIST_RENDER_W x MUSICA_HIST_RENDER_H x 4 bytes, row-major, top row first. */
#define MUSICA_HIST_RENDER_W 512u
#define MUSICA_HIST_RENDER_H 128u
void musica_oracle_render_noise_hist(const musica_oracle* o, uint8_t* rgba);
void musica_oracle_render_grad_hist(const musica_oracle* o, uint8_t* rgba);
/* stbi_write_bmp(path, w, h, 4, data) restated (stb_image_write.h:501-509). */
int musica_oracle_write_bmp_rgba(const char* path, uint32_t w, uint32_t h, const uint8_t* data);
/* stbi_write_bmp(path, w, h, 1, data) restated (stb_image_write.h:492-500). */
int musica_oracle_write_bmp_gray(const char* path, uint32_t w, uint32_t h, const uint8_t* data);
/* Raw reader of test/standalone/main.cpp:54-75. */
int musica_oracle_read_raw(const char* path, uint32_t image_size, uint16_t* dst);

/* ---- single-shader entry points (KATs and kernel-level parity) ---------
 * Images are dense row-major side x side; `out` must not alias `in`. */
void musica_oracle_k_sqrt(const uint16_t* in, uint32_t side, float* out);                      /* img_sqrt.comp */
/* One link of the chain: out side = ceil(side/8). */
void musica_oracle_k_max_reduce(const float* in, uint32_t side, float* out);                   /* img_max_reduce.comp */
void musica_oracle_k_min_reduce(const float* in, uint32_t side, float* out);                   /* min_reduce.comp */
void musica_oracle_k_normalize(const float* in, uint32_t side, float minv, float maxv, float* out); /* img_normalize.comp */
void musica_oracle_k_smooth(const float* in, uint32_t side, float* out, int order);            /* img_smooth.comp */
void musica_oracle_k_downsample(const float* in, uint32_t side, float* out);                   /* img_downsample.comp; out side ceil(side/2) */
void musica_oracle_k_upsample(const float* in, uint32_t in_side, float* out, uint32_t out_side); /* img_upsample.comp; out must be pre-zeroed */
void musica_oracle_k_smooth_upsampled(const float* in, uint32_t side, float* out, int order);  /* img_smooth_upsampled.comp */
void musica_oracle_k_difference(const float* a, const float* b, uint32_t side, float* out);   /* img_difference.comp */
void musica_oracle_k_addition(const float* a, const float* b, uint32_t side, float* out);      /* img_addition.comp */
void musica_oracle_k_sdev(const float* in, uint32_t side, float* out, int order);              /* img_sdev.comp */
/* groups = workgroup count per axis the reference dispatches (imageSize / 512, .cpp:2293-2295). */
void musica_oracle_k_noise_hist(const float* sdev, uint32_t side, uint32_t groups, uint32_t* hist /*2048, accumulated into*/); /* noise_hist.comp */
void musica_oracle_k_histogram_max(const uint32_t* hist, uint32_t bins, musica_hist_max_point* out); /* img_histogram_max.comp */
void musica_oracle_k_contrast_curve_generate(musica_hist_max_point mp, musica_contrast_params cp, musica_contrast_curve* out); /* contrast_curve_generate.comp */
void musica_oracle_k_contrast_curve_apply(const float* band, const float* sdev, uint32_t side, const musica_contrast_curve* curve, float* out); /* contrast_curve_apply.comp */
void musica_oracle_k_cnr(const float* sdev, uint32_t side, musica_hist_max_point mp, float* out); /* img_cnr.comp */
void musica_oracle_k_noise_reduction(const float* band, uint32_t side, const float* cnr, uint32_t cnr_side, musica_nr_params p, float* out); /* noise_reduction.comp */
void musica_oracle_k_relevant(const float* normalized, uint32_t side, const float* cnr, uint32_t cnr_side, float* out); /* img_relevant.comp */
/* groups = ceil(imageSize / 512) (.cpp:2492-2494). */
void musica_oracle_k_gradation_histogram(const float* img, const float* relevant, uint32_t side, uint32_t groups, uint32_t* hist /*1024, accumulated into*/); /* gradation_histogram.comp */
void musica_oracle_k_gradation_curve_generate(const uint32_t* hist, musica_grad_curve* out);   /* gradation_curve_generate.comp */
void musica_oracle_k_apply_gradation_curve(const float* in, uint32_t side, const musica_grad_curve* curve, float* out); /* img_apply_gradation_curve.comp */
float musica_oracle_get_y(const musica_point* points, uint32_t count, float x);                /* getY of contrast_curve_apply.comp:27-36 */
/* CLAHE trio (disabled in the reference, restated from shader text only). */
void musica_oracle_k_clahe_histogram(const float* img, const float* relevant, uint32_t side, uint32_t* hist /*4*4*256 [tx][ty][bin], accumulated*/); /* clahe_histogram.comp */
void musica_oracle_k_clahe_grad_curve(const uint32_t* hist, musica_point* points /*4*4*256 [tx][ty][i]*/); /* clahe_grad_curve.comp */
void musica_oracle_k_clahe_grad_curve_apply(const float* in, uint32_t side, const musica_point* points, float* out); /* clahe_grad_curve_apply.comp */

/* Host parameter formulas of initMemory (src/vk_processing.cpp:259-297, 321-325). */
musica_contrast_params musica_oracle_host_contrast_params(uint32_t level, uint32_t levels);
musica_nr_params musica_oracle_host_nr_params(uint32_t i);
musica_contrast_params musica_oracle_host_contrast_params_ex(uint32_t level, uint32_t levels, const musica_tunables* t);
musica_nr_params musica_oracle_host_nr_params_ex(uint32_t i, const musica_tunables* t);

#ifdef __cplusplus
}
#endif
#endif
