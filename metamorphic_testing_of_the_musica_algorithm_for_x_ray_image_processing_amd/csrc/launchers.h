// launchers.h — host-callable launch wrappers implemented in the kernel files, plus the
// argument blocks they share with the dispatch script in musica_ctx.hip.
#pragma once

#include <stdlib.h>
#include "musica_device.h"

namespace musica {

// How the contrast gain of a level is obtained (contrast_curve_apply.comp:61 with the curve of
// contrast_curve_generate.comp:56-94):
//   GAIN_CONST : levels >= 4 — the sdev image is never written there (src/vk_processing.cpp:2285), so
//                getY(0) == points[0].y == highContrastFactor exactly;
//   GAIN_RANGE : level 3 — two-point constant curve but a real sdev: getY(s) = 0*s + high for
//                0 <= s <= 1 (== high for finite s), 0 for s > 1 or NaN;
//   GAIN_CURVE : levels 0..2 — the 33-point polyline.
enum { GAIN_CONST = 0, GAIN_RANGE = 1, GAIN_CURVE = 2 };

struct ExpandArgs {
    const float* prev;   // coarse reconstruction (or downsampled[L-1] for the first slot)
    const float* band;
    const float* sdev;   // GAIN_RANGE / GAIN_CURVE
    const float* cnr;    // noise reduction (levels 0, 1)
    float* recon;
    const DevCurve* curves;  // curve of this level for image 0 (GAIN_CURVE)
    const DevCurveLut* luts; // its bucket table (GAIN_CURVE), stride MUSICA_COARSER_LEVELS_START per image
    int S, pitch; size_t plane;
    int Sc, cpitch; size_t cplane;
    int cnrS, cnrPitch; size_t cnrPlane;
    int cnrScale;        // uint(ceil(S / float(cnrS))), noise_reduction.comp:38
    float high;          // highContrastFactor of the level
    float lowCnr, lowFactor, highCnr, highFactor;  // NoiseReductionParams of the level
    int rows_per_wave;
    size_t curve_stride; // DevCurve elements between consecutive images
    // Level 0 only, NULL otherwise: the launch also accumulates the gradation histogram (img_relevant.comp +
    // gradation_histogram.comp) of the texels it has just reconstructed, see k_expand_fast<.., GH = true>.
    const uint16_t* raw;     // raw pixels (dense rows of S): `normalized <= 0.9` is tested as raw <= thr090[image]
    uint32_t* ghist;         // [batch][1024]
    uint32_t* gzero;         // [batch]: set when a reconstructed texel is exactly 0 (the `return` of gradation_histogram.comp:24
                             // then cuts the scan of its 16 x 16 area short: k_grad_hist redoes that image literally)
    const uint16_t* le090;   // or: [batch][Sc][S / 8] bits of `normalized <= 0.9` written by launch_reduce_band_u16 (then raw / thr090 are not read)
    uint32_t* chist;         // CLAHE contexts, with le090: [batch][4][4][256], the launch also counts clahe_histogram.comp (k_expand_fast<.., CH = true>)
    int swz;                 // XCD-aware tile mapping (kernels_common.h xcd_tile)
    const int* thr090;       // [batch]: largest raw value whose normalized value is <= 0.9 (k_curves_cnr)
    int ref_order;           // generic kernels: the shaders' literal 25-tap order (MUSICA_FLAG_REFERENCE_ORDER)
};

struct GradArgs {
    const float* img;        // expandImageStates[L-1]: the contrast-enhanced image
    const float* normalized;
    const float* cnr;
    uint32_t* hist;          // [batch][1024]
    int N, pitch; size_t plane;
    int cnrS, cnrPitch; size_t cnrPlane;
    int cnrScale;            // uint(ceil(N / float(cnrS))), img_relevant.comp:32
    int groups_per_wave;     // 16-row groups each wavefront walks
    const uint16_t* raw;     // non-NULL: test `normalized <= 0.9` on the raw pixels (dense rows of N) instead of reading `normalized`
    const uint32_t* minmax;
    int min_chain_exact;
    const uint32_t* only_if; // non-NULL: images whose word is 0 are skipped (fix-up launch behind the fused expand kernel)
};

// kernels_pyramid.hip
// ref != 0 (generic kernels only): the shaders' literal 25-tap accumulation order (MUSICA_FLAG_REFERENCE_ORDER)
void launch_reduce(hipStream_t st, const float* in, const LevelDesc& li, float* out, const LevelDesc& lo, int batch, bool force_generic, int tag, int ref = 0);
void launch_band(hipStream_t st, const float* fine, const float* coarse, float* band, const LevelDesc& lf, const LevelDesc& lc, int batch, int ref = 0);
void launch_reduce_band_u16(hipStream_t st, const uint16_t* px, float* down, float* band, const LevelDesc& lf, const LevelDesc& lc, int batch,
                            int rows_per_wave, const uint32_t* minmax, int min_chain_exact, uint16_t* le090);
void launch_reduce_band(hipStream_t st, const float* fine, float* down, float* band, const LevelDesc& lf, const LevelDesc& lc, int batch, int rows_per_wave);
void launch_lowpass(hipStream_t st, const float* coarse, float* low, const LevelDesc& lf, const LevelDesc& lc, int batch, int ref = 0);
void launch_expand(hipStream_t st, const ExpandArgs& a, int gain_mode, bool nr, int batch, bool force_generic);
// reduce + band of level i + 1 AND the sdev + noise-histogram pass of level i in one launch (both read what the reduce + band launch of level i wrote
// and nothing of each other; kernels_expand_sd.hip). Workgroups 0 .. of grid.x take the sdev pass (sl.first = 0), the rest the reduce + band launch.
struct RbSdevArgs {
    const float* fine; float* down; float* band;   // reduce + band of level i + 1: its fine image, coarse image, band image
    int S, pitch; size_t plane; int Sc, cpitch; size_t cplane; int rows_rb;
    int rb_strips, rb_blocks, rb_first;            // that role's workgroups: rb_strips * rb_blocks from rb_first on (a multiple of 8)
    SdevRunLevel sl;                               // level i's pass: march (sl.rows > 0) or one run per workgroup; sl.sdev == nullptr: histogram only
    size_t hist_stride; int cov, swz;
};
void launch_rb_sdev(hipStream_t st, RbSdevArgs a, const LevelDesc& ls, int batch);
void launch_expand_sd(hipStream_t st, const ExpandArgs& a, bool nr, int batch);   // a.sdev == nullptr: sdev computed by the launch (kernels_expand_sd.hip)
// reduce + band of the levels in `a`, then their expand slots, one workgroup per image (levels of side <= kTailSide)
void launch_tiny_tail(hipStream_t st, const TailArgs& a, int batch);
void launch_exp_band(hipStream_t st, const ExpandArgs& a, int gain_mode, bool nr, int batch);
// kernels_analysis.hip
void launch_clear(hipStream_t st, uint32_t* minmax, uint32_t* noise_hist, uint32_t* grad_hist, uint32_t* clahe_hist, int batch, uint32_t* grad_hist_b = nullptr, uint32_t* gzero = nullptr);
// min / max of the raw pixels (slots: [batch][kMinMaxSlots] words, ticket: [batch] words, zero at first use) and, where the pointers are
// given, the clears of src/vk_processing.cpp:2153-2162 in the same launch
constexpr int kMinMaxSlots = 4096;
void launch_minmax(hipStream_t st, const uint16_t* px, int N, uint32_t* minmax, uint32_t* slots, uint32_t* ticket, int batch,
                   uint32_t* noise_hist = nullptr, uint32_t* grad_hist = nullptr, uint32_t* grad_hist_b = nullptr, uint32_t* gzero = nullptr,
                   uint32_t* clahe_hist = nullptr);
void launch_normalize(hipStream_t st, const uint16_t* px, float* out, const LevelDesc& l0, const uint32_t* minmax, int min_chain_exact, int batch);
void launch_sqrt(hipStream_t st, const uint16_t* px, float* out, const LevelDesc& l0, int batch);
void launch_sdev_hist(hipStream_t st, const float* band, float* sdev, const LevelDesc& l, uint32_t* hist, size_t hist_stride, int cov, int batch, int rows_per_wave);
// the 16-row-run form of launch_sdev_hist for n <= kSdevRunLevelsMax levels in ONE launch (hist[k]: image 0's histogram of level k)
void launch_sdev_hist_runs(hipStream_t st, int n, const float* const* band, float* const* sdev, const LevelDesc* lv, uint32_t* const* hist,
                           size_t hist_stride, int cov, int batch);
// every level's sdev + noise-histogram pass in ONE launch, each level in its own form (rows[k] > 0: the march with that many rows per
// wavefront, 0: one 16-row run per workgroup); sdev[k] == nullptr: histogram only
void launch_sdev_hist_levels(hipStream_t st, int n, const float* const* band, float* const* sdev, const LevelDesc* lv, uint32_t* const* hist,
                             const int* rows, size_t hist_stride, int cov, int batch);
void launch_noise_hist_only(hipStream_t st, const float* sdev, const LevelDesc& l, uint32_t* hist, size_t hist_stride, int cov, int batch);
// img_sdev.comp:10-35 with the 25 squares accumulated in the shader's order (one thread per texel; MUSICA_FLAG_REFERENCE_ORDER)
void launch_sdev_literal(hipStream_t st, const float* band, float* sdev, const LevelDesc& l, int batch);
void launch_sdev_only(hipStream_t st, const float* band, float* sdev, const LevelDesc& l, int batch);   // the fast order's sdev image alone (no histogram)
void launch_noise_curves(hipStream_t st, const uint32_t* hist, size_t hist_stride, musica_hist_max_point* maxpts, DevCurve* curves, const musica_contrast_params* cparams, int levels, int batch, DevCurveLut* luts, const uint32_t* minmax, int min_chain_exact, int* thr090, int lev0 = 0, int nlev = 0);
void launch_curves_cnr(hipStream_t st, const uint32_t* hist, size_t hist_stride, musica_hist_max_point* maxpts, DevCurve* curves,
                       const musica_contrast_params* cparams, int levels, int batch, DevCurveLut* luts, const float* sdev, float* cnr,
                       const LevelDesc& l3, const uint32_t* minmax, int min_chain_exact, int* thr090, int lev0 = 0);
void launch_cnr(hipStream_t st, const float* sdev, float* cnr, const LevelDesc& l3, const musica_hist_max_point* maxpts, int levels, int batch);
void launch_selftest_exact_math(hipStream_t st, unsigned long long* d_bad4);
// the RGBA plots of RENDER_HISTS (kernels_gradation.hip): out = MUSICA_HIST_RENDER_WIDTH x MUSICA_HIST_RENDER_HEIGHT packed texels
void launch_render_noise_hist(hipStream_t st, const uint32_t* hist, const musica_hist_max_point* maxpt, uint32_t* out);
void launch_render_grad_hist(hipStream_t st, const uint32_t* hist, const musica_hist_max_point* maxpt, const DevCurve* curve, uint32_t* out);
constexpr int kStatsMaxBlocks = 64;
// XCD-aware workgroup -> tile mapping of the marching kernels (kernels_common.h xcd_tile); MUSICA_XCD_SWIZZLE=0 turns it off
int xcd_swizzle_on();
void launch_stats(hipStream_t st, const float* cnr, const LevelDesc& l3, const uint32_t* minmax, int min_chain_exact,
                  const musica_hist_max_point* noise_max, int levels, const musica_hist_max_point* grad_max, const DevCurve* gcurve,
                  musica_stats* out, uint32_t image_id_base, uint32_t image_id_stride, int batch, double* partial);
// kernels_gradation.hip
void launch_grad_hist(hipStream_t st, const GradArgs& a, int batch);
void launch_grad_hist_ref(hipStream_t st, const float* img, const float* relevant, const LevelDesc& l0, uint32_t* hist, int batch);
void launch_relevant(hipStream_t st, const float* normalized, const float* cnr, float* out, const LevelDesc& l0, const LevelDesc& l3,
                     int cnrScale, int batch, const uint16_t* raw = nullptr, const int* thr090 = nullptr);
// recount (images whose a.only_if word is set, into a.hist) + tone curve in one launch; ticket: [batch][kGradTicketStride] words, zero
constexpr int kGradTicketStride = 32;
void launch_grad_recount_curve(hipStream_t st, const GradArgs& a, uint32_t* hist, musica_hist_max_point* gmax, DevCurve* curves, uint32_t* ticket, int batch);
void launch_grad_curve(hipStream_t st, uint32_t* hist, musica_hist_max_point* gmax, DevCurve* curves, int batch, const uint32_t* hist_b = nullptr, const uint32_t* gzero = nullptr);
void launch_grad_apply(hipStream_t st, const float* in, float* out, const LevelDesc& l0, const DevCurve* curves, int batch);
// crop + quantise of saveOutImage for ONE image plane: out = (S - 2 margin)^2 bytes, dense
void launch_out_pixels(hipStream_t st, const float* graded, const LevelDesc& l0, int margin, uint8_t* out);
void launch_out_bmp24(hipStream_t st, const float* graded, const LevelDesc& l0, int margin, uint32_t* out);   // the BMP file's pixel array (24 bpp, bottom-up, padded rows)
// kernels_bench.hip (measurement aid)
void launch_copy41(hipStream_t st, const float* in, float* out, int side);
// kernels_clahe.hip
void launch_clahe(hipStream_t st, const float* img, const float* relevant, float* out, const LevelDesc& l0, uint32_t* hist, musica_point* pts, int batch,
                  const uint16_t* raw = nullptr, const int* thr090 = nullptr, const float* cnr = nullptr, const LevelDesc* l3 = nullptr, int cnrScale = 0,
                  bool hist_done = false /* the histogram is already in `hist` (counted by the level-0 expand launch) */,
                  bool with_apply = true /* false: histogram + curves only, launch_grad_clahe_apply follows */);
// K21 + K24 in one pass (l0.S % 4 == 0): out_graded = tone curve of `curves`, out_clahe = CLAHE blend of `pts`
void launch_grad_clahe_apply(hipStream_t st, const float* img, float* out_clahe, float* out_graded, const LevelDesc& l0, const musica_point* pts,
                             const DevCurve* curves, int batch);

}  // namespace musica
