// musica_device.h — device-side conventions shared by the kernel files and the host pipeline.
//
// Layout in HBM (one musica_ctx, batch B):
//   every f32 image of pyramid level i is stored as B planes of S_i rows with a row pitch
//   (in floats) rounded up to a multiple of 4, so every row starts 16-byte aligned and every
//   16-byte vector access inside [0, pitch) is in bounds; plane stride = pitch * S_i.
//   The raw input is dense uint16 (pitch N), exactly what the ABI hands over.
//
// Arithmetic contract (must match oracle/musica_oracle.c, MUSICA_ORDER_FAST, bit for bit):
//   IEEE binary32, no FMA contraction (this directory is compiled with -ffp-contract=off),
//   correctly rounded division and sqrt (hipcc default), left-to-right 5-tap chains that start
//   from the first product, vertical pass before horizontal pass.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/musica.h"

namespace musica {

// img_smooth.comp:23-30 — glslang folds the constant array in double and narrows once.
__device__ constexpr float W0 = 0.1f, W1 = 0.25f, W2 = 0.3f, W3 = 0.25f, W4 = 0.1f;

constexpr int kLaneCols = 8;                    // fine columns owned by one lane in the streaming kernels
constexpr int kStripCols = 64 * kLaneCols;      // fine columns covered by one 64-lane wavefront
constexpr int kWavesPerBlock = 4;
constexpr int kBlockThreads = 64 * kWavesPerBlock;

constexpr float kMaxNoiseValue = 0.1f;          // noise_hist.comp:7
constexpr float kMaxCnrValue = 256.0f;          // img_cnr.comp:6
constexpr int kHistArea = 16;                   // noise_hist.comp:5
constexpr int kMinMaxStride = 32;                // uint32 words per image in the min/max scratch (128 B)
constexpr int kMaxWord = 16;                     // the max lives one 64-byte line after the min
constexpr int kCurveCap = 64;                   // >= 33 (contrast) and 22 (gradation) points + 1 guard

// A polyline (T3 / T5 of the reference) in structure-of-arrays form plus the per-segment
// slope (p2.y - p1.y) / (p2.x - p1.x) the shaders recompute per pixel (contrast_curve_apply.comp:22-25).
// `monotone` = 1 when x[] is non-decreasing and NaN-free, which makes the shaders' first-match
// linear search equal to a binary search (see curve_eval in kernels_common.h).
struct DevCurve {
    float x[kCurveCap];
    float y[kCurveCap];
    float m[kCurveCap];
    uint32_t count;
    uint32_t monotone;
    float t0, ta, t1;       // gradation window (unused for contrast curves)
    uint32_t pad;
};

// Exact accelerator for getY() on a 33-point contrast curve (levels 0..2): j = #{x[i] < s} from ONE 16-byte table read
// and two compares, then the segment (x[j-1], y[j-1], slope[j-1]) from a second 16-byte read — no branches.
//   fine table   : s * inv_w < kLutBuckets (s below 1.75 p): uniform buckets over [0, 1.75 p); a bucket stores how many
//                  abscissae lie in lower buckets (jlo) and the at most two inside it (xa <= xb, +inf when absent);
//   coarse table : every other s: buckets of width 1 / 256 over [0, 1] (index min(int(s * 256), kLutCoarse - 1)); only the
//                  10 abscissae of the last Bezier span (x[23..32], spaced >= 0.01 apart) can lie at or above 1.75 p, so
//                  jlo = 23 + (tail abscissae in lower coarse buckets) and the bucket holds at most two of them;
//   #{x[i] < s} = jlo + (xa < s) + (xb < s).
// Bucket membership is decided by the same float expressions when the table is built and when it is read, and both are
// monotone in x, so the count is exact whatever the rounding: an abscissa in a lower bucket is < s, one in a higher bucket
// is > s (proof in DESIGN.md, "Exactness notes"). The builder verifies that exactly abscissae 0..22 fall inside the fine
// table, that no bucket holds three and that x[0] == 0; otherwise (e.g. maxBin == 0) ok = 0 and the level takes the
// literal scan.
constexpr int kLutBuckets = 256;    // fine buckets
constexpr int kLutCoarse = 258;     // coarse buckets: int(s * 256) = 0 .. 256, and one for everything above
constexpr int kLutTailFirst = 23;   // abscissae 0..22 are <= 1.4 p and always inside the fine table's range
constexpr int kLutPoints = 33;      // 3 x generateCurve(i <= 10), contrast_curve_generate.comp:72-86
struct DevCurveLut {
    float inv_w;
    uint32_t ok;
    uint32_t pad0, pad1;
    float4 bucket[kLutBuckets + kLutCoarse];   // {16 * jlo (integer bits: byte offset of seg[jlo]), xa, xb, unused}
    float4 seg[kLutPoints + 1];                 // seg[j] = {x[j-1], y[j-1], slope[j-1], 0}; seg[0] = {x[0], y[0], 0, 0}; seg[33] = 0
};

struct LevelDesc {
    int S;          // side
    int pitch;      // row pitch in floats (multiple of 4)
    size_t plane;   // floats per image plane = pitch * S
};

// the levels of one k_sdev_hist_runs launch (kernels_analysis.hip): workgroups first .. of grid.x take level k's runs
constexpr int kSdevRunLevelsMax = 4;
struct SdevRunLevel {
    const float* band;
    float* sdev;
    uint32_t* hist;   // image 0's histogram of this level
    size_t plane;
    int S, pitch, strips, first;
    int rows;         // k_sdev_hist_levels: rows per wavefront of a level that marches, 0 for a level that takes one 16-row run per workgroup
    int blocks;       // k_sdev_hist_levels: workgroups per strip
};
struct SdevRunLevels {
    SdevRunLevel l[kSdevRunLevelsMax];
    int n;
    int swz;          // k_sdev_hist_levels: XCD-aware workgroup -> tile mapping inside a level
};

// the levels of one k_tiny_tail launch (kernels_pyramid.hip): level T + k reads `fine`, writes its reduced image `down`, its
// band-pass image and its reconstruction; Sl / lpitch / lplane: the geometry of the last level's `down` (the coarsest image)
constexpr int kTailMax = 8;
constexpr int kTailSide = 32;   // finest level the tail takes
struct TailLevel {
    const float* fine;
    float* down;
    float* band;
    float* recon;
    size_t plane;
    int S, pitch;
    float high;   // highContrastFactor of the level (GAIN_CONST)
};
struct TailArgs {
    TailLevel l[kTailMax];
    size_t lplane;
    int n, Sl, lpitch, ref;
};

static inline int round_up4(int v) { return (v + 3) & ~3; }

}  // namespace musica
