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

// Exact accelerator for getY() on a 33-point contrast curve (levels 0..2). Texels with
// s * inv_w < kLutBuckets fall into a uniform bucket that stores how many curve abscissae lie in
// lower buckets (jlo) and the at most two abscissae inside it (xa <= xb, +inf when absent):
//   #{x[i] < s} = jlo + (xa < s) + (xb < s).
// Bucket membership is decided by the same float expression (int)(x * inv_w) when the table is built
// and when it is read, and that expression is monotone in x, so the count is exact whatever the
// rounding. Larger s only has the 10 abscissae of the last Bezier span left to compare with.
// ok == 0 (degenerate curve, e.g. maxBin == 0) sends the level down the literal scan instead.
constexpr int kLutBuckets = 256;
constexpr int kLutTailFirst = 23;   // abscissae 0..22 are <= 1.4 p and always inside the table's range
struct DevCurveLut {
    float inv_w;
    uint32_t ok;
    uint32_t pad0, pad1;
    float4 bucket[kLutBuckets];     // {jlo (as float), xa, xb, unused}
};

struct LevelDesc {
    int S;          // side
    int pitch;      // row pitch in floats (multiple of 4)
    size_t plane;   // floats per image plane = pitch * S
};

static inline int round_up4(int v) { return (v + 3) & ~3; }

}  // namespace musica
