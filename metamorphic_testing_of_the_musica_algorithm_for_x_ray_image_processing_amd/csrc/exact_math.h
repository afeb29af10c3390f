// exact_math.h — cheaper instruction sequences that return EXACTLY what the plain IEEE expression
// returns, for every non-negative float input. Used by the HIP kernels; compiled for the host by
// oracle/exhaustive.c, which checks both functions against the plain expressions over all 2^31
// non-negative floats (tests/test_exact_math.py).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MUSICA_HD __host__ __device__ __forceinline__
#else
#define MUSICA_HD static inline
#endif

// x / 25.0f (img_sdev.comp:30, `sum / count`). One Newton step on q0 = x * RN(1/25) with an exact FMA
// residual; equal to the correctly rounded quotient for every non-negative float, denormals and
// infinity included (exhaustively verified).
MUSICA_HD float musica_div25(float x) {
    const float y = 0.04f;  // RN(1/25)
    const float q = x * y;
    const float r = fmaf(-25.0f, q, x);
    return r == r ? fmaf(r, y, q) : q;  // x = +inf: the residual is NaN, the quotient is q = +inf
}

// noise_hist.comp:29-45 for one texel value `cur`:
//   if (cur == 0) break;  a = cur / 0.1f;  if (a > 1) break;  bin = int(a * 2048 + 0.5);  if (bin == 0) break;
// Returns the bin (1 .. 2048; 2048 is out of the image and dropped by the caller) or 0 for "break".
MUSICA_HD int musica_noise_bin_exact(float cur) {
    if (cur != cur) return 0;  // int(NaN) is undefined in GLSL; restated as 0, i.e. the `binPosition == 0` break
    if (cur == 0.0f) return 0;
    const float a = cur / 0.1f;
    if (a > 1.0f) return 0;
    return (int)(a * 2048.0f + 0.5f);
}
// The same in 7 instruction slots and without a branch. The division by the constant 0.1f is one reciprocal multiply
// (RN(1 / 0.1f) is exactly 10.0f) corrected once through the exact FMA residual, like musica_div25; a * 2048 is exact (a power
// of two), so a * 2048 + 0.5 is one FMA; cur == 0 gives 0.5 -> bin 0 = break; `!(a <= 1)` covers a > 1 and NaN (a huge cur
// overflows the estimate to inf, the residual to NaN: also a break, as the literal a > 1). Equal to musica_noise_bin_exact for
// every non-negative float and every NaN (exhaustive, tests/test_exact_math.py).
MUSICA_HD int musica_noise_bin(float cur) {
    const float q = cur * 10.0f;
    const float r = fmaf(-0.1f, q, cur);
    const float a = fmaf(r, 10.0f, q);
    const float t = fmaf(a, 2048.0f, 0.5f);
    return (a <= 1.0f) ? (int)t : 0;
}

// img_normalize.comp:24, `(sqrt - min) / (max - min)`: both chain scalars are integer-valued floats in
// 0 .. 255 (kernels_common.h chain_scalars), so den = max - min is an integer 1 .. 255 (0: flat image, the
// caller keeps the literal division for it) and x = sqrtf(v) - min for a 16-bit v. With rden = RN(1 / den)
// (one literal division per image) the quotient is q0 = x * rden corrected once through the exact FMA
// residual. Equal to x / den for EVERY (v, min, den) triple of that domain — 65536 x 256 x 255 cases, all
// checked by oracle/exhaustive.c musica_check_norm_div (tests/test_exact_math.py).
MUSICA_HD float musica_norm_div(float x, float den, float rden) {
    const float q = x * rden;
    const float r = fmaf(-den, q, x);
    return fmaf(r, rden, q);
}

#if defined(__HIPCC__)
// sqrtf(x) from one v_rsq_f32 and one FMA residual step: y = rsq(x), s = x * y, s' = s + (x - s * s) * (y / 2).
// On gfx950 this equals the correctly rounded sqrtf for EVERY float in [2^-100, FLT_MAX] and for +0 (rsq(0) = inf
// is clamped so that s = 0); below 2^-100 the residual underflows, and +inf / NaN / negatives go wrong — those
// take the literal sqrtf. v_rsq_f32 is a hardware approximation, so the exhaustive check runs on the GPU itself:
// musica_selftest_sqrt (tests/test_gpu_parity.py) compares both functions below with sqrtf over all 2^32 patterns.
__device__ __forceinline__ float musica_sqrt_core(float x) {
    const float y = fminf(__builtin_amdgcn_rsqf(x), 3.0e38f);
    const float s = x * y, h = 0.5f * y;
    const float r = fmaf(-s, s, x);
    return fmaf(r, h, s);
}
// True when musica_sqrt_core(x) is proven exact: x == +0 or 2^-100 <= x <= FLT_MAX.
__device__ __forceinline__ bool musica_sqrt_core_ok(float x) {
    const uint32_t u = __float_as_uint(x);
    return u == 0u || (u - 0x0D800000u) < (0x7F800000u - 0x0D800000u);
}
// Eight at once with one range test for the group: every bit pattern below +inf (this also rejects NaN, -0 and
// negatives, whose patterns are larger) and every non-zero one at least 2^-100 (pattern - 1 wraps for +0).
__device__ __forceinline__ void musica_sqrt8(float s[8]) {
    uint32_t mx = 0u, mn = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t u = __float_as_uint(s[j]);
        mx = max(mx, u);
        mn = min(mn, u - 1u);
    }
    const bool ok = mx < 0x7F800000u && mn >= 0x0D800000u - 1u;
    if (__builtin_expect(ok, 1)) {
#pragma unroll
        for (int j = 0; j < 8; j++) s[j] = musica_sqrt_core(s[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) s[j] = sqrtf(s[j]);
    }
}
__device__ __forceinline__ float musica_sqrt(float x) {
    const float s = musica_sqrt_core(x);
    if (__builtin_expect(!musica_sqrt_core_ok(x), 0)) return sqrtf(x);
    return s;
}
#endif
