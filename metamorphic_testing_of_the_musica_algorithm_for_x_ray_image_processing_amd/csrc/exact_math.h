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
// Fast path: a' = cur * 10 differs from a by at most 1.4e-7 relative (0.1f is 1.5e-8 above 1/10, two
// roundings), so t' = a' * 2048 + 0.5 differs from the exact t by less than 3e-7 * t'; when neither an
// integer boundary of t nor the a > 1 boundary is that close the truncated t' is the exact bin.
// Otherwise (a few texels in ten thousand) the literal sequence runs.
MUSICA_HD int musica_noise_bin_exact(float cur) {
    if (cur != cur) return 0;  // int(NaN) is undefined in GLSL; restated as 0, i.e. the `binPosition == 0` break
    if (cur == 0.0f) return 0;
    const float a = cur / 0.1f;
    if (a > 1.0f) return 0;
    return (int)(a * 2048.0f + 0.5f);
}
MUSICA_HD int musica_noise_bin(float cur) {
    const float a1 = cur * 10.0f;
    const float t1 = a1 * 2048.0f + 0.5f;
    const float fl = floorf(t1);
    const float fr = t1 - fl;
    const float m = t1 * 4.0e-7f + 1.0e-30f;
    const int safe = (fr > m) && (fr < 1.0f - m) && (a1 < 0.9999990f);
    if (safe) return (int)fl;        // cur == 0 gives t1 = 0.5 -> bin 0 == break; NaN fails every comparison
    return musica_noise_bin_exact(cur);
}
