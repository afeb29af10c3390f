// grad_parts.h — getY() of the tone curve on LDS tables, shared by k_grad_apply (kernels_gradation.hip) and the CLAHE
// context's one-pass apply of both curves (kernels_clahe.hip k_grad_clahe_apply4).
#pragma once
#include "kernels_common.h"

namespace musica {

// getY() for a monotone polyline without a branch: j = #{x[i] < s} by a 6-step binary search over x[] padded with +inf (no
// `probe <= count` test), one 16-byte read of seg[j] = {x[j-1], y[j-1], slope[j-1]} and curve_eval()'s arithmetic; j = 0 and
// j >= count (and NaN, which counts 0) take curve_eval()'s `x[0] == s ? y[0] : 0`. Same values as curve_eval() for every s.
struct GradLds {
    float xs[kCurveCap];          // x[i], +inf at i >= count
    float4 seg[kCurveCap + 1];    // seg[j] for 1 <= j < count; zeros elsewhere
};
// STEP0: the first probe distance — 16 for curves of fewer than 32 points (the 22-point tone curve), 32 otherwise.
template <int STEP0>
__device__ __forceinline__ float grad_eval_mono(const GradLds& t, uint32_t last_b /* (count - 1) * 4 */, float x0, float y0, float s) {
    uint32_t jb = 0u;   // 4 * j
#pragma unroll
    for (int step = STEP0; step >= 1; step >>= 1) {
        const float xv = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(t.xs) + jb + (uint32_t)(step - 1) * 4u);
        jb += xv < s ? (uint32_t)step * 4u : 0u;
    }
    const float4 g = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(t.seg) + jb * 4u);
    const float r = g.z * (s - g.x) + g.y;
    const float alt = s == x0 ? y0 : 0.0f;
    return (jb - 4u) < last_b ? r : alt;   // 1 <= j <= count - 1
}

template <int MONO>   // 0: literal scan, 16 / 32: branch-free search with that first probe distance
__device__ __forceinline__ float4 grad_eval4(const CurveLds& tab, const GradLds& gl, uint32_t last_b, float x0, float y0, float4 v) {
    float4 o;
    if (MONO) {
        o.x = grad_eval_mono<MONO ? MONO : 32>(gl, last_b, x0, y0, v.x);
        o.y = grad_eval_mono<MONO ? MONO : 32>(gl, last_b, x0, y0, v.y);
        o.z = grad_eval_mono<MONO ? MONO : 32>(gl, last_b, x0, y0, v.z);
        o.w = grad_eval_mono<MONO ? MONO : 32>(gl, last_b, x0, y0, v.w);
    } else {
        o.x = curve_eval(tab, v.x);                                      // img_apply_gradation_curve.comp:44
        o.y = curve_eval(tab, v.y);
        o.z = curve_eval(tab, v.z);
        o.w = curve_eval(tab, v.w);
    }
    return o;
}

// the tone curve of one image into the two LDS forms (every thread of the workgroup; the caller's barrier follows)
__device__ __forceinline__ void grad_tables_to_lds(CurveLds& tab, GradLds& gl, const DevCurve* __restrict__ cv) {
    curve_to_lds(tab, cv);
    const uint32_t count = cv->count;
    for (int k = threadIdx.x; k <= kCurveCap; k += blockDim.x) {
        if (k < kCurveCap) gl.xs[k] = (uint32_t)k < count ? cv->x[k] : __int_as_float(0x7F800000);
        gl.seg[k] = (k >= 1 && (uint32_t)k < count) ? make_float4(cv->x[k - 1], cv->y[k - 1], cv->m[k - 1], 0.0f) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

}  // namespace musica
