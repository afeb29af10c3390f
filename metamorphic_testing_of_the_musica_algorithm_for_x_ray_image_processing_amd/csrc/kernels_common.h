// kernels_common.h — device helpers shared by the HIP kernels (gfx950 / CDNA4 only).
#pragma once

#include "musica_device.h"
#include "exact_math.h"

namespace musica {

// ---- cross-lane moves on the 64-wide wavefront (DPP wave shifts, no LDS) ----
// wave_shr:1 — lane i receives lane i-1 (lane 0 receives 0); wave_shl:1 — lane i receives lane i+1.
// (bound_ctrl = 1: the lane without a source receives 0 from the instruction itself — with bound_ctrl = 0 and an `old` operand of 0
// the compiler had to zero the destination with a v_mov in front of every DPP move.)
__device__ __forceinline__ float from_left_lane(float v) {
    const int s = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(s, s, 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float from_right_lane(float v) {
    const int s = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(s, s, 0x130, 0xF, 0xF, true));
}

// ---- branch-free memory access through buffer descriptors ----
// A raw buffer load whose byte offset lies outside the descriptor returns 0 and touches no memory;
// a store there is dropped. Lanes that must not access memory therefore just carry kOob as their
// offset: no `if` around a load, so hipcc never parks an s_waitcnt inside a divergent block and a
// wavefront can keep every load of a loop trip in flight. It is also exactly the reference's
// "out-of-bounds imageLoad returns 0 / imageStore is dropped" rule.
// Descriptors are built from wave-uniform values only (one image plane, < 2 GiB).
// (ROCm 7.2's __builtin_amdgcn_raw_buffer_load_b64/_b128 lower to a single-dword load that is then
// splatted, so the LLVM intrinsics are bound directly by their names.)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ v4f llvm_buffer_load_v4f32(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ v2f llvm_buffer_load_v2f32(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ float llvm_buffer_load_f32(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ void llvm_buffer_store_v4f32(v4f data, v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");
__device__ short llvm_buffer_load_i16(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i16");
__device__ void llvm_buffer_store_i16(short data, v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.i16");
constexpr uint32_t kOob = 0x80000000u;

struct Buf {
    v4i r;
};
// {base address, num_records (bytes), 0x00020000 = 32-bit untyped data format on gfx9}
__device__ __forceinline__ Buf make_buf(const void* p, size_t bytes) {
    union {
        v4i v;
        struct { const void* p; uint32_t range; uint32_t cfg; } s;
    } u;
    u.s.p = p;
    u.s.range = (uint32_t)(bytes < 0x7FFFFFF0u ? bytes : 0x7FFFFFF0u);
    u.s.cfg = 0x00020000u;
    Buf b;
    b.r.x = __builtin_amdgcn_readfirstlane(u.v.x);
    b.r.y = __builtin_amdgcn_readfirstlane(u.v.y);
    b.r.z = __builtin_amdgcn_readfirstlane(u.v.z);
    b.r.w = __builtin_amdgcn_readfirstlane(u.v.w);
    return b;
}
__device__ __forceinline__ float4 bload4(const Buf& b, uint32_t off) {
    const v4f v = llvm_buffer_load_v4f32(b.r, (int)off, 0, 0);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float2 bload2(const Buf& b, uint32_t off) {
    const v2f v = llvm_buffer_load_v2f32(b.r, (int)off, 0, 0);
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ float bload1(const Buf& b, uint32_t off) { return llvm_buffer_load_f32(b.r, (int)off, 0, 0); }
__device__ __forceinline__ uint32_t bload_u16(const Buf& b, uint32_t off) { return (uint32_t)(uint16_t)llvm_buffer_load_i16(b.r, (int)off, 0, 0); }
__device__ __forceinline__ void bstore_u16(const Buf& b, uint32_t off, uint32_t v) { llvm_buffer_store_i16((short)v, b.r, (int)off, 0, 0); }
__device__ __forceinline__ void bstore4(const Buf& b, uint32_t off, float4 v) {
    v4f u;
    u.x = v.x; u.y = v.y; u.z = v.z; u.w = v.w;
    llvm_buffer_store_v4f32(u, b.r, (int)off, 0, 0);
}
// Non-temporal form (aux bit 1 = nt): for kernels that read several bytes per byte they write and do not read their output
// again (the metric kernel, reduce + band). Measured per kernel, not a blanket rule: alone on the chip the sdev launch is 5 - 20 %
// slower with it (round 3), and so were the expand launches and the gradation apply then; with steps in flight
// (round 4, same-box A/B of whole steps) the reconstruction and graded stores are 4 % of a step faster non-temporal and a lone context is no slower.
__device__ __forceinline__ void bstore4_nt(const Buf& b, uint32_t off, float4 v) {
    v4f u;
    u.x = v.x; u.y = v.y; u.z = v.z; u.w = v.w;
    llvm_buffer_store_v4f32(u, b.r, (int)off, 0, 2);
}

// ---- which tile a workgroup works on ----
// The hardware hands workgroups to the 8 XCDs round-robin by linear workgroup id, and every XCD has an L2 of its own: with the
// plain (blockIdx.x = strip, blockIdx.y = block of segments) mapping, the strips left and right of a strip and — unless the grid
// is 8 wide — the segments above and below it belong to other XCDs, so the halo lines two neighbours both read are fetched
// once per XCD. With swz the workgroups of XCD k take the k-th eighth of the segment blocks, strip by strip: neighbours in
// both directions share one L2 (and are dispatched back to back). Needs gridDim.y % 8 == 0 (else the plain mapping).
struct Tile { int strip, segblock; };
__device__ __forceinline__ Tile xcd_tile(int swz) {
    Tile t;
    t.strip = (int)blockIdx.x; t.segblock = (int)blockIdx.y;
    const unsigned gx = gridDim.x, gy = gridDim.y;
    if (swz && (gy & 7u) == 0u) {
        const unsigned lin = blockIdx.x + gx * blockIdx.y;
        const unsigned xcd = lin & 7u, j = lin >> 3;
        t.strip = (int)(j % gx);
        t.segblock = (int)(xcd * (gy >> 3) + j / gx);
    }
    return t;
}

// ---- arithmetic in the oracle's MUSICA_ORDER_FAST order ----
// ((((w0*a + w1*b) + w2*c) + w3*d) + w4*e), products and sums rounded separately.
__device__ __forceinline__ float chain5(float a, float b, float c, float d, float e) {
    float acc = W0 * a;
    acc = acc + W1 * b;
    acc = acc + W2 * c;
    acc = acc + W3 * d;
    acc = acc + W4 * e;
    return acc;
}
// zero-inserted grid, even phase: taps 0, 2, 4 of the 5-tap kernel (the zero taps add +0 exactly).
__device__ __forceinline__ float chain_even(float a, float b, float c) {
    float acc = W0 * a;
    acc = acc + W2 * b;
    acc = acc + W4 * c;
    return acc;
}
// zero-inserted grid, odd phase: taps 1, 3.
__device__ __forceinline__ float chain_odd(float a, float b) {
    float acc = W1 * a;
    acc = acc + W3 * b;
    return acc;
}

// mirror() of img_smooth.comp:10-16 (reflect-101; the clamp there is a no-op).
__device__ __forceinline__ int mirror_idx(int n, int hi) {
    int v = n;
    if (v > hi) v = hi - (v - hi);
    else if (v < 0) v = -v;
    return v;
}

// float -> uint as the GPU's v_cvt_u32_f32 does it: truncate, negative / NaN -> 0, saturate.
__device__ __forceinline__ uint32_t f2u(float v) {
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}

// The two scalars img_normalize.comp:17-18 reads from the 1x1 ends of the min / max chains, from the
// integer extrema k_minmax_u16 left in `minmax` (see kernels_analysis.hip).
__device__ __forceinline__ void chain_scalars(const uint32_t* __restrict__ minmax, int img, int min_chain_exact, float& minv, float& maxv) {
    const uint32_t mnu = minmax[kMinMaxStride * img], mxu = minmax[kMinMaxStride * img + kMaxWord];
    maxv = (float)f2u(sqrtf((float)mxu));
    minv = min_chain_exact ? (float)f2u(sqrtf((float)mnu)) : 0.0f;
}
// img_sqrt.comp:15 + img_normalize.comp:24 for one raw pixel (den = max - min), literally.
__device__ __forceinline__ float norm_px(uint32_t v, float minv, float den) { return (sqrtf((float)v) - minv) / den; }
// The same value from the exact shortcuts of exact_math.h: 9 + 1 + 4 VALU slots instead of 18 + 1 + 14.
// Both chain scalars are integers 0 .. 255, so den is an integer 0 .. 255: den >= 1 is the exhaustively checked
// domain of musica_norm_div; den == 0 (flat image) gives x / 0 = x * inf, which is q itself (rden = 1 / 0 = inf).
struct NormK {
    float minv, den, rden;
    bool flat;
};
__device__ __forceinline__ NormK make_norm(float minv, float maxv) {
    NormK k;
    k.minv = minv;
    k.den = maxv - minv;
    k.rden = 1.0f / k.den;
    k.flat = !(k.den >= 1.0f);
    return k;
}
__device__ __forceinline__ float norm_px(uint32_t v, const NormK& k) {
    const float x = musica_sqrt_core((float)v) - k.minv;  // v <= 65535: inside the core's exact range
    const float q = x * k.rden;
    const float r = fmaf(-k.den, q, x);
    const float f = fmaf(r, k.rden, q);
    return k.flat ? q : f;
}
// Largest raw value whose normalized value is <= 0.90 (img_relevant.comp:56), -1 if there is none. norm_px is
// non-decreasing in v (sqrt, subtraction and division by den >= 0 are monotone; den == 0 gives inf / NaN, for
// which the comparison is false for every v), so `normalized <= 0.9` is exactly `raw <= threshold`.
__device__ __forceinline__ int norm_threshold_090(float minv, float den) {
    int lo = -1, hi = 65535;  // invariant: P(lo) true (or lo == -1), P(hi + 1) false
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (norm_px((uint32_t)mid, minv, den) <= 0.90f) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// Relevance weight uint(relevant * 100) (img_relevant.comp:44-63, gradation_histogram.comp:30) of one cnr texel, split
// into the part that depends only on cnr (shared by all the pixels under that texel) and the per-pixel tests:
//   w = inside ? (ramp ? w_ramp : (high && pixel <= 0.9 ? 100 : 0)) : 0.     `c` is cnr * 256 already;
// pow(r, 5.0) is restated as ((r*r)*(r*r))*r (oracle Q5).
struct CnrClass {
    uint32_t w_ramp;  // uint(((r*r)*(r*r))*r * 100) for 1 <= cnr <= 6, r = cnr / 6
    bool ramp, high;  // 1 <= cnr <= 6 (first branch wins at cnr == 6) ; 6 <= cnr <= 256
};
__device__ __forceinline__ CnrClass classify_cnr(float c) {
    CnrClass k;
    k.ramp = c >= 1.0f && c <= 6.0f;
    k.high = c >= 6.0f && c <= kMaxCnrValue;
    const float r = c / 6.0f;
    k.w_ramp = f2u((((r * r) * (r * r)) * r) * 100.0f);
    return k;
}

// 16-byte load of 4 consecutive floats of a row; columns >= valid_cols come back as 0.
// `row` must be 16-byte aligned at column x (x % 4 == 0) and x < pitch.
__device__ __forceinline__ float4 load4_guard(const float* __restrict__ row, int x, int valid_cols) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x < valid_cols) {
        v = *reinterpret_cast<const float4*>(row + x);
        if (x + 1 >= valid_cols) v.y = 0.f;
        if (x + 2 >= valid_cols) v.z = 0.f;
        if (x + 3 >= valid_cols) v.w = 0.f;
    }
    return v;
}

// ---- polyline evaluation: getY() of contrast_curve_apply.comp:27-36 ----
// LDS copy of a DevCurve.
struct CurveLds {
    float x[kCurveCap];
    float y[kCurveCap];
    float m[kCurveCap];
    uint32_t count;
    uint32_t monotone;
};

__device__ __forceinline__ void curve_to_lds(CurveLds& dst, const DevCurve* __restrict__ src) {
    for (int i = threadIdx.x; i < kCurveCap; i += blockDim.x) {
        dst.x[i] = src->x[i];
        dst.y[i] = src->y[i];
        dst.m[i] = src->m[i];
    }
    if (threadIdx.x == 0) {
        dst.count = src->count;
        dst.monotone = src->monotone;
    }
}

// The shader scans i = 0 .. count-1 and returns at the first i with x[i] == s, or
// x[i] <= s <= x[i+1] (x[count] is the never-written entry behind the curve: 0).
// For a non-decreasing, NaN-free x[] that first match is i = j - 1 where j = #{x[i] < s}
// (proof in DESIGN.md §Kernels/curve lookup), found here by a 6-step branch-free binary search;
// otherwise the literal scan runs.
__device__ __forceinline__ float curve_eval(const CurveLds& t, float s) {
    const int count = (int)t.count;
    if (t.monotone) {
        int j = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
            int probe = j + step;
            if (probe <= count && t.x[probe - 1] < s) j = probe;
        }
        if (j == 0) return (t.x[0] == s) ? t.y[0] : 0.0f;
        if (j >= count) return 0.0f;
        return t.m[j - 1] * (s - t.x[j - 1]) + t.y[j - 1];
    }
    for (int i = 0; i < count; i++) {
        float xi = t.x[i];
        if (xi == s) return t.y[i];
        float xn = (i + 1 < count) ? t.x[i + 1] : 0.0f;
        float yn = (i + 1 < count) ? t.y[i + 1] : 0.0f;
        if (xi <= s && xn >= s) {
            float m = (yn - t.y[i]) / (xn - xi);
            return m * (s - xi) + t.y[i];
        }
    }
    return 0.0f;
}

// Fills m[] and monotone for a curve whose x[], y[], count are set (one thread).
__device__ __forceinline__ void curve_finish(DevCurve* c) {
    const int count = (int)c->count;
    uint32_t mono = 1;
    for (int i = 0; i < kCurveCap; i++) {
        if (i + 1 < count) {
            c->m[i] = (c->y[i + 1] - c->y[i]) / (c->x[i + 1] - c->x[i]);
            if (!(c->x[i] <= c->x[i + 1])) mono = 0;
        } else {
            c->m[i] = 0.0f;
        }
        if (i >= count) {
            c->x[i] = 0.0f;
            c->y[i] = 0.0f;
        }
    }
    if (count > 0 && !(c->x[0] == c->x[0])) mono = 0;
    // the binary search probes up to 63 entries: keep count within that
    if (count > 63) mono = 0;
    c->monotone = mono;
}

// interpolate() of contrast_curve_generate.comp:28-31
__device__ __forceinline__ float interpolate(float from, float to, float percent) {
    float difference = to - from;
    return from + (difference * percent);
}

// generateCurve(): contrast_curve_generate.comp:39-54 (steps 11) / gradation_curve_generate.comp:30-46 (steps 10)
__device__ __forceinline__ void generate_curve(DevCurve* c, uint32_t& n, float sx, float sy, float mx, float my,
                                               float ex, float ey, uint32_t steps) {
    for (uint32_t i = 0; i < steps; i++) {
        float t = (float)i / 10.0f;
        float xa = interpolate(sx, mx, t);
        float ya = interpolate(sy, my, t);
        float xb = interpolate(mx, ex, t);
        float yb = interpolate(my, ey, t);
        c->x[n] = interpolate(xa, xb, t);
        c->y[n] = interpolate(ya, yb, t);
        n++;
    }
}

// One point of generateCurve() (same arithmetic as generate_curve above): step k of the quadratic Bezier
// (s, m, e), t = k / 10.
__device__ __forceinline__ void bezier_point(float sx, float sy, float mx, float my, float ex, float ey, uint32_t k, float& x, float& y) {
    const float t = (float)k / 10.0f;
    const float xa = interpolate(sx, mx, t);
    const float ya = interpolate(sy, my, t);
    const float xb = interpolate(mx, ex, t);
    const float yb = interpolate(my, ey, t);
    x = interpolate(xa, xb, t);
    y = interpolate(ya, yb, t);
}

// Block-parallel curve_finish(): threads 0 .. kCurveCap-1 hold point i in (x, y) (zeros at i >= count) and
// have published them in sx / sy (LDS); fills slopes + monotone flag and writes the curve. Must be called by
// every thread of the block (contains barriers).
__device__ __forceinline__ void curve_store_parallel(DevCurve* c, const float* sx, const float* sy, int* s_mono, int count, float t0, float ta, float t1) {
    const int i = threadIdx.x;
    if (i == 0) *s_mono = 1;
    __syncthreads();
    if (i < kCurveCap) {
        float m = 0.0f;
        if (i + 1 < count) {
            m = (sy[i + 1] - sy[i]) / (sx[i + 1] - sx[i]);   // linearFunction slope, contrast_curve_apply.comp:22-25
            if (!(sx[i] <= sx[i + 1])) *s_mono = 0;
        }
        if (i == 0 && count > 0 && !(sx[0] == sx[0])) *s_mono = 0;
        c->x[i] = sx[i];
        c->y[i] = sy[i];
        c->m[i] = m;
    }
    __syncthreads();
    if (i == 0) {
        c->count = (uint32_t)count;
        c->monotone = (*s_mono && count <= 63) ? 1u : 0u;
        c->t0 = t0; c->ta = ta; c->t1 = t1;
        c->pad = 0;
    }
}

// img_relevant.comp:28-64. `c` is cnr * 256 already. pow(r, 5.0) is restated as ((r*r)*(r*r))*r (oracle Q5).
__device__ __forceinline__ float relevant_of(float pixel, float c, uint32_t x, uint32_t y, uint32_t N) {
    const uint32_t border = 100u, lim = N - border;  // uint arithmetic as in the shader (wraps for N < 100)
    const bool inside = x > border && x < lim && y > border && y < lim;
    if (!inside) return 0.0f;
    if (c >= 1.0f && c <= 6.0f) {
        const float r = c / 6.0f;
        return ((r * r) * (r * r)) * r;
    }
    if (c >= 6.0f && c <= kMaxCnrValue && pixel <= 0.90f) return 1.0f;
    return 0.0f;
}

__device__ __forceinline__ float cnr_at(const float* __restrict__ cnr, int cnrS, int cnrPitch, int scale, int x, int y) {
    const int cx = x / scale, cy = y / scale;
    return ((cx < cnrS && cy < cnrS) ? cnr[(size_t)cy * cnrPitch + cx] : 0.0f) * kMaxCnrValue;
}


}  // namespace musica
