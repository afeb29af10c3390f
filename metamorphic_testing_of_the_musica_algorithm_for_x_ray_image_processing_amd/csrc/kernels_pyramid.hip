// kernels_pyramid.hip — the Laplacian-pyramid kernels of the MUSICA path for gfx950 (CDNA4).
//
//   k_reduce_*  : img_smooth.comp + img_downsample.comp fused            (K5 + K6, the metric kernel)
//   k_band_*    : img_upsample.comp + img_smooth_upsampled.comp + img_difference.comp fused (K7 + K8 + K9)
//   k_expand_*  : contrast_curve_apply.comp + noise_reduction.comp + img_upsample.comp +
//                 img_smooth_upsampled.comp + img_addition.comp fused     (K14 + K16 + K7 + K8 + K17)
//
// Streaming ("fast") form, used when the level side S is a multiple of 8: one 64-lane wavefront
// owns a strip of 512 fine columns (8 per lane, two 16-byte loads per row) and marches down a
// segment of rows keeping the 5-row window in registers; the vertical 5-tap pass runs straight
// from the loaded registers, the horizontal pass takes its 2+1 neighbour columns from the
// adjacent lanes with DPP wave shifts (no LDS, no barrier) and from two predicated halo loads at
// the strip edges. Every HBM byte is touched by exactly one 16-byte access of one lane, plus the
// 3-row / 3-column halos that stay in the XCD's L2. Reflect-101 borders are register selects.
//
// Generic form (any S, including the 1..7-pixel tail of the reference's 12-level pyramid):
// one thread per output texel with the shaders' mirror()/out-of-bounds rules applied per tap.
//
// Both forms evaluate exactly the oracle's MUSICA_ORDER_FAST arithmetic.
#include "kernels_common.h"
#include "launchers.h"

namespace musica {

// ======================================================================================
// K5 + K6: out(xo, yo) = sum_m w[m] * ( sum_n w[n] * in(mirror(2xo+m-2), mirror(2yo+n-2)) )
// ======================================================================================

struct RowR {
    float v[8];
    float hl0, hl1;  // columns c0-2, c0-1 of the strip (lane 0 only)
    float hr;        // column c0+512 of the strip (lane 63 only)
};

struct LaneCfg {
    int c;            // first fine column of this lane
    bool active;      // c < S
    bool left_load;   // lane 0 of a strip that is not the first: halo comes from memory
    bool left_mirror; // lane 0 of the first strip: columns -2, -1 mirror onto 2, 1
    bool last_active; // the lane holding column S-1: column S mirrors onto S-2
    bool right_load;  // lane 63 with more image to its right
    bool lane0, lane63;
};

__device__ __forceinline__ LaneCfg make_cfg(int strip, int lane, int S) {
    LaneCfg g;
    const int c0 = strip * kStripCols;
    g.c = c0 + lane * kLaneCols;
    g.active = g.c < S;
    g.lane0 = lane == 0;
    g.lane63 = lane == 63;
    g.left_load = g.lane0 && c0 > 0;
    g.left_mirror = g.lane0 && c0 == 0;
    g.last_active = g.active && (g.c + kLaneCols >= S);
    g.right_load = g.lane63 && (g.c + kLaneCols < S);
    return g;
}

__device__ __forceinline__ void load_row(RowR& r, const float* __restrict__ base, int pitch, int row, const LaneCfg& g) {
    const float* p = base + (size_t)row * pitch;
    if (g.active) {
        float4 a = *reinterpret_cast<const float4*>(p + g.c);
        float4 b = *reinterpret_cast<const float4*>(p + g.c + 4);
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
        r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    }
    if (g.left_load) {
        float2 h = *reinterpret_cast<const float2*>(p + g.c - 2);
        r.hl0 = h.x; r.hl1 = h.y;
    }
    if (g.right_load) r.hr = p[g.c + kLaneCols];
}

__device__ __forceinline__ void zero_row(RowR& r) {
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = 0.f;
    r.hl0 = r.hl1 = r.hr = 0.f;
}

// grid: x = strips, y = ceil(segments / 4), z = batch. One wavefront = one (strip, segment).
__global__ __launch_bounds__(kBlockThreads) void k_reduce_fast(const float* __restrict__ in, float* __restrict__ out,
                                                               int S, int pitch, size_t in_plane, int So, int opitch,
                                                               size_t out_plane, int rows_per_wave) {
    const int lane = threadIdx.x & 63;
    const int seg = blockIdx.y * kWavesPerBlock + (threadIdx.x >> 6);
    const int yo0 = seg * rows_per_wave;
    if (yo0 >= So) return;  // wave-uniform
    const int yo1 = min(yo0 + rows_per_wave, So);
    in += (size_t)blockIdx.z * in_plane;
    out += (size_t)blockIdx.z * out_plane;
    const LaneCfg g = make_cfg(blockIdx.x, lane, S);
    const int hi = S - 1;

    RowR r0, r1, r2, r3, r4, n3, n4;
    zero_row(r0); zero_row(r1); zero_row(r2); zero_row(r3); zero_row(r4); zero_row(n3); zero_row(n4);
    load_row(r0, in, pitch, mirror_idx(2 * yo0 - 2, hi), g);
    load_row(r1, in, pitch, mirror_idx(2 * yo0 - 1, hi), g);
    load_row(r2, in, pitch, 2 * yo0, g);
    load_row(r3, in, pitch, mirror_idx(2 * yo0 + 1, hi), g);
    load_row(r4, in, pitch, mirror_idx(2 * yo0 + 2, hi), g);

    for (int yo = yo0; yo < yo1; yo++) {
        if (yo + 1 < yo1) {  // prefetch the two new rows of the next output row
            load_row(n3, in, pitch, mirror_idx(2 * yo + 3, hi), g);
            load_row(n4, in, pitch, mirror_idx(2 * yo + 4, hi), g);
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = chain5(r0.v[j], r1.v[j], r2.v[j], r3.v[j], r4.v[j]);
        const float vh0 = chain5(r0.hl0, r1.hl0, r2.hl0, r3.hl0, r4.hl0);
        const float vh1 = chain5(r0.hl1, r1.hl1, r2.hl1, r3.hl1, r4.hl1);
        const float vhr = chain5(r0.hr, r1.hr, r2.hr, r3.hr, r4.hr);
        float vl6 = from_left_lane(v[6]);
        float vl7 = from_left_lane(v[7]);
        float vr0 = from_right_lane(v[0]);
        if (g.lane0) {
            vl6 = g.left_mirror ? v[2] : vh0;  // column -2 -> 2, -1 -> 1 (img_smooth.comp:13)
            vl7 = g.left_mirror ? v[1] : vh1;
        }
        if (g.last_active) vr0 = v[6];         // column S -> S-2 (img_smooth.comp:12)
        else if (g.lane63) vr0 = vhr;
        if (g.active) {
            float4 o;
            o.x = chain5(vl6, vl7, v[0], v[1], v[2]);
            o.y = chain5(v[0], v[1], v[2], v[3], v[4]);
            o.z = chain5(v[2], v[3], v[4], v[5], v[6]);
            o.w = chain5(v[4], v[5], v[6], v[7], vr0);
            *reinterpret_cast<float4*>(out + (size_t)yo * opitch + (g.c >> 1)) = o;
        }
        r0 = r2; r1 = r3; r2 = r4; r3 = n3; r4 = n4;
    }
}

// One thread per output texel, any S >= 1. OOB loads (only reachable when S < 3, where the
// reference's unclamped mirror() leaves the image) return 0.
__device__ __forceinline__ float ld0(const float* __restrict__ im, int pitch, int S, int x, int y) {
    return (x >= 0 && y >= 0 && x < S && y < S) ? im[(size_t)y * pitch + x] : 0.0f;
}

__global__ void k_reduce_generic(const float* __restrict__ in, float* __restrict__ out, int S, int pitch,
                                 size_t in_plane, int So, int opitch, size_t out_plane) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x;
    const int yo = blockIdx.y * blockDim.y + threadIdx.y;
    if (xo >= So || yo >= So) return;
    in += (size_t)blockIdx.z * in_plane;
    out += (size_t)blockIdx.z * out_plane;
    const int hi = S - 1;
    float v[5];
#pragma unroll
    for (int m = 0; m < 5; m++) {
        const int x = mirror_idx(2 * xo + m - 2, hi);
        if (x < 0 || x >= S) { v[m] = 0.0f; continue; }
        v[m] = chain5(ld0(in, pitch, S, x, mirror_idx(2 * yo - 2, hi)), ld0(in, pitch, S, x, mirror_idx(2 * yo - 1, hi)),
                      ld0(in, pitch, S, x, mirror_idx(2 * yo, hi)), ld0(in, pitch, S, x, mirror_idx(2 * yo + 1, hi)),
                      ld0(in, pitch, S, x, mirror_idx(2 * yo + 2, hi)));
    }
    out[(size_t)yo * opitch + xo] = chain5(v[0], v[1], v[2], v[3], v[4]);
}

// ======================================================================================
// Zero-inserted upsample + x4 smooth (K7 + K8) of a coarse image, shared by band and expand.
// Fine (x, y) on the S grid; coarse c on the Sc = ceil(S/2) grid.
//   V(j, y) : vertical pass on coarse column j      even y = 2k: (w0*c[km1] + w2*c[k]) + w4*c[kp1]
//                                                   odd  y     :  w1*c[k] + w3*c[kp1]
//   H(x, y) : horizontal pass on V                  even x = 2j: (w0*V[jm1] + w2*V[j]) + w4*V[jp1]
//                                                   odd  x     :  w1*V[j] + w3*V[jp1]
//   lowpass = 4 * H
// km1 / kp1 follow the reflect-101 mirror on the FINE grid (img_smooth_upsampled.comp:10-16).
// ======================================================================================

// Coarse index hit by fine tap index f after mirroring, or -1 when the tap reads a zero
// (odd texel, Q2) or leaves the image (Q1).
__device__ __forceinline__ int coarse_of_fine(int f, int S) {
    const int m = mirror_idx(f, S - 1);
    if (m < 0 || m >= S || (m & 1)) return -1;
    return m >> 1;
}

struct CRow {
    float v[4];  // coarse columns j0 .. j0+3
    float hl;    // coarse column j0-1 (lane 0 of a strip that is not the first)
    float hr;    // coarse column j0+4 (lane 63 with more image to its right)
};

__device__ __forceinline__ void load_crow(CRow& r, const float* __restrict__ base, int pitch, int row, const LaneCfg& g) {
    const float* p = base + (size_t)row * pitch;
    const int j0 = g.c >> 1;
    if (g.active) {
        float4 a = *reinterpret_cast<const float4*>(p + j0);
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    }
    if (g.left_load) r.hl = p[j0 - 1];
    if (g.right_load) r.hr = p[j0 + 4];
}

__device__ __forceinline__ void zero_crow(CRow& r) {
    r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0.f;
    r.hl = r.hr = 0.f;
}

// Horizontal pass for the 8 fine columns of a lane from its 4 coarse V values + neighbours; returns 4 * H.
__device__ __forceinline__ void hpass8(const float V[4], float Vl, float Vr, float low[8]) {
    low[0] = 4.0f * chain_even(Vl, V[0], V[1]);
    low[1] = 4.0f * chain_odd(V[0], V[1]);
    low[2] = 4.0f * chain_even(V[0], V[1], V[2]);
    low[3] = 4.0f * chain_odd(V[1], V[2]);
    low[4] = 4.0f * chain_even(V[1], V[2], V[3]);
    low[5] = 4.0f * chain_odd(V[2], V[3]);
    low[6] = 4.0f * chain_even(V[2], V[3], Vr);
    low[7] = 4.0f * chain_odd(V[3], Vr);
}

// Neighbour exchange for one row of V values (S % 8 == 0, so S is even and the fine column S
// mirrors onto S-2 = coarse j0+3 of the last active lane; fine column -2 mirrors onto coarse 1).
__device__ __forceinline__ void exchange(const float V[4], float Vhl, float Vhr, const LaneCfg& g, float& Vl, float& Vr) {
    Vl = from_left_lane(V[3]);
    Vr = from_right_lane(V[0]);
    if (g.lane0) Vl = g.left_mirror ? V[1] : Vhl;
    if (g.last_active) Vr = V[3];
    else if (g.lane63) Vr = Vhr;
}

// Lowpass rows 2k and 2k+1 (8 fine columns each) from coarse rows a = km1, b = k, c = kp1.
__device__ __forceinline__ void lowpass_pair(const CRow& a, const CRow& b, const CRow& c, const LaneCfg& g,
                                             float lowE[8], float lowO[8]) {
    float Ve[4], Vo[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        Ve[j] = chain_even(a.v[j], b.v[j], c.v[j]);
        Vo[j] = chain_odd(b.v[j], c.v[j]);
    }
    const float Vehl = chain_even(a.hl, b.hl, c.hl), Vohl = chain_odd(b.hl, c.hl);
    const float Vehr = chain_even(a.hr, b.hr, c.hr), Vohr = chain_odd(b.hr, c.hr);
    float l, r;
    exchange(Ve, Vehl, Vehr, g, l, r);
    hpass8(Ve, l, r, lowE);
    exchange(Vo, Vohl, Vohr, g, l, r);
    hpass8(Vo, l, r, lowO);
}

__device__ __forceinline__ void load8(float d[8], const float* __restrict__ p) {
    float4 a = *reinterpret_cast<const float4*>(p);
    float4 b = *reinterpret_cast<const float4*>(p + 4);
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
}
__device__ __forceinline__ void store8(float* __restrict__ p, const float d[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(d[0], d[1], d[2], d[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(d[4], d[5], d[6], d[7]);
}

// K7 + K8 + K9: band = fine - lowpass(coarse).  rows_per_wave counts COARSE rows (2 fine rows each).
__global__ __launch_bounds__(kBlockThreads) void k_band_fast(const float* __restrict__ fine, const float* __restrict__ coarse,
                                                             float* __restrict__ band, int S, int pitch, size_t plane,
                                                             int Sc, int cpitch, size_t cplane, int rows_per_wave) {
    const int lane = threadIdx.x & 63;
    const int seg = blockIdx.y * kWavesPerBlock + (threadIdx.x >> 6);
    const int k0 = seg * rows_per_wave;
    if (k0 >= Sc) return;
    const int k1 = min(k0 + rows_per_wave, Sc);
    fine += (size_t)blockIdx.z * plane;
    band += (size_t)blockIdx.z * plane;
    coarse += (size_t)blockIdx.z * cplane;
    const LaneCfg g = make_cfg(blockIdx.x, lane, S);

    CRow ca, cb, cc, cn;
    zero_crow(ca); zero_crow(cb); zero_crow(cc); zero_crow(cn);
    load_crow(ca, coarse, cpitch, coarse_of_fine(2 * k0 - 2, S), g);
    load_crow(cb, coarse, cpitch, k0, g);
    load_crow(cc, coarse, cpitch, coarse_of_fine(2 * k0 + 2, S), g);
    for (int k = k0; k < k1; k++) {
        if (k + 1 < k1) load_crow(cn, coarse, cpitch, coarse_of_fine(2 * k + 4, S), g);
        float fe[8], fo[8];
        if (g.active) {
            load8(fe, fine + (size_t)(2 * k) * pitch + g.c);
            load8(fo, fine + (size_t)(2 * k + 1) * pitch + g.c);
        }
        float lowE[8], lowO[8];
        lowpass_pair(ca, cb, cc, g, lowE, lowO);
        if (g.active) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                fe[j] = fe[j] - lowE[j];   // img_difference.comp:15
                fo[j] = fo[j] - lowO[j];
            }
            store8(band + (size_t)(2 * k) * pitch + g.c, fe);
            store8(band + (size_t)(2 * k + 1) * pitch + g.c, fo);
        }
        ca = cb; cb = cc; cc = cn;
    }
}

// lowpass value at fine (x, y) for any S (generic form).
__device__ __forceinline__ float lowpass_generic(const float* __restrict__ coarse, int cpitch, int Sc, int S, int x, int y) {
    float V[5];
#pragma unroll
    for (int m = 0; m < 5; m++) {
        const int j = coarse_of_fine(x + m - 2, S);
        float acc = 0.0f;
        bool first = true;
        if (j >= 0) {
#pragma unroll
            for (int n = 0; n < 5; n++) {
                const int k = coarse_of_fine(y + n - 2, S);
                const float w = n == 0 ? W0 : n == 1 ? W1 : n == 2 ? W2 : n == 3 ? W3 : W4;
                const float t = w * (k >= 0 ? ld0(coarse, cpitch, Sc, j, k) : 0.0f);
                acc = first ? t : acc + t;
                first = false;
            }
        }
        V[m] = acc;
    }
    return 4.0f * chain5(V[0], V[1], V[2], V[3], V[4]);
}

__global__ void k_band_generic(const float* __restrict__ fine, const float* __restrict__ coarse, float* __restrict__ band,
                               int S, int pitch, size_t plane, int Sc, int cpitch, size_t cplane) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= S || y >= S) return;
    fine += (size_t)blockIdx.z * plane;
    band += (size_t)blockIdx.z * plane;
    coarse += (size_t)blockIdx.z * cplane;
    const float low = lowpass_generic(coarse, cpitch, Sc, S, x, y);
    band[(size_t)y * pitch + x] = fine[(size_t)y * pitch + x] - low;
}

// lowpass only (debugProcess' red_lowpass_i / exp_lowpass_i dumps, kernel-level tests)
__global__ void k_lowpass_generic(const float* __restrict__ coarse, float* __restrict__ low, int S, int pitch, size_t plane,
                                  int Sc, int cpitch, size_t cplane) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= S || y >= S) return;
    low += (size_t)blockIdx.z * plane;
    coarse += (size_t)blockIdx.z * cplane;
    low[(size_t)y * pitch + x] = lowpass_generic(coarse, cpitch, Sc, S, x, y);
}

// ======================================================================================
// K14 + K16 + K7 + K8 + K17: recon = lowpass(prev) + band * gain(sdev) [* nr(cnr)]
// ======================================================================================

// linearFunction() of noise_reduction.comp:24-31 — m * x, not m * (x - p1.x).
__device__ __forceinline__ float nr_factor(float c, float lowCnr, float lowFactor, float highCnr, float highFactor) {
    if (c < lowCnr) return lowFactor;
    else if (c > highCnr) return highFactor;
    const float m = (highFactor - lowFactor) / (highCnr - lowCnr);
    return m * c + lowFactor;
}

template <int GAIN>
__device__ __forceinline__ float gain_of(float s, float high, const CurveLds& t) {
    if (GAIN == GAIN_CONST) return high;
    if (GAIN == GAIN_RANGE) {
        if (s == 0.0f) return high;                     // points[0].x == x
        if (s >= 0.0f && s <= 1.0f) return 0.0f * s + high;  // m = (high - high) / (1 - 0) = 0
        return 0.0f;
    }
    return curve_eval(t, s);
}

template <int GAIN, bool NR>
__global__ __launch_bounds__(kBlockThreads) void k_expand_fast(ExpandArgs a) {
    __shared__ CurveLds tab;
    const int img = blockIdx.z;
    if (GAIN == GAIN_CURVE) {
        curve_to_lds(tab, a.curves + (size_t)img * a.curve_stride);
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int seg = blockIdx.y * kWavesPerBlock + (threadIdx.x >> 6);
    const int k0 = seg * a.rows_per_wave;
    if (k0 >= a.Sc) return;
    const int k1 = min(k0 + a.rows_per_wave, a.Sc);
    const float* band = a.band + (size_t)img * a.plane;
    const float* sdev = (GAIN != GAIN_CONST) ? a.sdev + (size_t)img * a.plane : nullptr;
    float* recon = a.recon + (size_t)img * a.plane;
    const float* prev = a.prev + (size_t)img * a.cplane;
    const float* cnr = NR ? a.cnr + (size_t)img * a.cnrPlane : nullptr;
    const int S = a.S, pitch = a.pitch, cpitch = a.cpitch;
    const LaneCfg g = make_cfg(blockIdx.x, lane, S);

    CRow ca, cb, cc, cn;
    zero_crow(ca); zero_crow(cb); zero_crow(cc); zero_crow(cn);
    load_crow(ca, prev, cpitch, coarse_of_fine(2 * k0 - 2, S), g);
    load_crow(cb, prev, cpitch, k0, g);
    load_crow(cc, prev, cpitch, coarse_of_fine(2 * k0 + 2, S), g);
    for (int k = k0; k < k1; k++) {
        if (k + 1 < k1) load_crow(cn, prev, cpitch, coarse_of_fine(2 * k + 4, S), g);
        float be[8], bo[8], se[8], so[8];
        if (g.active) {
            load8(be, band + (size_t)(2 * k) * pitch + g.c);
            load8(bo, band + (size_t)(2 * k + 1) * pitch + g.c);
            if (GAIN != GAIN_CONST) {
                load8(se, sdev + (size_t)(2 * k) * pitch + g.c);
                load8(so, sdev + (size_t)(2 * k + 1) * pitch + g.c);
            }
        }
        float lowE[8], lowO[8];
        lowpass_pair(ca, cb, cc, g, lowE, lowO);
        if (g.active) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float pe = be[j] * gain_of<GAIN>(GAIN != GAIN_CONST ? se[j] : 0.0f, a.high, tab);  // contrast_curve_apply.comp:61
                float po = bo[j] * gain_of<GAIN>(GAIN != GAIN_CONST ? so[j] : 0.0f, a.high, tab);
                if (NR) {
                    const int cx = (g.c + j) / a.cnrScale;                                       // noise_reduction.comp:39-45
                    const float ce = cnr[(size_t)((2 * k) / a.cnrScale) * a.cnrPitch + cx] * kMaxCnrValue;
                    const float co = cnr[(size_t)((2 * k + 1) / a.cnrScale) * a.cnrPitch + cx] * kMaxCnrValue;
                    pe = pe * nr_factor(ce, a.lowCnr, a.lowFactor, a.highCnr, a.highFactor);     // noise_reduction.comp:57
                    po = po * nr_factor(co, a.lowCnr, a.lowFactor, a.highCnr, a.highFactor);
                }
                be[j] = lowE[j] + pe;   // img_addition.comp:15
                bo[j] = lowO[j] + po;
            }
            store8(recon + (size_t)(2 * k) * pitch + g.c, be);
            store8(recon + (size_t)(2 * k + 1) * pitch + g.c, bo);
        }
        ca = cb; cb = cc; cc = cn;
    }
}

template <int GAIN, bool NR>
__global__ void k_expand_generic(ExpandArgs a) {
    __shared__ CurveLds tab;
    const int img = blockIdx.z;
    if (GAIN == GAIN_CURVE) {
        const int tid = threadIdx.y * blockDim.x + threadIdx.x;
        const DevCurve* src = a.curves + (size_t)img * a.curve_stride;
        for (int i = tid; i < kCurveCap; i += blockDim.x * blockDim.y) {
            tab.x[i] = src->x[i]; tab.y[i] = src->y[i]; tab.m[i] = src->m[i];
        }
        if (tid == 0) { tab.count = src->count; tab.monotone = src->monotone; }
        __syncthreads();
    }
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= a.S || y >= a.S) return;
    const size_t o = (size_t)img * a.plane + (size_t)y * a.pitch + x;
    const float low = lowpass_generic(a.prev + (size_t)img * a.cplane, a.cpitch, a.Sc, a.S, x, y);
    const float s = (GAIN != GAIN_CONST) ? a.sdev[o] : 0.0f;
    float p = a.band[o] * gain_of<GAIN>(s, a.high, tab);
    if (NR) {
        const int cx = x / a.cnrScale, cy = y / a.cnrScale;
        const float c = ((cx < a.cnrS && cy < a.cnrS) ? a.cnr[(size_t)img * a.cnrPlane + (size_t)cy * a.cnrPitch + cx] : 0.0f) * kMaxCnrValue;
        p = p * nr_factor(c, a.lowCnr, a.lowFactor, a.highCnr, a.highFactor);
    }
    a.recon[o] = low + p;
}

// band after contrast curve (+ noise reduction): what img_addition.comp reads as inputImageB
// (debugProcess' exp_bandpass_i dump; not materialised on the hot path).
template <int GAIN, bool NR>
__global__ void k_exp_band_generic(ExpandArgs a) {
    __shared__ CurveLds tab;
    const int img = blockIdx.z;
    if (GAIN == GAIN_CURVE) {
        const int tid = threadIdx.y * blockDim.x + threadIdx.x;
        const DevCurve* src = a.curves + (size_t)img * a.curve_stride;
        for (int i = tid; i < kCurveCap; i += blockDim.x * blockDim.y) {
            tab.x[i] = src->x[i]; tab.y[i] = src->y[i]; tab.m[i] = src->m[i];
        }
        if (tid == 0) { tab.count = src->count; tab.monotone = src->monotone; }
        __syncthreads();
    }
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= a.S || y >= a.S) return;
    const size_t o = (size_t)img * a.plane + (size_t)y * a.pitch + x;
    const float s = (GAIN != GAIN_CONST) ? a.sdev[o] : 0.0f;
    float p = a.band[o] * gain_of<GAIN>(s, a.high, tab);
    if (NR) {
        const int cx = x / a.cnrScale, cy = y / a.cnrScale;
        const float c = ((cx < a.cnrS && cy < a.cnrS) ? a.cnr[(size_t)img * a.cnrPlane + (size_t)cy * a.cnrPitch + cx] : 0.0f) * kMaxCnrValue;
        p = p * nr_factor(c, a.lowCnr, a.lowFactor, a.highCnr, a.highFactor);
    }
    a.recon[o] = p;
}

// ======================================================================================
// host-side launchers
// ======================================================================================

static inline dim3 stream_grid(int S, int rows, int rows_per_wave, int batch) {
    const int strips = (S + kStripCols - 1) / kStripCols;
    const int segs = (rows + rows_per_wave - 1) / rows_per_wave;
    return dim3(strips, (segs + kWavesPerBlock - 1) / kWavesPerBlock, batch);
}
static inline dim3 generic_grid(int S, int batch) { return dim3((S + 31) / 32, (S + 7) / 8, batch); }
static const dim3 kGenericBlock(32, 8, 1);

static inline bool fast_ok(int S) { return S >= 8 && (S % 8) == 0; }

void launch_reduce(hipStream_t st, const float* in, const LevelDesc& li, float* out, const LevelDesc& lo, int batch,
                   int rows_per_wave, bool force_generic) {
    if (fast_ok(li.S) && !force_generic) {
        hipLaunchKernelGGL(k_reduce_fast, stream_grid(li.S, lo.S, rows_per_wave, batch), dim3(kBlockThreads), 0, st, in, out,
                           li.S, li.pitch, li.plane, lo.S, lo.pitch, lo.plane, rows_per_wave);
    } else {
        hipLaunchKernelGGL(k_reduce_generic, generic_grid(lo.S, batch), kGenericBlock, 0, st, in, out, li.S, li.pitch,
                           li.plane, lo.S, lo.pitch, lo.plane);
    }
}

void launch_band(hipStream_t st, const float* fine, const float* coarse, float* band, const LevelDesc& lf, const LevelDesc& lc,
                 int batch, int rows_per_wave, bool force_generic) {
    if (fast_ok(lf.S) && !force_generic) {
        hipLaunchKernelGGL(k_band_fast, stream_grid(lf.S, lc.S, rows_per_wave, batch), dim3(kBlockThreads), 0, st, fine, coarse,
                           band, lf.S, lf.pitch, lf.plane, lc.S, lc.pitch, lc.plane, rows_per_wave);
    } else {
        hipLaunchKernelGGL(k_band_generic, generic_grid(lf.S, batch), kGenericBlock, 0, st, fine, coarse, band, lf.S, lf.pitch,
                           lf.plane, lc.S, lc.pitch, lc.plane);
    }
}

void launch_lowpass(hipStream_t st, const float* coarse, float* low, const LevelDesc& lf, const LevelDesc& lc, int batch) {
    hipLaunchKernelGGL(k_lowpass_generic, generic_grid(lf.S, batch), kGenericBlock, 0, st, coarse, low, lf.S, lf.pitch, lf.plane,
                       lc.S, lc.pitch, lc.plane);
}

template <int GAIN, bool NR>
static void launch_expand_t(hipStream_t st, const ExpandArgs& a, int batch, bool force_generic) {
    if (fast_ok(a.S) && !force_generic) {
        hipLaunchKernelGGL((k_expand_fast<GAIN, NR>), stream_grid(a.S, a.Sc, a.rows_per_wave, batch), dim3(kBlockThreads), 0, st, a);
    } else {
        hipLaunchKernelGGL((k_expand_generic<GAIN, NR>), generic_grid(a.S, batch), kGenericBlock, 0, st, a);
    }
}

void launch_expand(hipStream_t st, const ExpandArgs& a, int gain_mode, bool nr, int batch, bool force_generic) {
    if (gain_mode == GAIN_CONST) launch_expand_t<GAIN_CONST, false>(st, a, batch, force_generic);
    else if (gain_mode == GAIN_RANGE) launch_expand_t<GAIN_RANGE, false>(st, a, batch, force_generic);
    else if (nr) launch_expand_t<GAIN_CURVE, true>(st, a, batch, force_generic);
    else launch_expand_t<GAIN_CURVE, false>(st, a, batch, force_generic);
}

void launch_exp_band(hipStream_t st, const ExpandArgs& a, int gain_mode, bool nr, int batch) {
    const dim3 g = generic_grid(a.S, batch);
    if (gain_mode == GAIN_CONST) hipLaunchKernelGGL((k_exp_band_generic<GAIN_CONST, false>), g, kGenericBlock, 0, st, a);
    else if (gain_mode == GAIN_RANGE) hipLaunchKernelGGL((k_exp_band_generic<GAIN_RANGE, false>), g, kGenericBlock, 0, st, a);
    else if (nr) hipLaunchKernelGGL((k_exp_band_generic<GAIN_CURVE, true>), g, kGenericBlock, 0, st, a);
    else hipLaunchKernelGGL((k_exp_band_generic<GAIN_CURVE, false>), g, kGenericBlock, 0, st, a);
}

}  // namespace musica
