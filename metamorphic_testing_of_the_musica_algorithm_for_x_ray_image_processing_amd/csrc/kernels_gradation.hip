// kernels_gradation.hip — gradation stage of the MUSICA path (gfx950).
//
//   k_grad_hist   : img_relevant.comp + gradation_histogram.comp fused        (K18 + K19)
//   k_grad_curve  : img_histogram_max.comp + gradation_curve_generate.comp    (K12 + K20)
//   k_grad_apply  : img_apply_gradation_curve.comp                            (K21)
//   k_relevant    : img_relevant.comp alone (debug image, CLAHE input)
#include <algorithm>
#include "kernels_common.h"
#include "launchers.h"
#include "grad_parts.h"

namespace musica {

// gradation_histogram.comp:14-34 — the reference's thread (gx, gy) walks its 16x16 area column
// by column (m = x outer, n = y inner) and `return`s at the first pixel that is exactly 0, so a
// pixel at scan position m*16+n counts iff it precedes the area's first zero. Here a lane owns 4
// columns of a 16-row group (one 16-byte load per row), finds its columns' first zeros, the 4 lanes
// of an area combine them with two DPP/shuffle steps, and every pixel before that position is
// binned with weight uint(relevant * 100) into an LDS-private histogram.
// SHARED8: the cnr scale is a multiple of 8 (always the case for N >= 57: scale = ceil(N / ceil(N/8)) = 8), so
// the 4 columns of a lane and each half (8 rows) of a 16-row group sit under ONE cnr texel: two cnr loads
// and two classifications per lane and group instead of 64.
// RAW: `normalized <= 0.9` (img_relevant.comp:56) is tested on the raw uint16 pixel against the per-image
// threshold norm_threshold_090() — exact, because the normalisation is monotone — so the kernel reads
// 2 B/px of raw input instead of a stored 4 B/px normalized image.
// The body works for any workgroup of whole wavefronts: blockDim.x / 64 wavefronts side by side (256 columns each).
template <bool SHARED8, bool RAW>
__device__ __forceinline__ void grad_hist_body(const GradArgs& a, int img, uint32_t* lh /* [MUSICA_GRAD_BINS + 64] */, int& s_thr) {
    const int waves_per_block = (int)(blockDim.x >> 6);
    for (int i = threadIdx.x; i < MUSICA_GRAD_BINS + 64; i += blockDim.x) lh[i] = 0u;
    if (RAW && threadIdx.x == 0) {
        float minv, maxv;
        chain_scalars(a.minmax, img, a.min_chain_exact, minv, maxv);
        s_thr = norm_threshold_090(minv, maxv - minv);
    }
    __syncthreads();
    const int thr = RAW ? s_thr : 0;
    const Buf ib = make_buf(a.img + (size_t)img * a.plane, a.plane * 4);
    const Buf nb = RAW ? make_buf(a.raw + (size_t)img * a.N * a.N, (size_t)a.N * a.N * 2)
                       : make_buf(a.normalized + (size_t)img * a.plane, a.plane * 4);
    const Buf cb = make_buf(a.cnr + (size_t)img * a.cnrPlane, a.cnrPlane * 4);
    const int lane = threadIdx.x & 63;
    const int N = a.N;
    // the four wavefronts of a workgroup sit side by side (1024 columns x 16 rows per group): a workgroup reads 4 KiB
    // of every row it touches instead of four separate 1 KiB pieces
    const int c = (blockIdx.x * waves_per_block + (threadIdx.x >> 6)) * 256 + lane * 4;
    const int valid = min(max(N - c, 0), 4);                  // in-image columns among the lane's 4
    const uint32_t coff = c < N ? (uint32_t)c * 4u : kOob;
    const uint32_t rb = (uint32_t)a.pitch * 4u, crb = (uint32_t)a.cnrPitch * 4u;
    const uint32_t ucoff = c < N ? (uint32_t)c * 2u : kOob, urb = (uint32_t)N * 2u;   // dense uint16 rows (RAW; N % 4 == 0)
    const int mbase = (lane & 3) * 4;
    const uint32_t border = 100u, lim = (uint32_t)N - border;  // uint arithmetic of img_relevant.comp:46-49
    bool colin[4];
#pragma unroll
    for (int j = 0; j < 4; j++) colin[j] = (uint32_t)(c + j) > border && (uint32_t)(c + j) < lim;
    // byte offset of the lane's cnr column (out of the cnr image reads 0, Q1)
    const int cx0 = c / a.cnrScale;
    const uint32_t cxoff = (c < N && cx0 < a.cnrS) ? (uint32_t)cx0 * 4u : kOob;
    const int g0 = blockIdx.y * a.groups_per_wave;
    for (int gi = 0; gi < a.groups_per_wave; gi++) {
        const int yb = (g0 + gi) * kHistArea;
        if (yb >= N) break;  // wave-uniform
        float4 v[kHistArea];
#pragma unroll
        for (int n = 0; n < kHistArea; n++) {
            const int y = yb + n;
            v[n] = bload4(ib, (y < N ? (uint32_t)y * rb : kOob) + coff);   // out of image reads 0 (Q1)
            if (valid < 4) {
                if (valid < 1) v[n].x = 0.f;
                if (valid < 2) v[n].y = 0.f;
                if (valid < 3) v[n].z = 0.f;
                v[n].w = 0.f;
            }
        }
        // first zero of each owned column, as a scan position m*16 + n (256 = none)
        int q = 256;
#pragma unroll
        for (int n = kHistArea - 1; n >= 0; n--) {
            if (v[n].x == 0.0f) q = min(q, (mbase + 0) * 16 + n);
            if (v[n].y == 0.0f) q = min(q, (mbase + 1) * 16 + n);
            if (v[n].z == 0.0f) q = min(q, (mbase + 2) * 16 + n);
            if (v[n].w == 0.0f) q = min(q, (mbase + 3) * 16 + n);
        }
        q = min(q, __shfl_xor(q, 1));
        q = min(q, __shfl_xor(q, 2));
        // two halves of 8 rows: the normalized rows of a half are loaded back to back, then binned
#pragma unroll
        for (int half = 0; half < 2; half++) {
            float4 pn[8];
            float2 pr[8];
#pragma unroll
            for (int n = 0; n < 8; n++) {
                const int y = yb + half * 8 + n;
                if (RAW) pr[n] = bload2(nb, (y < N ? (uint32_t)y * urb : kOob) + ucoff);
                else pn[n] = bload4(nb, (y < N ? (uint32_t)y * rb : kOob) + coff);
            }
            CnrClass kc;
            if (SHARED8) {
                const int cy = (yb + half * 8) / a.cnrScale;   // wave-uniform
                kc = classify_cnr(bload1(cb, (cy < a.cnrS ? (uint32_t)cy * crb : kOob) + cxoff) * kMaxCnrValue);
            }
#pragma unroll
            for (int n8 = 0; n8 < 8; n8++) {
                const int n = half * 8 + n8;
                const int y = yb + n;
                const bool rowin = (uint32_t)y > border && (uint32_t)y < lim;
                const float vv[4] = {v[n].x, v[n].y, v[n].z, v[n].w};
                bool le090[4];
                if (RAW) {
                    const uint32_t r0 = __float_as_uint(pr[n8].x), r1 = __float_as_uint(pr[n8].y);
                    le090[0] = (int)(r0 & 0xFFFFu) <= thr; le090[1] = (int)(r0 >> 16) <= thr;
                    le090[2] = (int)(r1 & 0xFFFFu) <= thr; le090[3] = (int)(r1 >> 16) <= thr;
                } else {
                    le090[0] = pn[n8].x <= 0.90f; le090[1] = pn[n8].y <= 0.90f; le090[2] = pn[n8].z <= 0.90f; le090[3] = pn[n8].w <= 0.90f;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (!SHARED8) {
                        const int cx = (c + j) / a.cnrScale, cy = y / a.cnrScale;
                        const uint32_t off = (c + j < N && y < N && cx < a.cnrS && cy < a.cnrS) ? (uint32_t)cy * crb + (uint32_t)cx * 4u : kOob;
                        kc = classify_cnr(bload1(cb, off) * kMaxCnrValue);
                    }
                    const float cur = vv[j];
                    const float scaled = cur * (float)MUSICA_GRAD_BINS;                // gradation_histogram.comp:26
                    // NaN never indexes (oracle Q6); bins outside [0, 1024) are dropped (Q1)
                    const bool inrange = scaled > -1.0f && scaled < (float)MUSICA_GRAD_BINS;   // int(scaled) in [0, 1023]
                    const int bin = inrange ? (int)scaled : 0;
                    const uint32_t w = (rowin && colin[j]) ? (kc.ramp ? kc.w_ramp : ((kc.high && le090[j]) ? 100u : 0u)) : 0u;  // :28-30
                    const bool add = inrange && ((mbase + j) * 16 + n < q) && w != 0u;
                    atomicAdd(&lh[add ? bin : MUSICA_GRAD_BINS + lane], add ? w : 0u);
                }
            }
        }
    }
    __syncthreads();
    uint32_t* gh = a.hist + (size_t)img * MUSICA_GRAD_BINS;
    for (int i = threadIdx.x; i < MUSICA_GRAD_BINS; i += blockDim.x) {
        const uint32_t v = lh[i];
        if (v) atomicAdd(&gh[i], v);
    }
}
template <bool SHARED8, bool RAW>
__global__ __launch_bounds__(kBlockThreads) void k_grad_hist(GradArgs a) {
    __shared__ uint32_t lh[MUSICA_GRAD_BINS + 64];  // + one scratch word per lane for the branch-free adds
    __shared__ int s_thr;
    const int img = blockIdx.z;
    if (a.only_if && a.only_if[img] == 0u) return;   // block-uniform: nothing to redo for this image
    grad_hist_body<SHARED8, RAW>(a, img, lh, s_thr);
}

// histogram from a stored relevant image (kernel-level parity tests, one thread per 16x16 area like the reference)
__global__ void k_grad_hist_ref(const float* __restrict__ img, const float* __restrict__ relevant, int N, int pitch, size_t plane,
                                uint32_t* __restrict__ hist) {
    const int gx = blockIdx.x * blockDim.x + threadIdx.x, gy = blockIdx.y * blockDim.y + threadIdx.y;
    const int bx = gx * kHistArea, by = gy * kHistArea;
    if (bx >= N || by >= N) return;
    img += (size_t)blockIdx.z * plane;
    relevant += (size_t)blockIdx.z * plane;
    uint32_t* gh = hist + (size_t)blockIdx.z * MUSICA_GRAD_BINS;
    for (int m = 0; m < kHistArea; m++)
        for (int n = 0; n < kHistArea; n++) {
            const int x = bx + m, y = by + n;
            const float cur = (x < N && y < N) ? img[(size_t)y * pitch + x] : 0.0f;
            if (cur == 0.0f) return;
            if (cur != cur) continue;
            const float scaled = cur * (float)MUSICA_GRAD_BINS;
            if (!(scaled > -2147483648.0f && scaled < 2147483648.0f)) continue;
            const int bin = (int)scaled;
            if (bin < 0 || bin >= MUSICA_GRAD_BINS) continue;
            const uint32_t w = f2u(relevant[(size_t)y * pitch + x] * 100.0f);
            if (w) atomicAdd(&gh[bin], w);
        }
}

__global__ void k_relevant(const float* __restrict__ normalized, const float* __restrict__ cnr, float* __restrict__ out, int N, int pitch,
                           size_t plane, int cnrS, int cnrPitch, size_t cnrPlane, int cnrScale) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= N || y >= N) return;
    const size_t o = (size_t)blockIdx.z * plane + (size_t)y * pitch + x;
    const float cc = cnr_at(cnr + (size_t)blockIdx.z * cnrPlane, cnrS, cnrPitch, cnrScale, x, y);
    out[o] = relevant_of(normalized[o], cc, (uint32_t)x, (uint32_t)y, (uint32_t)N);
}

// Four texels per thread (one 16-byte load / store) when the side and the cnr scale are multiples of 4: the four share a
// cnr texel, its row index is the workgroup's (blockIdx.y = y). Same relevant_of() per texel. RAW: `normalized <= 0.9` is
// tested as raw <= thr090[image] (norm_threshold_090: exact, the normalisation is monotone) on the raw uint16 pixels, so a
// context that normalises on the fly needs no stored normalized image for its CLAHE block either.
template <bool RAW>
__global__ __launch_bounds__(256) void k_relevant4(const float* __restrict__ normalized, const uint16_t* __restrict__ raw, const int* __restrict__ thr090,
                                                   const float* __restrict__ cnr, float* __restrict__ out, int N, int pitch,
                                                   size_t plane, int cnrS, int cnrPitch, size_t cnrPlane, int cnrScale) {
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y;
    if (x >= N) return;
    const size_t o = (size_t)blockIdx.z * plane + (size_t)y * pitch + x;
    float4 v;
    if (RAW) {
        const uint2 q = *reinterpret_cast<const uint2*>(raw + ((size_t)blockIdx.z * N + y) * N + x);
        const int thr = thr090[blockIdx.z];
        // stand-ins with the same truth value of `pixel <= 0.90f`
        v.x = (int)(q.x & 0xFFFFu) <= thr ? 0.0f : 1.0f;
        v.y = (int)(q.x >> 16) <= thr ? 0.0f : 1.0f;
        v.z = (int)(q.y & 0xFFFFu) <= thr ? 0.0f : 1.0f;
        v.w = (int)(q.y >> 16) <= thr ? 0.0f : 1.0f;
    } else {
        v = *reinterpret_cast<const float4*>(normalized + o);
    }
    const float cc = cnr_at(cnr + (size_t)blockIdx.z * cnrPlane, cnrS, cnrPitch, cnrScale, x, y);
    float4 r;
    r.x = relevant_of(v.x, cc, (uint32_t)x, (uint32_t)y, (uint32_t)N);
    r.y = relevant_of(v.y, cc, (uint32_t)x + 1u, (uint32_t)y, (uint32_t)N);
    r.z = relevant_of(v.z, cc, (uint32_t)x + 2u, (uint32_t)y, (uint32_t)N);
    r.w = relevant_of(v.w, cc, (uint32_t)x + 3u, (uint32_t)y, (uint32_t)N);
    *reinterpret_cast<float4*>(out + o) = r;
}

// ---- K12 + K20 ------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_sum_u32(uint32_t v, uint32_t* scratch /*[16]*/) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) scratch[wv] = v;
    __syncthreads();
    uint32_t r = 0;
    for (int i = 0; i < nw; i++) r += scratch[i];
    __syncthreads();
    return r;
}
__device__ __forceinline__ unsigned long long block_max_u64g(unsigned long long k, unsigned long long* scratch /*[16]*/) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long other = __shfl_xor(k, o);
        k = other > k ? other : k;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) scratch[wv] = k;
    __syncthreads();
    unsigned long long r = 0ull;
    for (int i = 0; i < nw; i++) r = scratch[i] > r ? scratch[i] : r;
    __syncthreads();
    return r;
}

// Several reductions behind one pair of barriers (the kernel is a chain of block-wide reductions: each pair of
// barriers over 16 wavefronts costs about a microsecond of the step's critical path).
__device__ __forceinline__ void block_max_sum_sum(unsigned long long& k, uint32_t& a, uint32_t& b, unsigned long long* s64, uint32_t* sa, uint32_t* sb) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long other = __shfl_xor(k, o);
        k = other > k ? other : k;
        a += (uint32_t)__shfl_xor((int)a, o);
        b += (uint32_t)__shfl_xor((int)b, o);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) { s64[wv] = k; sa[wv] = a; sb[wv] = b; }
    __syncthreads();
    k = 0ull; a = 0u; b = 0u;
    for (int i = 0; i < nw; i++) { k = s64[i] > k ? s64[i] : k; a += sa[i]; b += sb[i]; }
    __syncthreads();
}
__device__ __forceinline__ void block_max_max(uint32_t& a, uint32_t& b, uint32_t* sa, uint32_t* sb) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        a = max(a, (uint32_t)__shfl_xor((int)a, o));
        b = max(b, (uint32_t)__shfl_xor((int)b, o));
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) { sa[wv] = a; sb[wv] = b; }
    __syncthreads();
    a = 0u; b = 0u;
    for (int i = 0; i < nw; i++) { a = max(a, sa[i]); b = max(b, sb[i]); }
    __syncthreads();
}

// One block of 1024 threads per image: thread i owns bin i.
// gradation_curve_generate.comp:50-193. All integer arithmetic is uint32 and wraps, like GLSL's uint.
// hist_b / gzero (may be NULL): images whose gzero word is set take their histogram from hist_b — the literal
// k_grad_hist pass that ran behind the fused expand kernel — and copy it into hist so that every getter sees it.
// `raw`: bin threadIdx.x of the image's histogram (1024 threads)
__device__ __forceinline__ void grad_curve_body(uint32_t raw, int img, musica_hist_max_point* __restrict__ gmax, DevCurve* __restrict__ curves) {
    __shared__ uint32_t cnt[MUSICA_GRAD_BINS];
    __shared__ uint32_t s32[16], s32b[16];
    __shared__ unsigned long long s64[16];
    const uint32_t i = threadIdx.x;
    const uint32_t count = raw / 100u;                                           // :68
    cnt[i] = count;
    // K12 on the raw histogram (src/vk_processing.cpp:2499, first maximum wins) and the two window sums, one pass
    const uint32_t lowest = 10u;                                                 // :48
    uint32_t meanCount = i >= lowest ? count * i : 0u;                           // :70 (wraps)
    uint32_t meanSum = i >= lowest ? count : 0u;                                 // :71
    {
        unsigned long long k = raw ? (((unsigned long long)raw << 32) | (unsigned long long)(0xFFFFFFFFu - i)) : 0ull;
        block_max_sum_sum(k, meanCount, meanSum, s64, s32, s32b);
        if (i == 0) {
            musica_hist_max_point mp;
            mp.maxValue = k ? (uint32_t)(k >> 32) : 0u;
            mp.maxBin = k ? 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull) : 0u;
            gmax[img] = mp;
        }
    }
    const uint32_t meanQuot = meanSum ? meanCount / meanSum : 0u;                // :74 (x / 0 restated as 0)
    const float meanHistPos = (float)meanQuot / (float)MUSICA_GRAD_BINS;
    const uint32_t upper = f2u(meanHistPos * (float)MUSICA_GRAD_BINS);           // :77
    // argmax over [lowest, upper), strict '>' scanning upwards: lowest index among the maxima; none -> (0, 0)
    unsigned long long k = (i >= lowest && i < upper && count) ? (((unsigned long long)count << 32) | (unsigned long long)(0xFFFFFFFFu - i)) : 0ull;
    k = block_max_u64g(k, s64);
    const uint32_t maxCount = k ? (uint32_t)(k >> 32) : 0u;
    const uint32_t maxPosition = k ? 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull) : 0u;
    const uint32_t lowThreshold = f2u((float)maxCount * 0.05f);                  // :88
    // t0 (:94-105): walk down from maxPosition to 1 while count >= lowThreshold; t0 = lowest bin reached.
    // fail0 = highest bin in [1, maxPosition] violating the condition (0 if none).
    uint32_t f0 = (i >= 1 && i <= maxPosition && !(count >= lowThreshold)) ? i : 0u;
    // t1 (:108-119): walk up from maxPosition while count > 0; t1 = highest bin reached.
    // fail1 = lowest bin in [maxPosition, 1023] with count == 0 (1024 if none): max of (1024 - i).
    uint32_t f1 = (i >= maxPosition && count == 0u) ? (uint32_t)MUSICA_GRAD_BINS - i : 0u;
    block_max_max(f0, f1, s32, s32b);
    __shared__ float cx[kCurveCap], cy[kCurveCap];
    __shared__ int s_mono;
    // the window scalars are block-uniform: every thread derives them (same arithmetic as one thread would)
    const uint32_t fail0 = f0;
    const uint32_t fail1 = f1 ? (uint32_t)MUSICA_GRAD_BINS - f1 : (uint32_t)MUSICA_GRAD_BINS;
    float t0 = 0.0f, t1 = 0.0f;
    if (maxPosition >= 1u && fail0 < maxPosition) t0 = (float)(fail0 + 1u) * (1.0f / (float)MUSICA_GRAD_BINS);   // :96-100
    if (fail1 > maxPosition) t1 = (float)(fail1 - 1u) * (1.0f / (float)MUSICA_GRAD_BINS);                       // :110-115
    float m = 3.0f;                                                              // :52
    const float y_m = 0.5f;                                                      // :57
    const float ta = (float)maxPosition * (1.0f / (float)MUSICA_GRAD_BINS);      // :121,:133
    t0 -= 0.01f;                                                                 // :140
    if (t0 < 0.0f) t0 = 0.0f;
    if (t1 > 1.0f) t1 = 1.0f;                                                    // :144
    float tf = -(0.5f / m) + ta;                                                 // :146
    if (tf < t0) tf = t0;                                                        // :149
    if (tf == t0) m = y_m / (ta - tf);                                           // :162-163
    const float ts = (y_m / m) + ta;                                             // :165
    // 22 points, one thread each: (0,0), Bezier[(t0,0),(tf,0),(ta,.5)] i = 0..9, Bezier[(ta,.5),(ts,1),(t1,1)] i = 0..9, (1,1)
    if (i < (uint32_t)kCurveCap) {
        float x = 0.0f, y = 0.0f;
        if (i >= 1 && i <= 10) bezier_point(t0, 0.0f, tf, 0.0f, ta, y_m, i - 1, x, y);        // :156-160
        else if (i >= 11 && i <= 20) bezier_point(ta, y_m, ts, 1.0f, t1, 1.0f, i - 11, x, y); // :169-173
        else if (i == 21) { x = 1.0f; y = 1.0f; }                                              // :179
        cx[i] = x;
        cy[i] = y;
    }
    __syncthreads();
    curve_store_parallel(curves + img, cx, cy, &s_mono, 22, t0, ta, t1);                       // :181
}
__global__ __launch_bounds__(1024) void k_grad_curve(uint32_t* __restrict__ hist, musica_hist_max_point* __restrict__ gmax,
                                                     DevCurve* __restrict__ curves, const uint32_t* __restrict__ hist_b,
                                                     const uint32_t* __restrict__ gzero) {
    const int img = blockIdx.x;
    const uint32_t i = threadIdx.x;
    uint32_t raw = hist[(size_t)img * MUSICA_GRAD_BINS + i];
    if (gzero && gzero[img] != 0u) {
        raw = hist_b[(size_t)img * MUSICA_GRAD_BINS + i];
        hist[(size_t)img * MUSICA_GRAD_BINS + i] = raw;
    }
    grad_curve_body(raw, img, gmax, curves);
}

// The recount and the tone curve in ONE launch, behind a level-0 expand launch that binned every texel (GH): for an image whose gzero
// word is clear — no reconstructed texel is exactly 0, the usual case — workgroup (0, 0) builds the curve from that histogram and the
// others return at once (the recount used to be a launch of its own that returned at once: 5 us of every step). For an image that
// holds a zero every workgroup recounts its share literally (grad_hist_body, 16 wavefronts side by side) into a.hist (= hist_b), draws
// a ticket once its atomics have landed, and the workgroup that draws the last one reads hist_b back (agent-scope loads: the adds
// were performed memory-side, across the XCDs), copies it into `hist` for the getters and builds the curve. `ticket`: one word per
// image, kGradTicketStride apart, zero before and after.
template <bool SHARED8, bool RAW>
__global__ __launch_bounds__(1024) void k_grad_recount_curve(GradArgs a, uint32_t* __restrict__ hist, musica_hist_max_point* __restrict__ gmax,
                                                             DevCurve* __restrict__ curves, uint32_t* __restrict__ ticket) {
    __shared__ uint32_t lh[MUSICA_GRAD_BINS + 64];
    __shared__ int s_thr, s_last;
    const int img = blockIdx.z;
    uint32_t raw;
    if (a.only_if[img] == 0u) {   // block-uniform
        if (blockIdx.x != 0 || blockIdx.y != 0) return;
        raw = hist[(size_t)img * MUSICA_GRAD_BINS + threadIdx.x];
    } else {
        grad_hist_body<SHARED8, RAW>(a, img, lh, s_thr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's adds have been performed
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t* tk = ticket + (size_t)img * kGradTicketStride;
            const uint32_t t = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = t + 1u == gridDim.x * gridDim.y ? 1 : 0;
            if (s_last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!s_last) return;   // block-uniform
        raw = __hip_atomic_load(a.hist + (size_t)img * MUSICA_GRAD_BINS + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hist[(size_t)img * MUSICA_GRAD_BINS + threadIdx.x] = raw;
    }
    grad_curve_body(raw, img, gmax, curves);
}

// ---- K21 (getY of the tone curve: grad_parts.h) --------------------------------------------
template <int MONO, int U>
__device__ __forceinline__ void grad_apply_loop(const CurveLds& tab, const GradLds& gl, const float4* __restrict__ src, float4* __restrict__ dst,
                                                size_t i, size_t stride, size_t total, float4 (&v)[U]) {
    const uint32_t last_b = tab.count ? (tab.count - 1u) * 4u : 0u;
    const float x0 = tab.x[0], y0 = tab.y[0];
    while (i < total) {
        const size_t nxt = i + U * stride;
        float4 w[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            if (nxt + u * stride < total) w[u] = src[nxt + u * stride];   // the next trip's groups, in flight during the lookups
#pragma unroll
        for (int u = 0; u < U; u++)
            // non-temporal (round 4): nothing of the step reads the graded image again; with three steps in flight 8 x 2048^2 -2.1 % per step, and
            // -4.2 % together with the reconstruction stores of the expand launches (same-box A/B, profiles/r04_expand_pipeline.txt); a lone context:
            // equal or 1 % better (round 3 had measured this launch 5 - 10 % slower alone with non-temporal stores: before its loads were prefetched)
            if (i + u * stride < total) {
                const float4 r_ = grad_eval4<MONO>(tab, gl, last_b, x0, y0, v[u]);
                v4f q_; q_.x = r_.x; q_.y = r_.y; q_.z = r_.z; q_.w = r_.w;
                __builtin_nontemporal_store(q_, reinterpret_cast<v4f*>(&dst[i + u * stride]));
            }
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = w[u];
        i = nxt;
    }
}

__global__ __launch_bounds__(256) void k_grad_apply(const float* __restrict__ in, float* __restrict__ out, int N, int pitch, size_t plane,
                                                    const DevCurve* __restrict__ curves) {
    __shared__ CurveLds tab;
    __shared__ __attribute__((aligned(16))) GradLds gl;
    const int img = blockIdx.z;
    in += (size_t)img * plane;
    out += (size_t)img * plane;
    // rows are pitched to a multiple of 4 floats, so a plane is a dense run of 16-byte groups (pad columns are processed too:
    // they hold zeros and nobody reads them). U groups per thread and trip, all requested before the first lookup — and the
    // first trip's before the curve is copied to LDS: a workgroup lives for one or two trips (the launcher sizes the grid so),
    // and with the copy + barrier in front of the first load every workgroup started with ~1.5 us of nothing in flight
    // (57.8 us per launch at C4 against 46 us for a plain 1:1 stream of the same bytes, devtools/stream11.hip).
    constexpr int U = 2;
    const size_t total = (size_t)(pitch >> 2) * N;
    const float4* src = reinterpret_cast<const float4*>(in);
    float4* dst = reinterpret_cast<float4*>(out);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++)
        if (i + u * stride < total) v[u] = src[i + u * stride];
    grad_tables_to_lds(tab, gl, curves + img);
    __syncthreads();
    // the tone curve is monotone unless t1 < ts ("Exactness notes"); one LDS word, the same for the whole workgroup
    const int mono = __builtin_amdgcn_readfirstlane((int)tab.monotone), cnt = __builtin_amdgcn_readfirstlane((int)tab.count);
    if (mono != 0 && cnt < 32) grad_apply_loop<16, U>(tab, gl, src, dst, i, stride, total, v);
    else if (mono != 0) grad_apply_loop<32, U>(tab, gl, src, dst, i, stride, total, v);
    else grad_apply_loop<0, U>(tab, gl, src, dst, i, stride, total, v);
}

// ---- the two RGBA plots of #define RENDER_HISTS (include/vk_processing.h:22) -------------------------------
// noise_hist_render.comp and gradation_curve_debug_render.comp: one workgroup of 512 invocations each on a 512 x 128 rgba8
// image (src/vk_processing.cpp:2347, :2508; include/vk_processing.h:31-32), invocation x draws column x. The reference renders
// them on every execute and debugProcess writes them out (:2758-2806); here they are rendered when asked for
// (musica_render_*_hist, musica_debug_process). A texel is one packed word 0xAABBGGRR; stores outside the image are dropped (Q1).
constexpr uint32_t kPlotW = MUSICA_HIST_RENDER_WIDTH, kPlotH = MUSICA_HIST_RENDER_HEIGHT;
constexpr uint32_t kBlack = 0xFF000000u, kWhite = 0xFFFFFFFFu, kRed = 0xFF0000FFu, kGreen = 0xFF00FF00u, kBlue = 0xFFFF0000u, kMagenta = 0xFFFF00FFu;
__device__ __forceinline__ void plot_store(uint32_t* __restrict__ img, uint32_t x, uint32_t y, uint32_t rgba) {
    if (x < kPlotW && y < kPlotH) img[y * kPlotW + x] = rgba;
}
// uint(float(value) * (float(imageSize.y) / float(maxValue + 1))), clipped as the shaders clip it
__device__ __forceinline__ uint32_t plot_bar_height(uint32_t value, uint32_t maxValue) {
    uint32_t h = f2u((float)value * ((float)kPlotH / (float)(maxValue + 1u)));
    if (h > kPlotH) h = kPlotH - 1u;
    return h;
}

// noise_hist_render.comp:17-76 on the cnr level's histogram and argmax (src/vk_processing.cpp:1260-1266); the conversion
// factor is 1.0: bins 0 .. 511 of the 2048.
__global__ __launch_bounds__(512) void k_render_noise_hist(const uint32_t* __restrict__ hist, const musica_hist_max_point* __restrict__ maxpt,
                                                           uint32_t* __restrict__ out) {
    const uint32_t pos = threadIdx.x;
    const float factor = 1.0f;
    const uint32_t bin = f2u((float)pos * factor);
    const uint32_t value = bin < (uint32_t)MUSICA_NOISE_BINS ? hist[bin] : 0u;
    const uint32_t bar = plot_bar_height(value, maxpt->maxValue);
    const uint32_t startY = kPlotH - bar - 1u;
    for (uint32_t y = 0; y < kPlotH; y++) plot_store(out, pos, y, kBlack);                 // :62-64
    plot_store(out, pos, kPlotH - 1u, kRed);                                               // :66
    const bool at_max = bin <= maxpt->maxBin && (float)bin + factor > (float)maxpt->maxBin;
    for (uint32_t y = startY; y < startY + bar; y++) plot_store(out, pos, y, at_max ? kGreen : kWhite);   // :68-76
}

// gradation_curve_debug_render.comp:48-123 on the gradation histogram, its argmax and the tone curve
// (src/vk_processing.cpp:1668-1675). getY (:31-46) is the literal scan with the slope computed in place; x[count], y[count]
// are the zeros behind the curve.
__global__ __launch_bounds__(512) void k_render_grad_hist(const uint32_t* __restrict__ hist, const musica_hist_max_point* __restrict__ maxpt,
                                                          const DevCurve* __restrict__ curve, uint32_t* __restrict__ out) {
    const uint32_t pos = threadIdx.x;
    const float factor = (float)MUSICA_GRAD_BINS / 512.0f;
    const uint32_t bin = f2u((float)pos * factor);
    const uint32_t value = bin < (uint32_t)MUSICA_GRAD_BINS ? hist[bin] : 0u;
    const uint32_t bar = plot_bar_height(value, maxpt->maxValue);
    const uint32_t startY = kPlotH - bar - 1u;
    plot_store(out, pos, kPlotH - 1u, kRed);                                               // :77
    const bool at_max = bin <= maxpt->maxBin && (float)bin + factor > (float)maxpt->maxBin;
    for (uint32_t y = 0; y < kPlotH; y++)                                                  // :79-91
        plot_store(out, pos, y, (y >= startY && y < startY + bar) ? (at_max ? kMagenta : kWhite) : kBlack);
    const float step = 1.0f / 512.0f;
    const float cp = (float)pos * step;                                                    // :94
    float gy = 0.0f;
    const uint32_t count = curve->count;
    for (uint32_t i = 0; i < count; i++) {
        const float xi = curve->x[i], xn = (i + 1u < (uint32_t)kCurveCap) ? curve->x[i + 1u] : 0.0f, yn = (i + 1u < (uint32_t)kCurveCap) ? curve->y[i + 1u] : 0.0f;
        if (xi == cp) { gy = curve->y[i]; break; }
        if (xi <= cp && xn >= cp) { gy = (yn - curve->y[i]) / (xn - xi) * (cp - xi) + curve->y[i]; break; }
    }
    const uint32_t posX = f2u(cp * 512.0f * ((float)kPlotW / 512.0f));                     // :99
    const uint32_t posY = (kPlotH - 1u) - f2u(gy * (float)(kPlotH - 1u));                  // :100
    const float next = (float)(pos + 1u) * step;
    if (cp <= curve->t0 && curve->t0 < next) for (uint32_t i = 0; i < kPlotW; i++) plot_store(out, posX, i, kRed);     // :103-107
    if (cp <= curve->ta && curve->ta < next) for (uint32_t i = 0; i < kPlotW; i++) plot_store(out, posX, i, kGreen);   // :110-114
    if (cp <= curve->t1 && curve->t1 < next) for (uint32_t i = 0; i < kPlotW; i++) plot_store(out, posX, i, kRed);     // :117-121
    plot_store(out, posX, posY, kBlue);                                                    // :123
}

// ======================================================================================
// host-side launchers
// ======================================================================================

void launch_grad_hist(hipStream_t st, const GradArgs& a, int batch) {
    const int col_blocks = (a.N + 256 * kWavesPerBlock - 1) / (256 * kWavesPerBlock);
    const int groups = (a.N + kHistArea - 1) / kHistArea;
    const int wave_rows = (groups + a.groups_per_wave - 1) / a.groups_per_wave;
    const dim3 grid(col_blocks, wave_rows, batch);
    const bool raw = a.raw != nullptr && (a.N & 3) == 0;
    if ((a.cnrScale & 7) == 0) {
        if (raw) hipLaunchKernelGGL((k_grad_hist<true, true>), grid, dim3(kBlockThreads), 0, st, a);
        else hipLaunchKernelGGL((k_grad_hist<true, false>), grid, dim3(kBlockThreads), 0, st, a);
    } else {
        if (raw) hipLaunchKernelGGL((k_grad_hist<false, true>), grid, dim3(kBlockThreads), 0, st, a);
        else hipLaunchKernelGGL((k_grad_hist<false, false>), grid, dim3(kBlockThreads), 0, st, a);
    }
}

void launch_grad_hist_ref(hipStream_t st, const float* img, const float* relevant, const LevelDesc& l0, uint32_t* hist, int batch) {
    const int areas = (l0.S + kHistArea - 1) / kHistArea;
    hipLaunchKernelGGL(k_grad_hist_ref, dim3((areas + 15) / 16, (areas + 15) / 16, batch), dim3(16, 16), 0, st, img, relevant, l0.S, l0.pitch,
                       l0.plane, hist);
}

void launch_relevant(hipStream_t st, const float* normalized, const float* cnr, float* out, const LevelDesc& l0, const LevelDesc& l3,
                     int cnrScale, int batch, const uint16_t* raw, const int* thr090) {
    const bool vec = (l0.S & 3) == 0 && (cnrScale & 3) == 0 && cnrScale > 0;
    const dim3 grid4((l0.S / 4 + 255) / 256, l0.S, batch);
    if (vec && raw && thr090)
        hipLaunchKernelGGL(k_relevant4<true>, grid4, dim3(256), 0, st, normalized, raw, thr090, cnr, out, l0.S, l0.pitch, l0.plane, l3.S, l3.pitch, l3.plane, cnrScale);
    else if (vec)
        hipLaunchKernelGGL(k_relevant4<false>, grid4, dim3(256), 0, st, normalized, raw, thr090, cnr, out, l0.S, l0.pitch, l0.plane, l3.S, l3.pitch, l3.plane, cnrScale);
    else
        hipLaunchKernelGGL(k_relevant, dim3((l0.S + 31) / 32, (l0.S + 7) / 8, batch), dim3(32, 8), 0, st, normalized, cnr, out, l0.S, l0.pitch,
                           l0.plane, l3.S, l3.pitch, l3.plane, cnrScale);
}

void launch_grad_recount_curve(hipStream_t st, const GradArgs& a, uint32_t* hist, musica_hist_max_point* gmax, DevCurve* curves, uint32_t* ticket, int batch) {
    const int col_blocks = (a.N + 256 * 16 - 1) / (256 * 16);   // 16 wavefronts side by side
    const int groups = (a.N + kHistArea - 1) / kHistArea;
    const int wave_rows = (groups + a.groups_per_wave - 1) / a.groups_per_wave;
    const dim3 grid(col_blocks, wave_rows, batch);
    const bool raw = a.raw != nullptr && (a.N & 3) == 0;
    if ((a.cnrScale & 7) == 0) {
        if (raw) hipLaunchKernelGGL((k_grad_recount_curve<true, true>), grid, dim3(1024), 0, st, a, hist, gmax, curves, ticket);
        else hipLaunchKernelGGL((k_grad_recount_curve<true, false>), grid, dim3(1024), 0, st, a, hist, gmax, curves, ticket);
    } else {
        if (raw) hipLaunchKernelGGL((k_grad_recount_curve<false, true>), grid, dim3(1024), 0, st, a, hist, gmax, curves, ticket);
        else hipLaunchKernelGGL((k_grad_recount_curve<false, false>), grid, dim3(1024), 0, st, a, hist, gmax, curves, ticket);
    }
}

void launch_grad_curve(hipStream_t st, uint32_t* hist, musica_hist_max_point* gmax, DevCurve* curves, int batch, const uint32_t* hist_b, const uint32_t* gzero) {
    hipLaunchKernelGGL(k_grad_curve, dim3(batch), dim3(1024), 0, st, hist, gmax, curves, hist_b, gzero);
}

// saveOutImage (src/vk_processing.cpp:2624-2634): crop `margin` texels on every side and quantise, (uint8_t)(255.0f * (v - 0) / (1 - 0)),
// on the device, so that the read-back is 1 byte per output pixel instead of 4 bytes per input pixel. The C cast is undefined
// outside [0, 256): restated as the x86 lowering the reference's build gets (cvttss2si to int32, low byte kept; NaN and
// values outside int32 give 0x80000000 -> 0), the same statement as the oracle's and dump_image's.
__global__ __launch_bounds__(256) void k_out_pixels(const float* __restrict__ graded, int pitch, int margin, int nw, uint8_t* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= nw) return;
    const float maxValue = 1.0f, minValue = 0.0f;
    const float q = 255.0f * (graded[(size_t)(y + margin) * pitch + x + margin] - minValue) / (maxValue - minValue);
    out[(size_t)y * nw + x] = (q == q && q > -2147483648.0f && q < 2147483648.0f) ? (uint8_t)(int32_t)q : (uint8_t)0;
}
void launch_out_pixels(hipStream_t st, const float* graded, const LevelDesc& l0, int margin, uint8_t* out) {
    const int nw = l0.S - 2 * margin;
    hipLaunchKernelGGL(k_out_pixels, dim3((nw + 255) / 256, nw), dim3(256), 0, st, graded, l0.pitch, margin, nw, out);
}

// The same pixels as the pixel array of the BMP file saveOutImage writes — stbi_write_bmp(path, w, h, comp = 1, data), stb_image_write.h:492-500:
// 24 bpp, gray replicated to B, G, R, rows bottom-up, each row padded with zeros to a multiple of 4 bytes — one thread per 32-bit word
// of a file row, so that the host writes the file with one call instead of expanding 1 -> 3 bytes per pixel row by row (14 - 19 ms of the
// drop-in's 26 - 29 ms `save` at 3072^2). `out` may be page-locked host memory (the stores then go over the link: 28 MB, ~1 ms).
__global__ __launch_bounds__(256) void k_out_bmp24(const float* __restrict__ graded, int pitch, int margin, int nw, int row_words, uint32_t* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;              // file row: 0 is the bottom row of the image
    if (k >= row_words) return;
    const float* src = graded + (size_t)(nw - 1 - r + margin) * pitch + margin;
    const float maxValue = 1.0f, minValue = 0.0f;
    uint32_t word = 0u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int px = (4 * k + j) / 3;
        uint32_t v = 0u;
        if (px < nw) {
            const float q = 255.0f * (src[px] - minValue) / (maxValue - minValue);
            v = (q == q && q > -2147483648.0f && q < 2147483648.0f) ? (uint32_t)(uint8_t)(int32_t)q : 0u;   // k_out_pixels' statement of the cast
        }
        word |= v << (8 * j);
    }
    out[(size_t)r * row_words + k] = word;
}
void launch_out_bmp24(hipStream_t st, const float* graded, const LevelDesc& l0, int margin, uint32_t* out) {
    const int nw = l0.S - 2 * margin;
    const int row_words = (nw * 3 + 3) / 4;
    hipLaunchKernelGGL(k_out_bmp24, dim3((row_words + 255) / 256, nw), dim3(256), 0, st, graded, l0.pitch, margin, nw, row_words, out);
}

void launch_grad_apply(hipStream_t st, const float* in, float* out, const LevelDesc& l0, const DevCurve* curves, int batch) {
    // two 16-byte groups per thread and trip; ~16384 workgroups per launch is where a plain 1:1 stream peaks on this part
    // (devtools/stream11.hip: 46 us for 2 x 134 MB; 2048 workgroups 52 us, 65536 51 us)
    const size_t total = (size_t)(l0.pitch >> 2) * l0.S;
    const size_t want = (total + 511) / 512;
    const size_t cap = std::max<size_t>(16384 / (size_t)std::max(batch, 1), 256);
    const int blocks = (int)std::max<size_t>(std::min(want, cap), 1);
    hipLaunchKernelGGL(k_grad_apply, dim3(blocks, 1, batch), dim3(256), 0, st, in, out, l0.S, l0.pitch, l0.plane, curves);
}

void launch_render_noise_hist(hipStream_t st, const uint32_t* hist, const musica_hist_max_point* maxpt, uint32_t* out) {
    hipLaunchKernelGGL(k_render_noise_hist, dim3(1), dim3(512), 0, st, hist, maxpt, out);
}
void launch_render_grad_hist(hipStream_t st, const uint32_t* hist, const musica_hist_max_point* maxpt, const DevCurve* curve, uint32_t* out) {
    hipLaunchKernelGGL(k_render_grad_hist, dim3(1), dim3(512), 0, st, hist, maxpt, curve, out);
}

}  // namespace musica
