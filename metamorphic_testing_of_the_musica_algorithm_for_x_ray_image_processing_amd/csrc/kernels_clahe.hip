// kernels_clahe.hip — the CLAHE gradation alternative (clahe_histogram.comp, clahe_grad_curve.comp,
// clahe_grad_curve_apply.comp; K22-K24). The reference keeps it behind `#ifdef ENABLE_CLAHE`
// (include/vk_processing.h:13, commented out) and its host wiring does not compile, so these
// kernels restate the shader text only (oracle: musica_oracle_k_clahe_*). Decisions shared with the
// oracle: relevantImage is read as the f32 it really is; float -> uint of a negative tile
// coordinate saturates to 0; points[256] (one past the per-tile curve) reads as (0, 0); the
// histogram image is cleared at the start of every execute.
#include "kernels_common.h"
#include "launchers.h"
#include "grad_parts.h"

namespace musica {

constexpr int kT = MUSICA_CLAHE_TILES;
constexpr int kB = MUSICA_CLAHE_BINS;

// K22 clahe_histogram.comp:13-45 — hist[tx][ty][bin] += 1 where relevant == 1.0.
// One workgroup per band of `band` consecutive rows (8 at most: a band touches one or two tile rows, so its flush is short; the
// LDS copy still holds all 16 tiles; fewer rows where 8 would leave the launch under ~2048 workgroups — one 4096^2 image:
// 512 workgroups of 8 rows walked their 32 dependent load trips at 2 TB/s, 51 -> 36 us), 4 columns per thread with 16-byte loads.
// (Measured and not kept: an LDS copy of only the two tile rows a band can touch — 41 us; the two serial float chains of
// k_clahe_curve on v_readlane instead of thread 0 walking LDS — 22 against 13 us; 16 instead of 8 rows per k_clahe_apply4 band.)
// RAWREL: the relevant value is computed here (relevant_of() on the cnr texel and `raw <= thr090` standing for
// `normalized <= 0.9`, as in k_relevant4<true>) instead of read from a stored relevant image: 6 B/px instead of 8, and the
// context's hot path writes no relevant image at all (musica_get_image computes it on demand). N % 4 == 0.
template <bool RAWREL>
__global__ __launch_bounds__(256) void k_clahe_hist(const float* __restrict__ img, const float* __restrict__ relevant, int N, int pitch,
                                                    size_t plane, uint32_t* __restrict__ hist, const uint16_t* __restrict__ raw,
                                                    const int* __restrict__ thr090, const float* __restrict__ cnr, int cnrS, int cnrPitch,
                                                    size_t cnrPlane, int cnrScale, int band) {
    __shared__ uint32_t lh[kT * kT * kB];
    for (int i = threadIdx.x; i < kT * kT * kB; i += blockDim.x) lh[i] = 0u;
    __syncthreads();
    img += (size_t)blockIdx.z * plane;
    if (!RAWREL) relevant += (size_t)blockIdx.z * plane;
    const int thr = RAWREL ? thr090[blockIdx.z] : 0;
    const float fN = (float)N;
    for (int y = blockIdx.x * band; y < min(blockIdx.x * band + band, N); y++) {
        const uint32_t ty = f2u((float)y / fN * (float)kT);                      // :35
        const float* irow = img + (size_t)y * pitch;
        for (int x0 = threadIdx.x * 4; x0 < N; x0 += blockDim.x * 4) {
            const float4 c4 = load4_guard(irow, x0, N);
            float rv[4];
            if (RAWREL) {
                const uint2 q = *reinterpret_cast<const uint2*>(raw + ((size_t)blockIdx.z * N + y) * N + x0);
                const float cc = cnr_at(cnr + (size_t)blockIdx.z * cnrPlane, cnrS, cnrPitch, cnrScale, x0, y);   // one texel for the four (scale % 4 == 0)
                const int pxl[4] = {(int)(q.x & 0xFFFFu), (int)(q.x >> 16), (int)(q.y & 0xFFFFu), (int)(q.y >> 16)};
#pragma unroll
                for (int j = 0; j < 4; j++) rv[j] = relevant_of(pxl[j] <= thr ? 0.0f : 1.0f, cc, (uint32_t)(x0 + j), (uint32_t)y, (uint32_t)N);
            } else {
                const float4 r4 = load4_guard(relevant + (size_t)y * pitch, x0, N);
                rv[0] = r4.x; rv[1] = r4.y; rv[2] = r4.z; rv[3] = r4.w;
            }
            const float cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int x = x0 + j;
                const float cur = cv[j];
                const float scaled = cur * (float)(kB - 1) + 0.5f;               // :20
                // NaN never indexes (Q6); int(scaled) must land in [0, 256): scaled in (-1, 256)
                const bool inrange = scaled > -1.0f && scaled < (float)kB;
                const int bin = inrange ? (int)scaled : 0;
                const uint32_t tx = f2u((float)x / fN * (float)kT);             // :34
                if (x < N && inrange && rv[j] == 1.0f && tx < (uint32_t)kT && ty < (uint32_t)kT)
                    atomicAdd(&lh[(tx * kT + ty) * kB + bin], 1u);               // :39-44
            }
        }
    }
    __syncthreads();
    uint32_t* gh = hist + (size_t)blockIdx.z * kT * kT * kB;
    for (int i = threadIdx.x; i < kT * kT * kB; i += blockDim.x) {
        const uint32_t v = lh[i];
        if (v) atomicAdd(&gh[i], v);
    }
}

// K23 clahe_grad_curve.comp:21-100 — the reference runs one thread per tile; here one 256-thread workgroup per
// tile stages the histogram and the per-bin quotients in LDS, and thread 0 performs the two float accumulations
// (clip excess, running sum) in the shader's index order, so the sums round exactly as the serial loops do.
__global__ __launch_bounds__(kB) void k_clahe_curve(const uint32_t* __restrict__ hist, musica_point* __restrict__ points) {
    __shared__ float ny[kB];
    __shared__ uint32_t part[kB];
    __shared__ float s_clipAdd;
    const int tile = blockIdx.x, i = threadIdx.x;                                // tile = tx * kT + ty
    const uint32_t* h = hist + ((size_t)blockIdx.y * kT * kT + tile) * kB;
    musica_point* pts = points + ((size_t)blockIdx.y * kT * kT + tile) * kB;
    const uint32_t hv = h[i];
    part[i] = hv;
    __syncthreads();
    for (int st = kB / 2; st >= 1; st >>= 1) {                                   // :31-43 (uint sum: any order)
        if (i < st) part[i] += part[i + st];
        __syncthreads();
    }
    const uint32_t count = part[0];
    const float clipLimit = 1.0f / 32.0f;                                        // :60
    const float q = (float)hv / (float)count;
    ny[i] = q;
    __syncthreads();
    if (i == 0) {
        float clipCount = 0.0f;
        for (int k = 0; k < kB; k++)                                             // :47-57, :63-69
            if (ny[k] > clipLimit) clipCount += ny[k] - clipLimit;
        s_clipAdd = clipCount / (float)kB;                                       // :76
    }
    __syncthreads();
    float v = q;
    if (v > clipLimit) v = clipLimit;                                            // :78-93
    v += s_clipAdd;
    __syncthreads();
    ny[i] = v;
    __syncthreads();
    if (i == 0) {
        float curr = 0.0f;
        for (int k = 0; k < kB; k++) { curr += ny[k]; ny[k] = curr; }
    }
    __syncthreads();
    pts[i].x = i == kB - 1 ? 1.0f : (float)i * (1.0f / (float)kB);
    pts[i].y = ny[i];
}

// getY() of clahe_grad_curve_apply.comp:27-36 on one tile's 256 points. The abscissae are x[i] = i/256 for
// i < 255 and x[255] = 1: strictly increasing, so the first match is i = j - 1 with j = #{x[i] < s} (same
// argument as curve_eval), and j has a closed form: i/256 < s <=> i < 256 s (exact: power-of-two scaling)
// <=> i <= ceil(256 s) - 1, capped at the 255 regular abscissae, plus one if 1 < s. NaN compares false: j = 0.
// The slope divides by x[j] - x[j-1] = 1/256 (2/256 for the last segment): multiplying by 256 (128) is the
// same correctly rounded value.
__device__ __forceinline__ float clahe_x(int i) { return i == kB - 1 ? 1.0f : (float)i * (1.0f / (float)kB); }
__device__ __forceinline__ float clahe_get_y(const musica_point* __restrict__ pts, float s) {
    int j = 0;
    if (s > 0.0f) j = (int)fminf(ceilf(s * (float)kB), (float)(kB - 1));
    if (s > 1.0f) j += 1;
    if (j == 0) return (s == 0.0f) ? pts[0].y : 0.0f;
    if (j >= kB) return 0.0f;
    const float y0 = pts[j - 1].y, y1 = pts[j].y, x0 = clahe_x(j - 1);
    const float m = (y1 - y0) * (j == kB - 1 ? (float)(kB / 2) : (float)kB);
    return m * (s - x0) + y0;
}

__device__ __forceinline__ float signf_(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

// K24 clahe_grad_curve_apply.comp:38-161
__global__ void k_clahe_apply(const float* __restrict__ in, float* __restrict__ out, int N, int pitch, size_t plane,
                              const musica_point* __restrict__ points) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= N || y >= N) return;
    const musica_point* P = points + (size_t)blockIdx.z * kT * kT * kB;
    const size_t o = (size_t)blockIdx.z * plane + (size_t)y * pitch + x;
    const float pixel = in[o];
    const uint32_t G = (uint32_t)N / (uint32_t)kT;                               // :43
    const float px = (float)x / (float)G, py = (float)y / (float)G;              // :45-48
    const float bx = (float)f2u(px) + 0.5f, by = (float)f2u(py) + 0.5f;          // :50-53
    const float dx = px - bx, dy = py - by;                                      // :55-58
    float combined = 0.0f;
    if (dx == 0.0f && dy == 0.0f) {
        const uint32_t tx = f2u(floorf(bx)), ty = f2u(floorf(by));               // :63
        if (tx < (uint32_t)kT && ty < (uint32_t)kT) combined = clahe_get_y(P + ((size_t)tx * kT + ty) * kB, pixel);
    } else {
        const bool usex = dx != 0.0f, usey = dy != 0.0f;
        float cx[4], cy[4];
        int cnt;
        cx[0] = bx; cy[0] = by;
        if (usex && usey) {
            cnt = 4;
            cx[1] = bx + signf_(dx); cy[1] = by;
            cx[2] = bx; cy[2] = by + signf_(dy);
            cx[3] = bx + signf_(dx); cy[3] = by + signf_(dy);
        } else if (usey) { cnt = 2; cx[1] = bx; cy[1] = by + signf_(dy); }
        else { cnt = 2; cx[1] = bx + signf_(dx); cy[1] = by; }
        for (int i = 0; i < cnt; i++) {
            const float tdx = cx[i] - px, tdy = cy[i] - py;
            uint32_t tx = f2u(floorf(cx[i])), ty = f2u(floorf(cy[i]));
            if (tx > (uint32_t)kT - 1) tx = kT - 1;                              // :78-79
            if (ty > (uint32_t)kT - 1) ty = kT - 1;
            const float g = clahe_get_y(P + ((size_t)tx * kT + ty) * kB, pixel);
            if (usex && usey) combined += (1.0f - fabsf(tdx)) * (1.0f - fabsf(tdy)) * g;   // :138-146
            else if (usey) combined += (1.0f - fabsf(tdy)) * g;                  // :81-88
            else combined += (1.0f - fabsf(tdx)) * g;                            // :107-114
        }
    }
    out[o] = combined;
}

// K24 again, for sides that are a multiple of 4: the same per-texel arithmetic, organised so that nothing is computed twice.
//  * the 16 tiles' curve ordinates are staged in LDS (16 KB): getY() reads two of them per tile and a texel blends up to four
//    tiles — 8 scattered global reads per texel in k_clahe_apply;
//  * everything that depends on the column only (px = x / G with its float division, the base tile, the neighbour tile, the two
//    horizontal weights 1 - |cx - px|) is computed once per thread — a thread owns 4 columns and walks kApplyRows rows — and
//    everything that depends on the row only (py, by, the vertical neighbour and weights) once per row, where it is wave-uniform;
//  * per texel remain the four getY() lookups and the blend, in the shader's order: (bx,by), (bx+sx,by), (bx,by+sy), (bx+sx,by+sy).
constexpr int kApplyRows = 8;
// one axis of clahe_grad_curve_apply.comp:45-79 for texel coordinate v: p = v / G, base b = uint(p) + 0.5, d = p - b, the
// neighbour b + sign(d); tile indices (the base one unclamped for the d == 0 case, both clamped for the blend) and the
// weights 1 - |c - p| of the two candidates.
struct ClaheAxis {
    float w0, w1;          // 1 - |b - p|, 1 - |(b + sign d) - p|
    uint32_t t0, t1;       // clamped tile index of b and of b + sign d
    uint32_t t0_raw;       // uint(floor(b)), unclamped (:63)
    bool use;              // d != 0
};
__device__ __forceinline__ ClaheAxis clahe_axis(int v, uint32_t G) {
    ClaheAxis a;
    const float p = (float)v / (float)G;                                         // :45-48
    const float b = (float)f2u(p) + 0.5f;                                        // :50-53
    const float d = p - b;                                                       // :55-58
    const float n = b + signf_(d);
    a.use = d != 0.0f;
    a.w0 = 1.0f - fabsf(b - p);
    a.w1 = 1.0f - fabsf(n - p);
    a.t0_raw = f2u(floorf(b));
    a.t0 = min(a.t0_raw, (uint32_t)kT - 1u);                                     // :78-79
    a.t1 = min(f2u(floorf(n)), (uint32_t)kT - 1u);
    return a;
}
// ---- the apply kernels' hot path --------------------------------------------------------------------------------
// The literal form (k_clahe_apply) costs ~100 vector instructions and a dozen branches per texel: at 4 columns per thread the
// apply ran instruction-bound at 2.6 TB/s. Here:
//  * the four getY() calls of a texel look the SAME s up in four tiles, so the segment index, s - x[j-1] and the slope factor are
//    computed once per texel, and getY()'s three outcomes become one expression on a padded table — a tile's 256 ordinates are
//    followed by two zeros:
//      1 <= j < 256 : idx = j - 1,  t = s - idx / 256   ->  ((y[idx+1] - y[idx]) * mult) * t + y[idx]        (getY proper)
//      s == 0       : idx = 0,      t = 0               ->  (..) * 0 + y[0] = y[0]                            (:29-31, x[0] == s)
//      otherwise    : idx = 256,    t = 0               ->  (0 * mult) * 0 + 0 = 0                            (no segment: 0; NaN too)
//    the same operations on the same values as clahe_get_y for the first case, the same results for the others (ordinates are >= 0
//    or, for a tile without a relevant texel, NaN);
//  * the shader's four cases (dx, dy zero or not: texels of the 4 columns / 4 rows through tile centres blend two tiles or take
//    one) are the four-term sum with weights (1, 0) on an axis without a neighbour and the neighbour's tile index replaced by the
//    base one: 1 * w = w and 0 + a = a are exact, a term 0 * w * g is +0 (g >= 0) or, for a NaN tile, the NaN the kept term of the
//    same tile already put into the sum. So every texel runs the same straight-line code.
constexpr int kYs = kB + 2;   // ordinates of one tile in LDS + two zeros
struct ClaheRowLds { float w0, w1; int b0, b1; };   // one row: the two weights and kYs * tile row of the two candidates
__device__ __forceinline__ void clahe_tables_to_lds(float* __restrict__ ys, const musica_point* __restrict__ P) {
    for (int i = threadIdx.x; i < kT * kT * kB; i += blockDim.x) ys[(i / kB) * kYs + (i % kB)] = P[i].y;
    if (threadIdx.x < kT * kT * 2) ys[(threadIdx.x >> 1) * kYs + kB + (threadIdx.x & 1)] = 0.0f;
}
struct ClaheLookup {
    int idx;         // see above
    float t, mult;   // s - x[idx] (0 where no segment); kB (kB / 2 for the last segment)
};
__device__ __forceinline__ ClaheLookup clahe_lookup(float s) {
    const float cj = fminf(ceilf(s * (float)kB), (float)(kB - 1));
    int j = s > 0.0f ? (int)cj : 0;
    j += s > 1.0f ? 1 : 0;
    const bool valid = (uint32_t)(j - 1) < (uint32_t)(kB - 1);   // 1 <= j < kB
    ClaheLookup q;
    q.idx = valid ? j - 1 : (s == 0.0f ? 0 : kB);
    q.t = q.idx == kB ? 0.0f : s - (float)q.idx * (1.0f / (float)kB);   // clahe_x(j - 1), j - 1 <= 254
    q.mult = j == kB - 1 ? (float)(kB / 2) : (float)kB;
    return q;
}
__device__ __forceinline__ float clahe_y_at(const float* __restrict__ ys, int base, const ClaheLookup& q) {
    const float y0 = ys[base + q.idx], y1 = ys[base + q.idx + 1];
    return ((y1 - y0) * q.mult) * q.t + y0;
}
// one texel (:61-146): wx0 / wx1, bx0 / bx1 = the column's weights and kT * kYs * tile column of its two candidates, r = the row's
__device__ __forceinline__ float clahe_blend_xy(const float* __restrict__ ys, float pixel, float wx0, float wx1, int bx0, int bx1, const ClaheRowLds& r) {
    const ClaheLookup q = clahe_lookup(pixel);
    float combined = 0.0f;
    combined += wx0 * r.w0 * clahe_y_at(ys, bx0 + r.b0, q);
    combined += wx1 * r.w0 * clahe_y_at(ys, bx1 + r.b0, q);
    combined += wx0 * r.w1 * clahe_y_at(ys, bx0 + r.b1, q);
    combined += wx1 * r.w1 * clahe_y_at(ys, bx1 + r.b1, q);
    return combined;
}

// K24 (GRAD = 0) or K21 + K24 in one pass over the reconstruction (GRAD = 16 / 32 / 1: the tone curve by grad_eval4<16 / 32 / 0>).
// The reference applies the tone curve (img_apply_gradation_curve.comp) and the CLAHE curves (clahe_grad_curve_apply.comp) in two
// dispatches that each read the whole image; with GRAD a texel is read once and both results are stored (12 instead of 16 bytes
// per texel, one launch).
template <int GRAD>
__device__ __forceinline__ void apply_rows(const float* __restrict__ in, float* __restrict__ out_clahe, float* __restrict__ out_graded, int N, int pitch, int x,
                                           int y0, const float* __restrict__ ys, const ClaheRowLds* __restrict__ rows, const ClaheAxis (&ax)[4],
                                           const CurveLds& tab, const GradLds& gl, float4 v) {
    const uint32_t last_b = tab.count ? (tab.count - 1u) * 4u : 0u;
    const float gx0 = tab.x[0], gy0 = tab.y[0];
    float wx0[4], wx1[4];
    int bx0[4], bx1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        wx0[j] = ax[j].use ? ax[j].w0 : 1.0f;
        wx1[j] = ax[j].use ? ax[j].w1 : 0.0f;
        bx0[j] = (int)ax[j].t0 * kT * kYs;
        bx1[j] = (int)(ax[j].use ? ax[j].t1 : ax[j].t0) * kT * kYs;
    }
    for (int r = 0; r < kApplyRows; r++) {
        const int y = y0 + r;
        if (y >= N) break;
        float4 nxt = v;
        if (r + 1 < kApplyRows && y + 1 < N) nxt = *reinterpret_cast<const float4*>(in + (size_t)(y + 1) * pitch + x);   // next row, in flight during the lookups
        // non-temporal stores (round 4: nothing of the step reads either image again; 4096^2 + CLAHE -1 % with steps in flight, -2.5 % on a lone context)
        if (GRAD) { const float4 r_ = grad_eval4<(GRAD == 1 ? 0 : GRAD)>(tab, gl, last_b, gx0, gy0, v); v4f q_; q_.x = r_.x; q_.y = r_.y; q_.z = r_.z; q_.w = r_.w;
                    __builtin_nontemporal_store(q_, reinterpret_cast<v4f*>(out_graded + (size_t)y * pitch + x)); }
        const ClaheRowLds rw = rows[r];
        float4 c;
        c.x = clahe_blend_xy(ys, v.x, wx0[0], wx1[0], bx0[0], bx1[0], rw);
        c.y = clahe_blend_xy(ys, v.y, wx0[1], wx1[1], bx0[1], bx1[1], rw);
        c.z = clahe_blend_xy(ys, v.z, wx0[2], wx1[2], bx0[2], bx1[2], rw);
        c.w = clahe_blend_xy(ys, v.w, wx0[3], wx1[3], bx0[3], bx1[3], rw);
        { v4f q_; q_.x = c.x; q_.y = c.y; q_.z = c.z; q_.w = c.w; __builtin_nontemporal_store(q_, reinterpret_cast<v4f*>(out_clahe + (size_t)y * pitch + x)); }
        v = nxt;
    }
}
// grid: x = 1024-column chunks, y = bands of kApplyRows rows, z = batch. BOTH = false: K24 alone (curves / out_graded unused).
template <bool BOTH, int W>
__global__ __launch_bounds__(256, W) void k_clahe_apply4(const float* __restrict__ in, float* __restrict__ out_clahe, float* __restrict__ out_graded, int N, int pitch,
                                                      size_t plane, const musica_point* __restrict__ points, const DevCurve* __restrict__ curves) {
    __shared__ float ys[kT * kT * kYs];
    __shared__ ClaheRowLds rows[kApplyRows];
    __shared__ CurveLds tab;
    __shared__ __attribute__((aligned(16))) GradLds gl;
    const int y0 = blockIdx.y * kApplyRows;
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const uint32_t G = (uint32_t)N / (uint32_t)kT;                               // :43
    in += (size_t)blockIdx.z * plane;
    out_clahe += (size_t)blockIdx.z * plane;
    if (BOTH) out_graded += (size_t)blockIdx.z * plane;
    const bool active = x < N;
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (active) v = *reinterpret_cast<const float4*>(in + (size_t)y0 * pitch + x);   // the first row's group is requested before the tables are staged
    clahe_tables_to_lds(ys, points + (size_t)blockIdx.z * kT * kT * kB);
    if (BOTH) grad_tables_to_lds(tab, gl, curves + blockIdx.z);
    if (threadIdx.x < kApplyRows) {   // the rows' axis values, once per workgroup
        const ClaheAxis ay = clahe_axis(y0 + (int)threadIdx.x, G);
        ClaheRowLds rw;
        rw.w0 = ay.use ? ay.w0 : 1.0f; rw.w1 = ay.use ? ay.w1 : 0.0f;
        rw.b0 = (int)ay.t0 * kYs; rw.b1 = (int)(ay.use ? ay.t1 : ay.t0) * kYs;
        rows[threadIdx.x] = rw;
    }
    ClaheAxis ax[4];
#pragma unroll
    for (int j = 0; j < 4; j++) ax[j] = clahe_axis(x + j, G);
    __syncthreads();
    if (!active) return;
    const int mono = BOTH ? __builtin_amdgcn_readfirstlane((int)tab.monotone) : 0, cnt = BOTH ? __builtin_amdgcn_readfirstlane((int)tab.count) : 0;
    if (!BOTH) apply_rows<0>(in, out_clahe, out_graded, N, pitch, x, y0, ys, rows, ax, tab, gl, v);
    else if (mono != 0 && cnt < 32) apply_rows<16>(in, out_clahe, out_graded, N, pitch, x, y0, ys, rows, ax, tab, gl, v);   // the 22-point tone curve, monotone unless t1 < ts
    else if (mono != 0) apply_rows<32>(in, out_clahe, out_graded, N, pitch, x, y0, ys, rows, ax, tab, gl, v);
    else apply_rows<1>(in, out_clahe, out_graded, N, pitch, x, y0, ys, rows, ax, tab, gl, v);
}

void launch_grad_clahe_apply(hipStream_t st, const float* img, float* out_clahe, float* out_graded, const LevelDesc& l0, const musica_point* pts,
                             const DevCurve* curves, int batch) {
    // W = 8: register allocation capped at 64 (8 wavefronts per SIMD, 52 bytes of scratch): 46 us for one 4096^2 image against 51 us at 88 registers
    const dim3 grid((l0.S / 4 + 255) / 256, (l0.S + kApplyRows - 1) / kApplyRows, batch);
    hipLaunchKernelGGL((k_clahe_apply4<true, 8>), grid, dim3(256), 0, st, img, out_clahe, out_graded, l0.S, l0.pitch, l0.plane, pts, curves);
}

void launch_clahe(hipStream_t st, const float* img, const float* relevant, float* out, const LevelDesc& l0, uint32_t* hist, musica_point* pts,
                  int batch, const uint16_t* raw, const int* thr090, const float* cnr, const LevelDesc* l3, int cnrScale, bool hist_done, bool with_apply) {
    int band = 8;
    while (band > 1 && (long)((l0.S + band - 1) / band) * batch < 2048) band >>= 1;
    const dim3 hgrid((l0.S + band - 1) / band, 1, batch);
    if (hist_done) {}   // the level-0 expand launch counted it (k_expand_fast<.., CH = true>)
    else if (raw && thr090 && cnr && l3 && (l0.S & 3) == 0 && cnrScale > 0 && (cnrScale & 3) == 0)   // relevant image computed on the fly
        hipLaunchKernelGGL(k_clahe_hist<true>, hgrid, dim3(256), 0, st, img, relevant, l0.S, l0.pitch, l0.plane, hist, raw, thr090, cnr, l3->S, l3->pitch,
                           l3->plane, cnrScale, band);
    else
        hipLaunchKernelGGL(k_clahe_hist<false>, hgrid, dim3(256), 0, st, img, relevant, l0.S, l0.pitch, l0.plane, hist, raw, thr090, cnr, 0, 0, (size_t)0, 0, band);
    hipLaunchKernelGGL(k_clahe_curve, dim3(kT * kT, batch), dim3(kB), 0, st, hist, pts);
    if (!with_apply) return;   // launch_grad_clahe_apply follows
    if ((l0.S & 3) == 0)
        hipLaunchKernelGGL((k_clahe_apply4<false, 5>), dim3((l0.S / 4 + 255) / 256, (l0.S + kApplyRows - 1) / kApplyRows, batch), dim3(256), 0, st, img, out, (float*)nullptr,
                           l0.S, l0.pitch, l0.plane, pts, (const DevCurve*)nullptr);
    else
        hipLaunchKernelGGL(k_clahe_apply, dim3((l0.S + 31) / 32, (l0.S + 7) / 8, batch), dim3(32, 8), 0, st, img, out, l0.S, l0.pitch, l0.plane, pts);
}

}  // namespace musica
