// kernels_clahe.hip — the CLAHE gradation alternative (clahe_histogram.comp, clahe_grad_curve.comp,
// clahe_grad_curve_apply.comp; K22-K24). The reference keeps it behind `#ifdef ENABLE_CLAHE`
// (include/vk_processing.h:13, commented out) and its host wiring does not compile, so these
// kernels restate the shader text only (oracle: musica_oracle_k_clahe_*). Decisions shared with the
// oracle: relevantImage is read as the f32 it really is; float -> uint of a negative tile
// coordinate saturates to 0; points[256] (one past the per-tile curve) reads as (0, 0); the
// histogram image is cleared at the start of every execute.
#include "kernels_common.h"
#include "launchers.h"

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
__device__ __forceinline__ float clahe_get_y_lds(const float* __restrict__ ys /* [kB] ordinates of one tile */, float s) {
    int j = 0;
    if (s > 0.0f) j = (int)fminf(ceilf(s * (float)kB), (float)(kB - 1));
    if (s > 1.0f) j += 1;
    if (j == 0) return (s == 0.0f) ? ys[0] : 0.0f;
    if (j >= kB) return 0.0f;
    const float y0 = ys[j - 1], y1 = ys[j], x0 = clahe_x(j - 1);
    const float m = (y1 - y0) * (j == kB - 1 ? (float)(kB / 2) : (float)kB);
    return m * (s - x0) + y0;
}
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
__device__ __forceinline__ float clahe_blend(const float* __restrict__ ys, float pixel, const ClaheAxis& ax, const ClaheAxis& ay) {
    if (!ax.use && !ay.use) {                                                    // :61-66
        if (ax.t0_raw < (uint32_t)kT && ay.t0_raw < (uint32_t)kT) return clahe_get_y_lds(ys + ((size_t)ax.t0_raw * kT + ay.t0_raw) * kB, pixel);
        return 0.0f;
    }
    float combined = 0.0f;
    if (ax.use && ay.use) {                                                      // :116-146: (bx,by) (bx+sx,by) (bx,by+sy) (bx+sx,by+sy)
        combined += ax.w0 * ay.w0 * clahe_get_y_lds(ys + ((size_t)ax.t0 * kT + ay.t0) * kB, pixel);
        combined += ax.w1 * ay.w0 * clahe_get_y_lds(ys + ((size_t)ax.t1 * kT + ay.t0) * kB, pixel);
        combined += ax.w0 * ay.w1 * clahe_get_y_lds(ys + ((size_t)ax.t0 * kT + ay.t1) * kB, pixel);
        combined += ax.w1 * ay.w1 * clahe_get_y_lds(ys + ((size_t)ax.t1 * kT + ay.t1) * kB, pixel);
    } else if (ay.use) {                                                         // :68-88
        combined += ay.w0 * clahe_get_y_lds(ys + ((size_t)ax.t0 * kT + ay.t0) * kB, pixel);
        combined += ay.w1 * clahe_get_y_lds(ys + ((size_t)ax.t0 * kT + ay.t1) * kB, pixel);
    } else {                                                                     // :94-114
        combined += ax.w0 * clahe_get_y_lds(ys + ((size_t)ax.t0 * kT + ay.t0) * kB, pixel);
        combined += ax.w1 * clahe_get_y_lds(ys + ((size_t)ax.t1 * kT + ay.t0) * kB, pixel);
    }
    return combined;
}
// grid: x = 1024-column chunks, y = bands of kApplyRows rows, z = batch
__global__ __launch_bounds__(256) void k_clahe_apply4(const float* __restrict__ in, float* __restrict__ out, int N, int pitch, size_t plane,
                                                      const musica_point* __restrict__ points) {
    __shared__ float ys[kT * kT * kB];
    const musica_point* P = points + (size_t)blockIdx.z * kT * kT * kB;
    const int y0 = blockIdx.y * kApplyRows;
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const uint32_t G = (uint32_t)N / (uint32_t)kT;                               // :43
    in += (size_t)blockIdx.z * plane;
    out += (size_t)blockIdx.z * plane;
    const bool active = x < N;
    // the first row's group is requested before the ordinates are staged
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (active) v = *reinterpret_cast<const float4*>(in + (size_t)y0 * pitch + x);
    for (int i = threadIdx.x; i < kT * kT * kB; i += blockDim.x) ys[i] = P[i].y;
    ClaheAxis ax[4];
#pragma unroll
    for (int j = 0; j < 4; j++) ax[j] = clahe_axis(x + j, G);
    __syncthreads();
    if (!active) return;
    for (int r = 0; r < kApplyRows; r++) {
        const int y = y0 + r;
        if (y >= N) break;
        float4 nxt = v;
        if (r + 1 < kApplyRows && y + 1 < N) nxt = *reinterpret_cast<const float4*>(in + (size_t)(y + 1) * pitch + x);   // next row, in flight during the lookups
        const ClaheAxis ay = clahe_axis(y, G);
        float4 c;
        c.x = clahe_blend(ys, v.x, ax[0], ay);
        c.y = clahe_blend(ys, v.y, ax[1], ay);
        c.z = clahe_blend(ys, v.z, ax[2], ay);
        c.w = clahe_blend(ys, v.w, ax[3], ay);
        *reinterpret_cast<float4*>(out + (size_t)y * pitch + x) = c;
        v = nxt;
    }
}

void launch_clahe(hipStream_t st, const float* img, const float* relevant, float* out, const LevelDesc& l0, uint32_t* hist, musica_point* pts,
                  int batch, const uint16_t* raw, const int* thr090, const float* cnr, const LevelDesc* l3, int cnrScale) {
    int band = 8;
    while (band > 1 && (long)((l0.S + band - 1) / band) * batch < 2048) band >>= 1;
    const dim3 hgrid((l0.S + band - 1) / band, 1, batch);
    if (raw && thr090 && cnr && l3 && (l0.S & 3) == 0 && cnrScale > 0 && (cnrScale & 3) == 0)   // relevant image computed on the fly
        hipLaunchKernelGGL(k_clahe_hist<true>, hgrid, dim3(256), 0, st, img, relevant, l0.S, l0.pitch, l0.plane, hist, raw, thr090, cnr, l3->S, l3->pitch,
                           l3->plane, cnrScale, band);
    else
        hipLaunchKernelGGL(k_clahe_hist<false>, hgrid, dim3(256), 0, st, img, relevant, l0.S, l0.pitch, l0.plane, hist, raw, thr090, cnr, 0, 0, (size_t)0, 0, band);
    hipLaunchKernelGGL(k_clahe_curve, dim3(kT * kT, batch), dim3(kB), 0, st, hist, pts);
    if ((l0.S & 3) == 0)
        hipLaunchKernelGGL(k_clahe_apply4, dim3((l0.S / 4 + 255) / 256, (l0.S + kApplyRows - 1) / kApplyRows, batch), dim3(256), 0, st, img, out, l0.S, l0.pitch, l0.plane, pts);
    else
        hipLaunchKernelGGL(k_clahe_apply, dim3((l0.S + 31) / 32, (l0.S + 7) / 8, batch), dim3(32, 8), 0, st, img, out, l0.S, l0.pitch, l0.plane, pts);
}

}  // namespace musica
