// kernels_analysis.hip — normalisation and image-analysis kernels of the MUSICA path (gfx950).
//
//   k_clear         : vkCmdClearColorImage of the histograms (src/vk_processing.cpp:2153-2162) for the single-stage debug entry points
//                     (on the hot path the clears ride in k_minmax_u16)
//   k_minmax_u16    : img_sqrt.comp + img_max_reduce.comp chain + min_reduce.comp chain   (K1 + K2 + K3)
//   k_normalize     : img_sqrt.comp + img_normalize.comp                                    (K1 + K4)
//   k_sdev_hist     : img_sdev.comp + noise_hist.comp fused                                 (K10 + K11)
//   k_noise_curves  : img_histogram_max.comp + contrast_curve_generate.comp                 (K12 + K13)
//   k_cnr           : img_cnr.comp                                                          (K15)
//   k_sqrt          : img_sqrt.comp alone (debug image)
#include <stdlib.h>
#include <algorithm>
#include "kernels_common.h"
#include "exact_math.h"
#include "launchers.h"
#include "sdev_parts.h"

namespace musica {

// ---- clears ---------------------------------------------------------------------------
// minmax[b] = {min = 0xFFFFFFFF at word 0, max = 0 at word kMaxWord} (one 64-byte line each: the atomics of different
// images and of min / max then go to different L2 channels); noise_hist[b][4][2048] = 0; grad_hist[b][1024] = 0; clahe hist = 0.
__global__ void k_clear(uint32_t* __restrict__ minmax, uint32_t* __restrict__ noise_hist, uint32_t* __restrict__ grad_hist,
                        uint32_t* __restrict__ clahe_hist, int batch, uint32_t* __restrict__ grad_hist_b, uint32_t* __restrict__ gzero) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nh = batch * 4 * MUSICA_NOISE_BINS, gh = batch * MUSICA_GRAD_BINS;
    const int ch = clahe_hist ? batch * MUSICA_CLAHE_TILES * MUSICA_CLAHE_TILES * MUSICA_CLAHE_BINS : 0;
    if (minmax && i < batch) { minmax[kMinMaxStride * i] = 0xFFFFFFFFu; minmax[kMinMaxStride * i + kMaxWord] = 0u; }
    if (noise_hist && i < nh) noise_hist[i] = 0u;
    if (grad_hist && i < gh) grad_hist[i] = 0u;
    if (grad_hist_b && i < gh) grad_hist_b[i] = 0u;
    if (gzero && i < batch) gzero[i] = 0u;
    if (i < ch) clahe_hist[i] = 0u;
}

// ---- K1 + K2 + K3 ---------------------------------------------------------------------
// sqrt is monotone and the chains' float(uint(.)) truncation (img_max_reduce.comp:53, min_reduce.comp:30)
// is monotone too, so the chain results are functions of the integer extrema of the raw pixels:
//   max chain  -> floor(sqrt(max u16))            (every link floors; zeros from partial blocks never win)
//   min chain  -> floor(sqrt(min u16)) when every link has only full 8x8 blocks (N a power of 8),
//                 0 otherwise (a partial block reads out-of-image zeros, min_reduce.comp:22-27 + Q1).
// So this kernel reduces the uint16 pixels exactly (wavefront DPP reduction, one atomic pair per block).
__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o));
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
    return v;
}

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void minmax_acc(us2& mn, us2& mx, const uint4& q) {   // v_pk_min_u16 / v_pk_max_u16: two pixels per slot
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const us2 v = __builtin_bit_cast(us2, w[k]);
        mn = __builtin_elementwise_min(mn, v);
        mx = __builtin_elementwise_max(mx, v);
    }
}

// Two stages, no same-address atomics (a block used to end with an atomic pair on the image's two words; those retire ~11 ns
// apart, which capped the grid at 256 blocks per image and left a serial tail): every block reduces its share — one trip of
// kMinMaxLoads 16-byte loads per lane, all in flight at once — and stores ONE word {min | max << 16} into its own slot; the
// block that draws the last ticket of its image folds the slots into the two words img_normalize.comp reads and puts the ticket
// back to 0 for the next launch. Slot hand-off across XCDs: sc1 (write-through) store -> vmcnt(0) -> ticket add by the same lane;
// last block: barrier -> sc1 loads of every slot. The launch also zeroes the image's histograms (the vkCmdClearColorImage
// calls of src/vk_processing.cpp:2153-2162): every kernel that adds to or reads them runs behind this one.
constexpr int kMinMaxLoads = 8;
constexpr int kMinMaxThreads = 256;
constexpr int kMinMaxMaxBlocks = 4096;   // slots per image (launch_minmax uses at most 256)
constexpr int kTicketStride = 32;        // words between the tickets of two images: one 128-byte line each (same-line atomics serialise)
struct ClearArgs {
    uint32_t* noise_hist;    // [batch][4][2048] or null
    uint32_t* grad_hist;     // [batch][1024] or null
    uint32_t* grad_hist_b;   // [batch][1024] or null
    uint32_t* gzero;         // [batch] or null
    uint32_t* clahe_hist;    // [batch][4 * 4 * 256] or null
};
__device__ __forceinline__ void zero_words(uint32_t* p, int n, int first, int stride) {
    if (!p) return;
    for (int i = first; i < n; i += stride) p[i] = 0u;
}
template <int LOADS>
__global__ __launch_bounds__(kMinMaxThreads) void k_minmax_u16(const uint16_t* __restrict__ px, size_t count, uint32_t* __restrict__ minmax,
                                                              uint32_t* __restrict__ slots, uint32_t* __restrict__ ticket, ClearArgs ca) {
    const int img = blockIdx.z, nb = gridDim.x;
    {
        const int first = blockIdx.x * blockDim.x + threadIdx.x, stride = nb * blockDim.x;
        zero_words(ca.noise_hist ? ca.noise_hist + (size_t)img * 4 * MUSICA_NOISE_BINS : nullptr, 4 * MUSICA_NOISE_BINS, first, stride);
        zero_words(ca.grad_hist ? ca.grad_hist + (size_t)img * MUSICA_GRAD_BINS : nullptr, MUSICA_GRAD_BINS, first, stride);
        zero_words(ca.grad_hist_b ? ca.grad_hist_b + (size_t)img * MUSICA_GRAD_BINS : nullptr, MUSICA_GRAD_BINS, first, stride);
        zero_words(ca.gzero ? ca.gzero + img : nullptr, 1, first, stride);
        constexpr int kClaheWords = MUSICA_CLAHE_TILES * MUSICA_CLAHE_TILES * MUSICA_CLAHE_BINS;
        zero_words(ca.clahe_hist ? ca.clahe_hist + (size_t)img * kClaheWords : nullptr, kClaheWords, first, stride);
    }
    const uint16_t* p = px + (size_t)img * count;
    const size_t nvec = (((uintptr_t)p & 15u) == 0) ? count / 8 : 0;  // 8 pixels per 16-byte load
    const uint4* pv = reinterpret_cast<const uint4*>(p);
    // a block reads CONTIGUOUS chunks of LOADS * blockDim 16-byte vectors (its lanes' successive loads one block-width apart): with the
    // loads of a lane a whole grid-width apart — a power-of-two number of bytes — they all fell into the same memory channels
    // (3.1 TB/s at 8 x 2048^2; this form: see DESIGN.md section 6)
    const size_t chunk = (size_t)LOADS * blockDim.x, stride = (size_t)nb * chunk;
    us2 mn2 = {0xFFFFu, 0xFFFFu}, mx2 = {0u, 0u};
    size_t i = (size_t)blockIdx.x * chunk + threadIdx.x;
    for (; i + (LOADS - 1) * (size_t)blockDim.x < nvec; i += stride) {
        uint4 q[LOADS];
#pragma unroll
        for (int k = 0; k < LOADS; k++) q[k] = pv[i + (size_t)k * blockDim.x];
#pragma unroll
        for (int k = 0; k < LOADS; k++) minmax_acc(mn2, mx2, q[k]);
    }
    if (i < nvec) {   // the image's last, partial chunk
#pragma unroll
        for (int k = 0; k < LOADS; k++)
            if (i + (size_t)k * blockDim.x < nvec) minmax_acc(mn2, mx2, pv[i + (size_t)k * blockDim.x]);
    }
    uint32_t mn = min((uint32_t)mn2.x, (uint32_t)mn2.y), mx = max((uint32_t)mx2.x, (uint32_t)mx2.y);
    for (size_t k = nvec * 8 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (size_t)nb * blockDim.x) {
        const uint32_t v = p[k];
        mn = min(mn, v);
        mx = max(mx, v);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    const int kWaves = blockDim.x / 64;
    __shared__ uint32_t smn[kMinMaxThreads / 64], smx[kMinMaxThreads / 64];
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { smn[wv] = mn; smx[wv] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kWaves; k++) { mn = min(mn, smn[k]); mx = max(mx, smx[k]); }
        // write-through (sc1) slot store, drained, then the ticket; the last block reads every slot with sc1 loads: no agent
        // release / acquire fence (a release per block writes back the whole XCD L2: 128 us per launch with 4096 blocks)
        __hip_atomic_store(&slots[(size_t)img * kMinMaxMaxBlocks + blockIdx.x], mn | (mx << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t old = __hip_atomic_fetch_add(&ticket[(size_t)img * kTicketStride], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == (uint32_t)(nb - 1);
    }
    __syncthreads();   // the other wavefronts load only behind the barrier the ticket holder joined after its add returned
    if (!s_last) return;   // block-uniform
    mn = 0xFFFFu; mx = 0u;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) {
        const uint32_t v = __hip_atomic_load(&slots[(size_t)img * kMinMaxMaxBlocks + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mn = min(mn, v & 0xFFFFu);
        mx = max(mx, v >> 16);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    __syncthreads();   // smn / smx were read by thread 0 above
    if (lane == 0) { smn[wv] = mn; smx[wv] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kWaves; k++) { mn = min(mn, smn[k]); mx = max(mx, smx[k]); }
        minmax[kMinMaxStride * img] = mn;
        minmax[kMinMaxStride * img + kMaxWord] = mx;
        ticket[(size_t)img * kTicketStride] = 0u;   // the next launch counts from 0 again (launches of one context are ordered)
    }
}

// ---- self-test of exact_math.h on the device (musica_selftest_exact_math) ----------------
__device__ __forceinline__ bool same_float(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }
__global__ __launch_bounds__(256) void k_selftest_sqrt(unsigned long long* __restrict__ bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long b0 = 0, b1 = 0, b2 = 0;
    for (uint64_t grp = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; grp < (1ull << 29); grp += stride) {
        float want[8], s[8], z[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float x = __uint_as_float((uint32_t)(grp * 8 + j));
            want[j] = sqrtf(x);
            s[j] = x;
            z[j] = j == (int)(grp & 7) ? 0.0f : x;
            if (!same_float(musica_sqrt(x), want[j])) b0++;
        }
        musica_sqrt8(s);
        musica_sqrt8(z);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (!same_float(s[j], want[j])) b1++;
            if (!same_float(z[j], j == (int)(grp & 7) ? 0.0f : want[j])) b2++;
        }
    }
    if (b0) atomicAdd(&bad[0], b0);
    if (b1) atomicAdd(&bad[1], b1);
    if (b2) atomicAdd(&bad[2], b2);
}
// blockIdx.x = min, blockIdx.y = max (both as the chains deliver them: integer-valued floats 0 .. 255)
__global__ __launch_bounds__(256) void k_selftest_norm(unsigned long long* __restrict__ bad) {
    if (blockIdx.y < blockIdx.x) return;
    const float minv = (float)blockIdx.x, maxv = (float)blockIdx.y;
    const NormK nk = make_norm(minv, maxv);
    unsigned long long b = 0;
    for (uint32_t v = threadIdx.x; v < 65536u; v += blockDim.x)
        if (!same_float(norm_px(v, nk), norm_px(v, minv, maxv - minv))) b++;
    if (b) atomicAdd(&bad[3], b);
}

// ---- K1 + K4 --------------------------------------------------------------------------
// out = (sqrt(float(px)) - min) / (max - min), unclamped (img_normalize.comp:24-27).
// Vector path: 8 pixels per thread when N % 8 == 0 (dense u16 rows and pitched f32 rows both 16-byte aligned).
__global__ __launch_bounds__(256) void k_normalize(const uint16_t* __restrict__ px, float* __restrict__ out, int N, int pitch,
                                                   size_t plane, const uint32_t* __restrict__ minmax, int min_chain_exact) {
    const int img = blockIdx.z;
    float minv, maxv;
    chain_scalars(minmax, img, min_chain_exact, minv, maxv);
    const NormK nk = make_norm(minv, maxv);
    const uint16_t* p = px + (size_t)img * N * N;
    float* o = out + (size_t)img * plane;
    if ((N & 7) == 0 && ((uintptr_t)p & 15u) == 0) {
        const int vec_per_row = N >> 3;
        const size_t total = (size_t)vec_per_row * N;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
            const int y = (int)(i / vec_per_row), xv = (int)(i % vec_per_row);
            const uint4 q = *reinterpret_cast<const uint4*>(p + (size_t)y * N + xv * 8);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
            float r[8];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                r[2 * k] = norm_px(w[k] & 0xFFFFu, nk);
                r[2 * k + 1] = norm_px(w[k] >> 16, nk);
            }
            float* d = o + (size_t)y * pitch + xv * 8;
            *reinterpret_cast<float4*>(d) = make_float4(r[0], r[1], r[2], r[3]);
            *reinterpret_cast<float4*>(d + 4) = make_float4(r[4], r[5], r[6], r[7]);
        }
    } else {
        const size_t total = (size_t)N * N;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
            const int y = (int)(i / N), x = (int)(i % N);
            o[(size_t)y * pitch + x] = norm_px((uint32_t)p[i], nk);
        }
    }
}

__global__ void k_sqrt(const uint16_t* __restrict__ px, float* __restrict__ out, int N, int pitch, size_t plane) {
    const uint16_t* p = px + (size_t)blockIdx.z * N * N;
    float* o = out + (size_t)blockIdx.z * plane;
    const size_t total = (size_t)N * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / N), x = (int)(i % N);
        o[(size_t)y * pitch + x] = sqrtf((float)p[i]);
    }
}

// ---- K10 + K11: row pieces in sdev_parts.h ------------------------------------------------
// rows_per_wave must be a multiple of 16 (histogram runs start at y % 16 == 0).
// cov = (imageSize / 512) * 512: the part of the grid the reference's dispatch covers (src/vk_processing.cpp:2293-2295).
// Software-pipelined form: one row per trip; the raw row of trip y+1 is requested before trip y is
// computed and squared only when it enters the window (so the request never blocks the arithmetic).
// (sdev_march_block — one workgroup of the march — lives in sdev_parts.h: the paired reduce + band / sdev launch of kernels_expand_sd.hip uses it too)
template <bool HIST, bool A8>
__global__ __launch_bounds__(kBlockThreads) void k_sdev_hist_pf(const float* __restrict__ band, float* __restrict__ sdev, int S, int pitch,
                                                                size_t plane, uint32_t* __restrict__ hist, size_t hist_stride, int cov,
                                                                int rows_per_wave, int swz) {
    __shared__ uint32_t lh[kHistLdsWords];
    const size_t img = blockIdx.z;
    sdev_march_block<HIST, A8>(band + img * plane, sdev ? sdev + img * plane : nullptr, S, pitch, plane, HIST ? hist + img * hist_stride : nullptr, cov, rows_per_wave,
                               xcd_tile(swz), lh);
}

// ---- K10 + K11, one 16-row histogram run per workgroup ------------------------------------------------------
// The march above gives a wavefront a whole 16-row run (the `break` of noise_hist.comp:29-39 makes a run an ordered scan), i.e.
// 20 dependent row trips: 16 - 17 us per launch however small the level, and 16 rows x ~290 vector instructions would cost a
// lone wavefront ~9 us even with every load in flight. Here the four wavefronts of a workgroup share ONE run of 512 columns:
// each takes 4 of its rows, requests its 8 input rows at once (one round of latency), computes its 4 sdev rows and the
// per-row "bin != 0" lane masks; the wavefronts then exchange, through LDS and one barrier, the AND of their masks per owned
// column — the run of a column is alive at row r exactly when every earlier row of the run had a non-zero bin — and add
// their texels to the workgroup's LDS histogram. Same expressions (sdev_values, musica_noise_bin) as the march: same bits.
// Input rows are read twice (8 per 4 instead of 20 per 16), from the XCD's L2 the second time.
// (sdev_run_block lives in sdev_parts.h, like sdev_march_block)
template <bool A8>
__global__ __launch_bounds__(kBlockThreads) void k_sdev_hist_run(const float* __restrict__ band, float* __restrict__ sdev, int S, int pitch,
                                                                 size_t plane, uint32_t* __restrict__ hist, size_t hist_stride, int cov, int swz) {
    __shared__ uint32_t lh[kHistLdsWords];
    __shared__ unsigned long long nzw[kWavesPerBlock][8];
    const size_t img = blockIdx.z;
    sdev_run_block<A8>(band + img * plane, sdev ? sdev + img * plane : nullptr, S, pitch, plane, hist + img * hist_stride, cov, xcd_tile(swz), lh, nzw);
}

// The runs of SEVERAL levels in one launch (levels whose launch of their own would be a few dozen workgroups: a context
// that runs alone pays ~5 us of fixed cost per launch and the four sdev launches of a step depend on nothing but their
// own band image). Workgroups first .. first + strips * blocks - 1 of grid.x belong to level k; the finest level first.
template <bool A8>
__global__ __launch_bounds__(kBlockThreads) void k_sdev_hist_runs(const SdevRunLevels a, size_t hist_stride, int cov) {
    __shared__ uint32_t lh[kHistLdsWords];
    __shared__ unsigned long long nzw[kWavesPerBlock][8];
    int k = 0;
    for (int j = 1; j < a.n; j++) k = (int)blockIdx.x >= a.l[j].first ? j : k;   // block-uniform
    const SdevRunLevel& l = a.l[k];
    const int local = (int)blockIdx.x - l.first;
    Tile tile;
    tile.strip = local % l.strips;
    tile.segblock = local / l.strips;
    const size_t img = blockIdx.z;
    sdev_run_block<A8>(l.band + img * l.plane, l.sdev ? l.sdev + img * l.plane : nullptr, l.S, l.pitch, l.plane, l.hist + img * hist_stride, cov, tile, lh, nzw);
}

// EVERY level's sdev + noise-histogram pass in one launch, each level in the form its launch of its own would take (a march of l.rows rows per
// wavefront, or one 16-row run per workgroup where l.rows == 0). The four passes of a step depend on nothing but their own band image, and
// alone on the chip the marches of levels 0 and 1 are 2 and 1 wavefronts per SIMD of dependent arithmetic (8 x 2048^2: 45 + 23 us, and 14 us
// for the runs of levels 2 + 3): side by side they fill each other's issue slots.
template <bool A8>
__global__ __launch_bounds__(kBlockThreads) void k_sdev_hist_levels(const SdevRunLevels a, size_t hist_stride, int cov) {
    __shared__ uint32_t lh[kHistLdsWords];
    __shared__ unsigned long long nzw[kWavesPerBlock][8];
    int k = 0;
    for (int j = 1; j < a.n; j++) k = (int)blockIdx.x >= a.l[j].first ? j : k;   // block-uniform
    const SdevRunLevel& l = a.l[k];
    const int local = (int)blockIdx.x - l.first;
    // a level starts at a multiple of 8 workgroups and the grid is a multiple of 8 wide (the launcher pads), so workgroup `local` of a level
    // runs on XCD local % 8: the mapping of xcd_tile() — XCD x takes the x-th eighth of the level's blocks, strip by strip — where the blocks
    // divide by 8, the plain one elsewhere
    const int blocks = l.blocks;
    Tile tile;
    if (a.swz && (blocks & 7) == 0) {
        const int xcd = local & 7, j = local >> 3;
        tile.strip = j % l.strips;
        tile.segblock = xcd * (blocks >> 3) + j / l.strips;
    } else {
        tile.strip = local % l.strips;
        tile.segblock = local / l.strips;
    }
    if (tile.segblock >= blocks) return;   // padding (block-uniform)
    const size_t img = blockIdx.z;
    float* sd = l.sdev ? l.sdev + img * l.plane : nullptr;
    if (l.rows > 0) sdev_march_block<true, A8>(l.band + img * l.plane, sd, l.S, l.pitch, l.plane, l.hist + img * hist_stride, cov, l.rows, tile, lh);
    else sdev_run_block<A8>(l.band + img * l.plane, sd, l.S, l.pitch, l.plane, l.hist + img * hist_stride, cov, tile, lh, nzw);
}

// histogram only (kernel-level parity tests feed a foreign sdev image): same scan, no stencil.
// img_sdev.comp:10-35 literally: sum += pixel * pixel over m (x) outer, n (y) inner, out-of-image taps 0 (Q1), sqrt(sum / 25).
__global__ void k_sdev_literal(const float* __restrict__ band, float* __restrict__ sdev, int S, int pitch, size_t plane) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= S || y >= S) return;
    band += (size_t)blockIdx.z * plane;
    sdev += (size_t)blockIdx.z * plane;
    float sum = 0.0f;
#pragma unroll
    for (int m = 0; m < 5; m++) {
#pragma unroll
        for (int n = 0; n < 5; n++) {
            const int px = x + m - 2, py = y + n - 2;
            const float p = (px >= 0 && py >= 0 && px < S && py < S) ? band[(size_t)py * pitch + px] : 0.0f;
            sum = sum + p * p;   // :23
        }
    }
    sdev[(size_t)y * pitch + x] = sqrtf(sum / 25.0f);   // :30
}

__global__ __launch_bounds__(kBlockThreads) void k_noise_hist_only(const float* __restrict__ sdev, int S, int pitch, size_t plane,
                                                                   uint32_t* __restrict__ hist, size_t hist_stride, int cov) {
    __shared__ uint32_t lh[MUSICA_NOISE_BINS];
    for (int i = threadIdx.x; i < MUSICA_NOISE_BINS; i += blockDim.x) lh[i] = 0u;
    __syncthreads();
    const int img = blockIdx.z;
    sdev += (size_t)img * plane;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;   // one column per thread
    const int y0 = blockIdx.y * kHistArea;
    if (x < S && x < cov && y0 < cov) {
        for (int n = 0; n < kHistArea; n++) {
            const int y = y0 + n;
            const float cur = (y < S && y < cov) ? sdev[(size_t)y * pitch + x] : 0.0f;
            if (cur == 0.0f) break;
            const float adj = cur / kMaxNoiseValue;
            if (adj > 1.0f) break;
            const int bin = (int)(adj * (float)MUSICA_NOISE_BINS + 0.5f);
            if (bin == 0) break;
            if (bin < MUSICA_NOISE_BINS) atomicAdd(&lh[bin], 1u);
        }
    }
    __syncthreads();
    uint32_t* gh = hist + (size_t)img * hist_stride;
    for (int i = threadIdx.x; i < MUSICA_NOISE_BINS; i += blockDim.x) {
        const uint32_t v = lh[i];
        if (v) atomicAdd(&gh[i], v);
    }
}

// ---- K12 + K13 ------------------------------------------------------------------------
// Block-wide argmax with the lowest index winning ties (img_histogram_max.comp:20-29: strict `>`
// while scanning upwards; an all-zero histogram yields (0, 0)).
__device__ __forceinline__ unsigned long long argmax_key(uint32_t value, uint32_t index) {
    return value ? (((unsigned long long)value << 32) | (unsigned long long)(0xFFFFFFFFu - index)) : 0ull;
}
__device__ __forceinline__ unsigned long long block_max_u64(unsigned long long k, unsigned long long* scratch /*[16]*/) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long other = __shfl_xor(k, o);
        k = other > k ? other : k;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) scratch[wv] = k;
    __syncthreads();
    unsigned long long r = 0ull;
    for (int i = 0; i < nw; i++) r = scratch[i] > r ? scratch[i] : r;
    __syncthreads();
    return r;
}

// grid (levels, batch), 256 threads. Levels 0..3 take the argmax of their noise histogram; every
// level then builds its contrast curve (src/vk_processing.cpp:2284-2320).
__device__ __forceinline__ void noise_curves_block(int level, int img, const uint32_t* __restrict__ hist, size_t hist_stride,
                                                   musica_hist_max_point* __restrict__ maxpts, DevCurve* __restrict__ curves,
                                                   const musica_contrast_params* __restrict__ cparams, int levels,
                                                   DevCurveLut* __restrict__ luts, const uint32_t* __restrict__ minmax, int min_chain_exact,
                                                   int* __restrict__ thr090) {
    __shared__ unsigned long long scratch[16];
    // the `normalized <= 0.9` test of img_relevant.comp:56 as a threshold on the raw pixel (norm_threshold_090), once per image:
    // the level-0 expand kernel reads it when it accumulates the gradation histogram. The block of the coarsest level has the
    // least work of its own (two-point curve, no table).
    if (thr090 && level == levels - 1 && threadIdx.x == 255) {
        float minv, maxv;
        chain_scalars(minmax, img, min_chain_exact, minv, maxv);
        thr090[img] = norm_threshold_090(minv, maxv - minv);
    }
    __shared__ float sx[kCurveCap];
    __shared__ int sbucket[kCurveCap];
    __shared__ int sok;
    musica_hist_max_point mp;
    mp.maxValue = 0; mp.maxBin = 0;
    if (level <= MUSICA_CNR_LEVEL) {
        const uint32_t* h = hist + (size_t)img * hist_stride + (size_t)level * MUSICA_NOISE_BINS;
        unsigned long long k = 0ull;
        for (int i = threadIdx.x; i < MUSICA_NOISE_BINS; i += blockDim.x) {
            const unsigned long long ki = argmax_key(h[i], (uint32_t)i);
            k = ki > k ? ki : k;
        }
        k = block_max_u64(k, scratch);
        if (k) { mp.maxValue = (uint32_t)(k >> 32); mp.maxBin = 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull); }
    }
    DevCurve* c = curves + (size_t)img * levels + level;
    __shared__ float cx[kCurveCap], cy[kCurveCap];
    __shared__ int s_mono;
    if (threadIdx.x == 0) maxpts[(size_t)img * levels + level] = mp;
    // contrast_curve_generate.comp:56-94, one thread per curve point
    const float low = cparams[level].lowContrastFactor, high = cparams[level].highContrastFactor;
    const bool constant = (low == 1.0f);                                                       // :59
    const int npts = constant ? 2 : 33;
    if ((int)threadIdx.x < kCurveCap) {
        const int i = threadIdx.x;
        float x = 0.0f, y = 0.0f;
        if (constant) {
            if (i == 0) { x = 0.0f; y = high; }                                                 // :68
            if (i == 1) { x = 1.0f; y = high; }                                                 // :69
        } else if (i < 33) {
            const float p = (float)mp.maxBin * (1.0f / (float)MUSICA_NOISE_BINS) * kMaxNoiseValue;  // :71
            const int seg = i / 11;
            const uint32_t k = (uint32_t)(i - seg * 11);
            if (seg == 0) bezier_point(0.0f, 1.0f, p * 4.0f / 5.0f, low, p, low, k, x, y);                               // :72-76
            else if (seg == 1) bezier_point(p, low, p * 6.0f / 5.0f, low, p * 7.0f / 5.0f, low * 4.0f / 5.0f, k, x, y);   // :77-81
            else bezier_point(p * 7.0f / 5.0f, low * 4.0f / 5.0f, p * 2.0f, 1.0f, 1.0f, 1.0f, k, x, y);                   // :82-86
        }
        cx[i] = x;
        cy[i] = y;
    }
    __syncthreads();
    curve_store_parallel(c, cx, cy, &s_mono, npts, 0.0f, 0.0f, 0.0f);
    if (level >= MUSICA_COARSER_LEVELS_START) return;   // only the 33-point curves get a lookup table (block-uniform)
    __syncthreads();
    // ---- lookup tables for the expand kernel (see DevCurveLut) ----
    DevCurveLut* lut = luts + (size_t)img * MUSICA_COARSER_LEVELS_START + level;
    const int count = npts;
    const float range = cx[kLutTailFirst - 1] * 1.25f;                // 1.75 p: above x[22] = 1.4 p, below x[23] >= 1.49 p + 0.01
    const float inv_w = (float)kLutBuckets / range;
    __shared__ int scoarse[kCurveCap];
    if (threadIdx.x == 0) sok = (s_mono && count == kLutPoints && range > 0.0f && inv_w < 3.0e38f && cx[0] == 0.0f) ? 1 : 0;
    if ((int)threadIdx.x < kCurveCap) {
        const float x = cx[threadIdx.x];
        sx[threadIdx.x] = x;
        const float kf = x * inv_w;                                    // the expand kernel's expression for the fine bucket
        sbucket[threadIdx.x] = ((int)threadIdx.x < count && kf < (float)kLutBuckets) ? (int)kf : kLutBuckets;
        const float cf = fminf(x * 256.0f, (float)(kLutCoarse - 1));  // ... and for the coarse bucket
        scoarse[threadIdx.x] = (int)cf;
    }
    __syncthreads();
    if ((int)threadIdx.x < count) {   // the fine table covers exactly the abscissae 0..22
        const bool inside = sbucket[threadIdx.x] < kLutBuckets;
        if (inside != ((int)threadIdx.x < kLutTailFirst)) sok = 0;
    }
    for (int k = threadIdx.x; k < kLutBuckets + kLutCoarse; k += blockDim.x) {
        const bool fine = k < kLutBuckets;
        const int kb = fine ? k : k - kLutBuckets;
        const int first = fine ? 0 : kLutTailFirst;                     // the coarse table only ever sees s above x[0..22]
        int jlo = first, inb = 0;
        float xa = __builtin_huge_valf(), xb = __builtin_huge_valf();
        for (int i = first; i < count && i < kCurveCap; i++) {
            const int b = fine ? sbucket[i] : scoarse[i];
            if (b < kb) jlo++;
            else if (b == kb) {
                if (inb == 0) xa = sx[i];
                else if (inb == 1) xb = sx[i];
                inb++;
            }
        }
        if (inb > 2) sok = 0;
        lut->bucket[k] = make_float4(__int_as_float(jlo * 16), xa, xb, 0.0f);   // 16 * jlo: the byte offset of seg[jlo]
    }
    if ((int)threadIdx.x <= kLutPoints) {   // segment table: same slopes as DevCurve::m (linearFunction, contrast_curve_apply.comp:22-25)
        const int j = threadIdx.x;
        float4 sg = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (j == 0) sg = make_float4(cx[0], cy[0], 0.0f, 0.0f);
        else if (j < count) sg = make_float4(cx[j - 1], cy[j - 1], (cy[j] - cy[j - 1]) / (cx[j] - cx[j - 1]), 0.0f);
        lut->seg[j] = sg;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        lut->inv_w = inv_w;
        lut->ok = (uint32_t)sok;
        lut->pad0 = lut->pad1 = 0;
    }
}

__global__ __launch_bounds__(256) void k_noise_curves(const uint32_t* __restrict__ hist, size_t hist_stride,
                                                      musica_hist_max_point* __restrict__ maxpts, DevCurve* __restrict__ curves,
                                                      const musica_contrast_params* __restrict__ cparams, int levels,
                                                      DevCurveLut* __restrict__ luts, const uint32_t* __restrict__ minmax, int min_chain_exact,
                                                      int* __restrict__ thr090, int lev0) {
    noise_curves_block(lev0 + (int)blockIdx.x, blockIdx.y, hist, hist_stride, maxpts, curves, cparams, levels, luts, minmax, min_chain_exact, thr090);
}

// K12 + K13 and K15 in one launch: workgroups 0 .. levels-1 of an image build that level's curve, the others each
// normalise one 32 x 8 tile of the level-3 sdev image. img_cnr.comp only needs the level-3 noise mode, so a tile
// workgroup takes the argmax of that histogram itself (2048 bins, the same first-maximum key as K12) instead of
// waiting for another workgroup: one launch and one dependency less on the step's critical path.
__global__ __launch_bounds__(256) void k_curves_cnr(const uint32_t* __restrict__ hist, size_t hist_stride,
                                                    musica_hist_max_point* __restrict__ maxpts, DevCurve* __restrict__ curves,
                                                    const musica_contrast_params* __restrict__ cparams, int levels,
                                                    DevCurveLut* __restrict__ luts, const float* __restrict__ sdev, float* __restrict__ cnr,
                                                    int S, int pitch, size_t plane, int tiles_x, const uint32_t* __restrict__ minmax,
                                                    int min_chain_exact, int* __restrict__ thr090, int lev0) {
    // workgroups 0 .. levels-lev0-1 build the curves of levels lev0 .. levels-1 (lev0 = 1 when level 0 has a launch of its own)
    const int img = blockIdx.y;
    const int nlev = levels - lev0;
    if ((int)blockIdx.x < nlev) {   // block-uniform
        noise_curves_block(lev0 + (int)blockIdx.x, img, hist, hist_stride, maxpts, curves, cparams, levels, luts, minmax, min_chain_exact, thr090);
        return;
    }
    __shared__ unsigned long long scratch2[16];
    const uint32_t* h = hist + (size_t)img * hist_stride + (size_t)MUSICA_CNR_LEVEL * MUSICA_NOISE_BINS;
    unsigned long long k = 0ull;
    for (int i = threadIdx.x; i < MUSICA_NOISE_BINS; i += blockDim.x) {
        const unsigned long long ki = argmax_key(h[i], (uint32_t)i);
        k = ki > k ? ki : k;
    }
    k = block_max_u64(k, scratch2);
    const uint32_t maxBin = k ? 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull) : 0u;
    float ref = (float)maxBin * (1.0f / (float)MUSICA_NOISE_BINS) * kMaxNoiseValue;            // img_cnr.comp:22
    if (ref == 0.0f) ref = (1.0f / (float)MUSICA_NOISE_BINS) * kMaxNoiseValue;                  // :25
    const int t = (int)blockIdx.x - nlev;
    const int x = (t % tiles_x) * 32 + (int)(threadIdx.x & 31), y = (t / tiles_x) * 8 + (int)(threadIdx.x >> 5);
    if (x >= S || y >= S) return;
    const size_t o = (size_t)img * plane + (size_t)y * pitch + x;
    const float v = sdev[o] / ref;                                                              // :31
    cnr[o] = v / kMaxCnrValue;                                                                  // :43
}

// ---- K15 ------------------------------------------------------------------------------
__global__ void k_cnr(const float* __restrict__ sdev, float* __restrict__ cnr, int S, int pitch, size_t plane,
                      const musica_hist_max_point* __restrict__ maxpts, int levels) {
    const int img = blockIdx.z;
    const musica_hist_max_point mp = maxpts[(size_t)img * levels + MUSICA_CNR_LEVEL];
    float ref = (float)mp.maxBin * (1.0f / (float)MUSICA_NOISE_BINS) * kMaxNoiseValue;          // img_cnr.comp:22
    if (ref == 0.0f) ref = (1.0f / (float)MUSICA_NOISE_BINS) * kMaxNoiseValue;                  // :25
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= S || y >= S) return;
    const size_t o = (size_t)img * plane + (size_t)y * pitch + x;
    const float v = sdev[o] / ref;                                                              // :31
    cnr[o] = v / kMaxCnrValue;                                                                  // :43
}

// Per-image summary (musica_stats) written on the device so the batch driver can all-gather it with
// RCCL without a host round trip. mean_cnr = mean(cnr image) * 256 — what test/mean_cnr/script.py:13-24
// prints for a cnr.bmp dump — summed in double with a fixed partition and a fixed tree (deterministic):
// k_stats_partial: nb workgroups per image, workgroup b sums the rows of its 16 wavefronts (wavefront b * 16 + w takes rows
// b * 16 + w, + 16 nb, ...; lanes across a row, 4 rows x 4 column chunks = 16 independent loads per trip) into partial[img][b];
// k_stats: one workgroup per image adds the nb partial sums in index order and fills the row. (One workgroup per image for
// everything took 20 us for the 256^2 cnr image of a 2048^2 input and 440 us for the 1024^2 one of an 8192^2 input.)
__global__ __launch_bounds__(1024) void k_stats_partial(const float* __restrict__ cnr, int S, int pitch, size_t plane, double* __restrict__ partial) {
    __shared__ double part[1024];
    const int img = blockIdx.y, nb = gridDim.x;
    const float* p = cnr + (size_t)img * plane;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x >> 6) * nb;
    double acc = 0.0;
    for (int y = blockIdx.x * (blockDim.x >> 6) + wv; y < S; y += 4 * nw) {
        for (int xb = 0; xb < S; xb += 256) {
            float v[4][4];
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int yy = y + k * nw, xx = xb + c * 64 + lane;
                    v[k][c] = (yy < S && xx < S) ? p[(size_t)yy * pitch + xx] : 0.0f;
                }
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc += (double)v[k][c];
        }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 512; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)img * nb + blockIdx.x] = part[0];
}

__global__ __launch_bounds__(64) void k_stats(const double* __restrict__ partial, int nb, int S,
                                              const uint32_t* __restrict__ minmax, int min_chain_exact,
                                              const musica_hist_max_point* __restrict__ noise_max, int levels,
                                              const musica_hist_max_point* __restrict__ grad_max, const DevCurve* __restrict__ gcurve,
                                              musica_stats* __restrict__ out, uint32_t image_id_base, uint32_t image_id_stride) {
    const int img = blockIdx.x;
    double part[1];
    part[0] = 0.0;
    if (threadIdx.x == 0)
        for (int b = 0; b < nb; b++) part[0] += partial[(size_t)img * nb + b];
    if (threadIdx.x != 0) return;
    musica_stats st;
    st.image_id = image_id_base + (uint32_t)img * image_id_stride;
    chain_scalars(minmax, img, min_chain_exact, st.min_sqrt, st.max_sqrt);
    for (int l = 0; l < 4; l++) {
        const musica_hist_max_point mp = noise_max[(size_t)img * levels + l];
        st.noise_max_bin[l] = mp.maxBin;
        st.noise_max_value[l] = mp.maxValue;
    }
    st.grad_max_bin = grad_max[img].maxBin;
    st.grad_max_value = grad_max[img].maxValue;
    st.mean_cnr = (float)(part[0] / ((double)S * (double)S) * 256.0);
    st.t0 = gcurve[img].t0;
    st.ta = gcurve[img].ta;
    st.t1 = gcurve[img].t1;
    out[img] = st;
}

// ======================================================================================
// host-side launchers
// ======================================================================================

void launch_clear(hipStream_t st, uint32_t* minmax, uint32_t* noise_hist, uint32_t* grad_hist, uint32_t* clahe_hist, int batch, uint32_t* grad_hist_b,
                  uint32_t* gzero) {
    int n = batch * 4 * MUSICA_NOISE_BINS;
    if (clahe_hist) n = max(n, batch * MUSICA_CLAHE_TILES * MUSICA_CLAHE_TILES * MUSICA_CLAHE_BINS);
    hipLaunchKernelGGL(k_clear, dim3((n + 255) / 256), dim3(256), 0, st, minmax, noise_hist, grad_hist, clahe_hist, batch, grad_hist_b, gzero);
}

void launch_minmax(hipStream_t st, const uint16_t* px, int N, uint32_t* minmax, uint32_t* slots, uint32_t* ticket, int batch,
                   uint32_t* noise_hist, uint32_t* grad_hist, uint32_t* grad_hist_b, uint32_t* gzero, uint32_t* clahe_hist) {
    const size_t count = (size_t)N * N;
    // one trip of kMinMaxLoads loads per lane (2048^2: 64 blocks of 128 KiB) up to 512 blocks per image, then more trips: a block
    // ends with one returning atomic on its image's ticket, ~11 ns apart on one address
    // 256 blocks per image at most (a block ends with one returning atomic on its image's ticket, ~11 ns apart on one address), 128
    // with a batch behind them. Measured with rocprofv3 (profiles/r03_minmax_shapes.txt, r04_minmax.txt): 8 x 2048^2 21 - 22 us, one
    // 2048^2 image 5.8 - 7.3 us, one 8192^2 image 39 - 40 us whatever the shape (256 ... 1024 threads, 1 ... 8 loads per lane, 64 ... 2048
    // blocks per image, chunks contiguous or a grid-width apart, images skewed against each other). Round 3 blamed the previous step's
    // gradation apply (268 MB of stores still draining); round 4: a plain read behind a write-heavy launch does take 15.6 us instead of 10.5,
    // and this launch's streaming loop alone (no clears, no ticket tail, the grid of a plain copy kernel) 19.5 us inside the pipeline.
    const size_t per_block = (size_t)kMinMaxThreads * kMinMaxLoads;
    size_t blocks = (count / 8 + per_block - 1) / per_block;
    const size_t maxb = batch >= 4 ? 128 : 256;
    if (blocks < 1) blocks = 1;
    if (blocks > maxb) blocks = maxb;
    ClearArgs ca{noise_hist, grad_hist, grad_hist_b, gzero, clahe_hist};
    hipLaunchKernelGGL(k_minmax_u16<kMinMaxLoads>, dim3((unsigned)blocks, 1, batch), dim3(kMinMaxThreads), 0, st, px, count, minmax, slots, ticket, ca);
}

void launch_normalize(hipStream_t st, const uint16_t* px, float* out, const LevelDesc& l0, const uint32_t* minmax,
                      int min_chain_exact, int batch) {
    const size_t items = (l0.S & 7) == 0 ? (size_t)l0.S * l0.S / 8 : (size_t)l0.S * l0.S;
    int blocks = (int)std::min<size_t>((items + 255) / 256, (size_t)4096);
    hipLaunchKernelGGL(k_normalize, dim3(blocks, 1, batch), dim3(256), 0, st, px, out, l0.S, l0.pitch, l0.plane, minmax,
                       min_chain_exact);
}

void launch_sqrt(hipStream_t st, const uint16_t* px, float* out, const LevelDesc& l0, int batch) {
    hipLaunchKernelGGL(k_sqrt, dim3(1024, 1, batch), dim3(256), 0, st, px, out, l0.S, l0.pitch, l0.plane);
}

void launch_sdev_hist(hipStream_t st, const float* band, float* sdev, const LevelDesc& l, uint32_t* hist, size_t hist_stride, int cov,
                      int batch, int rows_per_wave) {
    const int strips = (l.S + kStripCols - 1) / kStripCols;
    if (rows_per_wave <= 0) {   // one 16-row run per workgroup (k_sdev_hist_run)
        const dim3 grid(strips, (l.S + kHistArea - 1) / kHistArea, batch);
        if ((l.S & 7) == 0) hipLaunchKernelGGL((k_sdev_hist_run<true>), grid, dim3(kBlockThreads), 0, st, band, sdev, l.S, l.pitch, l.plane, hist, hist_stride, cov, xcd_swizzle_on());
        else hipLaunchKernelGGL((k_sdev_hist_run<false>), grid, dim3(kBlockThreads), 0, st, band, sdev, l.S, l.pitch, l.plane, hist, hist_stride, cov, xcd_swizzle_on());
        return;
    }
    const int segs = (l.S + rows_per_wave - 1) / rows_per_wave;
    const dim3 grid(strips, (segs + kWavesPerBlock - 1) / kWavesPerBlock, batch);
    if ((l.S & 7) == 0) hipLaunchKernelGGL((k_sdev_hist_pf<true, true>), grid, dim3(kBlockThreads), 0, st, band, sdev, l.S, l.pitch, l.plane, hist, hist_stride, cov, rows_per_wave, xcd_swizzle_on());
    else hipLaunchKernelGGL((k_sdev_hist_pf<true, false>), grid, dim3(kBlockThreads), 0, st, band, sdev, l.S, l.pitch, l.plane, hist, hist_stride, cov, rows_per_wave, xcd_swizzle_on());
}

void launch_sdev_hist_runs(hipStream_t st, int n, const float* const* band, float* const* sdev, const LevelDesc* lv, uint32_t* const* hist,
                           size_t hist_stride, int cov, int batch) {
    SdevRunLevels a;
    a.n = n;
    int first = 0;
    bool a8 = true;
    for (int k = 0; k < n; k++) {
        const LevelDesc& l = lv[k];
        const int strips = (l.S + kStripCols - 1) / kStripCols;
        a.l[k] = SdevRunLevel{band[k], sdev[k], hist[k], l.plane, l.S, l.pitch, strips, first, 0, 0};
        first += strips * ((l.S + kHistArea - 1) / kHistArea);
        a8 = a8 && (l.S & 7) == 0;
    }
    for (int k = n; k < kSdevRunLevelsMax; k++) a.l[k] = a.l[0];
    a.swz = 0;
    const dim3 grid(first, 1, batch);
    if (a8) hipLaunchKernelGGL((k_sdev_hist_runs<true>), grid, dim3(kBlockThreads), 0, st, a, hist_stride, cov);
    else hipLaunchKernelGGL((k_sdev_hist_runs<false>), grid, dim3(kBlockThreads), 0, st, a, hist_stride, cov);
}

void launch_sdev_hist_levels(hipStream_t st, int n, const float* const* band, float* const* sdev, const LevelDesc* lv, uint32_t* const* hist,
                             const int* rows, size_t hist_stride, int cov, int batch) {
    SdevRunLevels a;
    a.n = n;
    int first = 0;
    bool a8 = true;
    for (int k = 0; k < n; k++) {
        const LevelDesc& l = lv[k];
        const int strips = (l.S + kStripCols - 1) / kStripCols;
        int blocks;   // workgroups per strip
        if (rows[k] > 0) { const int segs = (l.S + rows[k] - 1) / rows[k]; blocks = (segs + kWavesPerBlock - 1) / kWavesPerBlock; }
        else blocks = (l.S + kHistArea - 1) / kHistArea;
        a.l[k] = SdevRunLevel{band[k], sdev[k], hist[k], l.plane, l.S, l.pitch, strips, first, rows[k] > 0 ? rows[k] : 0, blocks};
        first += (strips * blocks + 7) & ~7;   // every level starts on XCD 0
        a8 = a8 && (l.S & 7) == 0;
    }
    for (int k = n; k < kSdevRunLevelsMax; k++) a.l[k] = a.l[0];
    a.swz = xcd_swizzle_on();
    const dim3 grid(first, 1, batch);
    if (a8) hipLaunchKernelGGL((k_sdev_hist_levels<true>), grid, dim3(kBlockThreads), 0, st, a, hist_stride, cov);
    else hipLaunchKernelGGL((k_sdev_hist_levels<false>), grid, dim3(kBlockThreads), 0, st, a, hist_stride, cov);
}

// sdev alone (no histogram): the stored image of a level whose hot path does not store it, for getters / dumps / the stage entry points
void launch_sdev_only(hipStream_t st, const float* band, float* sdev, const LevelDesc& l, int batch) {
    const int strips = (l.S + kStripCols - 1) / kStripCols, rows = 16;
    const int segs = (l.S + rows - 1) / rows;
    const dim3 grid(strips, (segs + kWavesPerBlock - 1) / kWavesPerBlock, batch);
    if ((l.S & 7) == 0) hipLaunchKernelGGL((k_sdev_hist_pf<false, true>), grid, dim3(kBlockThreads), 0, st, band, sdev, l.S, l.pitch, l.plane, (uint32_t*)nullptr, (size_t)0, 0, rows, xcd_swizzle_on());
    else hipLaunchKernelGGL((k_sdev_hist_pf<false, false>), grid, dim3(kBlockThreads), 0, st, band, sdev, l.S, l.pitch, l.plane, (uint32_t*)nullptr, (size_t)0, 0, rows, xcd_swizzle_on());
}

void launch_sdev_literal(hipStream_t st, const float* band, float* sdev, const LevelDesc& l, int batch) {
    hipLaunchKernelGGL(k_sdev_literal, dim3((l.S + 31) / 32, (l.S + 7) / 8, batch), dim3(32, 8), 0, st, band, sdev, l.S, l.pitch, l.plane);
}

void launch_noise_hist_only(hipStream_t st, const float* sdev, const LevelDesc& l, uint32_t* hist, size_t hist_stride, int cov, int batch) {
    hipLaunchKernelGGL(k_noise_hist_only, dim3((l.S + kBlockThreads - 1) / kBlockThreads, (l.S + kHistArea - 1) / kHistArea, batch),
                       dim3(kBlockThreads), 0, st, sdev, l.S, l.pitch, l.plane, hist, hist_stride, cov);
}

void launch_noise_curves(hipStream_t st, const uint32_t* hist, size_t hist_stride, musica_hist_max_point* maxpts, DevCurve* curves,
                         const musica_contrast_params* cparams, int levels, int batch, DevCurveLut* luts, const uint32_t* minmax, int min_chain_exact,
                         int* thr090, int lev0, int nlev) {
    hipLaunchKernelGGL(k_noise_curves, dim3(nlev > 0 ? nlev : levels - lev0, batch), dim3(256), 0, st, hist, hist_stride, maxpts, curves, cparams, levels, luts,
                       minmax, min_chain_exact, thr090, lev0);
}

void launch_curves_cnr(hipStream_t st, const uint32_t* hist, size_t hist_stride, musica_hist_max_point* maxpts, DevCurve* curves,
                       const musica_contrast_params* cparams, int levels, int batch, DevCurveLut* luts, const float* sdev, float* cnr,
                       const LevelDesc& l3, const uint32_t* minmax, int min_chain_exact, int* thr090, int lev0) {
    const int tiles_x = (l3.S + 31) / 32, tiles_y = (l3.S + 7) / 8;
    hipLaunchKernelGGL(k_curves_cnr, dim3(levels - lev0 + tiles_x * tiles_y, batch), dim3(256), 0, st, hist, hist_stride, maxpts, curves, cparams, levels,
                       luts, sdev, cnr, l3.S, l3.pitch, l3.plane, tiles_x, minmax, min_chain_exact, thr090, lev0);
}

void launch_cnr(hipStream_t st, const float* sdev, float* cnr, const LevelDesc& l3, const musica_hist_max_point* maxpts, int levels,
                int batch) {
    hipLaunchKernelGGL(k_cnr, dim3((l3.S + 31) / 32, (l3.S + 7) / 8, batch), dim3(32, 8), 0, st, sdev, cnr, l3.S, l3.pitch, l3.plane,
                       maxpts, levels);
}

void launch_selftest_exact_math(hipStream_t st, unsigned long long* d_bad4) {
    hipLaunchKernelGGL(k_selftest_sqrt, dim3(8192), dim3(256), 0, st, d_bad4);
    hipLaunchKernelGGL(k_selftest_norm, dim3(256, 256), dim3(256), 0, st, d_bad4);
}

void launch_stats(hipStream_t st, const float* cnr, const LevelDesc& l3, const uint32_t* minmax, int min_chain_exact,
                  const musica_hist_max_point* noise_max, int levels, const musica_hist_max_point* grad_max, const DevCurve* gcurve,
                  musica_stats* out, uint32_t image_id_base, uint32_t image_id_stride, int batch, double* partial /* [batch][kStatsMaxBlocks] */) {
    const size_t px = (size_t)l3.S * l3.S;
    const int nb = (int)std::min<size_t>(std::max<size_t>(px / 16384, 1), (size_t)kStatsMaxBlocks);
    hipLaunchKernelGGL(k_stats_partial, dim3(nb, batch), dim3(1024), 0, st, cnr, l3.S, l3.pitch, l3.plane, partial);
    hipLaunchKernelGGL(k_stats, dim3(batch), dim3(64), 0, st, (const double*)partial, nb, l3.S, minmax, min_chain_exact, noise_max, levels,
                       grad_max, gcurve, out, image_id_base, image_id_stride);
}

}  // namespace musica
