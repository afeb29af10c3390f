// sdev_parts.h — the per-row pieces of the fused 5x5 RMS + noise-histogram march (img_sdev.comp + noise_hist.comp),
// shared by k_sdev_hist* (kernels_analysis.hip) and the band+sdev kernel (kernels_pyramid.hip).
#pragma once

#include "kernels_common.h"
#include "exact_math.h"

namespace musica {

// ---- K10 + K11 ------------------------------------------------------------------------
// sdev(x, y) = sqrt( sum_{5x5} band^2 / 25 ), taps outside the image are 0 and the divisor stays 25
// (img_sdev.comp:14-30). ORDER_FAST: vertical chain of squares, then horizontal chain.
// The noise histogram (noise_hist.comp) is accumulated on the fly: the reference's thread (gx, gy)
// walks its 16x16 area column by column (m = x outer, n = y inner) and `break`s out of a column at
// the first pixel that is 0, > 0.1 or lands in bin 0 — i.e. each (column, 16-row run) is an
// independent early-exit scan. A lane of this kernel owns 8 columns and marches down rows, so it
// sees every run in exactly that order: one `alive` bit per owned column, re-armed every 16 rows.
// Bins are privatised in LDS and flushed with one global atomic per non-empty bin. The noise values of an image crowd into a
// few bins (phantoms: the fullest bin takes 5 % of the texels at level 0, 17 % at level 1, 25-28 % at levels 2-3), and lanes
// of one ds_add that hit the same address are served one after the other — so the block keeps kHistCopies copies of the
// histogram, lane l adds into copy l % kHistCopies (copies kHistCopyStride words apart: same bin, different banks), and the
// flush sums them. Three copies = 24.9 KB per workgroup: six workgroups fit a CU's 160 KB, so the small sdev launches of
// levels 2 and 3 find room beside the four-per-CU workgroups of level 0 (with four copies, 33 KB, they waited it out).
constexpr int kHistCopies = 3;
constexpr int kHistCopyStride = MUSICA_NOISE_BINS + 8;                       // % 32 == 8: bin b of the copies sits in banks b, b+8, b+16
constexpr int kHistLdsWords = kHistCopies * kHistCopyStride + 64;           // + one scratch word per lane for the branch-free adds
__device__ __forceinline__ void hist_lds_clear(uint32_t* lh) {
    for (int i = threadIdx.x; i < kHistLdsWords; i += blockDim.x) lh[i] = 0u;
}
__device__ __forceinline__ void hist_lds_flush(const uint32_t* lh, uint32_t* __restrict__ gh) {
    for (int i = threadIdx.x; i < MUSICA_NOISE_BINS; i += blockDim.x) {
        uint32_t v = 0u;
#pragma unroll
        for (int k = 0; k < kHistCopies; k++) v += lh[k * kHistCopyStride + i];
        if (v) atomicAdd(&gh[i], v);
    }
}

struct SRow {
    float q[8];    // squares of columns c .. c+7
    float l0, l1;  // squares of columns c-2, c-1   (lane 0 of a strip that is not the first)
    float h0, h1;  // squares of columns c+8, c+9   (lane 63)
};

// Byte offsets of one lane inside a row; kOob where the access would leave the image (reads as 0: Q1).
struct SCfg {
    int c;
    uint32_t off0, off1;  // 16-byte groups c .. c+3, c+4 .. c+7
    uint32_t off_l;       // c-2, c-1 (lane 0)
    uint32_t off_r;       // c+8, c+9 (lane 63)
    int valid;            // number of in-image columns among the lane's 8 (pad columns of a pitched row must read as 0)
    bool lane0, lane63;
};

__device__ __forceinline__ SCfg make_scfg(int strip, int lane, int S) {
    SCfg g;
    g.c = strip * kStripCols + lane * kLaneCols;
    g.lane0 = lane == 0;
    g.lane63 = lane == 63;
    g.valid = min(max(S - g.c, 0), 8);
    g.off0 = g.c < S ? (uint32_t)g.c * 4u : kOob;
    g.off1 = g.c + 4 < S ? (uint32_t)(g.c + 4) * 4u : kOob;
    g.off_l = (g.lane0 && g.c >= 2 && g.c < S) ? (uint32_t)(g.c - 2) * 4u : kOob;
    g.off_r = (g.lane63 && g.c + 8 < S) ? (uint32_t)(g.c + 8) * 4u : kOob;
    return g;
}

// row_off = kOob for rows outside the image. Columns >= S inside the last 16-byte group and the odd
// right-halo column are masked to 0 by `valid` / the c+9 test when the row is squared.
struct SRaw {
    float4 a, d;
    float2 l, h;
};
__device__ __forceinline__ void load_sraw(SRaw& r, const Buf& b, uint32_t row_off, const SCfg& g) {
    // kOob has only bit 31 set and every in-image offset is < 2^31, so an OR keeps "either one out of range" out of range
    r.a = bload4(b, (g.off0 + row_off) | ((g.off0 | row_off) & kOob));
    r.d = bload4(b, (g.off1 + row_off) | ((g.off1 | row_off) & kOob));
    r.l = bload2(b, (g.off_l + row_off) | ((g.off_l | row_off) & kOob));
    r.h = bload2(b, (g.off_r + row_off) | ((g.off_r | row_off) & kOob));
}
// A8: the side is a multiple of 8, so a lane's 8 columns are all inside the image or all outside (then every load of the lane is
// out of range and reads 0) and the right-halo pair c+8, c+9 likewise: no per-column masks.
template <bool A8>
__device__ __forceinline__ void square_srow(SRow& r, const SRaw& w, const SCfg& g, int S) {
    const float v[8] = {w.a.x, w.a.y, w.a.z, w.a.w, w.d.x, w.d.y, w.d.z, w.d.w};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float t = (A8 || j < g.valid) ? v[j] : 0.0f;
        r.q[j] = t * t;
    }
    r.l0 = w.l.x * w.l.x; r.l1 = w.l.y * w.l.y;
    r.h0 = w.h.x * w.h.x;
    const float h1 = (A8 || g.c + 9 < S) ? w.h.y : 0.0f;
    r.h1 = h1 * h1;
}
template <bool A8>
__device__ __forceinline__ void load_srow(SRow& r, const Buf& b, uint32_t row_off, const SCfg& g, int S) {
    SRaw w;
    load_sraw(w, b, row_off, g);
    square_srow<A8>(r, w, g, S);
}

// `mask` is a wave-uniform lane mask (an SGPR pair): lane l gets if_set when bit l is set. Written as the one instruction it
// is — `(mask >> lane) & 1` costs a 64-bit shift, an AND and a compare per use.
__device__ __forceinline__ uint32_t select_by_lane_mask(unsigned long long mask, uint32_t if_set, uint32_t if_clear) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
    return r;
}

__device__ __forceinline__ float sum5(float a, float b, float c, float d, float e) {
    float acc = a;
    acc = acc + b;
    acc = acc + c;
    acc = acc + d;
    acc = acc + e;
    return acc;
}

// The 8 sdev values of a lane for one output row from its five rows of squares (img_sdev.comp:17-30 in MUSICA_ORDER_FAST).
__device__ __forceinline__ void sdev_values(const SRow& r0, const SRow& r1, const SRow& r2, const SRow& r3, const SRow& r4, const SCfg& g, float (&s)[8]) {
    float q[8];
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] = sum5(r0.q[j], r1.q[j], r2.q[j], r3.q[j], r4.q[j]);
    const float ql0 = sum5(r0.l0, r1.l0, r2.l0, r3.l0, r4.l0), ql1 = sum5(r0.l1, r1.l1, r2.l1, r3.l1, r4.l1);
    const float qh0 = sum5(r0.h0, r1.h0, r2.h0, r3.h0, r4.h0), qh1 = sum5(r0.h1, r1.h1, r2.h1, r3.h1, r4.h1);
    float a6 = from_left_lane(q[6]), a7 = from_left_lane(q[7]);
    float b0 = from_right_lane(q[0]), b1 = from_right_lane(q[1]);
    if (g.lane0) { a6 = ql0; a7 = ql1; }    // zeros at the image's left edge (loads out of range)
    if (g.lane63) { b0 = qh0; b1 = qh1; }   // zeros beyond the right edge
    // lanes right of the image hold q == 0, so the last in-image lane reads zeros from its neighbour
    s[0] = sum5(a6, a7, q[0], q[1], q[2]);
    s[1] = sum5(a7, q[0], q[1], q[2], q[3]);
    s[2] = sum5(q[0], q[1], q[2], q[3], q[4]);
    s[3] = sum5(q[1], q[2], q[3], q[4], q[5]);
    s[4] = sum5(q[2], q[3], q[4], q[5], q[6]);
    s[5] = sum5(q[3], q[4], q[5], q[6], q[7]);
    s[6] = sum5(q[4], q[5], q[6], q[7], b0);
    s[7] = sum5(q[5], q[6], q[7], b0, b1);
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = musica_div25(s[j]);  // img_sdev.comp:30 (exact x / 25, exact_math.h)
    musica_sqrt8(s);                                         // img_sdev.comp:30 (exact sqrt, exact_math.h)
}
template <bool A8>
__device__ __forceinline__ void sdev_store(const float (&s)[8], const SCfg& g, float* __restrict__ drow, const Buf& db, uint32_t row_off) {
    if (A8) {   // whole 16-byte groups, lanes outside the image carry out-of-range offsets: no branch
        bstore4(db, (g.off0 + row_off) | (g.off0 & kOob), make_float4(s[0], s[1], s[2], s[3]));
        bstore4(db, (g.off1 + row_off) | (g.off1 & kOob), make_float4(s[4], s[5], s[6], s[7]));
    } else if (g.valid == 8) {
        *reinterpret_cast<float4*>(drow + g.c) = make_float4(s[0], s[1], s[2], s[3]);
        *reinterpret_cast<float4*>(drow + g.c + 4) = make_float4(s[4], s[5], s[6], s[7]);
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (j < g.valid) drow[g.c + j] = s[j];
    }
}

// One output row of sdev from its five rows of squares + the histogram scan of that row. y is wave-uniform.
// alive[j] / start[j]: lane masks (wave-uniform 64-bit values, i.e. scalar registers) — bit l = the run of column c + j of lane l
// is still counting / starts alive. As per-lane bools the compiler packed them into bytes of two vector registers and spent
// ~7 vector instructions per texel unpacking, re-arming and repacking them; as lane masks the bookkeeping is two scalar ANDs.
template <bool HIST, bool A8>
__device__ __forceinline__ void sdev_row(const SRow& r0, const SRow& r1, const SRow& r2, const SRow& r3, const SRow& r4, const SCfg& g, int S,
                                         int y, int cov, float* __restrict__ drow, const Buf& db, uint32_t row_off, uint32_t* lh,
                                         unsigned long long (&alive)[8], const unsigned long long (&start)[8], bool store = true) {
    float s[8];
    sdev_values(r0, r1, r2, r3, r4, g, s);
    if (store) sdev_store<A8>(s, g, drow, db, row_off);   // wave-uniform: a level whose expand launch computes sdev itself only needs the histogram
    // noise_hist.comp:20-47, branch-free. A run adds until its first `break` (bin 0): alive[j] afterwards is exactly
    // "this texel is counted". A dead column adds into the lane's scratch word; bin 2048 (out of the histogram
    // image, dropped by Q1 without breaking) is a pad word behind the copy (never flushed), so it needs no test of its own. Columns
    // outside the image / the dispatch coverage start every run dead (their texel would read 0 -> break).
    if (HIST && y < cov) {
        const int lane = threadIdx.x & 63;
        const uint32_t copy_b = (uint32_t)((lane % kHistCopies) * kHistCopyStride) * 4u, scratch_b = (uint32_t)(kHistCopies * kHistCopyStride + lane) * 4u;
        const bool rearm = (y & (kHistArea - 1)) == 0;   // wave-uniform
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int bin = musica_noise_bin(s[j]);                           // 0 = break (:29, :33, :39); exact (exact_math.h)
            alive[j] = (rearm ? start[j] : alive[j]) & __ballot(bin != 0);
            const uint32_t addr = select_by_lane_mask(alive[j], copy_b + (uint32_t)bin * 4u, scratch_b);
            atomicAdd(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(lh) + addr), 1u);   // :45
        }
    }
}

// start[j] of sdev_row for this lane set: bit l = column c + j of lane l is inside the image and the dispatch coverage.
__device__ __forceinline__ void sdev_start_masks(unsigned long long (&start)[8], unsigned long long (&alive)[8], const SCfg& g, int cov) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        start[j] = __ballot(j < g.valid && g.c + j < cov);
        alive[j] = 0ull;
    }
}


// ---- one workgroup of each form of the pass (kernels_analysis.hip has the commentary of the forms) ----
// one workgroup of the march: `band` / `sdev` / `hist` are the image's own planes and histogram, `tile` its strip and block of four segments
template <bool HIST, bool A8>
__device__ __forceinline__ void sdev_march_block(const float* __restrict__ band, float* __restrict__ sdev, int S, int pitch, size_t plane,
                                                 uint32_t* __restrict__ hist, int cov, int rows_per_wave, const Tile tile, uint32_t* lh) {
    hist_lds_clear(lh);
    __syncthreads();
    const Buf bb = make_buf(band, plane * 4);
    const bool store = sdev != nullptr;   // nullptr: histogram only (the level's expand launch computes sdev itself, k_expand_fast<.., SD>)
    const Buf db = make_buf(store ? sdev : band, store ? plane * 4 : 0);
    const int lane = threadIdx.x & 63;
    const int seg = __builtin_amdgcn_readfirstlane((int)(tile.segblock * kWavesPerBlock + (threadIdx.x >> 6)));   // wave-uniform: row arithmetic stays on the scalar unit
    const int y0 = seg * rows_per_wave;
    const SCfg g = make_scfg(tile.strip, lane, S);
    const uint32_t rb = (uint32_t)pitch * 4u;
    auto roff = [&](int row) -> uint32_t { return (row >= 0 && row < S) ? (uint32_t)row * rb : kOob; };
    if (y0 < S) {
        const int y1 = min(y0 + rows_per_wave, S);
        // (Five window slots used round-robin in a trip unrolled five times — no `w0 = w1; ...` copies, 68 of the 284 vector
        // instructions of a row — was measured and dropped: 116 registers instead of 96, 4 wavefronts per SIMD instead of 5,
        // and the launch is 6 % slower. Two or four raw rows in flight per wavefront instead of one: no change either.)
        SRow w0, w1, w2, w3, w4;
        SRaw raw;
        load_srow<A8>(w0, bb, roff(y0 - 2), g, S);
        load_srow<A8>(w1, bb, roff(y0 - 1), g, S);
        load_srow<A8>(w2, bb, roff(y0), g, S);
        load_srow<A8>(w3, bb, roff(y0 + 1), g, S);
        load_sraw(raw, bb, roff(y0 + 2), g);
        unsigned long long alive[8], start[8];
        sdev_start_masks(start, alive, g, cov);
        for (int y = y0; y < y1; y++) {
            square_srow<A8>(w4, raw, g, S);
            // rows past the image carry an out-of-range offset: no access. (Non-temporal loads for the rows no neighbouring segment reads,
            // the trick that took 7 % off the metric kernel, change nothing here: 8 x 2048^2 33.6 - 37.7 us either way, 8192^2 55 -> 62 us.)
            load_sraw(raw, bb, roff(y + 3), g);
            sdev_row<HIST, A8>(w0, w1, w2, w3, w4, g, S, y, cov, store ? sdev + (size_t)y * pitch : nullptr, db, (uint32_t)y * rb, lh, alive, start, store);
            w0 = w1; w1 = w2; w2 = w3; w3 = w4;
        }
    }
    if (HIST) {
        __syncthreads();
        hist_lds_flush(lh, hist);
    }
}

constexpr int kRunRowsPerWave = kHistArea / kWavesPerBlock;   // 4
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// one run: `band` / `sdev` / `hist` are the image's own plane and histogram, `tile` the run's 512-column strip and 16-row block
template <bool A8>
__device__ __forceinline__ void sdev_run_block(const float* __restrict__ band, float* __restrict__ sdev, int S, int pitch, size_t plane,
                                               uint32_t* __restrict__ hist, int cov, const Tile tile, uint32_t* lh,
                                               unsigned long long (*nzw)[8]) {
    hist_lds_clear(lh);
    const Buf bb = make_buf(band, plane * 4);
    const bool store = sdev != nullptr;   // nullptr: histogram only
    const Buf db = make_buf(store ? sdev : band, store ? plane * 4 : 0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int y0 = tile.segblock * kHistArea + wave * kRunRowsPerWave;   // wave-uniform
    const SCfg g = make_scfg(tile.strip, lane, S);
    const uint32_t rb = (uint32_t)pitch * 4u;
    auto roff = [&](int row) -> uint32_t { return (row >= 0 && row < S) ? (uint32_t)row * rb : kOob; };
    unsigned long long nz[kRunRowsPerWave][8];
    int bin[kRunRowsPerWave][8];
    unsigned long long mine[8];
#pragma unroll
    for (int j = 0; j < 8; j++) mine[j] = ~0ull;
    const int nrows = min(max(S - y0, 0), kRunRowsPerWave);   // rows of this wavefront inside the image
    if (nrows > 0) {
        SRaw raw[kRunRowsPerWave + 4];
#pragma unroll
        for (int k = 0; k < kRunRowsPerWave + 4; k++) load_sraw(raw[k], bb, roff(y0 - 2 + k), g);
        SRow w[kRunRowsPerWave + 4];
#pragma unroll
        for (int k = 0; k < kRunRowsPerWave + 4; k++) square_srow<A8>(w[k], raw[k], g, S);
#pragma unroll
        for (int r = 0; r < kRunRowsPerWave; r++) {
            if (r < nrows) {   // wave-uniform
                float s[8];
                sdev_values(w[r], w[r + 1], w[r + 2], w[r + 3], w[r + 4], g, s);
                if (store) sdev_store<A8>(s, g, sdev + (size_t)(y0 + r) * pitch, db, (uint32_t)(y0 + r) * rb);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    bin[r][j] = musica_noise_bin(s[j]);      // 0 = break (noise_hist.comp:29, :33, :39)
                    nz[r][j] = __ballot(bin[r][j] != 0);
                    mine[j] &= nz[r][j];
                }
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) nzw[wave][j] = mine[j];
    }
    __syncthreads();   // the cleared histogram and every wavefront's masks
    if (nrows > 0 && y0 < cov) {   // the dispatch covers whole runs (cov is a multiple of 512)
        const uint32_t copy_b = (uint32_t)((lane % kHistCopies) * kHistCopyStride) * 4u, scratch_b = (uint32_t)(kHistCopies * kHistCopyStride + lane) * 4u;
        unsigned long long alive[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            alive[j] = __ballot(j < g.valid && g.c + j < cov);   // columns outside the image / the coverage start dead
            for (int v = 0; v < wave; v++) alive[j] &= uniform64(nzw[v][j]);   // the same value in every lane: back into scalar registers
        }
#pragma unroll
        for (int r = 0; r < kRunRowsPerWave; r++) {
            if (r < nrows) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    alive[j] &= nz[r][j];
                    const uint32_t addr = select_by_lane_mask(alive[j], copy_b + (uint32_t)bin[r][j] * 4u, scratch_b);
                    atomicAdd(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(lh) + addr), 1u);   // :45
                }
            }
        }
    }
    __syncthreads();
    hist_lds_flush(lh, hist);
}


}  // namespace musica
