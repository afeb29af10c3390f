// kernels_pyramid.hip — the Laplacian-pyramid kernels of the MUSICA path for gfx950 (CDNA4).
//
//   k_reduce_*  : img_smooth.comp + img_downsample.comp fused            (K5 + K6, the metric kernel)
//   k_band_*    : img_upsample.comp + img_smooth_upsampled.comp + img_difference.comp fused (K7 + K8 + K9)
//   k_expand_*  : contrast_curve_apply.comp + noise_reduction.comp + img_upsample.comp +
//                 img_smooth_upsampled.comp + img_addition.comp fused     (K14 + K16 + K7 + K8 + K17)
//
// Streaming ("fast") form, used when the level side S is a multiple of 8: one 64-lane wavefront
// owns a strip of 512 fine columns (8 per lane, two 16-byte loads per row) and marches down a
// segment of rows keeping the row window in registers; the vertical 5-tap pass runs straight
// from the loaded registers, the horizontal pass takes its 2+1 neighbour columns from the
// adjacent lanes with DPP wave shifts (no LDS, no barrier) and from two halo loads at the strip
// edges. All memory access goes through buffer descriptors with per-lane byte offsets, so a lane
// that has nothing to load carries an out-of-range offset instead of sitting behind an `if`:
// the loop bodies are branch-free and every load of a trip is in flight before the first use.
// Every HBM byte is touched by exactly one 16-byte access of one lane, plus the 3-row / 3-column
// halos that stay in the XCD's L2. Reflect-101 borders are register selects.
//
// Generic form (any S, including the 1..7-pixel tail of the reference's 12-level pyramid):
// one thread per output texel with the shaders' mirror()/out-of-bounds rules applied per tap.
//
// Both forms evaluate exactly the oracle's MUSICA_ORDER_FAST arithmetic.
#include <stdlib.h>
#include "kernels_common.h"
#include "sdev_parts.h"
#include "launchers.h"

// kernels_expand_sd.hip includes this file with MUSICA_PYRAMID_SD_ONLY defined: the device code of the expand march compiled a second time,
// without the SLP vectoriser, for the launches that compute sdev in registers (k_expand_fast<.., SD>: 172 registers that way, 211 with the
// pairs the vectoriser builds — 3 or 2 wavefronts per SIMD). Everything that is not a template — the generic kernels and every launcher —
// exists once, in the translation unit without that macro.
#ifndef MUSICA_PYRAMID_SD_ONLY
#define MUSICA_PYRAMID_FULL 1
#endif

namespace musica {

// Per-lane geometry of a 512-column strip, as byte offsets inside one row.
struct LaneCfg {
    int c;               // first fine column of this lane
    bool active;         // c < S
    bool left_mirror;    // lane 0 of the first strip: columns -2, -1 mirror onto 2, 1
    bool last_active;    // the lane holding column S-1: column S mirrors onto S-2
    bool lane0, lane63;
    uint32_t off;        // fine columns c .. c+7            (kOob when the lane is right of the image)
    uint32_t off_l;      // fine columns c-2, c-1            (lane 0 of a strip that is not the first, else kOob)
    uint32_t off_r;      // fine column  c+8                 (lane 63 with more image to its right, else kOob)
    uint32_t coff;       // coarse columns c/2 .. c/2+3
    uint32_t coff_l;     // coarse column  c/2-1
    uint32_t coff_r;     // coarse column  c/2+4
};

__device__ __forceinline__ LaneCfg make_cfg(int strip, int lane, int S) {
    LaneCfg g;
    const int c0 = strip * kStripCols;
    g.c = c0 + lane * kLaneCols;
    g.active = g.c < S;
    g.lane0 = lane == 0;
    g.lane63 = lane == 63;
    const bool left_load = g.lane0 && c0 > 0;
    const bool right_load = g.lane63 && (g.c + kLaneCols < S);
    g.left_mirror = g.lane0 && c0 == 0;
    g.last_active = g.active && (g.c + kLaneCols >= S);
    g.off = g.active ? (uint32_t)g.c * 4u : kOob;
    g.off_l = left_load ? (uint32_t)(g.c - 2) * 4u : kOob;
    g.off_r = right_load ? (uint32_t)(g.c + kLaneCols) * 4u : kOob;
    g.coff = g.active ? (uint32_t)(g.c >> 1) * 4u : kOob;
    g.coff_l = left_load ? (uint32_t)((g.c >> 1) - 1) * 4u : kOob;
    g.coff_r = right_load ? (uint32_t)((g.c >> 1) + 4) * 4u : kOob;
    return g;
}

// ======================================================================================
// K5 + K6: out(xo, yo) = sum_m w[m] * ( sum_n w[n] * in(mirror(2xo+m-2), mirror(2yo+n-2)) )
// ======================================================================================

struct RowR {
    float v[8];
    float hl0, hl1;  // columns c0-2, c0-1 of the strip (lane 0 only)
    float hr;        // column c0+512 of the strip (lane 63 only)
};

__device__ __forceinline__ void load8(float d[8], const Buf& b, uint32_t off) {
    const float4 a = bload4(b, off), c = bload4(b, off + 16u);
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = c.x; d[5] = c.y; d[6] = c.z; d[7] = c.w;
}
__device__ __forceinline__ void store8(const Buf& b, uint32_t off, const float d[8]) {
    bstore4(b, off, make_float4(d[0], d[1], d[2], d[3]));
    bstore4(b, off + 16u, make_float4(d[4], d[5], d[6], d[7]));
}
// a + b for two byte offsets either of which may be kOob ("nothing to access"): the sum of two markers wraps to 0, so the marker bit is ORed back in
__device__ __forceinline__ uint32_t oob_add(uint32_t a, uint32_t b) { return (a + b) | ((a | b) & kOob); }
__device__ __forceinline__ void store8_nt(const Buf& b, uint32_t off, const float d[8]) {
    bstore4_nt(b, off, make_float4(d[0], d[1], d[2], d[3]));
    bstore4_nt(b, off + 16u, make_float4(d[4], d[5], d[6], d[7]));
}

__device__ __forceinline__ void load_row(RowR& r, const Buf& b, uint32_t row_off, const LaneCfg& g) {
    load8(r.v, b, g.off + row_off);
    const float2 h = bload2(b, g.off_l + row_off);
    r.hl0 = h.x; r.hl1 = h.y;
    r.hr = bload1(b, g.off_r + row_off);
}

// One output row from its five input rows (vertical pass in registers, horizontal pass with the
// neighbour columns taken from the adjacent lanes). AUX: cache policy of the store (0 plain, 2 non-temporal).
template <int AUX = 0>
__device__ __forceinline__ void reduce_row(const RowR& r0, const RowR& r1, const RowR& r2, const RowR& r3, const RowR& r4,
                                           const LaneCfg& g, const Buf& ob, uint32_t out_off) {
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
    v4f o;
    o.x = chain5(vl6, vl7, v[0], v[1], v[2]);
    o.y = chain5(v[0], v[1], v[2], v[3], v[4]);
    o.z = chain5(v[2], v[3], v[4], v[5], v[6]);
    o.w = chain5(v[4], v[5], v[6], v[7], vr0);
    llvm_buffer_store_v4f32(o, ob.r, (int)(g.coff + out_off), 0, AUX);   // output columns c/2 .. c/2+3; dropped for lanes right of the image
}

// ---- the metric kernel: img_smooth.comp:18-50 + img_downsample.comp:10-20 of an f32 level -----------------------
// One 256-thread workgroup owns a tile of 512 input columns x kDmaRows output rows. Its 2 * kDmaRows + 3 input rows are
// brought in by LDS-DMA (`buffer_load_dwordx4 ... lds`: 1 KiB per wave-instruction straight from HBM / L2 into LDS, no
// VGPR destination and no register write-back), all of them requested before anything else happens — the four
// wavefronts issue the 2 KiB rows piecewise round-robin — plus one 4-byte DMA for the three halo columns (c0-2, c0-1,
// c0+512) of every row. Each wavefront then waits for its own DMAs, the workgroup meets at ONE barrier, and every
// wavefront computes one output row: five 16-byte-pair LDS reads per lane, the vertical 5-tap pass in registers, the
// horizontal pass with its 2 + 1 neighbour columns from the adjacent lanes by DPP wave shifts (halo columns on lane 0 /
// 63), one non-temporal 16-byte store per lane (the output is not read again by this launch: a plain store leaves the
// lines dirty in the XCD's L2 and their write-back competes with the incoming rows — 18.7 -> 15.9 us at 4096^2). The five rows
// of a tile that no other tile reads come in with the non-temporal policy (15.9 -> 14.85 us, see the loop).
// 22 KiB of LDS per workgroup: 7 workgroups = 154 KiB of rows in flight per CU, and the grid is fine-grained enough
// (4096 workgroups at 4096^2) that the tail of the launch is short. Measured from HBM on MI355X (DESIGN.md §6,
// profiles/r03_*): 4096^2 14.85 us = 0.706 of 8 TB/s (register-staged march of rounds 1-2: 19.3; a plain copy of the same
// traffic shape: 16.1, with non-temporal stores 15.4), 8192^2 51.6 us = 0.81 (march 65.6, copy 58.7).
// Rows the mirror sends outside [0, S) (only when S < 3 could that happen; the fast path needs S >= 8) do not occur; the
// last tile of an image whose So is not a multiple of kDmaRows re-requests valid rows and drops the surplus output rows.
// TAG only separates the launch sites in profiler output: 0 / 1 = level 0 / levels >= 1 of a pipeline that does not fuse
// reduce + band, 2 = stand-alone musica_k_reduce, 4 / 5 = stand-alone rotating over distinct planes at sides <= 4096 / above
// (musica_k_reduce_timed_rot: bench.py's 4096^2 and 8192^2 measurements stay apart in a profiler's per-symbol averages).
constexpr int kDmaRows = 4;                           // output rows per workgroup (one per wavefront)
constexpr int kDmaInRows = 2 * kDmaRows + 3;          // input rows of a tile
constexpr int kDmaPieces = 2 * kDmaInRows;            // 1 KiB pieces (half rows)
constexpr int kDmaPiecesPerWave = (kDmaPieces + kWavesPerBlock - 1) / kWavesPerBlock;
constexpr int kDmaHaloLanes = 24;                     // lanes of the halo DMA (2 per input row, rounded up to a multiple of 8)
static_assert(kDmaRows == kWavesPerBlock && kDmaInRows * 2 <= kDmaHaloLanes, "tile shape");

// Workgroup -> tile for the metric kernel when the strip count is a power of two of at least 2 and the tile rows divide: XCD k (the
// workgroups with linear id % 8 == k) owns a REGION of the image, two strips wide (region_width) and 1 / v of the tile rows tall where
// (strips / 2) v = 8, and walks it row of tiles by row of tiles,
// strip by strip: the tiles left / right and above / below a tile are on the same XCD and dispatched back to back, so the halo
// columns (the neighbouring strip's lines) and the shared halo rows are found in that XCD's L2 — with one strip per XCD (the plain
// mapping at 8 strips) every halo column was another XCD's line and came from the fabric a second time.
// strip PAIRS (quads from 32 strips): 4096^2 = 4 pairs across x 2 halves down, 8192^2 = 8 pairs x the full height, 2048^2 = 2 pairs x 4 quarters.
// Measured at 4096^2 / 8192^2 (us per launch from HBM, fetch + write traffic): pairs 14.62 / 52.5, 87.2 MB; regions of half the strips x
// a quarter of the rows 15.0 / 53.7, 85.2 MB (less fetch, slower); one strip per XCD (round 3) 14.86 / 52.0, 91.3 MB.
__host__ __device__ __forceinline__ unsigned region_width(unsigned strips) { return strips >= 32u ? strips >> 3 : 2u; }
__device__ __forceinline__ Tile xcd_region_tile() {
    const unsigned gx = gridDim.x, gy = gridDim.y;           // strips, tile rows
    const unsigned lin = blockIdx.x + gx * blockIdx.y;
    const unsigned xcd = lin & 7u, j = lin >> 3;
    const unsigned w = region_width(gx);                     // region width in strips
    const unsigned across = gx / w, v = 8u / across;          // regions across x regions down = 8
    const unsigned rh = gy / v;                               // tile rows per region
    const unsigned rx = xcd % across, ry = xcd / across;
    Tile t;
    t.strip = (int)(rx * w + j % w);
    t.segblock = (int)(ry * rh + j / w);
    return t;
}

// M0 = LDS byte address the wave-instruction writes to (wave-uniform); lane l lands at M0 + 16 l (dwordx4) / M0 + 4 l (dword).
// An asm load is invisible to hipcc's s_waitcnt bookkeeping: the kernel waits with its own vmcnt(0) below.
// (M0 is compiler-reserved: it is saved and restored inside the statement that changes it.)
__device__ __forceinline__ void dma16(const Buf& b, uint32_t voff, uint32_t lds_byte) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(voff), "s"(b.r) : "memory");
}
// the same with the non-temporal cache policy: for bytes no other workgroup will ask for
__device__ __forceinline__ void dma16_nt(const Buf& b, uint32_t voff, uint32_t lds_byte) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen nt lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(voff), "s"(b.r) : "memory");
}
__device__ __forceinline__ void dma4(const Buf& b, uint32_t voff, uint32_t lds_byte) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(voff), "s"(b.r) : "memory");
}

template <int TAG>
__global__ __launch_bounds__(kBlockThreads) void k_reduce_dma(const float* __restrict__ in, float* __restrict__ out,
                                                              int S, int pitch, size_t in_plane, int So, int opitch,
                                                              size_t out_plane, int swz) {
    __shared__ __attribute__((aligned(16))) float tile_s[kDmaInRows * kStripCols + kDmaHaloLanes * 4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const Tile tile = swz == 2 ? xcd_region_tile() : xcd_tile(swz);
    const int yo0 = tile.segblock * kDmaRows;
    const Buf ib = make_buf(in + (size_t)blockIdx.z * in_plane, in_plane * 4);
    const Buf ob = make_buf(out + (size_t)blockIdx.z * out_plane, out_plane * 4);
    const LaneCfg g = make_cfg(tile.strip, lane, S);
    const int c0 = tile.strip * kStripCols;
    const int hi = S - 1;
    const uint32_t rb = (uint32_t)pitch * 4u, orb = (uint32_t)opitch * 4u;
    const uint32_t lds0 = (uint32_t)(uintptr_t)tile_s;   // LDS byte address of the tile (low half of the flat address)
    // local row k of the tile = input row mirror(2 * yo0 - 2 + k) (img_smooth.comp:10-16), kept inside the image
    // halo columns of every row as two aligned 16-byte pieces — columns c0-4 .. c0-1 and c0+512 .. c0+515 — by ONE dwordx4 DMA of
    // 2 x 11 lanes (wavefront 0; entry 2 k + side, 4 floats each). Until round 4 a dword DMA gathered the three columns a row needs
    // one by one: 33 requests per tile, each a fabric request of its own on a line of the neighbouring strip (+9 % of fetch, PMC);
    // 22 now, and with the region mapping below the neighbour's line is usually in this XCD's L2 already.
    if (wave == 0 && lane < kDmaHaloLanes) {
        const int k = lane >> 1, side = lane & 1;
        const int y = min(max(mirror_idx(2 * yo0 - 2 + min(k, kDmaInRows - 1), hi), 0), hi);
        const int col = side ? c0 + kStripCols : c0 - 4;
        const bool ok = k < kDmaInRows && col >= 0 && col < S;
        dma16(ib, ok ? (uint32_t)y * rb + (uint32_t)col * 4u : kOob, lds0 + (uint32_t)(kDmaInRows * kStripCols) * 4u);
    }
    const uint32_t lane_off = (uint32_t)(c0 + lane * 4) * 4u;   // a piece = 256 columns, 4 per lane
#pragma unroll
    for (int i = 0; i < kDmaPiecesPerWave; i++) {
        const int q = min(i * kWavesPerBlock + wave, kDmaPieces - 1);   // a wave without a last piece repeats the tile's last one
        const int k = q >> 1, half = q & 1;
        const int y = min(max(mirror_idx(2 * yo0 - 2 + k, hi), 0), hi);
        const bool ok = c0 + half * 256 + lane * 4 < S;
        const uint32_t src = ok ? (uint32_t)y * rb + lane_off + (uint32_t)half * 1024u : kOob, dst = lds0 + (uint32_t)(k * kStripCols + half * 256) * 4u;
        // rows 0-2 and 8-10 of a tile are also the tile above's / below's: default policy, so the second reader finds them in the
        // XCD's L2; rows 3-7 are read by nobody else: non-temporal, they do not push the shared rows out (a CU's seven tiles x 32 CUs
        // are 4.9 MB in flight against 4 MiB of L2). 4096^2: 15.9 -> 14.85 us, 8192^2: 55.3 -> 51.6 us; nt on every row: 17.9 us.
        if (k >= 3 && k < 2 * kDmaRows) dma16_nt(ib, src, dst);   // wave-uniform
        else dma16(ib, src, dst);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float* hal = tile_s + kDmaInRows * kStripCols;
    RowR w[5];
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const int k = 2 * wave + r;
        const float4 a = *reinterpret_cast<const float4*>(tile_s + k * kStripCols + lane * 8);
        const float4 b = *reinterpret_cast<const float4*>(tile_s + k * kStripCols + lane * 8 + 4);
        w[r].v[0] = a.x; w[r].v[1] = a.y; w[r].v[2] = a.z; w[r].v[3] = a.w; w[r].v[4] = b.x; w[r].v[5] = b.y; w[r].v[6] = b.z; w[r].v[7] = b.w;
        w[r].hl0 = hal[8 * k + 2]; w[r].hl1 = hal[8 * k + 3]; w[r].hr = hal[8 * k + 4];   // columns c0-2, c0-1 | c0+512
    }
    const int yo = yo0 + wave;
    if (yo < So) reduce_row<2>(w[0], w[1], w[2], w[3], w[4], g, ob, (uint32_t)yo * orb);   // wave-uniform
}

// ---- level 0 straight from the raw pixels ------------------------------------------------------
// The reference materialises sqrt and normalized images (6 + 8 B/px of traffic) before the pyramid
// starts. Here the level-0 kernels read the uint16 pixels (2 B/px) and apply img_sqrt.comp:15 +
// img_normalize.comp:24 to every pixel they load — the same three IEEE operations k_normalize executes,
// so the values are bit-identical — and the normalized image is only produced on demand.
__device__ __forceinline__ void norm8(float d[8], float4 m, const NormK& nk) {
    const uint32_t w[4] = {__float_as_uint(m.x), __float_as_uint(m.y), __float_as_uint(m.z), __float_as_uint(m.w)};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        d[2 * k] = norm_px(w[k] & 0xFFFFu, nk);
        d[2 * k + 1] = norm_px(w[k] >> 16, nk);
    }
}

// One thread per output texel, any S >= 1. OOB loads (only reachable when S < 3, where the
// reference's unclamped mirror() leaves the image) return 0.
// Branch-free on purpose: the load always happens (texel (0, 0) stands in for an out-of-image tap) and the test only selects, so
// the 25 taps of a stencil are requested together instead of one branch + load + wait each (k_tiny_tail: 41 -> 14 us).
__device__ __forceinline__ float ld0(const float* __restrict__ im, int pitch, int S, int x, int y) {
    const bool in = x >= 0 && y >= 0 && x < S && y < S;
    const float v = im[(size_t)(in ? y : 0) * pitch + (in ? x : 0)];
    return in ? v : 0.0f;
}

// ---- the shaders' literal arithmetic order (MUSICA_FLAG_REFERENCE_ORDER) ---------------------------------------
// img_smooth.comp:32-45 / img_smooth_upsampled.comp:32-45 accumulate the 25 taps m (x) outer, n (y) inner,
// `pixel += weight[m] * weight[n] [* 4.0] * load`, starting from 0; img_sdev.comp:17-30 the 25 squares likewise.
// The generic kernels take a wave-uniform `ref` argument and then evaluate exactly that sequence (the oracle's
// MUSICA_ORDER_REFERENCE); speed is secondary here.
__device__ __forceinline__ float w5(int i) { return i == 0 ? W0 : i == 1 ? W1 : i == 2 ? W2 : i == 3 ? W3 : W4; }

__device__ __forceinline__ float smooth_literal(const float* __restrict__ in, int pitch, int S, int x, int y) {
    const int hi = S - 1;
    float pixel = 0.0f;
#pragma unroll
    for (int m = 0; m < 5; m++) {
        const int xm = mirror_idx(x + m - 2, hi);
#pragma unroll
        for (int n = 0; n < 5; n++) pixel = pixel + (w5(m) * w5(n)) * ld0(in, pitch, S, xm, mirror_idx(y + n - 2, hi));   // img_smooth.comp:43
    }
    return pixel;
}

// smooth + downsample at coarse (xo, yo) for any S (generic form)
__device__ __forceinline__ float reduce_generic_at(const float* __restrict__ in, int pitch, int S, int xo, int yo, int ref) {
    if (ref) return smooth_literal(in, pitch, S, 2 * xo, 2 * yo);   // img_downsample.comp:15 of the literally smoothed image
    const int hi = S - 1;
    float v[5];
#pragma unroll
    for (int m = 0; m < 5; m++) {
        const int x = mirror_idx(2 * xo + m - 2, hi);   // outside the image (S < 3): five zero taps, chain5 = +0
        v[m] = chain5(ld0(in, pitch, S, x, mirror_idx(2 * yo - 2, hi)), ld0(in, pitch, S, x, mirror_idx(2 * yo - 1, hi)),
                      ld0(in, pitch, S, x, mirror_idx(2 * yo, hi)), ld0(in, pitch, S, x, mirror_idx(2 * yo + 1, hi)),
                      ld0(in, pitch, S, x, mirror_idx(2 * yo + 2, hi)));
    }
    return chain5(v[0], v[1], v[2], v[3], v[4]);
}
#ifdef MUSICA_PYRAMID_FULL
__global__ void k_reduce_generic(const float* __restrict__ in, float* __restrict__ out, int S, int pitch,
                                 size_t in_plane, int So, int opitch, size_t out_plane, int ref) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x;
    const int yo = blockIdx.y * blockDim.y + threadIdx.y;
    if (xo >= So || yo >= So) return;
    in += (size_t)blockIdx.z * in_plane;
    out += (size_t)blockIdx.z * out_plane;
    out[(size_t)yo * opitch + xo] = reduce_generic_at(in, pitch, S, xo, yo, ref);
}
#endif

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

__device__ __forceinline__ void load_crow(CRow& r, const Buf& b, uint32_t row_off, const LaneCfg& g) {
    const float4 a = bload4(b, g.coff + row_off);
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    // one halo column per edge lane: lane 0 reads j0-1, lane 63 reads j0+4 (every other lane carries kOob on both) — one load, one register;
    // hl and hr hold the same value and the two chains the callers build on them are one
    r.hl = r.hr = bload1(b, (g.lane0 ? g.coff_l : g.coff_r) + row_off);
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

// ======================================================================================
// Level 0 in one march: K1 + K4 + K5 + K6 and K7 + K8 + K9 — smooth + downsample AND the band-pass image — from one read
// of the fine image. Separately, k_reduce_u16_pf and k_band_fast<., true> each read and normalise the whole image
// (2 x (2 B/px + ~14 instructions per pixel)) and the band launch reads the coarse image back (1 B/px); here a wavefront
// that marches down its strip keeps the last three coarse rows it produced in registers and emits the band rows 2k, 2k+1
// as soon as coarse row k+1 exists: 7 B/px instead of 3 + 7, one launch instead of two.
// What a strip cannot take from its neighbours' registers it recomputes: the coarse rows k0-1 and k1 above / below its
// segment (two extra fine rows at either end) and, on lane 0 / lane 63, the coarse column left / right of the strip
// (four raw halo pixels per row instead of two / one). Every value is produced by the expressions of reduce_row() and
// lowpass_pair(), so both outputs are bit-identical to the two-kernel path (tested against the oracle and against it).
// ======================================================================================
// (Round 4 measured the halo columns spread over the lanes of the first and last quad, one column per lane, with quad-broadcast DPP moves:
// 540 -> 503 vector instructions per trip and the same 49 - 53 us at 8 x 2048^2 — but 96 - 115 us instead of 89 - 91 us at 8192^2, eight 2-byte
// halo requests per row instead of two 8-byte ones; the round-3 form below stays. profiles/r04_rb0_experiments.txt)
struct FRow {
    float v[8];   // normalized pixels c .. c+7
    float h[4];   // c-4 .. c-1 on lane 0, c+8 .. c+11 on lane 63 (strips with a neighbour on that side; 0 elsewhere)
};
struct RawF {
    float4 m;     // 8 raw uint16 (bit pattern)
    float2 h;     // 4 raw halo pixels
};
__device__ __forceinline__ void load_raw_f(RawF& r, const Buf& b, uint32_t row_off, uint32_t off, uint32_t off_h) {
    r.m = bload4(b, off + row_off);
    r.h = bload2(b, off_h + row_off);
}
__device__ __forceinline__ void convert_f(FRow& r, const RawF& w, const NormK& nk) {
    norm8(r.v, w.m, nk);
    const uint32_t a = __float_as_uint(w.h.x), b = __float_as_uint(w.h.y);
    r.h[0] = norm_px(a & 0xFFFFu, nk);
    r.h[1] = norm_px(a >> 16, nk);
    r.h[2] = norm_px(b & 0xFFFFu, nk);
    r.h[3] = norm_px(b >> 16, nk);
}
// One coarse row (the lane's 4 columns + the halo column of an edge lane) from its five fine rows: reduce_row()'s arithmetic.
__device__ __forceinline__ void coarse_row(CRow& cr, const FRow& r0, const FRow& r1, const FRow& r2, const FRow& r3, const FRow& r4, const LaneCfg& g) {
    float v[8], vh[4];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = chain5(r0.v[j], r1.v[j], r2.v[j], r3.v[j], r4.v[j]);
#pragma unroll
    for (int j = 0; j < 4; j++) vh[j] = chain5(r0.h[j], r1.h[j], r2.h[j], r3.h[j], r4.h[j]);
    float vl6 = from_left_lane(v[6]);
    float vl7 = from_left_lane(v[7]);
    float vr0 = from_right_lane(v[0]);
    if (g.lane0) {
        vl6 = g.left_mirror ? v[2] : vh[2];  // column -2 -> 2, -1 -> 1 (img_smooth.comp:13); else columns c-2, c-1
        vl7 = g.left_mirror ? v[1] : vh[3];
    }
    if (g.last_active) vr0 = v[6];           // column S -> S-2 (img_smooth.comp:12)
    else if (g.lane63) vr0 = vh[0];          // column c+8
    cr.v[0] = chain5(vl6, vl7, v[0], v[1], v[2]);
    cr.v[1] = chain5(v[0], v[1], v[2], v[3], v[4]);
    cr.v[2] = chain5(v[2], v[3], v[4], v[5], v[6]);
    cr.v[3] = chain5(v[4], v[5], v[6], v[7], vr0);
    // the coarse column next to the strip: j0-1 (fine c-4 .. c) on lane 0, j0+4 (fine c+6 .. c+10) on lane 63 — what the
    // neighbouring strip's edge lane computes as its own column
    const bool r = g.lane63;
    const float ch = chain5(r ? v[6] : vh[0], r ? v[7] : vh[1], r ? vh[0] : vh[2], r ? vh[1] : vh[3], r ? vh[2] : v[0]);
    cr.hl = ch;
    cr.hr = ch;
}
__device__ __forceinline__ void band_pair(const CRow& a, const CRow& b, const CRow& c, const FRow& fe, const FRow& fo, const LaneCfg& g,
                                          const Buf& bb, uint32_t off_e, uint32_t off_o) {
    float lowE[8], lowO[8], be[8], bo[8];
    lowpass_pair(a, b, c, g, lowE, lowO);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        be[j] = fe.v[j] - lowE[j];   // img_difference.comp:15
        bo[j] = fo.v[j] - lowO[j];
    }
    store8_nt(bb, g.off + off_e, be);   // non-temporal: the band image is next read by another launch (C5 level 0: 108 -> 90 us)
    store8_nt(bb, g.off + off_o, bo);
}

// bit j: fe.v[j] <= 0.9, bit 8 + j: fo.v[j] <= 0.9 — the shader's own comparison on the normalized value (NaN: false)
__device__ __forceinline__ uint32_t le090_bits(const FRow& fe, const FRow& fo) {
    uint32_t m = 0u;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        m |= fe.v[j] <= 0.90f ? 1u << j : 0u;
        m |= fo.v[j] <= 0.90f ? 1u << (8 + j) : 0u;
    }
    return m;
}

__device__ __forceinline__ void load_f(FRow& r, const Buf& b, uint32_t row_off, uint32_t off, uint32_t off_h) {
    load8(r.v, b, off + row_off);
    const float4 h = bload4(b, off_h + row_off);
    r.h[0] = h.x; r.h[1] = h.y; r.h[2] = h.z; r.h[3] = h.w;
}

// rows_per_wave counts coarse rows. grid: x = strips, y = ceil(segments / 4), z = batch.
// U16: the fine image is the raw uint16 input, normalised on the fly (level 0); the two rows of the next trip are requested as
// raw pixels (6 registers) one trip ahead. Otherwise the fine image is f32 (levels >= 1): the rows are loaded where they are
// needed (a prefetch would cost 24 registers and a wavefront per SIMD; these levels are small and L2-resident).
// one workgroup of the launch: `tile` = its strip and block of four segments, `img` = its image (k_reduce_band below; the paired launch of
// kernels_expand_sd.hip gives some of its workgroups this role)
template <bool U16>
__device__ __forceinline__ void reduce_band_block(const void* __restrict__ fine, float* __restrict__ down, float* __restrict__ band,
                                                  int S, int pitch, size_t plane, int Sc, int cpitch, size_t cplane,
                                                  int rows_per_wave, const uint32_t* __restrict__ minmax, int min_chain_exact,
                                                  uint16_t* __restrict__ le090, const Tile tile, const int img) {
    const int lane = threadIdx.x & 63;
    const int seg = __builtin_amdgcn_readfirstlane((int)(tile.segblock * kWavesPerBlock + (threadIdx.x >> 6)));   // wave-uniform: row arithmetic stays on the scalar unit
    const int k0 = seg * rows_per_wave;
    if (k0 >= Sc) return;  // wave-uniform
    const int k1 = min(k0 + rows_per_wave, Sc);
    NormK nk = make_norm(0.0f, 1.0f);
    if (U16) {
        float minv, maxv;
        chain_scalars(minmax, img, min_chain_exact, minv, maxv);
        nk = make_norm(minv, maxv);
    }
    const Buf ib = U16 ? make_buf(reinterpret_cast<const uint16_t*>(fine) + (size_t)img * S * S, (size_t)S * S * 2)
                       : make_buf(reinterpret_cast<const float*>(fine) + (size_t)img * plane, plane * 4);
    const Buf db = make_buf(down + (size_t)img * cplane, cplane * 4);
    const Buf bb = make_buf(band + (size_t)img * plane, plane * 4);
    const LaneCfg g = make_cfg(tile.strip, lane, S);
    // U16 with le090: one uint16 per lane and row pair, bit j / 8 + j = `normalized <= 0.9` (img_relevant.comp:58) of column c + j in the
    // even / odd row — the level-0 expand launch bins the gradation histogram with it instead of reading the raw pixels again
    const bool want_mask = U16 && le090 != nullptr;
    const Buf mb = want_mask ? make_buf(le090 + (size_t)img * Sc * (S / 8), (size_t)Sc * (S / 8) * 2) : bb;
    const uint32_t moff = g.off == kOob ? kOob : (uint32_t)g.c >> 2, mrb = (uint32_t)S >> 2;
    // own pixels and four halo pixels: c-4 .. c-1 (lane 0 of a strip that is not the first) or c+8 .. c+11 (lane 63 with more image to its right)
    const uint32_t px_bytes = U16 ? 2u : 4u;
    const uint32_t foff = g.off == kOob ? kOob : (uint32_t)g.c * px_bytes;
    const uint32_t foff_h = g.off_l != kOob ? (uint32_t)(g.c - 4) * px_bytes : (g.off_r != kOob ? (uint32_t)(g.c + 8) * px_bytes : kOob);
    const int hi = S - 1;
    const uint32_t frb = U16 ? (uint32_t)S * 2u : (uint32_t)pitch * 4u, rb = (uint32_t)pitch * 4u, crb = (uint32_t)cpitch * 4u;
    const int ks = max(k0 - 1, 0), ke = min(k1, Sc - 1);   // coarse rows this wavefront computes (the first / last only feed its band rows)

    FRow w0, w1, w2, w3, w4;
    RawF ra, rc;
    if (U16) {
        load_raw_f(ra, ib, (uint32_t)mirror_idx(2 * ks - 2, hi) * frb, foff, foff_h);
        convert_f(w0, ra, nk);
        load_raw_f(ra, ib, (uint32_t)mirror_idx(2 * ks - 1, hi) * frb, foff, foff_h);
        convert_f(w1, ra, nk);
        load_raw_f(ra, ib, (uint32_t)(2 * ks) * frb, foff, foff_h);
        convert_f(w2, ra, nk);
        load_raw_f(ra, ib, (uint32_t)mirror_idx(2 * ks + 1, hi) * frb, foff, foff_h);
        load_raw_f(rc, ib, (uint32_t)mirror_idx(2 * ks + 2, hi) * frb, foff, foff_h);
    } else {
        load_f(w0, ib, (uint32_t)mirror_idx(2 * ks - 2, hi) * frb, foff, foff_h);
        load_f(w1, ib, (uint32_t)mirror_idx(2 * ks - 1, hi) * frb, foff, foff_h);
        load_f(w2, ib, (uint32_t)(2 * ks) * frb, foff, foff_h);
    }
    CRow c0, cm1, cm2;
    cm1 = CRow(); cm2 = CRow();
    for (int k = ks; k <= ke; k++) {
        if (U16) {
            convert_f(w3, ra, nk);   // the pair requested one trip ago
            convert_f(w4, rc, nk);
            const int kn = min(k + 1, ke);
            load_raw_f(ra, ib, (uint32_t)mirror_idx(2 * kn + 1, hi) * frb, foff, foff_h);
            load_raw_f(rc, ib, (uint32_t)mirror_idx(2 * kn + 2, hi) * frb, foff, foff_h);
        } else {
            load_f(w3, ib, (uint32_t)mirror_idx(2 * k + 1, hi) * frb, foff, foff_h);
            load_f(w4, ib, (uint32_t)mirror_idx(2 * k + 2, hi) * frb, foff, foff_h);
        }
        coarse_row(c0, w0, w1, w2, w3, w4, g);
        if (k >= k0 && k < k1)  // wave-uniform
            bstore4_nt(db, g.coff + (uint32_t)k * crb, make_float4(c0.v[0], c0.v[1], c0.v[2], c0.v[3]));
        // coarse row k completes the neighbourhood of row k-1: band rows 2(k-1), 2(k-1)+1 are the two oldest rows of the window.
        // km1(0) = coarse_of_fine(-2) = 1 (reflect-101 on the fine grid, img_smooth_upsampled.comp:10-16): row k itself.
        const int kp = k - 1;
        if (kp >= k0 && kp < k1) {  // wave-uniform
            if (want_mask) bstore_u16(mb, moff + (uint32_t)kp * mrb, le090_bits(w0, w1));
            band_pair(kp == 0 ? c0 : cm2, cm1, c0, w0, w1, g, bb, (uint32_t)(2 * kp) * rb, (uint32_t)(2 * kp + 1) * rb);
        }
        cm2 = cm1; cm1 = c0;
        w0 = w2; w1 = w3; w2 = w4;
    }
    // the last row pair of the image: kp1(Sc-1) = coarse_of_fine(S) = Sc-1 (fine row S mirrors onto S-2)
    if (k1 == Sc) {
        if (want_mask) bstore_u16(mb, moff + (uint32_t)(Sc - 1) * mrb, le090_bits(w0, w1));
        band_pair(cm2, cm1, cm1, w0, w1, g, bb, (uint32_t)(2 * (Sc - 1)) * rb, (uint32_t)(2 * (Sc - 1) + 1) * rb);
    }
}
template <bool U16>
__global__ __launch_bounds__(kBlockThreads, 4) void k_reduce_band(const void* __restrict__ fine, float* __restrict__ down, float* __restrict__ band,
                                                               int S, int pitch, size_t plane, int Sc, int cpitch, size_t cplane,
                                                               int rows_per_wave, const uint32_t* __restrict__ minmax, int min_chain_exact,
                                                               uint16_t* __restrict__ le090, int swz) {
    reduce_band_block<U16>(fine, down, band, S, pitch, plane, Sc, cpitch, cplane, rows_per_wave, minmax, min_chain_exact, le090, xcd_tile(swz), (int)blockIdx.z);
}

// lowpass value at fine (x, y) for any S (generic form).
__device__ __forceinline__ float lowpass_generic(const float* __restrict__ coarse, int cpitch, int Sc, int S, int x, int y, int ref) {
    if (ref) {   // img_smooth_upsampled.comp:32-45 on the zero-inserted image of img_upsample.comp (odd texels: 0, Q2), literally
        float pixel = 0.0f;
#pragma unroll
        for (int m = 0; m < 5; m++) {
            const int j = coarse_of_fine(x + m - 2, S);
#pragma unroll
            for (int n = 0; n < 5; n++) {
                const int k = coarse_of_fine(y + n - 2, S);
                const float tap = ld0(coarse, cpitch, Sc, j, k);   // j, k < 0 (odd fine index: the inserted zero) read 0
                pixel = pixel + ((w5(m) * w5(n)) * 4.0f) * tap;   // :43
            }
        }
        return pixel;
    }
    float V[5];
#pragma unroll
    for (int m = 0; m < 5; m++) {
        const int j = coarse_of_fine(x + m - 2, S);   // -1 (an inserted zero column): five zero taps, the sum is +0
        float acc = 0.0f;
#pragma unroll
        for (int n = 0; n < 5; n++) {
            const int k = coarse_of_fine(y + n - 2, S);
            const float w = n == 0 ? W0 : n == 1 ? W1 : n == 2 ? W2 : n == 3 ? W3 : W4;
            const float t = w * ld0(coarse, cpitch, Sc, j, k);   // j, k < 0 read 0
            acc = n == 0 ? t : acc + t;
        }
        V[m] = acc;
    }
    return 4.0f * chain5(V[0], V[1], V[2], V[3], V[4]);
}

#ifdef MUSICA_PYRAMID_FULL
__global__ void k_band_generic(const float* __restrict__ fine, const float* __restrict__ coarse, float* __restrict__ band,
                               int S, int pitch, size_t plane, int Sc, int cpitch, size_t cplane, int ref) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= S || y >= S) return;
    fine += (size_t)blockIdx.z * plane;
    band += (size_t)blockIdx.z * plane;
    coarse += (size_t)blockIdx.z * cplane;
    const float low = lowpass_generic(coarse, cpitch, Sc, S, x, y, ref);
    band[(size_t)y * pitch + x] = fine[(size_t)y * pitch + x] - low;
}

// lowpass only (debugProcess' red_lowpass_i / exp_lowpass_i dumps, kernel-level tests)
__global__ void k_lowpass_generic(const float* __restrict__ coarse, float* __restrict__ low, int S, int pitch, size_t plane,
                                  int Sc, int cpitch, size_t cplane, int ref) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= S || y >= S) return;
    low += (size_t)blockIdx.z * plane;
    coarse += (size_t)blockIdx.z * cplane;
    low[(size_t)y * pitch + x] = lowpass_generic(coarse, cpitch, Sc, S, x, y, ref);
}
#endif

// ======================================================================================
// K14 + K16 + K7 + K8 + K17: recon = lowpass(prev) + band * gain(sdev) [* nr(cnr)]
// ======================================================================================

// linearFunction() of noise_reduction.comp:24-31 with its slope m = (p2.y - p1.y) / (p2.x - p1.x) precomputed.
// selects, not branches (a NaN c fails both tests and yields m * NaN + lowFactor = NaN, like the if / else chain)
__device__ __forceinline__ float nr_factor_m(float c, float lowCnr, float lowFactor, float highCnr, float highFactor, float m) {
    float r = m * c + lowFactor;
    r = c > highCnr ? highFactor : r;
    r = c < lowCnr ? lowFactor : r;
    return r;
}
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
        if (s == 0.0f) return high;                          // points[0].x == x
        if (s >= 0.0f && s <= 1.0f) return 0.0f * s + high;  // m = (high - high) / (1 - 0) = 0
        return 0.0f;
    }
    return curve_eval(t, s);
}

// The streaming kernel's form of getY() for the 33-point contrast polyline (DevCurveLut): two 16-byte LDS reads and
// no branch. Exactly curve_eval()'s result for every float s:
//   * s is first clamped with sf = min(s, 2): NaN becomes 2 (v_min_f32 returns the other operand), as do +inf and every
//     s > 2; all of them lie above x[32] = 1, where getY() matches no interval and returns 0 — and so does the table
//     (j = 33, segment {0, 0, 0});
//   * j = #{x[i] < sf} from the fine or the coarse bucket (musica_device.h); j in 1..32 -> the interval [x[j-1], x[j]];
//   * j = 0 means sf <= x[0] = 0: getY() returns y[0] for sf == 0 (also -0) and 0 for negative sf.
// Degenerate curves (lut.ok == 0) take curve_eval()'s literal scan.
struct LutLds {
    float4 bucket[kLutBuckets + kLutCoarse];
    float4 seg[kLutPoints + 1];
    float inv_w;
    uint32_t ok;
};
// LUTOK is wave-uniform and decided once per wavefront (k_expand_fast picks the instantiation of the whole march): a per-texel
// `if (!ok) slow()` put a divergent branch and a call sequence around each of the 16 lookups of a trip.
template <bool LUTOK>
__device__ __forceinline__ float curve_eval_lut(const CurveLds& t, const LutLds& lut, float s) {
    if (!LUTOK) return curve_eval(t, s);
    const float sf = fminf(s, 2.0f);
    const float kf = sf * lut.inv_w;
    // bucket index without a branch: the fine bucket int(kf) when kf < 256, else the coarse bucket 256 + int(min(sf * 256, 257)).
    // The selection happens on the floats (both are <= 257, so one conversion serves either) and the +256 on the integers —
    // the same values as converting each on its own. A negative kf (s < 0) clamps to bucket 0; its result is replaced below.
    const bool in_fine = kf < (float)kLutBuckets;
    const float tsel = in_fine ? kf : fminf(sf * 256.0f, (float)(kLutCoarse - 1));
    const int idx = max((int)tsel + (in_fine ? 0 : kLutBuckets), 0);
    const float4 e = lut.bucket[idx];
    // e.x holds 16 * jlo as an integer (the byte offset of seg[jlo]); each abscissa of the bucket below sf moves one entry on
    const int joff = __float_as_int(e.x) + (e.y < sf ? 16 : 0) + (e.z < sf ? 16 : 0);
    const float4 g = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(lut.seg) + joff);
    const float r = g.z * (sf - g.x) + g.y;
    return sf < 0.0f ? 0.0f : r;
}

// GH (level 0 only): the launch also accumulates the gradation histogram — img_relevant.comp + gradation_histogram.comp —
// of the texels it reconstructs, while they are still in registers (the reference re-reads the whole image for it).
// The reference's thread walks a 16 x 16 area and `return`s at the first texel that is exactly 0
// (gradation_histogram.comp:24); here every texel is binned and a zero only raises the image's gzero word: the literal
// kernel (k_grad_hist) then recounts that image into a second histogram and k_grad_curve takes that one. The relevance
// weight uint(relevant * 100) is the one k_grad_hist computes: one cnr classification per lane and row pair (the cnr scale
// is 8 here, so a lane's 8 columns and both rows of a pair sit under one cnr texel), `normalized <= 0.9` on the raw pixel.
// `normalized <= 0.9` comes as the bit image k_reduce_band<true> wrote (2 bytes per lane and row pair; the raw-pixel form, 32 bytes
// against the threshold thr090, left in round 4).
// CNR48: the cnr scale is 4 or 8 (levels 1 and 0 of every image side that is a multiple of 8): columns c..c+3 and c+4..c+7 of a
// lane each sit under one cnr texel. Wave-uniform like LUTOK and chosen the same way (the other form divides per texel).
// CH (with GH, CLAHE contexts): the launch also accumulates clahe_histogram.comp:13-45 — hist[tx][ty][bin] += 1 where
// relevant == 1.0 — of the texels it reconstructs (k_clahe_hist<true> re-read the whole reconstruction and the raw pixels for it:
// 36 us for one 4096^2 image). relevant == 1.0 is relevant_of()'s value spelled out: inside the border and either the ramp
// value ((r*r)*(r*r))*r itself 1.0 or cnr in [6, 256] with `normalized <= 0.9`. A workgroup's 512 columns and its rows touch at
// most 2 x 2 tiles (the launcher checks: tile side >= 512 and >= the workgroup's rows): lc = 2 copies x 4 tile slots x 256 bins.
constexpr int kChSlots = 4, kChCopies = 2;
#ifndef MUSICA_SD_W
#define MUSICA_SD_W 3
#endif
#ifndef MUSICA_SD_CH_W
#define MUSICA_SD_CH_W 2   // with the CLAHE histogram on board the launch wants 189 registers: 64 bytes of scratch at 3 wavefronts per SIMD
#endif
// SD (levels 0 .. 2 of a context that does not store their sdev images): the launch computes the 5 x 5 RMS of the band image itself, from a
// window of six band rows it keeps in registers — the bits k_sdev_hist* would have stored (the same sum5 / musica_div25 / musica_sqrt8 on the same
// squares, sdev_parts.h) — instead of reading them: 4 B per texel less to read here and 4 B less for the sdev launch to write.
struct BRow {
    float v[8];     // band columns c .. c+7
    float e0, e1;   // lane 0: columns c-2, c-1; lane 63: columns c+8, c+9 (0 where the image ends); unused elsewhere
};
__device__ __forceinline__ void sdev_from_band(const BRow& r0, const BRow& r1, const BRow& r2, const BRow& r3, const BRow& r4, bool lane0, bool lane63, float (&s)[8]) {
    float q[8];
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] = sum5(r0.v[j] * r0.v[j], r1.v[j] * r1.v[j], r2.v[j] * r2.v[j], r3.v[j] * r3.v[j], r4.v[j] * r4.v[j]);
    const float qe0 = sum5(r0.e0 * r0.e0, r1.e0 * r1.e0, r2.e0 * r2.e0, r3.e0 * r3.e0, r4.e0 * r4.e0);
    const float qe1 = sum5(r0.e1 * r0.e1, r1.e1 * r1.e1, r2.e1 * r2.e1, r3.e1 * r3.e1, r4.e1 * r4.e1);
    float a6 = from_left_lane(q[6]), a7 = from_left_lane(q[7]);
    float b0 = from_right_lane(q[0]), b1 = from_right_lane(q[1]);
    if (lane0) { a6 = qe0; a7 = qe1; }
    if (lane63) { b0 = qe0; b1 = qe1; }
    s[0] = sum5(a6, a7, q[0], q[1], q[2]);
    s[1] = sum5(a7, q[0], q[1], q[2], q[3]);
    s[2] = sum5(q[0], q[1], q[2], q[3], q[4]);
    s[3] = sum5(q[1], q[2], q[3], q[4], q[5]);
    s[4] = sum5(q[2], q[3], q[4], q[5], q[6]);
    s[5] = sum5(q[3], q[4], q[5], q[6], q[7]);
    s[6] = sum5(q[4], q[5], q[6], q[7], b0);
    s[7] = sum5(q[5], q[6], q[7], b0, b1);
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = musica_div25(s[j]);
    musica_sqrt8(s);
}
template <int GAIN, bool NR, bool GH, bool LUTOK, bool CNR48, bool CH = false, bool SD = false>
__device__ __forceinline__ bool expand_march(const ExpandArgs& a, const CurveLds& tab, const LutLds& lut, uint32_t* lh, int img, uint32_t* lc = nullptr,
                                             uint32_t tx0 = 0u, uint32_t ty0 = 0u) {
    constexpr int kGhCopies = 4, kGhStride = MUSICA_GRAD_BINS + 8;
    // slope of noise_reduction.comp:28, the same value for every texel
    const float nr_m = (a.highFactor - a.lowFactor) / (a.highCnr - a.lowCnr);
    const int lane = threadIdx.x & 63;
    const Tile tile = xcd_tile(a.swz);
    const int seg = __builtin_amdgcn_readfirstlane((int)(tile.segblock * kWavesPerBlock + (threadIdx.x >> 6)));   // wave-uniform: row arithmetic stays on the scalar unit
    const int k0 = seg * a.rows_per_wave;
    bool saw_zero = false;
    if (k0 < a.Sc) {   // wave-uniform
    const int k1 = min(k0 + a.rows_per_wave, a.Sc);
    const int S = a.S;
    const int cnrScale = GH ? 8 : a.cnrScale;   // the launcher takes the GH variant only at scale 8
    const Buf bb = make_buf(a.band + (size_t)img * a.plane, a.plane * 4);
    const Buf sb = make_buf((GAIN != GAIN_CONST ? a.sdev : a.band) + (size_t)img * a.plane, a.plane * 4);
    const Buf ob = make_buf(a.recon + (size_t)img * a.plane, a.plane * 4);
    const Buf pb = make_buf(a.prev + (size_t)img * a.cplane, a.cplane * 4);
    const Buf wb = !GH ? bb : make_buf(a.le090 + (size_t)img * a.Sc * (S / 8), (size_t)a.Sc * (S / 8) * 2);   // `normalized <= 0.9` as k_reduce_band<true>'s bit image
    const float* cnr = NR ? a.cnr + (size_t)img * a.cnrPlane : nullptr;
    const LaneCfg g = make_cfg(tile.strip, lane, S);
    const uint32_t rb = (uint32_t)a.pitch * 4u, crb = (uint32_t)a.cpitch * 4u;
    const uint32_t moff = g.off == kOob ? kOob : (uint32_t)g.c >> 2, mrb = (uint32_t)S >> 2;
    // noise reduction: the 8 columns of a lane share ceil(8 / scale) cnr texels per row
    const int cxs[2] = {g.active ? g.c / cnrScale : 0, g.active ? (g.c + 4) / cnrScale : 0};
    // gradation histogram: img_relevant.comp:46-49 in uint arithmetic (wraps for N < 100 like the shader)
    const uint32_t border = 100u, lim = (uint32_t)S - border;
    uint32_t colin = 0u;
    if (GH) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (g.active && (uint32_t)(g.c + j) > border && (uint32_t)(g.c + j) < lim) colin |= 1u << j;
    }
    uint32_t* lhc = lh + (GH ? (lane & (kGhCopies - 1)) * kGhStride : 0);
    // CLAHE tile column of each of the lane's 8 columns, relative to the workgroup's first one (clahe_histogram.comp:34)
    const float fS = (float)S;
    uint32_t chx = 0u;
    if (CH) {
#pragma unroll
        for (int j = 0; j < 8; j++) chx |= (min(f2u((float)(g.c + j) / fS * (float)MUSICA_CLAHE_TILES) - tx0, 1u)) << j;
    }
    uint32_t* lcc = CH ? lc + (lane & (kChCopies - 1)) * (kChSlots * MUSICA_CLAHE_BINS) : nullptr;

    // One row of a pair once its operands are in registers: contrast gain, noise reduction, addition, store, histograms.
    // `src` is the band row, `b` the registers the reconstructed row is left in (the same array, or — SD — the lowpass row's: the band row stays
    // in the window)
    auto row_phase = [&](const int ph, float* b, const float* src, const float* sd, const float* low, const float* f, const uint32_t le_t,
                         const int kk, const uint32_t w_cnr, const uint32_t w_dark_or_ramp, const bool one_ramp, const bool one_dark) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            // contrast_curve_apply.comp:61
            float p = src[j] * (GAIN == GAIN_CURVE ? curve_eval_lut<LUTOK>(tab, lut, sd[j]) : gain_of<GAIN>(GAIN != GAIN_CONST ? sd[j] : 0.0f, a.high, tab));
            if (NR) p = p * f[j];   // noise_reduction.comp:57
            b[j] = low[j] + p;      // img_addition.comp:15
        }
        store8_nt(ob, g.off + (uint32_t)(2 * kk + ph) * rb, b);   // non-temporal since round 4: with steps in flight -2 % per step at 8 x 2048^2 (k_grad_apply has the numbers)
        if (GH) {
            const uint32_t dark = le_t >> (8 * ph);
            const uint32_t y = (uint32_t)(2 * kk + ph);
            const uint32_t m = (y > border && y < lim) ? colin : 0u;   // inside-the-border bits of the lane's 8 columns in this row
            const uint32_t chy = CH ? min(f2u((float)y / fS * (float)MUSICA_CLAHE_TILES) - ty0, 1u) : 0u;   // clahe_histogram.comp:35 (wave-uniform)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float cur = b[j];
                saw_zero = saw_zero || (cur == 0.0f);                        // gradation_histogram.comp:24
                // :26 int(cur * 1024) with "NaN never indexes" (oracle Q6) and "bins outside [0, 1024) are dropped" (Q1) folded into
                // one unsigned compare: max(NaN, -1) = -1 and every scaled <= -1 convert to a negative int, i.e. a huge unsigned;
                // (-1, 0) truncates to bin 0 like the shader's int(); everything out of range lands on the spare word 1024
                const uint32_t bin = min((uint32_t)(int)fminf(fmaxf(cur * (float)MUSICA_GRAD_BINS, -1.0f), 2048.0f), (uint32_t)MUSICA_GRAD_BINS);
                // :28-30 uint(relevant * 100): 0 outside the border; adding 0 leaves the histogram as it is
                const bool le090 = ((dark >> j) & 1u) != 0u;
                const uint32_t w = ((m >> j) & 1u) * (le090 ? w_dark_or_ramp : w_cnr);
                atomicAdd(&lhc[bin], w);
                if (CH) {
                    const float scaled = cur * (float)(MUSICA_CLAHE_BINS - 1) + 0.5f;                 // clahe_histogram.comp:20
                    const bool inrange = scaled > -1.0f && scaled < (float)MUSICA_CLAHE_BINS;       // NaN never indexes (Q6)
                    if (((m >> j) & 1u) && inrange && (one_ramp || (one_dark && le090)))
                        atomicAdd(&lcc[((((chx >> j) & 1u) * 2u + chy) * MUSICA_CLAHE_BINS) + (uint32_t)(int)scaled], 1u);   // :39-44
                }
            }
        }
    };
    if constexpr (SD) {
        // rows 2k - 2 .. 2k + 3 of the band image: six slots that rotate by two rows per trip; three trips are spelled out per loop iteration so
        // that a slot is a fixed set of registers (no moves). A row outside the image is requested out of range (zeros: Q1).
        const uint32_t off_e = g.lane0 ? g.off_l : g.off_r;
        auto load_brow = [&](BRow& r, const int y, const bool wanted) __attribute__((always_inline)) {
            const uint32_t row_off = (wanted && y >= 0 && y < S) ? (uint32_t)y * rb : kOob;   // wave-uniform
            load8(r.v, bb, (g.off + row_off) | ((g.off | row_off) & kOob));
            const float2 e = bload2(bb, (off_e + row_off) | ((off_e | row_off) & kOob));
            r.e0 = e.x; r.e1 = e.y;
        };
        CRow c0, c1, cn;
        load_crow(c0, pb, (uint32_t)coarse_of_fine(2 * k0 - 2, S) * crb, g);
        load_crow(c1, pb, (uint32_t)k0 * crb, g);
        load_crow(cn, pb, (uint32_t)coarse_of_fine(2 * k0 + 2, S) * crb, g);
        uint32_t le = 0u;
        float cq[2] = {0.0f, 0.0f};
        auto load_cnr2 = [&](const int kk) __attribute__((always_inline)) {
            const size_t re = (size_t)((2 * kk) / cnrScale) * a.cnrPitch;
            cq[0] = cnr[re + cxs[0]]; cq[1] = cnr[re + cxs[1]];
        };
        if (NR && CNR48) load_cnr2(k0);
        BRow w0, w1, w2, w3, w4, w5;
        load_brow(w0, 2 * k0 - 2, true); load_brow(w1, 2 * k0 - 1, true); load_brow(w2, 2 * k0, true);
        load_brow(w3, 2 * k0 + 1, true); load_brow(w4, 2 * k0 + 2, true); load_brow(w5, 2 * k0 + 3, true);
        if (GH) le = bload_u16(wb, moff + (uint32_t)k0 * mrb);
        auto trip = [&](const int k, BRow& r0, BRow& r1, BRow& r2, BRow& r3, BRow& r4, BRow& r5) __attribute__((always_inline)) {
            const bool more = k + 1 < k1;                     // wave-uniform
            const int kn = k + 1;
            uint32_t le_n = 0u;
            if (GH) le_n = bload_u16(wb, (more ? moff : kOob) + (uint32_t)kn * mrb);
            float lowE[8], lowO[8];
            lowpass_pair(c0, c1, cn, g, lowE, lowO);
            c0 = c1; c1 = cn;
            {
                const uint32_t row_off = (uint32_t)coarse_of_fine(2 * kn + 2, S) * crb;
                const float4 q = bload4(pb, (more ? g.coff : kOob) + row_off);
                cn.v[0] = q.x; cn.v[1] = q.y; cn.v[2] = q.z; cn.v[3] = q.w;
                cn.hl = cn.hr = bload1(pb, (more ? (g.lane0 ? g.coff_l : g.coff_r) : kOob) + row_off);
            }
            float fe[8], fo[8];
            float cnr_pair = 0.0f;
            if (NR) {
                if (CNR48) {
                    cnr_pair = cq[0] * kMaxCnrValue;
                    const float e0 = nr_factor_m(cnr_pair, a.lowCnr, a.lowFactor, a.highCnr, a.highFactor, nr_m);
                    const float e1 = nr_factor_m(cq[1] * kMaxCnrValue, a.lowCnr, a.lowFactor, a.highCnr, a.highFactor, nr_m);
#pragma unroll
                    for (int j = 0; j < 8; j++) { fe[j] = j < 4 ? e0 : e1; fo[j] = fe[j]; }
                    load_cnr2(more ? kn : k);
                } else {
                    const size_t re = (size_t)((2 * k) / cnrScale) * a.cnrPitch, ro = (size_t)((2 * k + 1) / cnrScale) * a.cnrPitch;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int cx = g.active ? (g.c + j) / cnrScale : 0;                 // noise_reduction.comp:39-45
                        fe[j] = nr_factor_m(cnr[re + cx] * kMaxCnrValue, a.lowCnr, a.lowFactor, a.highCnr, a.highFactor, nr_m);
                        fo[j] = nr_factor_m(cnr[ro + cx] * kMaxCnrValue, a.lowCnr, a.lowFactor, a.highCnr, a.highFactor, nr_m);
                    }
                }
            }
            const CnrClass kc = classify_cnr(cnr_pair);
            const uint32_t w_cnr = kc.ramp ? kc.w_ramp : 0u, w_dark_or_ramp = kc.ramp ? kc.w_ramp : (kc.high ? 100u : 0u);
            const float r6 = cnr_pair / 6.0f;
            const bool one_ramp = CH && kc.ramp && (((r6 * r6) * (r6 * r6)) * r6 == 1.0f);
            const bool one_dark = CH && !kc.ramp && kc.high;
            {   // even row 2k: window rows 2k - 2 .. 2k + 2, its band values are the centre row's
                float sd[8];
                sdev_from_band(r0, r1, r2, r3, r4, g.lane0, g.lane63, sd);
                row_phase(0, lowE, r2.v, sd, lowE, fe, le, k, w_cnr, w_dark_or_ramp, one_ramp, one_dark);
            }
            load_brow(r0, 2 * k + 4, more);   // the slot of row 2k - 2 takes row 2k + 4 (the next trip's; nothing if there is none)
            {   // odd row 2k + 1: rows 2k - 1 .. 2k + 3
                float sd[8];
                sdev_from_band(r1, r2, r3, r4, r5, g.lane0, g.lane63, sd);
                row_phase(1, lowO, r3.v, sd, lowO, fo, le, k, w_cnr, w_dark_or_ramp, one_ramp, one_dark);
            }
            load_brow(r1, 2 * k + 5, more);
            le = le_n;
        };
        for (int k = k0; k < k1; k += 3) {
            trip(k, w0, w1, w2, w3, w4, w5);
            if (k + 1 < k1) trip(k + 1, w2, w3, w4, w5, w0, w1);
            if (k + 2 < k1) trip(k + 2, w4, w5, w0, w1, w2, w3);
        }
        return saw_zero;
    }
    // The march is a software pipeline (round 4): while a wavefront computes one row of a pair, the operands of its NEXT row are in
    // flight — the odd row's band / sdev registers are requested before the even row is computed, the next trip's even row into the even
    // row's registers as soon as that row is stored, the next trip's coarse row into the registers of the coarse row that has just left
    // the window, the next trip's cnr texels where this trip's have been consumed. No register more than the form that requested a whole
    // trip at its top and waited for the first of them at once would need: 128 registers, 4 wavefronts per SIMD, no scratch (114 before).
    // A trip that does not exist (beyond k1) requests out of range: no traffic.
    CRow c0, c1, cn;
    load_crow(c0, pb, (uint32_t)coarse_of_fine(2 * k0 - 2, S) * crb, g);
    load_crow(c1, pb, (uint32_t)k0 * crb, g);
    load_crow(cn, pb, (uint32_t)coarse_of_fine(2 * k0 + 2, S) * crb, g);
    float be[8], bo[8], se[8], so[8];
    uint32_t le = 0u;   // the pair's `<= 0.9` bits (GH)
    // cnr texels of a trip (CNR48: scale 4 or 8): the left / right half of the lane's columns; rows 2k and 2k + 1 sit under the same
    // cnr row at an even scale ((2k) / s == (2k + 1) / s), so a pair needs two texels, not four
    float cq[2] = {0.0f, 0.0f};
    auto load_cnr4 = [&](const int kk) __attribute__((always_inline)) {
        const size_t re = (size_t)((2 * kk) / cnrScale) * a.cnrPitch;
        cq[0] = cnr[re + cxs[0]]; cq[1] = cnr[re + cxs[1]];
    };
    if (NR && CNR48) load_cnr4(k0);
    load8(be, bb, g.off + (uint32_t)(2 * k0) * rb);
    if (GAIN != GAIN_CONST) load8(se, sb, g.off + (uint32_t)(2 * k0) * rb);
    if (GH) le = bload_u16(wb, moff + (uint32_t)k0 * mrb);
    for (int k = k0; k < k1; k++) {
        const bool more = k + 1 < k1;                     // wave-uniform
        const int kn = k + 1;
        const uint32_t goff_n = more ? g.off : kOob;      // next trip's rows: the lane's columns, or nothing
        // the next pair's `<= 0.9` bits a whole trip ahead (2 bytes per lane): requested with the next even row they would make the
        // bottom of the loop wait for that row (the rotation le = le_n is a register move of a loaded value)
        uint32_t le_n = 0u;
        if (GH) le_n = bload_u16(wb, (more ? moff : kOob) + (uint32_t)kn * mrb);
        // the odd row of this trip: in flight while the even row is computed
        load8(bo, bb, g.off + (uint32_t)(2 * k + 1) * rb);
        if (GAIN != GAIN_CONST) load8(so, sb, g.off + (uint32_t)(2 * k + 1) * rb);
        float lowE[8], lowO[8];
        lowpass_pair(c0, c1, cn, g, lowE, lowO);
        c0 = c1; c1 = cn;
        {   // coarse row of the next trip (k + 2's neighbourhood), a whole trip ahead
            const LaneCfg& gg = g;
            const uint32_t row_off = (uint32_t)coarse_of_fine(2 * kn + 2, S) * crb;
            const float4 q = bload4(pb, (more ? gg.coff : kOob) + row_off);
            cn.v[0] = q.x; cn.v[1] = q.y; cn.v[2] = q.z; cn.v[3] = q.w;
            cn.hl = cn.hr = bload1(pb, (more ? (gg.lane0 ? gg.coff_l : gg.coff_r) : kOob) + row_off);
        }
        float fe[8], fo[8];  // noise-reduction factors of the two rows
        float cnr_pair = 0.0f;   // cnr * 256 of the texel above this lane's row pair (GH: scale 8)
        if (NR) {
            if (CNR48) {  // columns c..c+3 and c+4..c+7 each sit inside one cnr texel (c % 8 == 0)
                cnr_pair = cq[0] * kMaxCnrValue;
                const float e0 = nr_factor_m(cnr_pair, a.lowCnr, a.lowFactor, a.highCnr, a.highFactor, nr_m);
                const float e1 = nr_factor_m(cq[1] * kMaxCnrValue, a.lowCnr, a.lowFactor, a.highCnr, a.highFactor, nr_m);
#pragma unroll
                for (int j = 0; j < 8; j++) { fe[j] = j < 4 ? e0 : e1; fo[j] = fe[j]; }
                load_cnr4(more ? kn : k);   // branch-free: the last trip asks for its own texels again
            } else {
                const size_t re = (size_t)((2 * k) / cnrScale) * a.cnrPitch, ro = (size_t)((2 * k + 1) / cnrScale) * a.cnrPitch;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int cx = g.active ? (g.c + j) / cnrScale : 0;                 // noise_reduction.comp:39-45
                    fe[j] = nr_factor_m(cnr[re + cx] * kMaxCnrValue, a.lowCnr, a.lowFactor, a.highCnr, a.highFactor, nr_m);
                    fo[j] = nr_factor_m(cnr[ro + cx] * kMaxCnrValue, a.lowCnr, a.lowFactor, a.highCnr, a.highFactor, nr_m);
                }
            }
        }
        // img_relevant.comp:44-63 for the cnr texel above the row pair (rows 2k, 2k+1 and columns c..c+7 share it)
        const CnrClass kc = classify_cnr(cnr_pair);
        const uint32_t w_cnr = kc.ramp ? kc.w_ramp : 0u, w_dark_or_ramp = kc.ramp ? kc.w_ramp : (kc.high ? 100u : 0u);   // bright / dark pixel under this texel
        // relevant == 1.0 (CH): the ramp value itself, else 1.0 for a dark pixel under a high-cnr texel (relevant_of)
        const float r6 = cnr_pair / 6.0f;
        const bool one_ramp = CH && kc.ramp && (((r6 * r6) * (r6 * r6)) * r6 == 1.0f);
        const bool one_dark = CH && !kc.ramp && kc.high;
        // (Left alone, the machine scheduler moves the first instructions on the odd row's operands — the NaN-quieting v_max of the
        // lookup — up into the even row's phase, and the wait for the odd row with them. __builtin_amdgcn_sched_barrier(0) in front of
        // either phase keeps them apart, and costs more than it saves: 153 registers, i.e. 84 bytes of scratch at 4 wavefronts per SIMD
        // (116 us) or 3 wavefronts per SIMD (85.9 us, the same as this form: 85.2 - 87.2 against 89.4 - 89.7 us before the pipeline).)
        row_phase(0, be, be, se, lowE, fe, le, k, w_cnr, w_dark_or_ramp, one_ramp, one_dark);
        // the even row of the next trip into the registers the even row has just left
        load8(be, bb, goff_n + (uint32_t)(2 * kn) * rb);
        if (GAIN != GAIN_CONST) load8(se, sb, goff_n + (uint32_t)(2 * kn) * rb);
        row_phase(1, bo, bo, so, lowO, fo, le, k, w_cnr, w_dark_or_ramp, one_ramp, one_dark);
        le = le_n;
    }
    }
    return saw_zero;
}

template <int GAIN, bool NR, bool GH, int W = 1, bool CH = false, bool SD = false>
__global__ __launch_bounds__(kBlockThreads, W) void k_expand_fast(ExpandArgs a) {
    __shared__ CurveLds tab;
    __shared__ __attribute__((aligned(16))) LutLds lut;
    // four bank-staggered copies of the histogram (lane l adds into copy l % 4, see sdev_parts.h: neighbouring texels share bins,
    // and lanes of one ds_add that hit the same address are served one after the other); word 1024 of a copy takes what is out of range
    constexpr int kGhCopies = 4, kGhStride = MUSICA_GRAD_BINS + 8;
    __shared__ uint32_t lh[GH ? kGhCopies * kGhStride : 1];
    __shared__ uint32_t lc[CH ? kChCopies * kChSlots * MUSICA_CLAHE_BINS : 1];
    const int img = blockIdx.z;
    if (GH)
        for (int i = threadIdx.x; i < kGhCopies * kGhStride; i += blockDim.x) lh[i] = 0u;
    uint32_t tx0 = 0u, ty0 = 0u;   // CLAHE tile of the workgroup's first column / first row
    if (CH) {
        for (int i = threadIdx.x; i < kChCopies * kChSlots * MUSICA_CLAHE_BINS; i += blockDim.x) lc[i] = 0u;
        const Tile t = xcd_tile(a.swz);
        tx0 = f2u((float)(t.strip * kStripCols) / (float)a.S * (float)MUSICA_CLAHE_TILES);
        ty0 = f2u((float)(t.segblock * kWavesPerBlock * a.rows_per_wave * 2) / (float)a.S * (float)MUSICA_CLAHE_TILES);
    }
    if (GAIN == GAIN_CURVE) {
        const DevCurve* cv = a.curves + (size_t)img * a.curve_stride;
        const DevCurveLut* lv = a.luts + (size_t)img * MUSICA_COARSER_LEVELS_START;
        curve_to_lds(tab, cv);
        for (int i = threadIdx.x; i < kLutBuckets + kLutCoarse; i += blockDim.x) lut.bucket[i] = lv->bucket[i];
        for (int i = threadIdx.x; i <= kLutPoints; i += blockDim.x) lut.seg[i] = lv->seg[i];
        if (threadIdx.x == 0) { lut.inv_w = lv->inv_w; lut.ok = lv->ok; }
        __syncthreads();
    }
    // wave-uniform (one LDS word): readfirstlane tells the compiler so, and the whole march is instantiated per case
    const bool lut_ok = GAIN != GAIN_CURVE || __builtin_amdgcn_readfirstlane((int)lut.ok) != 0;
    const bool cnr48 = !NR || GH || a.cnrScale == 4 || a.cnrScale == 8;   // kernel argument: uniform
    bool saw_zero;
    if (lut_ok && cnr48) saw_zero = expand_march<GAIN, NR, GH, true, true, CH, SD>(a, tab, lut, lh, img, lc, tx0, ty0);
    else if (lut_ok) saw_zero = expand_march<GAIN, NR, GH, true, NR && !GH ? false : true, CH, SD>(a, tab, lut, lh, img, lc, tx0, ty0);
    else if (cnr48) saw_zero = expand_march<GAIN, NR, GH, GAIN != GAIN_CURVE, true, CH, SD>(a, tab, lut, lh, img, lc, tx0, ty0);
    else saw_zero = expand_march<GAIN, NR, GH, GAIN != GAIN_CURVE, NR && !GH ? false : true, CH, SD>(a, tab, lut, lh, img, lc, tx0, ty0);
    if (GH) {
        if (saw_zero) atomicOr(&a.gzero[img], 1u);
        __syncthreads();
        uint32_t* gh = a.ghist + (size_t)img * MUSICA_GRAD_BINS;
        for (int i = threadIdx.x; i < MUSICA_GRAD_BINS; i += blockDim.x) {
            uint32_t v = 0u;
#pragma unroll
            for (int k = 0; k < kGhCopies; k++) v += lh[k * kGhStride + i];
            if (v) atomicAdd(&gh[i], v);
        }
        if (CH) {
            uint32_t* ch = a.chist + (size_t)img * MUSICA_CLAHE_TILES * MUSICA_CLAHE_TILES * MUSICA_CLAHE_BINS;
            for (int i = threadIdx.x; i < kChSlots * MUSICA_CLAHE_BINS; i += blockDim.x) {
                const uint32_t slot = (uint32_t)i / MUSICA_CLAHE_BINS, tx = tx0 + (slot >> 1), ty = ty0 + (slot & 1u);
                uint32_t v = 0u;
#pragma unroll
                for (int k = 0; k < kChCopies; k++) v += lc[k * kChSlots * MUSICA_CLAHE_BINS + i];
                if (v && tx < (uint32_t)MUSICA_CLAHE_TILES && ty < (uint32_t)MUSICA_CLAHE_TILES)
                    atomicAdd(&ch[(tx * MUSICA_CLAHE_TILES + ty) * MUSICA_CLAHE_BINS + ((uint32_t)i % MUSICA_CLAHE_BINS)], v);
            }
        }
    }
}

__device__ __forceinline__ void curve_to_lds_2d(CurveLds& tab, const DevCurve* __restrict__ src) {
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    for (int i = tid; i < kCurveCap; i += blockDim.x * blockDim.y) {
        tab.x[i] = src->x[i]; tab.y[i] = src->y[i]; tab.m[i] = src->m[i];
    }
    if (tid == 0) { tab.count = src->count; tab.monotone = src->monotone; }
}

// band * gain [* nr] at one texel (generic form)
template <int GAIN, bool NR>
__device__ __forceinline__ float exp_band_at(const ExpandArgs& a, const CurveLds& tab, int img, int x, int y) {
    const size_t o = (size_t)img * a.plane + (size_t)y * a.pitch + x;
    const float s = (GAIN != GAIN_CONST) ? a.sdev[o] : 0.0f;
    float p = a.band[o] * gain_of<GAIN>(s, a.high, tab);
    if (NR) {
        const int cx = x / a.cnrScale, cy = y / a.cnrScale;
        const float c = ((cx < a.cnrS && cy < a.cnrS) ? a.cnr[(size_t)img * a.cnrPlane + (size_t)cy * a.cnrPitch + cx] : 0.0f) * kMaxCnrValue;
        p = p * nr_factor(c, a.lowCnr, a.lowFactor, a.highCnr, a.highFactor);
    }
    return p;
}

template <int GAIN, bool NR>
__global__ void k_expand_generic(ExpandArgs a) {
    __shared__ CurveLds tab;
    const int img = blockIdx.z;
    if (GAIN == GAIN_CURVE) {
        curve_to_lds_2d(tab, a.curves + (size_t)img * a.curve_stride);
        __syncthreads();
    }
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= a.S || y >= a.S) return;
    const float low = lowpass_generic(a.prev + (size_t)img * a.cplane, a.cpitch, a.Sc, a.S, x, y, a.ref_order);
    a.recon[(size_t)img * a.plane + (size_t)y * a.pitch + x] = low + exp_band_at<GAIN, NR>(a, tab, img, x, y);
}

// band after contrast curve (+ noise reduction): what img_addition.comp reads as inputImageB
// (debugProcess' exp_bandpass_i dump; not materialised on the hot path).
template <int GAIN, bool NR>
__global__ void k_exp_band_generic(ExpandArgs a) {
    __shared__ CurveLds tab;
    const int img = blockIdx.z;
    if (GAIN == GAIN_CURVE) {
        curve_to_lds_2d(tab, a.curves + (size_t)img * a.curve_stride);
        __syncthreads();
    }
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= a.S || y >= a.S) return;
    a.recon[(size_t)img * a.plane + (size_t)y * a.pitch + x] = exp_band_at<GAIN, NR>(a, tab, img, x, y);
}

// ---- the tiny tail of the pyramid in one launch ------------------------------------------------------------------
// Levels whose side is at most 32 (the last four or five of a full-depth pyramid: 24, 12, 6, 3, 2 for the reference's
// 3072^2 / L = 12) are a few hundred texels each, yet every one costs three or four launches of ~5 us (reduce, band, expand: 14
// launches, 72 us of a 270 us step at 3072^2 / L12). One 1024-thread workgroup per image walks them here: reduce and band of levels
// T .. L-1, then expand L-1 .. T (constant gain: these levels are above the cnr level), a texel per thread and phase, with the
// SAME device functions as the per-level generic launches (reduce_generic_at, lowpass_generic: same bits) and a workgroup
// barrier between dependent phases.
// The images live in LDS while the workgroup works on them (dense rows; ~4100 floats for sides 32, 16, 8, ..: a phase reads 25
// taps per texel — from the XCD's L2, where the workgroup's own stores land, that is 25 round trips of ~1 us) and every result is
// also stored to its global image (the getters, and the expand slot of level T - 1, read those).
constexpr int kTailPool = 3 * (kTailSide * kTailSide + (kTailSide / 2) * (kTailSide / 2) * 2) + 64;   // fine + band + recon of every level (sum of squares < 1.5 x the first)
#ifdef MUSICA_PYRAMID_FULL
__global__ __launch_bounds__(1024) void k_tiny_tail(const TailArgs a) {
    __shared__ float pool[kTailPool];
    // side of level k (k = n: the coarsest image) and where its fine, band and reconstruction images sit in the pool. Rolled
    // loops over LDS tables on purpose: unrolled per level, the three stencils are ~100 KB of code that runs once (44 us per
    // launch, most of it instruction fetch).
    __shared__ int side[kTailMax + 1], offF[kTailMax + 1], offB[kTailMax], offR[kTailMax];
    const size_t img = blockIdx.x;
    const int tid = threadIdx.x, nt = blockDim.x, n = a.n;
    if (tid == 0) {
        int o = 0;
        for (int k = 0; k <= n; k++) { side[k] = k < n ? a.l[k].S : a.Sl; offF[k] = o; o += side[k] * side[k]; }
        for (int k = 0; k < n; k++) { offB[k] = o; o += side[k] * side[k]; offR[k] = o; o += side[k] * side[k]; }
    }
    {
        const TailLevel& f = a.l[0];
        const float* fine = f.fine + img * f.plane;
        for (int i = tid; i < f.S * f.S; i += nt) {
            const int y = i / f.S, x = i - y * f.S;
            pool[i] = fine[(size_t)y * f.pitch + x];   // offF[0] = 0
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < n; k++) {
        const int S = side[k], Sc = side[k + 1];
        const float* fineL = pool + offF[k];
        float* downL = pool + offF[k + 1];
        for (int i = tid; i < Sc * Sc; i += nt) {
            const int yo = i / Sc, xo = i - yo * Sc;
            downL[i] = reduce_generic_at(fineL, S, S, xo, yo, a.ref);
        }
        __syncthreads();
        float* bandL = pool + offB[k];
        for (int i = tid; i < S * S; i += nt) {
            const int y = i / S, x = i - y * S;
            bandL[i] = fineL[i] - lowpass_generic(downL, Sc, Sc, S, x, y, a.ref);
        }
    }
#pragma unroll 1
    for (int k = n - 1; k >= 0; k--) {
        __syncthreads();   // the coarser reconstruction is complete (and, the first time, every band image)
        const int S = side[k], Sc = side[k + 1];
        const float* prevL = pool + (k + 1 == n ? offF[k + 1] : offR[k + 1]);   // src/vk_processing.cpp:930-934
        const float* bandL = pool + offB[k];
        float* reconL = pool + offR[k];
        const float high = a.l[k].high;
        for (int i = tid; i < S * S; i += nt) {
            const int y = i / S, x = i - y * S;
            const float low = lowpass_generic(prevL, Sc, Sc, S, x, y, a.ref);
            reconL[i] = low + bandL[i] * high;   // contrast_curve_apply.comp:61 (two-point curve), img_addition.comp:15
        }
    }
    __syncthreads();
    // every image to its global plane, once, behind the last phase
#pragma unroll 1
    for (int k = 0; k < n; k++) {
        const TailLevel& f = a.l[k];
        const bool last = k + 1 == n;
        const int S = side[k], Sc = side[k + 1], cpitch = last ? a.lpitch : a.l[k + 1].pitch;
        const size_t cplane = last ? a.lplane : a.l[k + 1].plane;
        float* down = f.down + img * cplane;
        for (int i = tid; i < Sc * Sc; i += nt) {
            const int yo = i / Sc, xo = i - yo * Sc;
            down[(size_t)yo * cpitch + xo] = pool[offF[k + 1] + i];
        }
        float* band = f.band + img * f.plane;
        float* recon = f.recon + img * f.plane;
        for (int i = tid; i < S * S; i += nt) {
            const int y = i / S, x = i - y * S;
            band[(size_t)y * f.pitch + x] = pool[offB[k] + i];
            recon[(size_t)y * f.pitch + x] = pool[offR[k] + i];
        }
    }
}

// ======================================================================================
// host-side launchers
// ======================================================================================

#endif   // MUSICA_PYRAMID_FULL (k_tiny_tail)
static inline dim3 stream_grid(int S, int rows, int rows_per_wave, int batch) {
    const int strips = (S + kStripCols - 1) / kStripCols;
    const int segs = (rows + rows_per_wave - 1) / rows_per_wave;
    return dim3(strips, (segs + kWavesPerBlock - 1) / kWavesPerBlock, batch);
}

#ifndef MUSICA_PYRAMID_FULL
// Two launches that depend on the same producer and not on each other, as one: the sdev + noise-histogram pass of level i and reduce + band of
// level i + 1 both wait for reduce + band of level i only. Alone on the chip each is 1 - 2 wavefronts per SIMD of dependent arithmetic (8 x 2048^2:
// 45 us and 20 us at i = 0); as the two roles of one grid they fill each other's issue slots, and the pyramid's dependent chain RB1 -> RB2 -> RB3 -> RB4
// runs in the shadow of the sdev passes instead of in front of them. A role is a contiguous range of workgroups starting at a multiple of 8 (the XCD a
// workgroup runs on is its index % 8: the tile mapping of xcd_tile() holds inside a role).
__device__ __forceinline__ bool role_tile(int local, int strips, int blocks, int swz, Tile& t) {
    if (swz && (blocks & 7) == 0) {
        const int xcd = local & 7, j = local >> 3;
        t.strip = j % strips;
        t.segblock = xcd * (blocks >> 3) + j / strips;
    } else {
        t.strip = local % strips;
        t.segblock = local / strips;
    }
    return t.segblock < blocks;   // false: padding
}
__global__ __launch_bounds__(kBlockThreads, 4) void k_rb_sdev(const RbSdevArgs a) {
    __shared__ uint32_t lh[kHistLdsWords];
    __shared__ unsigned long long nzw[kWavesPerBlock][8];
    const int img = (int)blockIdx.z;
    Tile tile;
    if ((int)blockIdx.x >= a.rb_first) {   // block-uniform
        if (!role_tile((int)blockIdx.x - a.rb_first, a.rb_strips, a.rb_blocks, a.swz, tile)) return;
        reduce_band_block<false>(a.fine, a.down, a.band, a.S, a.pitch, a.plane, a.Sc, a.cpitch, a.cplane, a.rows_rb, nullptr, 0, nullptr, tile, img);
        return;
    }
    const SdevRunLevel& l = a.sl;
    if (!role_tile((int)blockIdx.x, l.strips, l.blocks, a.swz, tile)) return;
    float* sd = l.sdev ? l.sdev + (size_t)img * l.plane : nullptr;
    if (l.rows > 0) sdev_march_block<true, true>(l.band + (size_t)img * l.plane, sd, l.S, l.pitch, l.plane, l.hist + (size_t)img * a.hist_stride, a.cov, l.rows, tile, lh);
    else sdev_run_block<true>(l.band + (size_t)img * l.plane, sd, l.S, l.pitch, l.plane, l.hist + (size_t)img * a.hist_stride, a.cov, tile, lh, nzw);
}
// `a`: the pointers, geometry of the reduce + band role and sl.{band, sdev, hist, rows} filled in by the caller; ls: the sdev level
void launch_rb_sdev(hipStream_t st, RbSdevArgs a, const LevelDesc& ls, int batch) {
    const int s_strips = (ls.S + kStripCols - 1) / kStripCols;
    int s_blocks;
    if (a.sl.rows > 0) { const int segs = (ls.S + a.sl.rows - 1) / a.sl.rows; s_blocks = (segs + kWavesPerBlock - 1) / kWavesPerBlock; }
    else s_blocks = (ls.S + kHistArea - 1) / kHistArea;
    a.sl.plane = ls.plane; a.sl.S = ls.S; a.sl.pitch = ls.pitch; a.sl.strips = s_strips; a.sl.blocks = s_blocks; a.sl.first = 0;
    a.rb_first = (s_strips * s_blocks + 7) & ~7;
    a.rb_strips = (a.S + kStripCols - 1) / kStripCols;
    const int segs = (a.Sc + a.rows_rb - 1) / a.rows_rb;
    a.rb_blocks = (segs + kWavesPerBlock - 1) / kWavesPerBlock;
    a.swz = xcd_swizzle_on();
    const int total = a.rb_first + ((a.rb_strips * a.rb_blocks + 7) & ~7);
    hipLaunchKernelGGL(k_rb_sdev, dim3(total, 1, batch), dim3(kBlockThreads), 0, st, a);
}

// The expand launch of a level whose sdev image is not stored (a.sdev == nullptr; GAIN_CURVE levels 0 .. 2): kernels_pyramid.hip's
// launch_expand hands those over to this translation unit.
void launch_expand_sd(hipStream_t st, const ExpandArgs& a, bool nr, int batch) {
    const dim3 grid = stream_grid(a.S, a.Sc, a.rows_per_wave, batch);
    if (nr && a.ghist) {
        if (a.chist) hipLaunchKernelGGL((k_expand_fast<GAIN_CURVE, true, true, MUSICA_SD_CH_W, true, true>), grid, dim3(kBlockThreads), 0, st, a);
        else hipLaunchKernelGGL((k_expand_fast<GAIN_CURVE, true, true, MUSICA_SD_W, false, true>), grid, dim3(kBlockThreads), 0, st, a);
    }
    else if (nr) hipLaunchKernelGGL((k_expand_fast<GAIN_CURVE, true, false, MUSICA_SD_W, false, true>), grid, dim3(kBlockThreads), 0, st, a);
    else hipLaunchKernelGGL((k_expand_fast<GAIN_CURVE, false, false, MUSICA_SD_W, false, true>), grid, dim3(kBlockThreads), 0, st, a);
}
#else
void launch_tiny_tail(hipStream_t st, const TailArgs& a, int batch) {
    hipLaunchKernelGGL(k_tiny_tail, dim3(batch), dim3(1024), 0, st, a);
}

int xcd_swizzle_on() {
    static const int on = getenv("MUSICA_XCD_SWIZZLE") ? atoi(getenv("MUSICA_XCD_SWIZZLE")) : 1;
    return on;
}

static int region_map_on() {   // MUSICA_XCD_REGIONS=0: the round-3 tile mapping of the metric kernel
    static const int on = getenv("MUSICA_XCD_REGIONS") ? atoi(getenv("MUSICA_XCD_REGIONS")) : 1;
    return on;
}
static inline dim3 generic_grid(int S, int batch) { return dim3((S + 31) / 32, (S + 7) / 8, batch); }
static const dim3 kGenericBlock(32, 8, 1);

static inline bool fast_ok(int S) { return S >= 8 && (S % 8) == 0; }

// tag: launch site (see k_reduce_dma)
void launch_reduce(hipStream_t st, const float* in, const LevelDesc& li, float* out, const LevelDesc& lo, int batch, bool force_generic, int tag, int ref) {
    if (fast_ok(li.S) && !force_generic) {
        const int strips = (li.S + kStripCols - 1) / kStripCols;
        const dim3 grid(strips, (lo.S + kDmaRows - 1) / kDmaRows, batch);
        // 8 k strips: workgroup id % 8 == strip % 8 already gives every strip one XCD, tile above tile in dispatch order; otherwise
        // the XCD-aware mapping does (2048^2: 6.0 -> 5.05 us; it costs 1.5 us at 4096^2 where the plain mapping has that property).
        // Round 4: where the geometry allows, 2-D regions per XCD (xcd_region_tile): horizontal neighbours share an L2 as well.
        int swz = (strips % 8) != 0 ? xcd_swizzle_on() : 0;
        {
            const unsigned gx = (unsigned)strips, gy = grid.y;
            const bool pow2 = gx >= 2 && (gx & (gx - 1)) == 0 && gx <= 32;
            const unsigned w = region_width(gx), across = pow2 ? gx / w : 1u, v = pow2 ? 8u / across : 1u;
            if (pow2 && across * v == 8u && gy % v == 0 && xcd_swizzle_on() && region_map_on()) swz = 2;
        }
        auto* kern = tag == 0 ? k_reduce_dma<0> : tag == 1 ? k_reduce_dma<1> : tag == 2 ? k_reduce_dma<2> : tag == 4 ? k_reduce_dma<4> : k_reduce_dma<5>;
        hipLaunchKernelGGL(kern, grid, dim3(kBlockThreads), 0, st, in, out, li.S, li.pitch, li.plane, lo.S, lo.pitch, lo.plane, swz);
    } else {
        hipLaunchKernelGGL(k_reduce_generic, generic_grid(lo.S, batch), kGenericBlock, 0, st, in, out, li.S, li.pitch,
                           li.plane, lo.S, lo.pitch, lo.plane, ref);
    }
}

void launch_reduce_band_u16(hipStream_t st, const uint16_t* px, float* down, float* band, const LevelDesc& lf, const LevelDesc& lc, int batch,
                            int rows_per_wave, const uint32_t* minmax, int min_chain_exact, uint16_t* le090) {
    hipLaunchKernelGGL(k_reduce_band<true>, stream_grid(lf.S, lc.S, rows_per_wave, batch), dim3(kBlockThreads), 0, st, (const void*)px, down, band, lf.S,
                       lf.pitch, lf.plane, lc.S, lc.pitch, lc.plane, rows_per_wave, minmax, min_chain_exact, le090, xcd_swizzle_on());
}
// levels >= 1 (f32 fine image); the side must be a multiple of 8 and at least 16 (the caller checks)
void launch_reduce_band(hipStream_t st, const float* fine, float* down, float* band, const LevelDesc& lf, const LevelDesc& lc, int batch, int rows_per_wave) {
    hipLaunchKernelGGL(k_reduce_band<false>, stream_grid(lf.S, lc.S, rows_per_wave, batch), dim3(kBlockThreads), 0, st, (const void*)fine, down, band, lf.S,
                       lf.pitch, lf.plane, lc.S, lc.pitch, lc.plane, rows_per_wave, (const uint32_t*)nullptr, 0, (uint16_t*)nullptr, xcd_swizzle_on());
}
// band of a level that does not take the fused reduce + band march (sides that are not a multiple of 8, generic / literal-order contexts)
void launch_band(hipStream_t st, const float* fine, const float* coarse, float* band, const LevelDesc& lf, const LevelDesc& lc, int batch, int ref) {
    hipLaunchKernelGGL(k_band_generic, generic_grid(lf.S, batch), kGenericBlock, 0, st, fine, coarse, band, lf.S, lf.pitch,
                       lf.plane, lc.S, lc.pitch, lc.plane, ref);
}

void launch_lowpass(hipStream_t st, const float* coarse, float* low, const LevelDesc& lf, const LevelDesc& lc, int batch, int ref) {
    hipLaunchKernelGGL(k_lowpass_generic, generic_grid(lf.S, batch), kGenericBlock, 0, st, coarse, low, lf.S, lf.pitch, lf.plane,
                       lc.S, lc.pitch, lc.plane, ref);
}

template <int GAIN, bool NR>
static void launch_expand_t(hipStream_t st, const ExpandArgs& a, int batch, bool force_generic) {
    if (GAIN == GAIN_CURVE && !a.sdev) {   // the level's sdev image is not stored: the launch computes sdev itself (k_expand_fast<.., SD>, kernels_expand_sd.hip)
        launch_expand_sd(st, a, NR, batch);
        return;
    }
    if (fast_ok(a.S) && !force_generic) {
        const dim3 grid = stream_grid(a.S, a.Sc, a.rows_per_wave, batch);
        if (GAIN == GAIN_CURVE && NR && a.ghist) {   // level 0 with the gradation histogram on board (the caller checked cnrScale == 8 and passes le090)
            // W = 4: register allocation capped at 128 (4 wavefronts per SIMD) against 143 registers and 3 wavefronts
            // a.chist: a tile side (S / 4) of at least a strip and at least the rows of a workgroup (the caller checks)
            if (a.chist) hipLaunchKernelGGL((k_expand_fast<GAIN_CURVE, true, true, 4, true>), grid, dim3(kBlockThreads), 0, st, a);
            else hipLaunchKernelGGL((k_expand_fast<GAIN_CURVE, true, true, 4>), grid, dim3(kBlockThreads), 0, st, a);
        }
        else hipLaunchKernelGGL((k_expand_fast<GAIN, NR, false, NR ? 4 : 1>), grid, dim3(kBlockThreads), 0, st, a);   // NR: 129 registers wanted, capped at 128 (4 wavefronts per SIMD)
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
#endif   // MUSICA_PYRAMID_FULL (launchers)

}  // namespace musica
