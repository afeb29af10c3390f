// Dev tool (round 3): candidate forms of the metric kernel (fused 5-tap smooth + 2x downsample) side by side,
// each checked bit for bit against the production kernel and timed from HBM (launches rotate over NBUF plane pairs).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I<pkg>/csrc devtools/reduce_lab.hip -o devtools/reduce_lab
//   devtools/reduce_lab [side] [variant ...]
#include "kernels_pyramid.hip"   // the production kernels + helpers (namespace musica)
#include "kernels_bench.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

namespace musica {

// ---- variant UP: every input row of the segment requested before the first use --------------------------------
// One wavefront = 512 columns x R output rows; its 2R + 3 input rows (8 floats per lane each) plus ONE halo register per
// row (lane 0: column c0-2, lane 1: column c0-1, lane 63: column c0+512) are all in flight at once; the compiler's counted
// vmcnt waits release the output rows one after the other as the rows land.
struct URow {
    float v[8];
    float h;
};
__device__ __forceinline__ void reduce_row_u(const URow& r0, const URow& r1, const URow& r2, const URow& r3, const URow& r4,
                                             const LaneCfg& g, const Buf& ob, uint32_t out_off, int aux) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = chain5(r0.v[j], r1.v[j], r2.v[j], r3.v[j], r4.v[j]);
    const float vh = chain5(r0.h, r1.h, r2.h, r3.h, r4.h);
    float vl6 = from_left_lane(v[6]);
    float vl7 = from_left_lane(v[7]);
    float vr0 = from_right_lane(v[0]);
    const float vh1 = from_right_lane(vh);   // lane 0 receives lane 1's column c0-1
    if (g.lane0) {
        vl6 = g.left_mirror ? v[2] : vh;
        vl7 = g.left_mirror ? v[1] : vh1;
    }
    if (g.last_active) vr0 = v[6];
    else if (g.lane63) vr0 = vh;
    v4f o;
    o.x = chain5(vl6, vl7, v[0], v[1], v[2]);
    o.y = chain5(v[0], v[1], v[2], v[3], v[4]);
    o.z = chain5(v[2], v[3], v[4], v[5], v[6]);
    o.w = chain5(v[4], v[5], v[6], v[7], vr0);
    if (aux == 0) llvm_buffer_store_v4f32(o, ob.r, (int)out_off, 0, 0);
    else llvm_buffer_store_v4f32(o, ob.r, (int)out_off, 0, 2);
}

template <int R, int OCC, int NT>
__global__ __launch_bounds__(kBlockThreads, OCC) void k_reduce_up(const float* __restrict__ in, float* __restrict__ out, int S, int pitch,
                                                                  size_t in_plane, int So, int opitch, size_t out_plane, int swz) {
    const int lane = threadIdx.x & 63;
    const Tile tile = xcd_tile(swz);
    const int seg = __builtin_amdgcn_readfirstlane((int)(tile.segblock * kWavesPerBlock + (threadIdx.x >> 6)));
    const int yo0 = seg * R;
    if (yo0 >= So) return;
    const Buf ib = make_buf(in + (size_t)blockIdx.z * in_plane, in_plane * 4);
    const Buf ob = make_buf(out + (size_t)blockIdx.z * out_plane, out_plane * 4);
    const LaneCfg g = make_cfg(tile.strip, lane, S);
    const int c0 = tile.strip * kStripCols;
    uint32_t hoff = kOob;
    if (lane == 0 && c0 > 0) hoff = (uint32_t)(c0 - 2) * 4u;
    if (lane == 1 && c0 > 0) hoff = (uint32_t)(c0 - 1) * 4u;
    if (lane == 63 && c0 + kStripCols < S) hoff = (uint32_t)(c0 + kStripCols) * 4u;
    const int hi = S - 1;
    const uint32_t rb = (uint32_t)pitch * 4u, orb = (uint32_t)opitch * 4u;
    URow w[2 * R + 3];
#pragma unroll
    for (int k = 0; k < 2 * R + 3; k++) {
        const int y = max(mirror_idx(2 * yo0 - 2 + k, hi), 0);   // rows past a short last segment are requested, never stored
        const uint32_t ro = (uint32_t)min(y, hi) * rb;
        if (NT & 2) {
            const v4f a = llvm_buffer_load_v4f32(ib.r, (int)(g.off + ro), 0, 2), b = llvm_buffer_load_v4f32(ib.r, (int)(g.off + ro + 16u), 0, 2);
            w[k].v[0] = a.x; w[k].v[1] = a.y; w[k].v[2] = a.z; w[k].v[3] = a.w; w[k].v[4] = b.x; w[k].v[5] = b.y; w[k].v[6] = b.z; w[k].v[7] = b.w;
        } else {
            load8(w[k].v, ib, g.off + ro);
        }
        w[k].h = bload1(ib, hoff + ro);
    }
    __builtin_amdgcn_sched_barrier(0);   // every load is issued before the first use (hipcc otherwise sinks the later rows below the first output row)
#pragma unroll
    for (int t = 0; t < R; t++) {
        const int yo = yo0 + t;
        const uint32_t oo = (yo < So && g.coff != kOob) ? g.coff + (uint32_t)yo * orb : kOob;
        reduce_row_u(w[2 * t], w[2 * t + 1], w[2 * t + 2], w[2 * t + 3], w[2 * t + 4], g, ob, oo, NT & 1);
    }
}

// ---- variant LDS: one workgroup = 512 columns x RT output rows, every input row brought in by LDS-DMA --------
// (buffer_load_dwordx4 ... lds: 1 KiB per wave-instruction, no VGPR destination). The 4 wavefronts issue the 2*RT+3 rows
// round-robin, wait for their own DMAs, meet at one barrier and then each computes RT/4 output rows from LDS with the
// register march of the production kernel (rows re-read from LDS, neighbours by DPP, halo columns from a small LDS array
// filled by two 4-byte DMAs). 2*RT+3 rows of 2 KiB: RT = 16 -> 70 KiB, two workgroups per CU.
__device__ __forceinline__ void dma16nt(const Buf& b, uint32_t voff, uint32_t lds_byte) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" ::"s"(lds_byte), "v"(voff), "s"(b.r) : "memory", "m0");
}

// RT output rows per workgroup of W wavefronts; NTS: non-temporal stores; NTL: non-temporal DMA loads;
// SPLIT: the first half of the wavefronts starts computing when the upper half of the rows has landed.
template <int RT, int W, int NTS, int NTL, int SPLIT>
__global__ __launch_bounds__(64 * W) void k_reduce_lds(const float* __restrict__ in, float* __restrict__ out, int S, int pitch,
                                                       size_t in_plane, int So, int opitch, size_t out_plane, int swz) {
    constexpr int NR = 2 * RT + 3;            // input rows of the tile
    constexpr int RW = RT / W;                // output rows per wavefront
    constexpr int NH = (NR * 3 + 63) / 64;    // 4-byte DMA instructions for the halo columns
    constexpr int NI = (2 * NR + W - 1) / W;  // 16-byte DMA instructions per wavefront (row-major over (row, half))
    static_assert(RT % W == 0 && NH <= W, "tile shape");
    __shared__ __attribute__((aligned(16))) float tile_s[NR * kStripCols + NH * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const Tile tile = xcd_tile(swz);
    const int strip = tile.strip;
    const int yo0 = tile.segblock * RT;
    const Buf ib = make_buf(in + (size_t)blockIdx.z * in_plane, in_plane * 4);
    const Buf ob = make_buf(out + (size_t)blockIdx.z * out_plane, out_plane * 4);
    const LaneCfg g = make_cfg(strip, lane, S);
    const int c0 = strip * kStripCols;
    const int hi = S - 1;
    const uint32_t rb = (uint32_t)pitch * 4u, orb = (uint32_t)opitch * 4u;
    const uint32_t lds0 = (uint32_t)(uintptr_t)tile_s;   // LDS byte address of the tile
    const uint32_t lane_off = (uint32_t)(c0 + lane * 4) * 4u;   // 4 floats per lane per instruction, 256 columns per instruction
    // halo columns first (they belong to every row): entry e = 3 * k + j, j = 0: c0-2, 1: c0-1, 2: c0+512
    if (wave < NH) {
        const int e = wave * 64 + lane;
        const int k = e / 3, j = e - 3 * k;
        const int y = min(max(mirror_idx(2 * yo0 - 2 + min(k, NR - 1), hi), 0), hi);
        const int col = j == 0 ? c0 - 2 : j == 1 ? c0 - 1 : c0 + kStripCols;
        const bool ok = k < NR && col >= 0 && col < S;
        dma4(ib, ok ? (uint32_t)y * rb + (uint32_t)col * 4u : kOob, lds0 + (uint32_t)(NR * kStripCols + wave * 64) * 4u);
    }
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int q = min(i * W + wave, 2 * NR - 1);   // 0 .. 2*NR-1 (a wave without a last piece repeats the tile's last one: every wave issues NI)
        {
            const int k = q >> 1, half = q & 1;
            const int y = min(max(mirror_idx(2 * yo0 - 2 + k, hi), 0), hi);
            const uint32_t col_off = lane_off + (uint32_t)half * 1024u;
            const bool ok = c0 + half * 256 + lane * 4 < S;
            if (NTL == 1 || (NTL == 2 && k >= 3 && k < 2 * RT)) dma16nt(ib, ok ? (uint32_t)y * rb + col_off : kOob, lds0 + (uint32_t)(k * kStripCols + half * 256) * 4u);   // NTL == 2: only the rows no other tile reads
            else dma16(ib, ok ? (uint32_t)y * rb + col_off : kOob, lds0 + (uint32_t)(k * kStripCols + half * 256) * 4u);
        }
    }
    const float* hal = tile_s + NR * kStripCols;
    const int t0 = wave * RW;
    auto lds_row = [&](RowR& r, int k) {
        const float4 a = *reinterpret_cast<const float4*>(tile_s + k * kStripCols + lane * 8);
        const float4 b = *reinterpret_cast<const float4*>(tile_s + k * kStripCols + lane * 8 + 4);
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
        r.hl0 = hal[3 * k]; r.hl1 = hal[3 * k + 1]; r.hr = hal[3 * k + 2];
    };
    auto compute = [&]() {
        RowR w[5];
        lds_row(w[0], 2 * t0);
        lds_row(w[1], 2 * t0 + 1);
        lds_row(w[2], 2 * t0 + 2);
#pragma unroll
        for (int t = 0; t < RW; t++) {
            lds_row(w[3], 2 * (t0 + t) + 3);
            lds_row(w[4], 2 * (t0 + t) + 4);
            const int yo = yo0 + t0 + t;
            if (yo < So) {
                if (NTS) {
                    // reduce_row with a non-temporal store
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] = chain5(w[0].v[j], w[1].v[j], w[2].v[j], w[3].v[j], w[4].v[j]);
                    const float vh0 = chain5(w[0].hl0, w[1].hl0, w[2].hl0, w[3].hl0, w[4].hl0);
                    const float vh1 = chain5(w[0].hl1, w[1].hl1, w[2].hl1, w[3].hl1, w[4].hl1);
                    const float vhr = chain5(w[0].hr, w[1].hr, w[2].hr, w[3].hr, w[4].hr);
                    float vl6 = from_left_lane(v[6]), vl7 = from_left_lane(v[7]), vr0 = from_right_lane(v[0]);
                    if (g.lane0) { vl6 = g.left_mirror ? v[2] : vh0; vl7 = g.left_mirror ? v[1] : vh1; }
                    if (g.last_active) vr0 = v[6]; else if (g.lane63) vr0 = vhr;
                    v4f o;
                    o.x = chain5(vl6, vl7, v[0], v[1], v[2]);
                    o.y = chain5(v[0], v[1], v[2], v[3], v[4]);
                    o.z = chain5(v[2], v[3], v[4], v[5], v[6]);
                    o.w = chain5(v[4], v[5], v[6], v[7], vr0);
                    llvm_buffer_store_v4f32(o, ob.r, (int)(g.coff + (uint32_t)yo * orb), 0, NTS);
                } else {
                    reduce_row(w[0], w[1], w[2], w[3], w[4], g, ob, (uint32_t)yo * orb);
                }
            }
            w[0] = w[2]; w[1] = w[3]; w[2] = w[4];
        }
    };
    if (SPLIT) {
        // rows 0 .. 2*(RT/2)+2 feed wavefronts 0 .. W/2-1: q < 2 * (RT + 3) -> this wave's first NA instructions
        constexpr int QA = 2 * (RT + 3);
        constexpr int NA = (QA + W - 1) / W;      // instructions (per wave) that must have landed; an upper bound for every wave
        constexpr int NB = NI - NA;               // may stay in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB > 0 ? NB : 0) : "memory");
        __builtin_amdgcn_s_barrier();
        if (wave < W / 2) compute();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wave >= W / 2) compute();
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        compute();
    }
}

// ---- copy ceilings of the same traffic shape -------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_copy41nt(const float4* __restrict__ in, float* __restrict__ out, int S4, int So4) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x;
    const int yo = blockIdx.y;
    if (xo >= So4) return;
    const float4* r0 = in + (size_t)(2 * yo) * S4 + 2 * xo;
    const float4* r1 = in + (size_t)(2 * yo + 1) * S4 + 2 * xo;
    const float4 a = r0[0], b = r0[1], c = r1[0], d = r1[1];
    v4f o;
    o.x = a.x + b.x + c.x + d.x; o.y = a.y + b.y + c.y + d.y; o.z = a.z + b.z + c.z + d.z; o.w = a.w + b.w + c.w + d.w;
    const Buf ob = make_buf(out, (size_t)So4 * 4 * (So4 * 4) * 4);
    llvm_buffer_store_v4f32(o, ob.r, (int)(((size_t)yo * So4 + xo) * 16), 0, 2);
}
// 8 x 16-byte loads in flight per thread (two output float4 per thread)
__global__ __launch_bounds__(256) void k_copy41x2(const float4* __restrict__ in, float4* __restrict__ out, int S4, int So4) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x;
    const int yo = blockIdx.y * 2;
    if (xo >= So4) return;
    float4 v[8];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float4* p = in + (size_t)(2 * yo + r) * S4 + 2 * xo;
        v[2 * r] = p[0]; v[2 * r + 1] = p[1];
    }
#pragma unroll
    for (int o = 0; o < 2; o++) {
        const float4 a = v[4 * o], b = v[4 * o + 1], c = v[4 * o + 2], d = v[4 * o + 3];
        float4 r;
        r.x = a.x + b.x + c.x + d.x; r.y = a.y + b.y + c.y + d.y; r.z = a.z + b.z + c.z + d.z; r.w = a.w + b.w + c.w + d.w;
        out[(size_t)(yo + o) * So4 + xo] = r;
    }
}

}  // namespace musica

using namespace musica;

struct Variant {
    const char* name;
    void (*launch)(hipStream_t, const float*, float*, int S);
};

template <int R, int OCC, int NT>
static void l_up(hipStream_t st, const float* in, float* out, int S) {
    const int So = S / 2;
    const int strips = (S + kStripCols - 1) / kStripCols, segs = (So + R - 1) / R;
    hipLaunchKernelGGL((k_reduce_up<R, OCC, NT>), dim3(strips, (segs + 3) / 4, 1), dim3(256), 0, st, in, out, S, S, (size_t)S * S, So, So, (size_t)So * So, 1);
}
template <int RT, int W, int NTS, int NTL, int SPLIT, int SWZ>
static void l_lds(hipStream_t st, const float* in, float* out, int S) {
    const int So = S / 2;
    const int strips = (S + kStripCols - 1) / kStripCols;
    hipLaunchKernelGGL((k_reduce_lds<RT, W, NTS, NTL, SPLIT>), dim3(strips, (So + RT - 1) / RT, 1), dim3(64 * W), 0, st, in, out, S, S, (size_t)S * S, So, So, (size_t)So * So, SWZ);
}
template <int RPW>
static void l_prod(hipStream_t st, const float* in, float* out, int S) {
    LevelDesc li{S, S, (size_t)S * S}, lo{S / 2, S / 2, (size_t)(S / 2) * (S / 2)};
    launch_reduce(st, in, li, out, lo, 1, false, 4);   // the production kernel (k_reduce_dma); RPW is unused since round 3
}
static void l_copy41(hipStream_t st, const float* in, float* out, int S) { launch_copy41(st, in, out, S); }
static void l_copy41nt(hipStream_t st, const float* in, float* out, int S) {
    const int So = S / 2, So4 = So / 4;
    hipLaunchKernelGGL(k_copy41nt, dim3((So4 + 255) / 256, So), dim3(256), 0, st, (const float4*)in, out, S / 4, So4);
}
static void l_copy41x2(hipStream_t st, const float* in, float* out, int S) {
    const int So = S / 2, So4 = So / 4;
    hipLaunchKernelGGL(k_copy41x2, dim3((So4 + 255) / 256, So / 2), dim3(256), 0, st, (const float4*)in, (float4*)out, S / 4, So4);
}

int main(int argc, char** argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 4096;
    const int So = S / 2;
    const int NBUF = S >= 8192 ? 3 : 8;
    const int iters = S >= 8192 ? 24 : 64;
    std::vector<Variant> vars = {
        {"prod", l_prod<4>},
        {"up_r4", l_up<4, 4, 0>}, {"up_r4_nts", l_up<4, 4, 1>}, {"up_r4_ntl", l_up<4, 4, 2>}, {"up_r6", l_up<6, 3, 0>}, {"up_r8", l_up<8, 2, 0>}, {"up_r2", l_up<2, 4, 0>}, {"up_r3", l_up<3, 4, 0>},
        {"lds_rt4_nts", l_lds<4, 4, 2, 0, 0, 0>}, {"lds_rt4_nts_ntli", l_lds<4, 4, 2, 2, 0, 0>}, {"lds_rt8_nts", l_lds<8, 4, 2, 0, 0, 0>}, {"lds_rt8_nts_ntli", l_lds<8, 4, 2, 2, 0, 0>},
        {"lds_rt8_w8", l_lds<8, 8, 2, 0, 0, 0>}, {"lds_rt8_w8_ntli", l_lds<8, 8, 2, 2, 0, 0>}, {"lds_rt3_w3_ntli", l_lds<3, 3, 2, 2, 0, 0>}, {"lds_rt5_w5_ntli", l_lds<5, 5, 2, 2, 0, 0>}, {"lds_rt4_w2_ntli", l_lds<4, 2, 2, 2, 0, 0>}, {"lds_rt2_w2_ntli", l_lds<2, 2, 2, 2, 0, 0>}, {"lds_rt6_w3", l_lds<6, 3, 2, 0, 0, 0>}, {"lds_rt6_w3_ntli", l_lds<6, 3, 2, 2, 0, 0>},
        {"copy41", l_copy41}, {"copy41x2", l_copy41x2}, {"copy41_nts", l_copy41nt},
    };
    const size_t ip = (size_t)S * S, op = (size_t)So * So;
    float *in, *out, *ref;
    CK(hipMalloc(&in, NBUF * ip * 4));
    CK(hipMalloc(&out, NBUF * op * 4));
    CK(hipMalloc(&ref, op * 4));
    {
        std::vector<float> h(ip);
        uint32_t s = 12345u;
        for (size_t i = 0; i < ip; i++) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) * (1.0f / 16777216.0f); }
        for (int b = 0; b < NBUF; b++) CK(hipMemcpy(in + b * ip, h.data(), ip * 4, hipMemcpyHostToDevice));
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    l_prod<4>(st, in, ref, S);
    CK(hipStreamSynchronize(st));
    std::vector<float> href(op), hout(op);
    CK(hipMemcpy(href.data(), ref, op * 4, hipMemcpyDeviceToHost));
    const double bytes = 5.0 * S * S;
    for (int rep = 0; rep < 2; rep++) {
        for (auto& v : vars) {
            bool want = argc <= 2;
            for (int i = 2; i < argc; i++) want |= !strcmp(argv[i], v.name);
            if (!want) continue;
            CK(hipMemset(out, 0xff, op * 4));
            v.launch(st, in, out, S);
            CK(hipStreamSynchronize(st));
            CK(hipGetLastError());
            CK(hipMemcpy(hout.data(), out, op * 4, hipMemcpyDeviceToHost));
            const bool same = !memcmp(hout.data(), href.data(), op * 4);
            for (int i = 0; i < 8; i++) v.launch(st, in + (size_t)(i % NBUF) * ip, out + (size_t)(i % NBUF) * op, S);
            CK(hipEventRecord(a, st));
            for (int i = 0; i < iters; i++) v.launch(st, in + (size_t)(i % NBUF) * ip, out + (size_t)(i % NBUF) * op, S);
            CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            const double us = ms * 1000.0 / iters;
            // one event pair per launch: the kernel without the inter-launch boundary
            std::vector<float> t;
            for (int i = 0; i < 24; i++) {
                CK(hipEventRecord(a, st));
                v.launch(st, in + (size_t)(i % NBUF) * ip, out + (size_t)(i % NBUF) * op, S);
                CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
                float m1; CK(hipEventElapsedTime(&m1, a, b)); t.push_back(m1 * 1000);
            }
            std::sort(t.begin(), t.end());
            printf("%-12s S=%d rep%d  %7.2f us back-to-back = %5.0f GB/s = %.3f of 8 TB/s | single-launch median %7.2f us | %s\n", v.name, S, rep, us,
                   bytes / (us * 1e-6) / 1e9, bytes / (us * 1e-6) / 8e12, t[t.size() / 2], same ? "bits==prod" : (strncmp(v.name, "copy", 4) ? "MISMATCH" : "-"));
            fflush(stdout);
        }
    }
    return 0;
}
