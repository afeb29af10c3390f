// Dev tool (round 4): what plain streaming kernels reach from HBM for the READ : WRITE mixes of the level-0 kernels of a C4 step
// (8 x 2048^2) — the ceilings the marching kernels are measured against. One thread = 8 pixels of a row; no stencil, no halo.
//   rb0   : read 2 B/px (uint16), write 4 B/px (band) + 1 B/px (coarse)          = k_reduce_band<true>   (7 B/px)
//   sdev  : read 4 B/px, write 4 B/px                                            = k_sdev_hist_pf        (8 B/px)
//   exp0  : read 4 + 4 B/px (band, sdev) + 1 B/px (coarse), write 4 B/px          = k_expand_fast<GH>     (13 B/px)
//   apply : read 4, write 4 (the 1:1 stream of devtools/stream11.hip)
//   rd / wr : read-only (sum into a dummy) / write-only
// Rotates over NB buffer sets (footprint beyond the 256 MiB Infinity Cache).
//   hipcc --offload-arch=gfx950 -O3 devtools/stream_mix.hip -o devtools/stream_mix && devtools/stream_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4(float* p, v4f v, int nt) {
    if (nt) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
    else *reinterpret_cast<v4f*>(p) = v;
}

// mode bits: 1 read u16 (2 B/px), 2 read f32 a (4 B/px), 4 read f32 b (4 B/px), 8 read coarse (1 B/px)
//            16 write f32 (4 B/px), 32 write coarse (1 B/px);  nt: non-temporal stores
template <int MODE, int NT>
__global__ __launch_bounds__(256) void k_mix(const uint16_t* __restrict__ u, const float* __restrict__ a, const float* __restrict__ b,
                                             const float* __restrict__ c, float* __restrict__ o, float* __restrict__ oc, float* __restrict__ sink, size_t groups) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += stride) {   // group = 8 pixels
        v4f x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0};
        if (MODE & 1) {
            const uint4 q = *reinterpret_cast<const uint4*>(u + i * 8);
            x0.x += (float)(q.x & 0xFFFF); x0.y += (float)(q.x >> 16); x0.z += (float)(q.y & 0xFFFF); x0.w += (float)(q.y >> 16);
            x1.x += (float)(q.z & 0xFFFF); x1.y += (float)(q.z >> 16); x1.z += (float)(q.w & 0xFFFF); x1.w += (float)(q.w >> 16);
        }
        if (MODE & 2) { x0 += *reinterpret_cast<const v4f*>(a + i * 8); x1 += *reinterpret_cast<const v4f*>(a + i * 8 + 4); }
        if (MODE & 4) { x0 += *reinterpret_cast<const v4f*>(b + i * 8); x1 += *reinterpret_cast<const v4f*>(b + i * 8 + 4); }
        if (MODE & 8) { const float2 t = *reinterpret_cast<const float2*>(c + i * 2); x0.x += t.x; x1.x += t.y; }
        if (MODE & 16) { st4(o + i * 8, x0, NT); st4(o + i * 8 + 4, x1, NT); }
        if (MODE & 32) { float2 t; t.x = x0.x + x0.y; t.y = x1.x + x1.y; *reinterpret_cast<float2*>(oc + i * 2) = t; }
        if (!(MODE & 16)) acc += x0.x + x0.y + x0.z + x0.w + x1.x + x1.y + x1.z + x1.w;
    }
    if (!(MODE & 16) && acc == 123.456f) sink[0] = acc;
}

struct Set { uint16_t* u; float *a, *b, *c, *o, *oc; };

template <int MODE, int NT>
static void run(const char* name, Set* s, int NB, float* sink, size_t px, hipStream_t st, hipEvent_t ea, hipEvent_t eb, double bytes_per_px) {
    const size_t groups = px / 8;
    for (int blocks : {4096, 16384, 65536}) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            const int iters = 24;
            CK(hipEventRecord(ea, st));
            for (int i = 0; i < iters; i++) {
                const Set& q = s[i % NB];
                hipLaunchKernelGGL((k_mix<MODE, NT>), dim3(blocks), dim3(256), 0, st, q.u, q.a, q.b, q.c, q.o, q.oc, sink, groups);
            }
            CK(hipEventRecord(eb, st)); CK(hipEventSynchronize(eb));
            float ms; CK(hipEventElapsedTime(&ms, ea, eb));
            if (rep && ms / iters < best) best = ms / iters;
        }
        printf("%-28s nt=%d %6d blocks: %7.2f us per launch  %6.0f GB/s  (%.3f of 8 TB/s)\n", name, NT, blocks, best * 1000, bytes_per_px * px / (best * 1e-3) / 1e9,
               bytes_per_px * px / (best * 1e-3) / 8e12);
    }
}

int main(int argc, char** argv) {
    const size_t px = (size_t)(argc > 1 ? atoi(argv[1]) : 8) * 2048 * 2048;
    const int NB = argc > 2 ? atoi(argv[2]) : 3;   // buffer sets rotated over (8 x (67 + 134 + ...) MB: beyond the Infinity Cache for every stream)
    Set s[16];
    for (int k = 0; k < NB; k++) {
        CK(hipMalloc(&s[k].u, px * 2)); CK(hipMalloc(&s[k].a, px * 4)); CK(hipMalloc(&s[k].b, px * 4)); CK(hipMalloc(&s[k].c, px)); CK(hipMalloc(&s[k].o, px * 4)); CK(hipMalloc(&s[k].oc, px));
        CK(hipMemset(s[k].u, 0x11, px * 2)); CK(hipMemset(s[k].a, 0x3c, px * 4)); CK(hipMemset(s[k].b, 0x3c, px * 4)); CK(hipMemset(s[k].c, 0x3c, px));
    }
    float* sink; CK(hipMalloc(&sink, 64));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    printf("pixels per launch: %zu\n", px);
    if (argc > 3 && atoi(argv[3]) == 2) {   // a read-only launch behind a write-heavy one, alternating (rocprofv3 --kernel-trace gives the per-kernel durations):
        const size_t groups = px / 8;      // does the 67 MB read slow down while the previous launch's stores drain?
        for (int i = 0; i < 60; i++) {
            const Set& q = s[i % NB];
            const Set& r = s[(i + NB / 2) % NB];
            hipLaunchKernelGGL((k_mix<2 | 16, 0>), dim3(16384), dim3(256), 0, st, q.u, q.a, q.b, q.c, q.o, q.oc, sink, groups);   // read 134 MB, write 134 MB
            hipLaunchKernelGGL((k_mix<1, 0>), dim3(4096), dim3(256), 0, st, r.u, r.a, r.b, r.c, r.o, r.oc, sink, groups);          // read 67 MB
        }
        CK(hipStreamSynchronize(st));
        return 0;
    }
    if (argc > 3) {   // reads only
        run<1, 0>("read u16 only (r2)", s, NB, sink, px, st, a, b, 2.0);
        run<2, 0>("read f32 only (r4)", s, NB, sink, px, st, a, b, 4.0);
        return 0;
    }
    run<1 | 16 | 32, 0>("rb0 (r2 w4+1)", s, NB, sink, px, st, a, b, 7.0);
    run<1 | 16 | 32, 1>("rb0 (r2 w4+1)", s, NB, sink, px, st, a, b, 7.0);
    run<2 | 16, 0>("sdev / apply (r4 w4)", s, NB, sink, px, st, a, b, 8.0);
    run<2 | 16, 1>("sdev / apply (r4 w4)", s, NB, sink, px, st, a, b, 8.0);
    run<2 | 4 | 8 | 16, 0>("exp0 (r4+4+1 w4)", s, NB, sink, px, st, a, b, 13.0);
    run<2 | 4 | 8 | 16, 1>("exp0 (r4+4+1 w4)", s, NB, sink, px, st, a, b, 13.0);
    run<1, 0>("read u16 only (r2)", s, NB, sink, px, st, a, b, 2.0);
    run<2, 0>("read f32 only (r4)", s, NB, sink, px, st, a, b, 4.0);
    run<2 | 4, 0>("read 2 x f32 (r8)", s, NB, sink, px, st, a, b, 8.0);
    run<16, 0>("write only (w4)", s, NB, sink, px, st, a, b, 4.0);
    run<16, 1>("write only (w4)", s, NB, sink, px, st, a, b, 4.0);
    return 0;
}
