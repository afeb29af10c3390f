// Dev tool: practical HBM ceiling for the metric kernel's traffic shape (read S*S f32, write (S/2)^2 f32)
// and per-launch timing of musica_k_reduce with one event pair per launch (no inter-launch gap).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "../include/musica.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// each thread: 4 x 16-byte loads (two rows x two groups = the 8x... footprint of 4 outputs) -> 1 x 16-byte store
__global__ __launch_bounds__(256) void k_copy41(const float4* __restrict__ in, float4* __restrict__ out, int S4 /*float4 per input row*/, int So4) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x;   // output float4 index in row
    const int yo = blockIdx.y;
    if (xo >= So4) return;
    const float4* r0 = in + (size_t)(2 * yo) * S4 + 2 * xo;
    const float4* r1 = in + (size_t)(2 * yo + 1) * S4 + 2 * xo;
    const float4 a = r0[0], b = r0[1], c = r1[0], d = r1[1];
    float4 o;
    o.x = a.x + b.x + c.x + d.x; o.y = a.y + b.y + c.y + d.y; o.z = a.z + b.z + c.z + d.z; o.w = a.w + b.w + c.w + d.w;
    out[(size_t)yo * So4 + xo] = o;
}

// grid-stride flat version: every thread streams 4 loads + 1 store per trip
__global__ __launch_bounds__(256) void k_copy41_flat(const float4* __restrict__ in, float4* __restrict__ out, size_t nout) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nout; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
        float4 o;
        o.x = a.x + b.x + c.x + d.x; o.y = a.y + b.y + c.y + d.y; o.z = a.z + b.z + c.z + d.z; o.w = a.w + b.w + c.w + d.w;
        out[i] = o;
    }
}

static double med(std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 4096;
    const int B = argc > 2 ? atoi(argv[2]) : 1;
    const int So = S / 2;
    float *in, *out;
    CK(hipMalloc(&in, (size_t)B * S * S * 4));
    CK(hipMalloc(&out, (size_t)B * So * So * 4));
    CK(hipMemset(in, 0x3c, (size_t)B * S * S * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double bytes = 5.0 * S * S * B;
    const int iters = 200;
    // (1) 2-D copy-shaped kernel
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a, st));
        for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_copy41, dim3((So / 4 + 255) / 256, So * B), dim3(256), 0, st, (const float4*)in, (float4*)out, S / 4, So / 4);
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep) printf("copy41 2-D      S=%d B=%d: %.2f us/launch back-to-back -> %.0f GB/s\n", S, B, ms * 1000 / iters, bytes / (ms / iters * 1e-3) / 1e9);
    }
    for (int blocks : {1024, 2048, 4096, 8192}) {
        CK(hipEventRecord(a, st));
        for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_copy41_flat, dim3(blocks), dim3(256), 0, st, (const float4*)in, (float4*)out, (size_t)B * So * So / 4);
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("copy41 flat %5d blocks: %.2f us/launch back-to-back -> %.0f GB/s\n", blocks, ms * 1000 / iters, bytes / (ms / iters * 1e-3) / 1e9);
    }
    // per-launch events (single kernel, no gap)
    {
        std::vector<float> t;
        for (int i = 0; i < 50; i++) {
            CK(hipEventRecord(a, st));
            hipLaunchKernelGGL(k_copy41_flat, dim3(2048), dim3(256), 0, st, (const float4*)in, (float4*)out, (size_t)B * So * So / 4);
            CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms * 1000);
        }
        printf("copy41 flat 2048 blocks, one event pair per launch: median %.2f us\n", med(t));
    }
    // (2) the metric kernel through the C ABI
    musica_params p = {64, 4, 1, 0, 0};
    musica_ctx* c = musica_create(&p);
    if (!c) return 1;
    double us = 0;
    musica_k_reduce_timed(c, in, S, S, out, So, B, 20, &us);
    musica_k_reduce_timed(c, in, S, S, out, So, B, iters, &us);
    printf("musica_k_reduce S=%d B=%d: %.2f us/launch back-to-back -> %.0f GB/s\n", S, B, us, bytes / (us * 1e-6) / 1e9);
    {
        std::vector<float> t;
        for (int i = 0; i < 50; i++) { double u; musica_k_reduce_timed(c, in, S, S, out, So, B, 1, &u); t.push_back((float)u); }
        printf("musica_k_reduce one event pair per launch: median %.2f us -> %.0f GB/s\n", med(t), bytes / (med(t) * 1e-6) / 1e9);
    }
    musica_destroy(c);
    return 0;
}
