// Dev tool: what a plain 1:1 read/write stream reaches from HBM on this part — the ceiling of k_grad_apply (reads 134 MB, writes 134 MB
// at C4). Rotates over 4 source / destination pairs (1 GB footprint, beyond the 256 MiB Infinity Cache).
//   hipcc --offload-arch=gfx950 -O3 devtools/stream11.hip -o devtools/stream11 && devtools/stream11 [MB per direction]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U>
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) { v[u].x += 1.0f; out[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) { float4 v = in[i]; v.x += 1.0f; out[i] = v; }
}

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 134;
    const size_t n = mb * 1000000 / 16;
    const int NB = 4;
    float4 *in[NB], *out[NB];
    for (int k = 0; k < NB; k++) { CK(hipMalloc(&in[k], n * 16)); CK(hipMalloc(&out[k], n * 16)); CK(hipMemset(in[k], 0x3c, n * 16)); }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 40;
    for (int blocks : {2048, 4096, 8192, 16384, 65536}) {
        for (int U : {1, 4}) {
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(a, st));
                for (int i = 0; i < iters; i++) {
                    if (U == 1) hipLaunchKernelGGL(k_stream<1>, dim3(blocks), dim3(256), 0, st, in[i % NB], out[i % NB], n);
                    else hipLaunchKernelGGL(k_stream<4>, dim3(blocks), dim3(256), 0, st, in[i % NB], out[i % NB], n);
                }
                CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (rep) printf("1:1 stream %zu MB each way, %5d blocks, U=%d: %.2f us per launch -> %.0f GB/s\n", mb, blocks, U, ms * 1000 / iters, 2.0 * n * 16 / (ms / iters * 1e-3) / 1e9);
            }
        }
    }
    // hipMemcpyAsync device-to-device for comparison
    CK(hipEventRecord(a, st));
    for (int i = 0; i < iters; i++) CK(hipMemcpyAsync(out[i % NB], in[i % NB], n * 16, hipMemcpyDeviceToDevice, st));
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("hipMemcpyAsync D2D %zu MB: %.2f us per copy -> %.0f GB/s\n", mb, ms * 1000 / iters, 2.0 * n * 16 / (ms / iters * 1e-3) / 1e9);
    return 0;
}
