// Probe: which cheap sqrt sequences equal the correctly rounded sqrtf for every non-negative float on gfx950?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
__device__ __forceinline__ float sq1(float x) {
    float y = __builtin_amdgcn_rsqf(x); y = fminf(y, 3.0e38f);
    float s = x * y, h = 0.5f * y;
    float r = fmaf(-s, s, x);
    return fmaf(r, h, s);
}
__device__ __forceinline__ float sq2(float x) {
    float y = __builtin_amdgcn_rsqf(x); y = fminf(y, 3.0e38f);
    float s = x * y, h = 0.5f * y;
    float r = fmaf(-s, s, x);
    s = fmaf(r, h, s);
    r = fmaf(-s, s, x);
    return fmaf(r, h, s);
}
__device__ __forceinline__ float sq3(float x) {   // v_sqrt + one residual step with h from rsq
    float s = __builtin_amdgcn_sqrtf(x);
    float h = 0.5f * fminf(__builtin_amdgcn_rsqf(x), 3.0e38f);
    float r = fmaf(-s, s, x);
    return fmaf(r, h, s);
}
__global__ void probe(unsigned long long* bad, uint32_t* first) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= 0x7f800000ull; b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const uint32_t want = __float_as_uint(sqrtf(x));
        const uint32_t g[3] = {__float_as_uint(sq1(x)), __float_as_uint(sq2(x)), __float_as_uint(sq3(x))};
        const int cls = b < 0x00800000u ? 0 : (b < 0x0d800000u ? 1 : (b > 0x71800000u ? 3 : 2));  // denormal, < 2^-100, > 2^100, middle
        for (int k = 0; k < 3; k++)
            if (g[k] != want) { atomicAdd(&bad[k * 4 + cls], 1ull); atomicMin(&first[k * 4 + cls], (uint32_t)b); }
    }
}
int main() {
    unsigned long long* d_bad; uint32_t* d_first;
    hipMalloc(&d_bad, 12 * 8); hipMalloc(&d_first, 12 * 4);
    hipMemset(d_bad, 0, 12 * 8); hipMemset(d_first, 0xff, 12 * 4);
    probe<<<4096, 256>>>(d_bad, d_first);
    unsigned long long bad[12]; uint32_t first[12];
    hipMemcpy(bad, d_bad, sizeof bad, hipMemcpyDeviceToHost); hipMemcpy(first, d_first, sizeof first, hipMemcpyDeviceToHost);
    const char* cls[4] = {"denormal", "<2^-100", "middle", ">2^100"};
    for (int k = 0; k < 3; k++) for (int c = 0; c < 4; c++) printf("variant %d %-9s mismatches %llu first 0x%08x\n", k + 1, cls[c], bad[k * 4 + c], first[k * 4 + c]);
    return 0;
}
