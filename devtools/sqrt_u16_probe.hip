// Is the hardware v_sqrt_f32 correctly rounded for every integer 0 .. 65535 (the raw pixel domain of img_sqrt.comp)?
// Answer on gfx950 (round 3): no — wrong by one ulp on 10168 of the 65536 integers (6, 11, 14, 24, 30, ...), so the level-0 kernels keep
// the rsq + residual form (exact_math.h musica_sqrt_core: ~9 vector-unit slots per pixel against 4).
// hipcc --offload-arch=gfx950 -O2 devtools/sqrt_u16_probe.hip -o devtools/sqrt_u16_probe && ./devtools/sqrt_u16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(float* hw, float* ref) {
    const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
    hw[v] = __builtin_amdgcn_sqrtf((float)v);
    ref[v] = sqrtf((float)v);
}
int main() {
    float *hw, *ref;
    hipMalloc(&hw, 65536 * 4); hipMalloc(&ref, 65536 * 4);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, hw, ref);
    static float a[65536], b[65536];
    hipMemcpy(a, hw, sizeof a, hipMemcpyDeviceToHost); hipMemcpy(b, ref, sizeof b, hipMemcpyDeviceToHost);
    int bad_hw = 0, bad_dev = 0;
    for (int v = 0; v < 65536; v++) {
        const float want = (float)std::sqrt((double)v);   // correctly rounded: double sqrt then one rounding is exact for 16-bit integers
        if (a[v] != want) { if (bad_hw < 5) printf("v_sqrt_f32(%d) = %.9g, want %.9g\n", v, a[v], want); bad_hw++; }
        if (b[v] != want) bad_dev++;
    }
    printf("v_sqrt_f32 wrong on %d of 65536 integers; device sqrtf wrong on %d\n", bad_hw, bad_dev);
    return 0;
}
