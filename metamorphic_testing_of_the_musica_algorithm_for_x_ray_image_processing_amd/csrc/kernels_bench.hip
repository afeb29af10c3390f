// kernels_bench.hip — measurement aid, not on the product path (gfx950).
//
//   k_copy41 : a plain streaming kernel with the traffic shape of the metric kernel (img_smooth.comp +
//              img_downsample.comp fused: read S^2 f32 once, write (S/2)^2 f32 once) and nothing else —
//              no halo rows, no halo columns, three adds per loaded float4. bench.py times it beside
//              k_reduce_dma, the same rotating-buffer way, so that the metric kernel's fraction of the
//              8 TB/s HBM peak can also be read against what a streaming kernel of this shape attains on
//              the same box (musica_k_copy41_timed_rot, include/musica.h). A grid-stride form (2048 / 4096 workgroups streaming
//              4 contiguous 16-byte loads per trip) measured the same 16.0 us at 4096^2 from HBM, non-temporal loads 22 us.
#include <stdlib.h>
#include "kernels_common.h"
#include "launchers.h"

namespace musica {

// One thread per output float4: the 2 x 2 float4 block of input above it (two rows, 32 contiguous bytes each),
// i.e. a 256-thread block reads 2 x 8 KiB of two consecutive input rows and writes 4 KiB of one output row.
// TAG: 0 = sides <= 4096, 1 = above (keeps bench.py's two sizes apart in a profiler's per-symbol averages)
template <int TAG>
__global__ __launch_bounds__(256) void k_copy41(const float4* __restrict__ in, float4* __restrict__ out, int S4 /* float4 per input row */, int So4) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x;
    const int yo = blockIdx.y;
    if (xo >= So4) return;
    const float4* r0 = in + (size_t)(2 * yo) * S4 + 2 * xo;
    const float4* r1 = in + (size_t)(2 * yo + 1) * S4 + 2 * xo;
    const float4 a = r0[0], b = r0[1], c = r1[0], d = r1[1];
    float4 o;
    o.x = a.x + b.x + c.x + d.x; o.y = a.y + b.y + c.y + d.y; o.z = a.z + b.z + c.z + d.z; o.w = a.w + b.w + c.w + d.w;
    out[(size_t)yo * So4 + xo] = o;
}

void launch_copy41(hipStream_t st, const float* in, float* out, int side) {
    const int So = side / 2, So4 = So / 4;
    auto* kern = side <= 4096 ? k_copy41<0> : k_copy41<1>;
    hipLaunchKernelGGL(kern, dim3((So4 + 255) / 256, So), dim3(256), 0, st, reinterpret_cast<const float4*>(in), reinterpret_cast<float4*>(out),
                       side / 4, So4);
}

}  // namespace musica
