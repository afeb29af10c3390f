// Dev tool (round 4): what a vector instruction costs on a gfx950 SIMD at the occupancy the level-0 marches run at (2 wavefronts per
// SIMD) — scalar-lane f32 ops, packed f32 ops, v_mov, v_cndmask, v_rsq, DPP moves — as shader cycles (s_memtime) per instruction per
// SIMD. Every CU gets W wavefronts per SIMD (one workgroup of 256 W threads per CU); each runs ITER x 64 independent instructions.
//   hipcc --offload-arch=gfx950 -O3 devtools/valu_cost.hip -o devtools/valu_cost && devtools/valu_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void k_cost(float* out, unsigned long long* cyc, int iters) {
    float a = threadIdx.x * 1.0f, b = 1.0001f, c = 0.5f, d = 2.0f, e = 3.0f, f = 4.0f, g = 5.0f, h = 6.0f;
    v2f p0 = {a, b}, p1 = {c, d}, p2 = {e, f}, p3 = {g, h}, q = {1.0001f, 0.9999f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {   // scalar-lane fma, 8 independent chains
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(q.x));
        } else if (KIND == 1) {   // packed fma, 4 independent chains
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                              "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
        } else if (KIND == 2) {   // v_mov
            asm volatile(REP8("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        } else if (KIND == 3) {   // v_rsq
            asm volatile(REP8("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        } else if (KIND == 4) {   // packed mul
            asm volatile(REP8("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                              "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
        } else if (KIND == 5) {   // scalar-lane add
            asm volatile(REP8("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                              "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(q.x));
        } else if (KIND == 6) {   // v_cndmask with an SGPR-pair mask (VOP3)
            asm volatile(REP8("v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc\n v_cndmask_b32_e64 %2, %2, %3, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc\n"
                              "v_cndmask_b32_e64 %4, %4, %5, vcc\n v_cndmask_b32_e64 %5, %5, %6, vcc\n v_cndmask_b32_e64 %6, %6, %7, vcc\n v_cndmask_b32_e64 %7, %7, %0, vcc\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : : "vcc");
        } else if (KIND == 7) {   // DPP wave shift move
            asm volatile(REP8("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %5, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        } else if (KIND == 8) {   // v_cvt_f32_u32 sdwa-free conversion
            asm volatile(REP8("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n v_cvt_f32_u32 %4, %4\n v_cvt_f32_u32 %5, %5\n v_cvt_f32_u32 %6, %6\n v_cvt_f32_u32 %7, %7\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(const char* name, float* out, unsigned long long* cyc, hipStream_t st) {
    const int iters = 2000, cus = 256;
    for (int W : {1, 2, 4}) {
        const int threads = 256 * W;   // 4 SIMDs x W wavefronts
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL((k_cost<KIND>), dim3(cus), dim3(threads), 0, st, out, cyc, 10);
        CK(hipEventRecord(a, st));
        hipLaunchKernelGGL((k_cost<KIND>), dim3(cus), dim3(threads), 0, st, out, cyc, iters);
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        unsigned long long h[4096];
        const int nw = cus * threads / 64;
        CK(hipMemcpy(h, cyc, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
        double mean = 0; for (int i = 0; i < nw; i++) mean += (double)h[i]; mean /= nw;
        const double insts = (double)iters * 64;
        // s_memtime ticks at a fixed 100 MHz on gfx9; report wall-derived ns and the tick count
        printf("%-14s W=%d  wall %.1f us  -> %.2f ns per instruction per wavefront, %.2f ns per instruction per SIMD (memtime ticks/inst/wave %.3f)\n", name, W, ms * 1000,
               ms * 1e6 / insts, ms * 1e6 / insts / W, mean / insts);
    }
}

int main() {
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&cyc, 4096 * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    run<0>("v_fma_f32", out, cyc, st);
    run<5>("v_add_f32", out, cyc, st);
    run<1>("v_pk_fma_f32", out, cyc, st);
    run<4>("v_pk_mul_f32", out, cyc, st);
    run<2>("v_mov_b32", out, cyc, st);
    run<6>("v_cndmask", out, cyc, st);
    run<7>("v_mov_dpp", out, cyc, st);
    run<8>("v_cvt_f32_u32", out, cyc, st);
    run<3>("v_rsq_f32", out, cyc, st);
    return 0;
}
