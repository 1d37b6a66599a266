// Microbenchmark (GPU box): issue rate of plain, packed and double-precision VALU instructions on gfx950,
// every SIMD busy with 4 waves: what a "VALU instruction" costs when the kernel's VALU utilisation is priced.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float F2 __attribute__((ext_vector_type(2)));
template <int kKind> __global__ __launch_bounds__(256) void k(float* out, int iters, float seed, unsigned long long* clk)
{
    extern __shared__ float lds_pad[];          // 40 KB of dynamic LDS per block: exactly four blocks fit a CU, so the launch spreads evenly
    if (seed == 12345.0f) lds_pad[threadIdx.x] = seed;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();     // s_memtime (shader clock), s_memrealtime (100 MHz)
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    F2 p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a4, a5 }, p3 = { a6, a7 }, p4 = p0 + 1.0f, p5 = p1 + 1.0f, p6 = p2 + 1.0f, p7 = p3 + 1.0f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    const float m = 1.0000001f; const F2 pm = { m, m }; const double dm = 1.0000001;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (kKind == 0) { asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m)); }
            if (kKind == 1) { asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm)); }
            if (kKind == 2) { asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dm)); }
            if (kKind == 3) { asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
            if (kKind == 4) { asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm)); }
            if (kKind == 5) { asm volatile("v_fma_f64 %0, %0, %8, %0\n v_fma_f64 %1, %1, %8, %1\n v_fma_f64 %2, %2, %8, %2\n v_fma_f64 %3, %3, %8, %3\n v_fma_f64 %4, %4, %8, %4\n v_fma_f64 %5, %5, %8, %5\n v_fma_f64 %6, %6, %8, %6\n v_fma_f64 %7, %7, %8, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dm)); }
            if (kKind == 6) { asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dm)); }
            if (kKind == 7) { asm volatile("v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 3, %1\n v_lshlrev_b64 %2, 3, %2\n v_lshlrev_b64 %3, 3, %3\n v_lshlrev_b64 %4, 3, %4\n v_lshlrev_b64 %5, 3, %5\n v_lshlrev_b64 %6, 3, %6\n v_lshlrev_b64 %7, 3, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)); }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}
template <int kKind> void run(const char* name, float* d, int blocks)
{
    unsigned long long* clk; hipMalloc(&clk, 16); unsigned long long h[2] = { 0, 0 };
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<kKind><<<blocks, 256, 40 * 1024>>>(d, 10, 1.0f, clk);
    hipEventRecord(e0); k<kKind><<<blocks, 256, 40 * 1024>>>(d, iters, 1.0f, clk); hipEventRecord(e1); hipEventSynchronize(e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); hipFree(clk);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 * iters * 16 * 8;   // wave-instructions
    // one wavefront's own clocks: it shares its SIMD with 3 others, so the SIMD issued 4 * iters * 128 instructions in h[0] shader cycles
    const double per_simd = (double)h[0] / (4.0 * iters * 16 * 8), ghz = (double)h[0] / ((double)h[1] * 10.0);
    printf("%-14s %8.3f ms  %8.1f G wave-instr/s  = %.2f cycles per instruction and SIMD at an assumed 2.4 GHz;  s_memtime: %.2f cycles per instruction and SIMD, shader clock %.3f GHz\n", name, ms,
           insts / ms / 1e6, 1024 * 2.4e9 / (insts / (ms * 1e-3)), per_simd, ghz);
}
int main()
{
    float* d; hipMalloc(&d, 256 * 4 * 4096 * 4);
    const int blocks = 256 * 4;     // 4 blocks of 4 waves per CU: 4 waves per SIMD
    run<0>("v_mul_f32", d, blocks); run<1>("v_pk_mul_f32", d, blocks); run<4>("v_pk_fma_f32", d, blocks); run<2>("v_mul_f64", d, blocks);
    run<5>("v_fma_f64", d, blocks); run<6>("v_add_f64", d, blocks); run<3>("v_rcp_f32", d, blocks); run<7>("v_lshlrev_b64", d, blocks);
    return 0;
}
