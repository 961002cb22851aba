// microbench.hip — issue-rate probes for the integer instructions the field multiplication is built from.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o gpurun_out/microbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 64
#define ITERS 2000
template <int KIND> __global__ void k(uint32_t *out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c0 = 1, c1 = 2, c2 = 3, c3 = 4, c4 = 5, c5 = 6, c6 = 7, c7 = 8;
    uint64_t d0 = a, d1 = b, d2 = a + 1, d3 = b + 1, d4 = 5, d5 = 6, d6 = 7, d7 = 8;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (KIND == 0) {   // v_mad_u64_u32, 8 independent chains
                asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                             "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b) : "vcc");
            } else if (KIND == 1) {   // v_mul_lo_u32
                asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                             "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a));
            } else if (KIND == 2) {   // v_add_u32 (full rate reference)
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a));
            } else if (KIND == 3) {   // v_lshl_add_u64
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %4\n"
                             "v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %5, %5, 0, %6\n v_lshl_add_u64 %6, %6, 0, %7\n v_lshl_add_u64 %7, %7, 0, %0\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
            } else if (KIND == 4) {   // v_mul_hi_u32
                asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
                             "v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a));
            } else if (KIND == 5) {   // v_mad_u32_u24 (24-bit multiply-add)
                asm volatile("v_mad_u32_u24 %0, %0, %8, %0\n v_mad_u32_u24 %1, %1, %8, %1\n v_mad_u32_u24 %2, %2, %8, %2\n v_mad_u32_u24 %3, %3, %8, %3\n"
                             "v_mad_u32_u24 %4, %4, %8, %4\n v_mad_u32_u24 %5, %5, %8, %5\n v_mad_u32_u24 %6, %6, %8, %6\n v_mad_u32_u24 %7, %7, %8, %7\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a));
            } else if (KIND == 6) {   // v_addc_co_u32 chain through vcc (dependent carries, hazard nops as the compiler pads them)
                asm volatile("v_add_co_u32 %0, vcc, %0, %8\n s_nop 1\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n s_nop 1\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n s_nop 1\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n s_nop 1\n"
                             "v_addc_co_u32 %4, vcc, %4, %8, vcc\n s_nop 1\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n s_nop 1\n v_addc_co_u32 %6, vcc, %6, %8, vcc\n s_nop 1\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n s_nop 1\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a) : "vcc");
            } else if (KIND == 7) {   // v_fma_f64
                double *e = (double *)&d0; (void)e;
                asm volatile("v_fma_f64 %0, %0, %0, %1\n v_fma_f64 %1, %1, %1, %2\n v_fma_f64 %2, %2, %2, %3\n v_fma_f64 %3, %3, %3, %4\n"
                             "v_fma_f64 %4, %4, %4, %5\n v_fma_f64 %5, %5, %5, %6\n v_fma_f64 %6, %6, %6, %7\n v_fma_f64 %7, %7, %7, %0\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}
template <int KIND> void run(const char *name, uint32_t *d_out, int waves_per_simd) {
    int cus = 256; hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); cus = p.multiProcessorCount;
    dim3 grid(cus * waves_per_simd), block(256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d_out, 1u);
    hipEventRecord(a); hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d_out, 2u); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double insts = (double)grid.x * 4 /*waves per block*/ * ITERS * REP;          // wave-instructions
    double per_simd = insts / (cus * 4.0);
    double cyc = ms * 1e-3 * p.clockRate * 1e3;                                     // at nominal clock
    printf("%-16s waves/SIMD=%d  %.3f ms  wave-instr/SIMD=%.0f  ~cycles/wave-instr/SIMD=%.2f  chip Ginstr*lanes/s=%.1f\n", name, waves_per_simd, ms, per_simd, cyc / per_simd, insts * 64 / ms / 1e6);
}
int main() {
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    for (int w : {1, 2, 4}) {
        run<0>("v_mad_u64_u32", d, w); run<1>("v_mul_lo_u32", d, w); run<4>("v_mul_hi_u32", d, w); run<2>("v_add_u32", d, w);
        run<3>("v_lshl_add_u64", d, w); run<5>("v_mad_u32_u24", d, w); run<6>("addc+nop1 chain", d, w); run<7>("v_fma_f64", d, w);
    }
    return 0;
}
