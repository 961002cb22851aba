// mul_variants.hip — what does a field product cost on gfx950, instruction by instruction, in REAL cycles?
//
// Round-2's roofline table priced the kernel with issue rates measured at the NOMINAL clock (tools/microbench.hip).  This probe
// (1) stamps s_memtime / s_memrealtime inside every kernel, so every figure below is in cycles of the clock the chip actually held,
// (2) times each instruction FORM the shipped Montgomery stream is built from (v_mad_u64_u32 with its carry-out in an SGPR pair,
//     v_addc_co_u32_e64 with its carry-in from an SGPR pair, ...) next to the plain forms,
// (3) times the shipped 8 x 32-bit product (csrc/mont_asm.inc) and a 9 x 29-bit unsaturated product (no carry folds at all: 18 partial
//     products of 58 bits fit a 64-bit column) at 1, 2, 4 and 8 wavefronts per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mul_variants.hip -o gpurun_out/mul_variants ; run on the GPU box.
// Output is copied to profiles/r3_mul_variants.txt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include "../zklaim_amd/csrc/fp.hip.hpp"
using namespace zk;

#define ITERS 1000
#define REP 8            // asm blocks per iteration, 8 instructions each

struct Stamp { uint64_t t0, t1, r0, r1; };
__device__ __forceinline__ void stamp_begin(Stamp &s) { s.t0 = __builtin_amdgcn_s_memtime(); s.r0 = __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ void stamp_end(Stamp &s) { s.t1 = __builtin_amdgcn_s_memtime(); s.r1 = __builtin_amdgcn_s_memrealtime(); }

template <int KIND> __global__ __launch_bounds__(256) void k_instr(uint32_t *out, Stamp *stamps, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c0 = 1, c1 = 2, c2 = 3, c3 = 4, c4 = 5, c5 = 6, c6 = 7, c7 = 8;
    uint64_t d0 = a, d1 = b, d2 = a + 1, d3 = b + 1, d4 = 5, d5 = 6, d6 = 7, d7 = 8;
    uint64_t m0 = 0x5555555555555555ull ^ seed, m1 = 0x3333333333333333ull ^ seed, m2 = 0x0f0f0f0f0f0f0f0full ^ seed;   // SGPR pairs (lane masks)
    Stamp st; stamp_begin(st);
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (KIND == 0) {          // v_mad_u64_u32, carry-out to vcc
                asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                             "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b) : "vcc");
            } else if (KIND == 1) {   // v_mad_u64_u32, carry-out to a rotating SGPR pair (the stream's form)
                asm volatile("v_mad_u64_u32 %0, %10, %8, %9, %0\n v_mad_u64_u32 %1, %11, %8, %9, %1\n v_mad_u64_u32 %2, %12, %8, %9, %2\n v_mad_u64_u32 %3, %10, %8, %9, %3\n"
                             "v_mad_u64_u32 %4, %11, %8, %9, %4\n v_mad_u64_u32 %5, %12, %8, %9, %5\n v_mad_u64_u32 %6, %10, %8, %9, %6\n v_mad_u64_u32 %7, %11, %8, %9, %7\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b), "s"(m0), "s"(m1), "s"(m2));
            } else if (KIND == 2) {   // v_addc_co_u32_e64 with carry-in from an SGPR pair (the stream's carry fold)
                asm volatile("v_addc_co_u32_e64 %0, vcc, 0, %0, %8\n v_addc_co_u32_e64 %1, vcc, 0, %1, %9\n v_addc_co_u32_e64 %2, vcc, 0, %2, %10\n v_addc_co_u32_e64 %3, vcc, 0, %3, %8\n"
                             "v_addc_co_u32_e64 %4, vcc, 0, %4, %9\n v_addc_co_u32_e64 %5, vcc, 0, %5, %10\n v_addc_co_u32_e64 %6, vcc, 0, %6, %8\n v_addc_co_u32_e64 %7, vcc, 0, %7, %9\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "s"(m0), "s"(m1), "s"(m2) : "vcc");
            } else if (KIND == 3) {   // v_add_u32 (full-rate reference)
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a));
            } else if (KIND == 4) {   // the stream's pairing: 4 x (mad with SGPR carry-out, fold of a carry written >= 2 instructions earlier)
                asm volatile("v_mad_u64_u32 %0, %14, %12, %13, %0\n v_addc_co_u32_e64 %4, vcc, 0, %4, %15\n v_mad_u64_u32 %1, %16, %12, %13, %1\n v_addc_co_u32_e64 %5, vcc, 0, %5, %14\n"
                             "v_mad_u64_u32 %2, %15, %12, %13, %2\n v_addc_co_u32_e64 %6, vcc, 0, %6, %16\n v_mad_u64_u32 %3, %14, %12, %13, %3\n v_addc_co_u32_e64 %7, vcc, 0, %7, %15\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
                             : "v"(a), "v"(b), "s"(m0), "s"(m1), "s"(m2) : "vcc");
            } else if (KIND == 5) {   // v_addc_co_u32 VOP2 form through vcc (carry-in vcc, independent destinations; vcc written two slots earlier by the partner)
                asm volatile("v_addc_co_u32_e32 %0, vcc, %8, %0, vcc\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_addc_co_u32_e32 %1, vcc, %8, %1, vcc\n"
                             "v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n v_addc_co_u32_e32 %2, vcc, %8, %2, vcc\n v_add_u32 %3, %3, %8\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a) : "vcc");
            } else if (KIND == 6) {   // v_lshrrev_b64 (column carry extraction of an unsaturated product)
                asm volatile("v_lshrrev_b64 %0, 29, %0\n v_lshrrev_b64 %1, 29, %1\n v_lshrrev_b64 %2, 29, %2\n v_lshrrev_b64 %3, 29, %3\n"
                             "v_lshrrev_b64 %4, 29, %4\n v_lshrrev_b64 %5, 29, %5\n v_lshrrev_b64 %6, 29, %6\n v_lshrrev_b64 %7, 29, %7\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
            } else if (KIND == 7) {   // v_alignbit_b32 (the 32-bit way to shift a column)
                asm volatile("v_alignbit_b32 %0, %1, %0, 29\n v_alignbit_b32 %1, %2, %1, 29\n v_alignbit_b32 %2, %3, %2, 29\n v_alignbit_b32 %3, %4, %3, 29\n"
                             "v_alignbit_b32 %4, %5, %4, 29\n v_alignbit_b32 %5, %6, %5, 29\n v_alignbit_b32 %6, %7, %6, 29\n v_alignbit_b32 %7, %0, %7, 29\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));
            } else if (KIND == 8) {   // v_mul_lo_u32
                asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                             "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a));
            } else if (KIND == 9) {   // v_mad_u64_u32 with an SGPR multiplicand (the m x p half reads the modulus limbs from SGPRs)
                asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                             "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "s"(seed) : "vcc");
            } else if (KIND == 10) {  // v_cndmask_b32_e64 with an SGPR-pair mask
                asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %10\n v_cndmask_b32_e64 %2, %2, %8, %11\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
                             "v_cndmask_b32_e64 %4, %4, %8, %10\n v_cndmask_b32_e64 %5, %5, %8, %11\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %10\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "s"(m0), "s"(m1), "s"(m2));
            } else if (KIND == 11) {  // v_add_co_u32_e64 with carry-out to an SGPR pair
                asm volatile("v_add_co_u32_e64 %0, %9, %0, %8\n v_add_co_u32_e64 %1, %10, %1, %8\n v_add_co_u32_e64 %2, %11, %2, %8\n v_add_co_u32_e64 %3, %9, %3, %8\n"
                             "v_add_co_u32_e64 %4, %10, %4, %8\n v_add_co_u32_e64 %5, %11, %5, %8\n v_add_co_u32_e64 %6, %9, %6, %8\n v_add_co_u32_e64 %7, %10, %7, %8\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "s"(m0), "s"(m1), "s"(m2));
            } else if (KIND == 12) {  // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %0, %1\n v_fma_f64 %1, %1, %1, %2\n v_fma_f64 %2, %2, %2, %3\n v_fma_f64 %3, %3, %3, %4\n"
                             "v_fma_f64 %4, %4, %4, %5\n v_fma_f64 %5, %5, %5, %6\n v_fma_f64 %6, %6, %6, %7\n v_fma_f64 %7, %7, %7, %0\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
            } else if (KIND == 13) {  // v_lshl_add_u64 (the 64-bit add an f64-split product needs per partial product)
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %4\n"
                             "v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %5, %5, 0, %6\n v_lshl_add_u64 %6, %6, 0, %7\n v_lshl_add_u64 %7, %7, 0, %0\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
            }
        }
    }
    stamp_end(st);
    if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = st;
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (uint32_t)(m0 + m1 + m2);
}

// ---- products ------------------------------------------------------------------------------------------------------------------
// 9 x 29-bit limbs, R' = 2^261, product scanning with ONE 64-bit column accumulator: no carry folds (18 x 2^58 < 2^63).
struct F29 { uint32_t v[9]; };
static constexpr uint32_t MASK29 = (1u << 29) - 1;
// alt_bn128 q in 29-bit limbs and -q^-1 mod 2^29, filled in by the host
__constant__ uint32_t P29[9];
__constant__ uint32_t INV29;
__device__ __forceinline__ F29 mul29(const F29 &a, const F29 &b, const uint32_t (&p)[9], uint32_t inv) {
    uint64_t acc = 0; uint32_t m[9]; F29 t;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * p[k - i];
        m[k] = ((uint32_t)acc * inv) & MASK29;
        acc += (uint64_t)m[k] * p[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i) { acc += (uint64_t)a.v[i] * b.v[k - i]; acc += (uint64_t)m[i] * p[k - i]; }
        t.v[k - 9] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    t.v[8] = (uint32_t)acc;
    return t;
}
__global__ __launch_bounds__(256) void k_chain29(F29 *io, int iters, Stamp *stamps) {
    uint32_t p[9]; for (int i = 0; i < 9; ++i) p[i] = P29[i];
    const uint32_t inv = INV29;
    F29 x = io[threadIdx.x], y = io[threadIdx.x + 256];
    Stamp st; stamp_begin(st);
    for (int i = 0; i < iters; ++i) x = mul29(x, y, p, inv);
    stamp_end(st);
    if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = st;
    if (blockIdx.x == 0) io[threadIdx.x + 512] = x;
}
__global__ __launch_bounds__(256) void k_chain32(Fq *io, int iters, Stamp *stamps) {
    Fq x = io[threadIdx.x], y = io[threadIdx.x + 256];
    Stamp st; stamp_begin(st);
    for (int i = 0; i < iters; ++i) x = x * y;
    stamp_end(st);
    if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = st;
    if (blockIdx.x == 0) io[threadIdx.x + 512] = x.normalized();
}

static double clock_ghz(const std::vector<Stamp> &st) {
    std::vector<double> r;
    for (const Stamp &s : st) if (s.r1 > s.r0) r.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 0.1);     // s_memrealtime ticks at 100 MHz
    if (r.empty()) return 0;
    std::sort(r.begin(), r.end());
    return r[r.size() / 2];
}

template <int KIND> void run_instr(const char *name, uint32_t *d_out, Stamp *d_st, int wps, int cus) {
    dim3 grid(cus * wps), block(256);
    const size_t waves = (size_t)grid.x * 4;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_instr<KIND>, grid, block, 0, 0, d_out, d_st, 1u);
    hipEventRecord(a); hipLaunchKernelGGL(k_instr<KIND>, grid, block, 0, 0, d_out, d_st, 2u); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<Stamp> st(waves); hipMemcpy(st.data(), d_st, waves * sizeof(Stamp), hipMemcpyDeviceToHost);
    const double ghz = clock_ghz(st), per_simd = (double)waves / (cus * 4.0) * ITERS * REP * 8;
    printf("%-44s waves/SIMD=%d  %.3f ms  clock %.3f GHz  cycles per wave-instruction per SIMD = %.2f\n", name, wps, ms, ghz, ms * 1e-3 * ghz * 1e9 / per_simd);
    hipEventDestroy(a); hipEventDestroy(b);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs, nominal %d MHz; cycles are at the clock each kernel measured for itself (s_memtime / s_memrealtime)\n", prop.gcnArchName, cus, prop.clockRate / 1000);
    uint32_t *d_out; Stamp *d_st;
    hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4); hipMalloc(&d_st, (size_t)cus * 8 * 4 * sizeof(Stamp));
    for (int w : {4, 8}) {
        run_instr<3>("v_add_u32", d_out, d_st, w, cus);
        run_instr<0>("v_mad_u64_u32 (carry -> vcc)", d_out, d_st, w, cus);
        run_instr<1>("v_mad_u64_u32 (carry -> SGPR pair)", d_out, d_st, w, cus);
        run_instr<9>("v_mad_u64_u32 (SGPR multiplicand)", d_out, d_st, w, cus);
        run_instr<2>("v_addc_co_u32_e64 (carry <- SGPR pair)", d_out, d_st, w, cus);
        run_instr<4>("mad(SGPR carry) + addc(SGPR carry) pairs", d_out, d_st, w, cus);
        run_instr<5>("v_addc_co_u32_e32 via vcc + 2 v_add_u32", d_out, d_st, w, cus);
        run_instr<11>("v_add_co_u32_e64 (carry -> SGPR pair)", d_out, d_st, w, cus);
        run_instr<10>("v_cndmask_b32_e64 (SGPR mask)", d_out, d_st, w, cus);
        run_instr<6>("v_lshrrev_b64", d_out, d_st, w, cus);
        run_instr<7>("v_alignbit_b32", d_out, d_st, w, cus);
        run_instr<8>("v_mul_lo_u32", d_out, d_st, w, cus);
        run_instr<12>("v_fma_f64", d_out, d_st, w, cus);
        run_instr<13>("v_lshl_add_u64", d_out, d_st, w, cus);
    }
    // ---- products: dependent chains, 1 .. 8 wavefronts per SIMD
    // q in 29-bit limbs
    const uint32_t q32[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    uint32_t p29[9];
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, limb = bit >> 5, sh = bit & 31;
        uint64_t x = limb < 8 ? q32[limb] : 0; if (limb + 1 < 8) x |= (uint64_t)q32[limb + 1] << 32;
        p29[i] = (uint32_t)(x >> sh) & MASK29;
    }
    uint32_t inv = 1; for (int i = 0; i < 6; ++i) inv *= 2 - p29[0] * inv;      // q^-1 mod 2^32 (Newton)
    const uint32_t inv29 = (0u - inv) & MASK29;
    hipMemcpyToSymbol(HIP_SYMBOL(P29), p29, sizeof(p29)); hipMemcpyToSymbol(HIP_SYMBOL(INV29), &inv29, 4);
    F29 *d29; Fq *d32;
    hipMalloc(&d29, 1024 * sizeof(F29)); hipMalloc(&d32, 1024 * sizeof(Fq));
    std::vector<F29> h29(1024); std::vector<Fq> h32(1024);
    uint64_t sm = 0x5A4B4C41494D0000ull;
    auto next = [&] { sm += 0x9E3779B97F4A7C15ull; uint64_t z = sm; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    for (auto &f : h29) { for (int i = 0; i < 9; ++i) f.v[i] = (uint32_t)next() & MASK29; f.v[8] &= (1u << 21) - 1; }      // < 2^253 < q
    for (auto &f : h32) { for (int i = 0; i < 8; ++i) f.v[i] = (uint32_t)next(); f.v[7] &= 0x1fffffffu; }
    hipMemcpy(d29, h29.data(), 1024 * sizeof(F29), hipMemcpyHostToDevice); hipMemcpy(d32, h32.data(), 1024 * sizeof(Fq), hipMemcpyHostToDevice);
    const int iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int kind = 0; kind < 2; ++kind)
        for (int wps : {1, 2, 4, 8}) {
            dim3 grid(cus * wps), block(256);
            const size_t waves = (size_t)grid.x * 4;
            for (int pass = 0; pass < 2; ++pass) {
                if (pass) hipEventRecord(a);
                if (kind == 0) hipLaunchKernelGGL(k_chain32, grid, block, 0, 0, d32, pass ? iters : 10, d_st);
                else hipLaunchKernelGGL(k_chain29, grid, block, 0, 0, d29, pass ? iters : 10, d_st);
                if (pass) { hipEventRecord(b); hipEventSynchronize(b); } else hipDeviceSynchronize();
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            std::vector<Stamp> st(waves); hipMemcpy(st.data(), d_st, waves * sizeof(Stamp), hipMemcpyDeviceToHost);
            const double ghz = clock_ghz(st);
            printf("%-28s waves/SIMD %d : %7.1f ns per product per SIMD, clock %.3f GHz, %7.1f cycles per product per SIMD\n",
                   kind == 0 ? "8 x 32-bit (shipped stream)" : "9 x 29-bit (no carry folds)", wps, ms * 1e6 / iters / wps, ghz, ms * 1e-3 * ghz * 1e9 / iters / wps);
        }
    // correctness vectors of the 29-bit product for the Python check in the runner (one iteration: x * y * 2^-261 mod q)
    hipMemcpy(d29, h29.data(), 1024 * sizeof(F29), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_chain29, dim3(1), dim3(256), 0, 0, d29, 1, d_st);
    std::vector<F29> o29(1024); hipMemcpy(o29.data(), d29, 1024 * sizeof(F29), hipMemcpyDeviceToHost);
    auto hex = [](const F29 &f) { unsigned __int128 lo = 0, hi = 0; (void)lo; (void)hi; static char buf[128]; char *w = buf;
        // big integer from 29-bit limbs, printed as hex (limb 8 may carry a few bits above 2^29: lazy top limb)
        uint32_t words[10] = {0};
        for (int i = 0; i < 9; ++i) { const int bit = 29 * i, l = bit >> 5, s = bit & 31; uint64_t x = (uint64_t)f.v[i] << s; uint64_t c = (uint64_t)words[l] + (uint32_t)x; words[l] = (uint32_t)c;
            uint64_t c2 = (uint64_t)words[l + 1] + (x >> 32) + (c >> 32); words[l + 1] = (uint32_t)c2; if (l + 2 < 10) words[l + 2] += (uint32_t)(c2 >> 32); }
        for (int i = 9; i >= 0; --i) w += sprintf(w, "%08x", words[i]);
        return buf; };
    for (int t = 0; t < 4; ++t) {
        printf("VEC29 a=%s", hex(h29[t])); printf(" b=%s", hex(h29[t + 256])); printf(" r=%s\n", hex(o29[t + 512]));
    }
    return 0;
}
