// Tuning aid: time of a dependent chain of Fq multiplications per wavefront at 1, 2, 4, 8 wavefronts per SIMD, and of two
// independent chains in one wavefront (does a lone wavefront reach the VALU issue rate?).  hipcc --offload-arch=gfx950 -O3 -std=c++17
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../zklaim_amd/csrc/fp.hip.hpp"
using namespace zk;
__global__ __launch_bounds__(256) void k_chain1(Fq *io, int iters) {
    Fq x = io[threadIdx.x], y = io[threadIdx.x + 256];
    for (int i = 0; i < iters; ++i) x = x * y;
    io[threadIdx.x + blockIdx.x * 0] = x.normalized();
}
__global__ __launch_bounds__(256) void k_chain2(Fq *io, int iters) {
    Fq x = io[threadIdx.x], y = io[threadIdx.x + 256], z = io[threadIdx.x + 512];
    for (int i = 0; i < iters; ++i) { x = x * y; z = z * y; }
    io[threadIdx.x] = (x + z).normalized();
}
int main() {
    Fq *d; hipMalloc(&d, 1024 * sizeof(Fq)); hipMemset(d, 1, 1024 * sizeof(Fq));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    for (int wgs_per_cu : {1, 2, 4, 8}) {
        for (int two = 0; two < 2; ++two) {
            dim3 grid(256 * wgs_per_cu), block(256);
            if (two) hipLaunchKernelGGL(k_chain2, grid, block, 0, 0, d, 10); else hipLaunchKernelGGL(k_chain1, grid, block, 0, 0, d, 10);
            hipDeviceSynchronize();
            hipEventRecord(a);
            if (two) hipLaunchKernelGGL(k_chain2, grid, block, 0, 0, d, iters); else hipLaunchKernelGGL(k_chain1, grid, block, 0, 0, d, iters);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            int muls = iters * (two ? 2 : 1);
            printf("waves/SIMD %d  chains/wave %d : %.1f ns per multiplication per wave, %.1f ns per multiplication per SIMD\n", wgs_per_cu, two + 1,
                   ms * 1e6 / muls, ms * 1e6 / muls / wgs_per_cu);
        }
    }
    return 0;
}
