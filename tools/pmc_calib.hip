// pmc_calib.hip — known-byte-count kernels for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950
// in the MSM's access pattern (64 B per lane gathered at random, via 4 x dwordx4), next to a plain 16 B/lane stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
struct alignas(16) P64 { uint4 a, b, c, d; };
__global__ void k_gather64(const P64 *src, const uint32_t *idx, size_t n, uint4 *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    P64 p = src[idx[i]];
    uint4 r = {p.a.x ^ p.b.x ^ p.c.x ^ p.d.x, p.a.y ^ p.b.y, p.c.z ^ p.d.z, p.a.w};
    if (r.x == 0x12345678u) out[0] = r;          // keep the loads alive, (almost) never store
}
__global__ void k_stream16(const uint4 *src, size_t n, uint4 *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 r = src[i];
    if (r.x == 0x12345678u && r.y == 0x9abcdef0u) out[0] = r;
}
__global__ void k_store16(uint4 *dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = uint4{(uint32_t)i, 1, 2, 3};
}
int main() {
    const size_t npts = (size_t)1 << 24;               // 1 GiB of 64-B records: far beyond the 256 MiB Infinity Cache
    const size_t ngather = (size_t)1 << 24;
    P64 *src; uint32_t *idx; uint4 *out;
    hipMalloc(&src, npts * sizeof(P64)); hipMalloc(&idx, ngather * 4); hipMalloc(&out, 1 << 20);
    hipMemset(src, 1, npts * sizeof(P64));
    std::vector<uint32_t> h(ngather);
    uint64_t s = 88172645463325252ull;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s % npts); }
    hipMemcpy(idx, h.data(), ngather * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_gather64, dim3((unsigned)(ngather / 256)), dim3(256), 0, 0, src, idx, ngather, out);
    hipLaunchKernelGGL(k_stream16, dim3((unsigned)(npts * 4 / 256)), dim3(256), 0, 0, (const uint4 *)src, npts * 4, out);
    hipLaunchKernelGGL(k_store16, dim3((unsigned)(npts * 4 / 256)), dim3(256), 0, 0, (uint4 *)src, npts * 4);
    hipDeviceSynchronize();
    printf("k_gather64: %zu gathers x 64 B = %zu bytes (+ %zu index bytes)\n", ngather, ngather * 64, ngather * 4);
    printf("k_stream16: %zu bytes read\nk_store16: %zu bytes written\n", npts * 64, npts * 64);
    return 0;
}
