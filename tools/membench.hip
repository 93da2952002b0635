// membench.hip -- dev tool: ceiling of a pure streaming READ on this GPU with the tile kernel's access pattern.
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o /tmp/membench && /tmp/membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int LOADS, bool NT>
__global__ void k_read(const u32x4* __restrict__ src, int64_t n_tiles_of_loads, uint32_t* __restrict__ out) {
    // one wave handles LOADS x 1 KiB contiguous per step, grid-stride over steps
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (int64_t t = wave; t < n_tiles_of_loads; t += n_waves) {
        const u32x4* p = src + t * (64 * LOADS) + lane;
        u32x4 v[LOADS];
#pragma unroll
        for (int i = 0; i < LOADS; ++i) v[i] = NT ? __builtin_nontemporal_load(p + 64 * i) : p[64 * i];
#pragma unroll
        for (int i = 0; i < LOADS; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    }
    if (acc == 0x12345678) out[0] = acc;
}

template <int LOADS, bool NT>
float run(const u32x4* d, size_t bytes, uint32_t* out, int blocks, int threads, int iters) {
    int64_t n = bytes / (1024 * LOADS);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_read<LOADS, NT>), dim3(blocks), dim3(threads), 0, 0, d, n, out);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_read<LOADS, NT>), dim3(blocks), dim3(threads), 0, 0, d, n, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    const size_t bytes = 512ull << 20;
    u32x4* d; uint32_t* out;
    hipMalloc(&d, bytes); hipMalloc(&out, 4);
    hipMemset(d, 1, bytes);
    struct Cfg { int blocks, threads; } cfgs[] = {{256, 1024}, {512, 512}, {1024, 256}, {2048, 256}, {256, 512}, {512, 1024}, {4096, 256}, {8192, 256}};
    for (auto c : cfgs) {
        float t16 = run<16, true>(d, bytes, out, c.blocks, c.threads, 20);
        float t16p = run<16, false>(d, bytes, out, c.blocks, c.threads, 20);
        float t8 = run<8, true>(d, bytes, out, c.blocks, c.threads, 20);
        float t4 = run<4, true>(d, bytes, out, c.blocks, c.threads, 20);
        float t32 = run<32, true>(d, bytes, out, c.blocks, c.threads, 20);
        printf("blocks %5d x %4d thr: 16nt %.1f us (%.2f TB/s) | 16plain %.1f | 8nt %.1f | 4nt %.1f | 32nt %.1f\n", c.blocks, c.threads,
               t16 * 1e3, bytes / t16 / 1e9, t16p * 1e3, t8 * 1e3, t4 * 1e3, t32 * 1e3);
    }
    return 0;
}
