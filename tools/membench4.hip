// membench4.hip -- dev tool: which on-chip work degrades the streaming read of the tile kernel's pattern?
// Same loads as the tile kernel (16 x 1 KiB nt per wave and tile, 12 waves per CU), plus optional synthetic work:
//   L = 64 table lookups in LDS per lane and tile (like the class lookup), V = n dependent-ish VALU ops per tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int LOOKUPS, int VALU, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const u32x4* __restrict__ src, int64_t n_tiles, unsigned long long* __restrict__ out) {
    __shared__ uint8_t tbl[32768 + 16 * 5120];
    for (int i = threadIdx.x; i < 32768; i += THREADS) tbl[i] = (uint8_t)(i * 7);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    uint8_t* stage = tbl + 32768 + wave_in_block * 5120;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    unsigned long long buf[8]; int nb = 0; int64_t tb[8];
    for (int64_t t = wave; t < n_tiles; t += n_waves) {
        u32x4 v[16];
        const u32x4* p = src + t * 1024 + lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(p + 64 * i);
        uint32_t a = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (LOOKUPS) {
                const uint32_t c = (uint32_t)tbl[v[i].x & 127] | ((uint32_t)tbl[v[i].y & 127] << 8) | ((uint32_t)tbl[v[i].z & 127] << 16) |
                                   ((uint32_t)tbl[v[i].w & 127] << 24);
                *reinterpret_cast<uint32_t*>(stage + 4 * lane + 16 * (lane >> 4) + 320 * i) = c;
            } else {
                a ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
            }
        }
        if (LOOKUPS) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                const uint4 q = *reinterpret_cast<const uint4*>(stage + 80 * lane + 16 * k2);
                a ^= q.x ^ q.y ^ q.z ^ q.w;
            }
        }
        unsigned long long x = a, y = a ^ 0x9E3779B97F4A7C15ull, z = ~x, u = x * 3;
#pragma unroll 8
        for (int i = 0; i < VALU / 8; ++i) {   // 8 64-bit-ish ops per iteration on four chains
            x = (x << 1) ^ (y >> 3); y = (y + z) | (u & x); z = (z >> 2) ^ u; u = (u << 5) + x;
        }
        x ^= y ^ z ^ u;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (j == nb) { buf[j] = x; tb[j] = t; }
        if (++nb == 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) out[tb[j] * 64 + lane] = buf[j];
            nb = 0;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j < nb) out[tb[j] * 64 + lane] = buf[j];
}

template <int LOOKUPS, int VALU, int THREADS>
float run(const u32x4* d, size_t bytes, unsigned long long* out, int iters) {
    int64_t n = bytes / 16384;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<LOOKUPS, VALU, THREADS>), dim3(256), dim3(THREADS), 0, 0, d, n, out);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k<LOOKUPS, VALU, THREADS>), dim3(256), dim3(THREADS), 0, 0, d, n, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters * 1e3;
}

int main() {
    const size_t bytes = 512ull << 20;
    u32x4* d; unsigned long long* out;
    hipMalloc(&d, bytes); hipMalloc(&out, bytes / 32 + 64);
    hipMemset(d, 0x41, bytes);
    printf("threads/CU:            512     768    1024\n");
    printf("read+store          %6.1f  %6.1f  %6.1f us\n", run<0, 0, 512>(d, bytes, out, 20), run<0, 0, 768>(d, bytes, out, 20), run<0, 0, 1024>(d, bytes, out, 20));
    printf("+lookups+400 VALU   %6.1f  %6.1f  %6.1f us\n", run<1, 400, 512>(d, bytes, out, 20), run<1, 400, 768>(d, bytes, out, 20), run<1, 400, 1024>(d, bytes, out, 20));
    printf("+lookups+600 VALU   %6.1f  %6.1f  %6.1f us\n", run<1, 600, 512>(d, bytes, out, 20), run<1, 600, 768>(d, bytes, out, 20), run<1, 600, 1024>(d, bytes, out, 20));
    printf("+lookups+800 VALU   %6.1f  %6.1f  %6.1f us\n", run<1, 800, 512>(d, bytes, out, 20), run<1, 800, 768>(d, bytes, out, 20), run<1, 800, 1024>(d, bytes, out, 20));
    printf("no loads, 800 VALU: see membench notes\n");
    return 0;
}
