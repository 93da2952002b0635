// membench2.hip -- dev tool: what does the tile kernel's scaffolding cost on top of a pure streaming read?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int LDSB, bool PROLOGUE, int STORE, bool EARLY>
__global__ __launch_bounds__(1024) void k(const u32x4* __restrict__ src, int64_t n_tiles, const uint4* __restrict__ tbl,
                                          unsigned long long* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[LDSB];
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    u32x4 v[16];
    int64_t t = wave;
    if (EARLY && t < n_tiles) {
        const u32x4* p = src + t * 1024 + lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(p + 64 * i);
    }
    if (PROLOGUE) {
        uint4* d = reinterpret_cast<uint4*>(lds);
        for (int i = threadIdx.x; i < 41360 / 16; i += 1024) d[i] = tbl[i];
        __syncthreads();
    }
    uint32_t acc = PROLOGUE ? lds[threadIdx.x] : 0;
    bool first = EARLY;
    for (; t < n_tiles; t += n_waves) {
        if (!first) {
            const u32x4* p = src + t * 1024 + lane;
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(p + 64 * i);
        }
        first = false;
        uint32_t a = acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) a ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
        if (STORE == 1) out[t * 64 + lane] = a;
        else if (STORE == 2) __builtin_nontemporal_store((unsigned long long)a, out + t * 64 + lane);
        else if (STORE == 3) { // 16 B per lane from 32 lanes
            unsigned int hi = __shfl_down(a, 1);
            if ((lane & 1) == 0) { uint4 w; w.x = a; w.y = 0; w.z = hi; w.w = 0; *reinterpret_cast<uint4*>(out + t * 64 + lane) = w; }
        }
        else if (STORE == 4) { if ((t & 7) == 7) out[t * 64 + lane] = a; }   // 1/8 of the stores
        else acc = a;
    }
    if (!STORE && acc == 0x12345678) out[0] = acc;
}

template <int LDSB, bool PROLOGUE, int STORE, bool EARLY>
float run(const u32x4* d, size_t bytes, const uint4* tbl, unsigned long long* out, int iters) {
    int64_t n = bytes / 16384;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<LDSB, PROLOGUE, STORE, EARLY>), dim3(256), dim3(1024), 0, 0, d, n, tbl, out);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k<LDSB, PROLOGUE, STORE, EARLY>), dim3(256), dim3(1024), 0, 0, d, n, tbl, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters * 1e3;
}

int main() {
    const size_t bytes = 512ull << 20;
    u32x4* d; unsigned long long* out; uint4* tbl;
    hipMalloc(&d, bytes); hipMalloc(&out, bytes / 32 + 64); hipMalloc(&tbl, 65536);
    hipMemset(d, 1, bytes); hipMemset(tbl, 2, 65536);
    printf("plain read                 : %.1f us\n", run<1024, false, 0, false>(d, bytes, tbl, out, 20));
    printf("store 8B/lane              : %.1f us\n", run<1024, false, 1, false>(d, bytes, tbl, out, 20));
    printf("store 8B/lane nontemporal  : %.1f us\n", run<1024, false, 2, false>(d, bytes, tbl, out, 20));
    printf("store 16B/lane x32 lanes   : %.1f us\n", run<1024, false, 3, false>(d, bytes, tbl, out, 20));
    printf("1/8 of the stores          : %.1f us\n", run<1024, false, 4, false>(d, bytes, tbl, out, 20));
    printf("plain read again           : %.1f us\n", run<1024, false, 0, false>(d, bytes, tbl, out, 20));
    return 0;
}
