// membench5.hip -- dev tool: what do the tile kernel's 1-bit-per-char output stores cost next to its read stream, and
// does it matter when they are issued?  Same loads as the tile kernel (16 x 1 KiB nt per wave and tile, 12 waves/CU,
// each CU its own contiguous segment of tiles, waves round-robin inside it).
//   STORE 0: never (one guarded store keeps the loads alive)     1: 512 B per tile as it is finished
//         2: 8 tiles combined in registers (the shipped kernel)   3: all of a wave's words kept in LDS, stored at the end
// each also with non-temporal stores, plus a kernel that only writes the 16 MB.
// MI355X, C2 shape: no store 76-80 us; per tile 92; 8 combined 87-89; at the end 86-88; nt: +1; write only 4.4 us.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kWaves = 12, kMaxPerWave = 16;

template <bool NT>
__device__ __forceinline__ void st(unsigned long long* p, unsigned long long v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// write-only: the same 512 B per tile, nothing read
template <bool NT>
__global__ __launch_bounds__(kWaves * 64) void kw(int64_t n_tiles, int seg_tiles, unsigned long long* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t T0 = (int64_t)blockIdx.x * seg_tiles;
    const int64_t T1 = T0 + seg_tiles < n_tiles ? T0 + seg_tiles : n_tiles;
    for (int64_t t = T0 + wave; t < T1; t += kWaves) st<NT>(out + t * 64 + lane, (unsigned long long)t);
}

template <int STORE, bool NT = false>
__global__ __launch_bounds__(kWaves * 64) void k(const u32x4* __restrict__ src, int64_t n_tiles, int seg_tiles,
                                                 unsigned long long* __restrict__ out) {
    __shared__ unsigned long long keep[STORE == 3 ? kWaves * kMaxPerWave * 64 : 64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t T0 = (int64_t)blockIdx.x * seg_tiles;
    const int64_t T1 = T0 + seg_tiles < n_tiles ? T0 + seg_tiles : n_tiles;
    unsigned long long buf[8];
    int nb = 0, cnt = 0;
    int64_t tb = 0;
    for (int64_t t = T0 + wave; t < T1; t += kWaves) {
        u32x4 v[16];
        const u32x4* p = src + t * 1024 + lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(p + 64 * i);
        uint32_t a = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) a ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
        const unsigned long long x = a * 0x9E3779B97F4A7C15ull;
        if (STORE == 0) {
            if (x == 0x1234567ull) out[t * 64 + lane] = x;
        } else if (STORE == 1) {
            st<NT>(out + t * 64 + lane, x);
        } else if (STORE == 2) {
            if (nb == 0) tb = t;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j == nb) buf[j] = x;
            if (++nb == 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) st<NT>(out + (tb + j * kWaves) * 64 + lane, buf[j]);
                nb = 0;
            }
        } else {
            keep[(wave * kMaxPerWave + cnt) * 64 + lane] = x;
            if (++cnt == kMaxPerWave) {
                for (int j = 0; j < cnt; ++j) st<NT>(out + (t - (int64_t)(cnt - 1 - j) * kWaves) * 64 + lane, keep[(wave * kMaxPerWave + j) * 64 + lane]);
                cnt = 0;
            }
            tb = t;
        }
    }
    if (STORE == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (j < nb) st<NT>(out + (tb + j * kWaves) * 64 + lane, buf[j]);
    }
    if (STORE == 3)
        for (int j = 0; j < cnt; ++j) st<NT>(out + (tb - (int64_t)(cnt - 1 - j) * kWaves) * 64 + lane, keep[(wave * kMaxPerWave + j) * 64 + lane]);
}

template <int STORE, bool NT = false>
float run(const u32x4* d, size_t bytes, unsigned long long* out, int iters) {
    const int64_t n = bytes / 16384;
    const int seg = (int)((n + 255) / 256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<STORE, NT>), dim3(256), dim3(kWaves * 64), 0, 0, d, n, seg, out);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k<STORE, NT>), dim3(256), dim3(kWaves * 64), 0, 0, d, n, seg, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters * 1e3;
}

int main() {
    const size_t bytes = 500ull << 20;   // C2: 31250 tiles -> 123 per CU, 10-11 per wave
    u32x4* d; unsigned long long* out;
    hipMalloc(&d, bytes + 65536); hipMalloc(&out, bytes / 32 + 65536);
    hipMemset(d, 0x41, bytes);
    {
        const int64_t n = bytes / 16384; const int seg = (int)((n + 255) / 256);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float ms;
        for (int nt = 0; nt < 2; ++nt) {
            for (int i = 0; i < 23; ++i) {
                if (i == 3) hipEventRecord(a);
                if (nt) hipLaunchKernelGGL((kw<true>), dim3(256), dim3(kWaves * 64), 0, 0, n, seg, out);
                else hipLaunchKernelGGL((kw<false>), dim3(256), dim3(kWaves * 64), 0, 0, n, seg, out);
            }
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            printf("write only (16 MB)%s %6.1f us\n", nt ? " nt" : "   ", ms / 20 * 1e3);
        }
    }
    for (int rep = 0; rep < 2; ++rep) {
        printf("per tile, nt      %6.1f us\n", run<1, true>(d, bytes, out, 30));
        printf("8 combined, nt    %6.1f us\n", run<2, true>(d, bytes, out, 30));
        printf("at the end, nt    %6.1f us\n", run<3, true>(d, bytes, out, 30));

        printf("no store          %6.1f us\n", run<0>(d, bytes, out, 30));
        printf("store per tile    %6.1f us\n", run<1>(d, bytes, out, 30));
        printf("8 tiles combined  %6.1f us\n", run<2>(d, bytes, out, 30));
        printf("at the end (LDS)  %6.1f us\n", run<3>(d, bytes, out, 30));
    }
    return 0;
}
