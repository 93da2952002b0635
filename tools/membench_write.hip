// membench_write.hip -- dev tool: ceiling of a pure streaming WRITE (16 B per lane, consecutive lanes -> consecutive addresses)
//   hipcc --offload-arch=gfx950 -O3 tools/membench_write.hip -o /tmp/membench_write && /tmp/membench_write
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef long long ll2 __attribute__((ext_vector_type(2)));

template <bool NT>
__global__ void k_write(ll2* __restrict__ dst, int64_t n_vec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
        ll2 v;
        v.x = i;
        v.y = ~i;
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}

int main() {
    const size_t bytes = 1024ull << 20;
    ll2* d;
    hipMalloc(&d, bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    int cfgs[][2] = {{256, 1024}, {512, 512}, {1024, 256}, {2048, 256}, {4096, 256}, {8192, 256}};
    for (auto& c : cfgs) {
        for (int nt = 0; nt < 2; ++nt) {
            for (int i = 0; i < 3; ++i) {
                if (nt) hipLaunchKernelGGL((k_write<true>), dim3(c[0]), dim3(c[1]), 0, 0, d, (int64_t)(bytes / 16));
                else hipLaunchKernelGGL((k_write<false>), dim3(c[0]), dim3(c[1]), 0, 0, d, (int64_t)(bytes / 16));
            }
            hipEventRecord(a);
            for (int i = 0; i < 10; ++i) {
                if (nt) hipLaunchKernelGGL((k_write<true>), dim3(c[0]), dim3(c[1]), 0, 0, d, (int64_t)(bytes / 16));
                else hipLaunchKernelGGL((k_write<false>), dim3(c[0]), dim3(c[1]), 0, 0, d, (int64_t)(bytes / 16));
            }
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            printf("blocks %5d x %4d thr, %s stores: %.1f us per GiB = %.2f TB/s\n", c[0], c[1], nt ? "nt" : "plain", ms / 10 * 1e3,
                   bytes / (ms / 10 * 1e-3) / 1e12);
        }
    }
    return 0;
}
