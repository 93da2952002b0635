// aux_kernels.hip -- everything on the device that is NOT the fused hot path:
//   * the reference's native functions one string at a time, as plain element-parallel kernels (compat surface):
//       _gen_parse_matrix      reference latok/core/src/latok/latok.c:31-138
//       _combine_matrix_rows   reference latok/core/src/latok/latok.c:275-370
//   * device-wide exclusive scan (used by the compaction passes and the UTF-8 decoder)
//   * the staged UTF-8 decoder (code-point units; the byte-space ingest lives in split_kernels.hip)
//   * synthetic corpus fill, UTF-8 size reduction and the streaming-read ceiling kernel for the benchmark
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "corpus_gen.h"
#include "kernels.h"
#include "utf8_decode.h"

namespace latok {

// ---- _gen_parse_matrix ------------------------------------------------------------------------------------------
// One thread per character; the 12 base features come from the class-word table (bit i = column i), the 13 context
// columns are the neighbours' base bits with the reference's edge conventions (latok.c:69-73,114-134).
__device__ __forceinline__ uint32_t base_word(const uint8_t* t1, const uint8_t* t2cls, const uint16_t* cw, uint32_t cp) {
    const uint32_t hi = min(cp >> kTblShift, (uint32_t)(kStage1Len - 1));
    return cw[t2cls[((uint32_t)t1[hi] << kTblShift) | (cp & ((1u << kTblShift) - 1u))]];
}

// 25 feature bits of one char from its base word `w` and the base words of its neighbours inside the string
// (p = previous, x = next, y = after next; 0 where they do not exist).  Bit c = reference column c (offsets.py:24-49).
__device__ __forceinline__ uint32_t feature_row_bits(uint32_t w, uint32_t p, uint32_t x, uint32_t y, bool first, bool last) {
    uint32_t r = w & 0xFFFu;
    r |= ((p >> 0) & 1u) << 12;                        // PREV_ALPHA
    r |= ((x >> 0) & 1u) << 13;                        // NEXT_ALPHA
    r |= ((p >> 1) & 1u) << 14;                        // PREV_ALPHA_NUM
    r |= ((x >> 1) & 1u) << 15;                        // NEXT_ALPHA_NUM
    r |= ((p >> 3) & 1u) << 16;                        // PREV_LOWER
    r |= ((x >> 3) & 1u) << 17;                        // NEXT_LOWER
    r |= (first ? 1u : (p >> 5) & 1u) << 18;           // PREV_SPACE (string start counts as space)
    r |= (last ? 1u : (x >> 5) & 1u) << 19;            // NEXT_SPACE (string end counts as space)
    r |= ((p >> 6) & 1u) << 20;                        // PREV_SYMBOL
    r |= ((x >> 8) & 1u) << 21;                        // NEXT_AT
    r |= ((x >> 10) & 1u) << 22;                       // NEXT_SLASH
    r |= ((y >> 0) & 1u) << 23;                        // AFTER_NEXT_ALPHA
    r |= ((y >> 10) & 1u) << 24;                       // AFTER_NEXT_SLASH
    return r;
}

__global__ void k_parse_matrix(const uint32_t* __restrict__ cps, int64_t n, const uint8_t* __restrict__ t1,
                               const uint8_t* __restrict__ t2cls, const uint16_t* __restrict__ cw,
                               int8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = base_word(t1, t2cls, cw, cps[i]);
    const uint32_t p = i > 0 ? base_word(t1, t2cls, cw, cps[i - 1]) : 0u;
    const uint32_t x = i + 1 < n ? base_word(t1, t2cls, cw, cps[i + 1]) : 0u;
    const uint32_t y = i + 2 < n ? base_word(t1, t2cls, cw, cps[i + 2]) : 0u;
    const uint32_t bits = feature_row_bits(w, p, x, y, i == 0, i + 1 == n);
    int8_t* row = out + i * 25;
#pragma unroll
    for (int c = 0; c < 25; ++c) row[c] = (int8_t)((bits >> c) & 1u);
}

hipError_t launch_parse_matrix(const uint32_t* cps, int64_t n, const uint8_t* t1, const uint8_t* t2cls,
                               const uint16_t* cw, int8_t* out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int threads = 256;
    hipLaunchKernelGGL(k_parse_matrix, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), 0, st, cps, n, t1,
                       t2cls, cw, out);
    return hipGetLastError();
}

// One string of at most kSmallMatrixChars chars whose chars and matrix live in pinned host memory (the reference's calling
// pattern: _gen_parse_matrix(text) per string, latok.c:46-146): ONE workgroup; every char's base word is looked up once
// (all loads in flight together), the 25-byte rows meet in LDS and leave as dwords -- byte stores over the bus are slow --
// and the last store is the completion word the host polls (api.cpp: wait_completion_word).
__global__ __launch_bounds__(256) void k_parse_matrix_small(const uint32_t* __restrict__ cps, int n, const uint8_t* __restrict__ t1,
                                                           const uint8_t* __restrict__ t2cls, const uint16_t* __restrict__ cw,
                                                           int8_t* __restrict__ out, unsigned long long* done,
                                                           unsigned long long seq) {
    __shared__ uint32_t s_w[kSmallMatrixChars];
    __shared__ __attribute__((aligned(16))) uint8_t s_rows[256 * 25];
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += 256) s_w[i] = base_word(t1, t2cls, cw, cps[i]);
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        if (i < n) {
            const uint32_t bits = feature_row_bits(s_w[i], i > 0 ? s_w[i - 1] : 0u, i + 1 < n ? s_w[i + 1] : 0u,
                                                   i + 2 < n ? s_w[i + 2] : 0u, i == 0, i + 1 == n);
#pragma unroll
            for (int c = 0; c < 25; ++c) s_rows[tid * 25 + c] = (uint8_t)((bits >> c) & 1u);
        }
        __syncthreads();
        const int n_bytes = min(256, n - i0) * 25;
        int8_t* dst = out + (size_t)i0 * 25;                 // (i0 * 25 is a multiple of 6400: dword aligned when out is)
        for (int k = tid; k < (n_bytes >> 2); k += 256)
            reinterpret_cast<uint32_t*>(dst)[k] = reinterpret_cast<const uint32_t*>(s_rows)[k];
        for (int k = (n_bytes & ~3) + tid; k < n_bytes; k += 256) dst[k] = (int8_t)s_rows[k];
        __syncthreads();
    }
    if (done) {
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_parse_matrix_small(const uint32_t* cps, int n, const uint8_t* t1, const uint8_t* t2cls, const uint16_t* cw,
                                     int8_t* out, unsigned long long* done, unsigned long long seq, hipStream_t st) {
    if (n <= 0 || n > kSmallMatrixChars) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_parse_matrix_small, dim3(1), dim3(256), 0, st, cps, n, t1, t2cls, cw, out, done, seq);
    return hipGetLastError();
}

// ---- _combine_matrix_rows -----------------------------------------------------------------------------------------
// One thread per output element k.  uint8 wrap-around; 2-D idx: sum over idx rows of the product over idx columns,
// -1 skipped; the running product is only re-initialised by column 0 (latok.c:328-333), so an idx row that starts
// with -1 keeps multiplying the previous row's product -- kept.  1-D idx: plain sum (latok.c:342-354).
__global__ void k_combine_rows(const uint8_t* __restrict__ m, int64_t stride_r, int64_t stride_c, int64_t cols,
                               const int8_t* __restrict__ idx, int idx_ndim, int irows, int icols,
                               int8_t* __restrict__ out, unsigned long long* done, unsigned long long seq) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < cols; k += (int64_t)gridDim.x * blockDim.x) {
        uint8_t acc = 0, prod = 0;
        if (idx_ndim == 2) {
            for (int i = 0; i < irows; ++i) {
                for (int j = 0; j < icols; ++j) {
                    const uint8_t r = (uint8_t)idx[i * icols + j];
                    if (r == 255) continue;
                    const uint8_t v = m[(int64_t)r * stride_r + k * stride_c];
                    prod = j == 0 ? v : (uint8_t)(prod * v);
                }
                acc = (uint8_t)(acc + prod);
            }
        } else {
            for (int j = 0; j < icols; ++j) {
                const uint8_t r = (uint8_t)idx[j];
                if (r == 255) continue;
                acc = (uint8_t)(acc + m[(int64_t)r * stride_r + k * stride_c]);
            }
        }
        out[k] = (int8_t)acc;
    }
    if (done) {   // one-workgroup launches only (small host arrays in pinned memory): the completion word the host polls
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_combine_rows(const uint8_t* m, int64_t stride_r, int64_t stride_c, int64_t cols, const int8_t* idx,
                               int idx_ndim, int irows, int icols, int8_t* out, hipStream_t st, unsigned long long* done,
                               unsigned long long seq) {
    if (cols <= 0) return hipSuccess;
    const int threads = 256;
    const unsigned blocks = done ? 1u : (unsigned)((cols + threads - 1) / threads);
    hipLaunchKernelGGL(k_combine_rows, dim3(blocks), dim3(threads), 0, st, m, stride_r, stride_c, cols, idx, idx_ndim, irows,
                       icols, out, done, seq);
    return hipGetLastError();
}

// ---- device-wide exclusive scan of int64 counts: per-block scan + scan of the block totals + fix-up -----------------
constexpr int kScanBlock = 1024;
constexpr int kScanItems = 4;                       // elements per thread
constexpr int kScanChunk = kScanBlock * kScanItems; // elements per block

__device__ __forceinline__ long long block_exclusive_scan_ll(long long v, long long* total, long long* lds /*[16]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    __syncthreads();
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    long long before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kScanBlock / 64; ++w) {
        const long long x = lds[w];
        if (w < wave) before += x;
        all += x;
    }
    *total = all;
    return before + inc - v;
}

// out[i] = exclusive prefix inside the block's chunk; block_tot[b] = sum of the chunk
__global__ __launch_bounds__(kScanBlock) void k_scan_local(const int64_t* __restrict__ in, int64_t n,
                                                           int64_t* __restrict__ out, int64_t* __restrict__ block_tot) {
    __shared__ long long lds[kScanBlock / 64];
    const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * kScanItems;
    long long v[kScanItems], sum = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        v[j] = base + j < n ? in[base + j] : 0;
        sum += v[j];
    }
    long long tot;
    long long run = block_exclusive_scan_ll(sum, &tot, lds);
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        if (base + j < n) out[base + j] = run;
        run += v[j];
    }
    if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
}

// exclusive scan of the block totals by one block (in place), grand total -> *total
__global__ __launch_bounds__(kScanBlock) void k_scan_totals(int64_t* __restrict__ block_tot, int64_t n_blocks,
                                                            int64_t* __restrict__ total, int64_t* __restrict__ total_host) {
    __shared__ long long lds[kScanBlock / 64];
    const int64_t per = (n_blocks + kScanBlock - 1) / kScanBlock;
    const int64_t lo = min((int64_t)threadIdx.x * per, n_blocks), hi = min(lo + per, n_blocks);
    long long sum = 0;
    for (int64_t i = lo; i < hi; ++i) sum += block_tot[i];
    long long tot;
    long long run = block_exclusive_scan_ll(sum, &tot, lds);
    for (int64_t i = lo; i < hi; ++i) {
        const long long v = block_tot[i];
        block_tot[i] = run;
        run += v;
    }
    if (threadIdx.x == 0) {
        *total = tot;
        if (total_host) *total_host = tot;   // pinned, device-mapped: the host reads it after the stream drains, no copy
    }
}

__global__ __launch_bounds__(kScanBlock) void k_scan_add(int64_t* __restrict__ out, int64_t n,
                                                         const int64_t* __restrict__ block_tot) {
    const long long add = block_tot[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j)
        if (base + j < n) out[base + j] += add;
}

// the whole scan in one workgroup (each thread owns a contiguous run): for the tile-level arrays (n = chars / 4096)
__global__ __launch_bounds__(kScanBlock) void k_scan_small(const int64_t* __restrict__ in, int64_t n, int64_t* __restrict__ out,
                                                           int64_t* __restrict__ total, int64_t* __restrict__ total_host) {
    __shared__ long long lds[kScanBlock / 64];
    const int64_t per = (n + kScanBlock - 1) / kScanBlock;
    const int64_t lo = min((int64_t)threadIdx.x * per, n), hi = min(lo + per, n);
    long long sum = 0;
    for (int64_t i = lo; i < hi; ++i) sum += in[i];
    long long tot;
    long long run = block_exclusive_scan_ll(sum, &tot, lds);
    for (int64_t i = lo; i < hi; ++i) {
        const long long v = in[i];
        out[i] = run;
        run += v;
    }
    if (threadIdx.x == 0) {
        *total = tot;
        if (total_host) *total_host = tot;
    }
}

constexpr int64_t kScanSmallMax = 4096;   // beyond that one workgroup is slower than three launches over many

hipError_t launch_exclusive_scan(const int64_t* in, int64_t n, int64_t* out, int64_t* total, int64_t* block_tot,
                                 hipStream_t st, int64_t* total_host) {
    if (n <= kScanSmallMax) {
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(kScanBlock), 0, st, in, n, out, total, total_host);
        return hipGetLastError();
    }
    const int64_t n_blocks = scan_blocks(n);
    hipLaunchKernelGGL(k_scan_local, dim3((unsigned)n_blocks), dim3(kScanBlock), 0, st, in, n, out, block_tot);
    hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(kScanBlock), 0, st, block_tot, n_blocks, total, total_host);
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)n_blocks), dim3(kScanBlock), 0, st, out, n, block_tot);
    return hipGetLastError();
}
// row offsets of a chunk of a batch, uploaded as they are: subtract the chunk's first offset (chunked host pipeline)
__global__ void k_rebase_rows(int64_t* __restrict__ row, int64_t n, int64_t base) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) row[i] -= base;
}
hipError_t launch_rebase_rows(int64_t* row, int64_t n, int64_t base, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_rebase_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, row, n, base);
    return hipGetLastError();
}
int64_t scan_blocks(int64_t n) { return n > 0 ? (n + kScanChunk - 1) / kScanChunk : 1; }

// ---- UTF-8 ingest: decode a CSR batch of UTF-8 strings to packed UTF-32 on the device --------------
// The reference never sees UTF-8: it reads CPython's PEP-393 buffer (latok.c:53-55,79).  Feeding the GPU path from
// UTF-8 cuts the host->device bytes ~4x for ASCII.  One thread per string (strings are short in batch workloads).
// One code point per lead byte; continuation bytes that do not belong to a lead byte are skipped, truncated or
// malformed sequences yield U+FFFD -- input is expected to be valid UTF-8 ("surrogatepass" forms decode as they are).
__device__ __forceinline__ bool utf8_is_lead(uint8_t b) { return (b & 0xC0u) != 0x80u; }

// Chunk-parallel decode: the byte stream is cut into 16-byte chunks (one per thread, 4 KiB per 256-thread block).
//   pass 1  k_utf8_block_counts : lead bytes per block                       -> block_cnt[b]
//   (device-wide exclusive scan of block_cnt -> block_base[b], grand total = number of code points)
//   pass 2  k_utf8_decode_chunks: per block, exclusive scan of the chunk counts; every lead byte of a chunk is decoded
//                                 and stored at cps[block_base + chunk prefix + rank]; chunk prefixes are kept (u16)
//   pass 3  k_utf8_cp_offsets   : code-point offset of every string start from block_base + chunk prefix + the leads
//                                 of its chunk before it
// Everything is coalesced; strings never have to be walked byte by byte.
constexpr int kU8Threads = 256;
constexpr int kU8Chunk = 16;
constexpr int kU8Block = kU8Threads * kU8Chunk;   // 4096 bytes

// 16 bytes of the stream starting at byte p (zero beyond `total`), as 4 little-endian dwords
__device__ __forceinline__ uint4 utf8_load16(const uint8_t* __restrict__ u8, int64_t p, int64_t total) {
    uint4 v = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);   // padding looks like continuation bytes
    if (p + 16 <= total && ((uintptr_t)(u8 + p) & 15) == 0) return *reinterpret_cast<const uint4*>(u8 + p);
    uint32_t w[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
    for (int i = 0; i < 16; ++i)
        if (p + i < total) w[i >> 2] = (w[i >> 2] & ~(0xFFu << (8 * (i & 3)))) | ((uint32_t)u8[p + i] << (8 * (i & 3)));
    v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
    return v;
}
__device__ __forceinline__ uint32_t utf8_lead_mask16(uint4 v) {
    return utf8_lead_nibble(v.x) | (utf8_lead_nibble(v.y) << 4) | (utf8_lead_nibble(v.z) << 8) | (utf8_lead_nibble(v.w) << 12);
}

__device__ __forceinline__ int block_exclusive_scan_int(int v, int* total, int* lds /*[kU8Threads/64]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    __syncthreads();
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kU8Threads / 64; ++w) {
        if (w < wave) before += lds[w];
        all += lds[w];
    }
    *total = all;
    return before + inc - v;
}

__global__ __launch_bounds__(kU8Threads) void k_utf8_block_counts(const uint8_t* __restrict__ u8, int64_t total,
                                                                  int64_t* __restrict__ block_cnt) {
    __shared__ int lds[kU8Threads / 64];
    const int64_t p = ((int64_t)blockIdx.x * kU8Threads + threadIdx.x) * kU8Chunk;
    const int c = p < total ? __popc(utf8_lead_mask16(utf8_load16(u8, p, total))) : 0;
    int tot;
    (void)block_exclusive_scan_int(c, &tot, lds);
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = tot;
}

// The block's code points are collected in LDS at their rank inside the block and then streamed out, so that the 4
// bytes per code point leave as full, consecutive lines (a thread storing its own 16 code points one by one touches
// 64 different lines per wave instruction: 5x slower).
__global__ __launch_bounds__(kU8Threads) void k_utf8_decode_chunks(const uint8_t* __restrict__ u8, int64_t total,
                                                                   const int64_t* __restrict__ block_base,
                                                                   uint16_t* __restrict__ chunk_pref,
                                                                   uint32_t* __restrict__ cps) {
    __shared__ int lds[kU8Threads / 64];
    __shared__ uint32_t out_s[kU8Block];
    const int64_t chunk = (int64_t)blockIdx.x * kU8Threads + threadIdx.x;
    const int64_t p = chunk * kU8Chunk;
    uint4 v = make_uint4(0, 0, 0, 0), vd = v;   // vd: the same bytes as the decode windows see them
    uint32_t leads = 0;
    if (p < total) {
        v = utf8_load16(u8, p, total);
        leads = utf8_lead_mask16(v);
        vd = v;
        if (total - p < 16) {
            // the batch ends inside my chunk: for the lead count the padding had to look like continuation bytes, for the
            // decode WINDOWS a byte that does not exist is "not a continuation byte" (a lead cut short by the end of the batch
            // is U+FFFD, as everywhere else -- the 0x80 padding used to complete it: b"\xc3" at the very end decoded as U+00C0)
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
            for (int i = (int)(total - p); i < 16; ++i) w[i >> 2] |= 0xFFu << (8 * (i & 3));
            vd = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    int tot;
    const int excl = block_exclusive_scan_int(__popc(leads), &tot, lds);
    if (p < total) chunk_pref[chunk] = (uint16_t)excl;
    // bytes p+16 .. p+18 (what a sequence that starts in my last bytes may need): the next thread's first dword, taken
    // while every lane is active; the last lane of a wave reads them from memory.  A byte that does not exist reads as 0xFF
    // ("not a continuation byte": the sequence is truncated -> U+FFFD).
    uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vd.x, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
    if ((threadIdx.x & 63) == 63) {
        nx = 0;
        for (int i = 0; i < 3; ++i) {
            const int64_t q = p + 16 + i;
            if (q < total) nx |= (uint32_t)u8[q] << (8 * i);
        }
    }
    {
        const int64_t have = total - (p + 16);                        // bytes that exist behind my chunk
        nx |= have >= 3 ? 0xFF000000u : (have <= 0 ? 0xFFFFFFFFu : (0xFFFFFFFFu << (8 * (int)have)));
    }
    if (leads) {
        uint32_t* dst = out_s + excl;
        if (((v.x | v.y | v.z | v.w) & 0x80808080u) == 0) {
            // 16 ASCII bytes = 16 code points
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dst[4 * j + 0] = w4[j] & 0xFFu;
                dst[4 * j + 1] = (w4[j] >> 8) & 0xFFu;
                dst[4 * j + 2] = (w4[j] >> 16) & 0xFFu;
                dst[4 * j + 3] = w4[j] >> 24;
            }
        } else {
            // A mixed chunk.  ASCII lead bytes are their own code points; only the NON-ASCII leads are decoded -- two
            // branch-free slots per dword (well-formed UTF-8 has at most two multi-byte leads in 4 bytes; more: the loop
            // below), as in the byte-space tile kernel (split_kernels.hip: bytes_phase1).  A lead's place in the output is
            // its rank among the chunk's leads.  (The old form decoded all 16 positions with the branchy utf8_decode_at:
            // 554 us for the 471 MB of C3.)
            // bytes p+16 .. p+18: the next thread's first dword; the last lane of a wave reads them from memory
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
            const uint32_t w[5] = {vd.x, vd.y, vd.z, vd.w, nx};
            uint32_t rest[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t hi = d[q] & 0x80808080u;
                const uint32_t nl = hi & (d[q] << 1);                     // 11xxxxxx: the leads that need a decode
                // ASCII leads of the dword
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * q + r;
                    const uint32_t b = (d[q] >> (8 * r)) & 0xFFu;
                    if (b < 0x80u && ((leads >> k) & 1u)) dst[__popc(leads & ((1u << k) - 1u))] = b;
                }
                uint32_t m = nl;
#pragma unroll
                for (int slot = 0; slot < 2; ++slot) {
                    const uint32_t r = ((uint32_t)__builtin_ctz(m | 0x80000000u) >> 3) & 3u;
                    const uint32_t cp = utf8_cp_of(__builtin_amdgcn_alignbyte(w[q + 1], w[q], r));
                    const uint32_t k = 4u * q + r;
                    if (m) dst[__popc(leads & ((1u << k) - 1u))] = cp;
                    m &= m - 1u;
                }
                rest[q] = m;
            }
            while ((rest[0] | rest[1] | rest[2] | rest[3]) != 0u) {       // malformed input only
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (rest[q]) {
                        const uint32_t r = ((uint32_t)__builtin_ctz(rest[q]) >> 3) & 3u;
                        const uint32_t k = 4u * q + r;
                        dst[__popc(leads & ((1u << k) - 1u))] = utf8_cp_of(__builtin_amdgcn_alignbyte(w[q + 1], w[q], r));
                        rest[q] &= rest[q] - 1u;
                    }
                }
            }
        }
    }
    __syncthreads();
    uint32_t* dst_g = cps + block_base[blockIdx.x];
    for (int i = threadIdx.x; i < tot; i += kU8Threads) dst_g[i] = out_s[i];
}

__global__ void k_utf8_cp_offsets(const uint8_t* __restrict__ u8, int64_t total, const int64_t* __restrict__ byte_off,
                                  int64_t n_str, const int64_t* __restrict__ block_base,
                                  const uint16_t* __restrict__ chunk_pref, int64_t total_cps,
                                  int64_t* __restrict__ cp_off) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_str) return;
    const int64_t p = byte_off[s];
    if (p >= total) { cp_off[s] = total_cps; return; }
    const int64_t chunk = p / kU8Chunk;
    const uint32_t leads = utf8_lead_mask16(utf8_load16(u8, chunk * kU8Chunk, total));
    const int within = (int)(p - chunk * kU8Chunk);
    cp_off[s] = block_base[chunk / kU8Threads] + chunk_pref[chunk] + __popc(leads & ((1u << within) - 1u));
}

int64_t utf8_blocks(int64_t total_bytes) { return total_bytes > 0 ? (total_bytes + kU8Block - 1) / kU8Block : 1; }

hipError_t launch_utf8_block_counts(const uint8_t* u8, int64_t total, int64_t* block_cnt, hipStream_t st) {
    hipLaunchKernelGGL(k_utf8_block_counts, dim3((unsigned)utf8_blocks(total)), dim3(kU8Threads), 0, st, u8, total, block_cnt);
    return hipGetLastError();
}
hipError_t launch_utf8_decode(const uint8_t* u8, int64_t total, const int64_t* byte_off, int64_t n_str,
                              const int64_t* block_base, uint16_t* chunk_pref, int64_t total_cps, uint32_t* cps,
                              int64_t* cp_off, hipStream_t st) {
    hipLaunchKernelGGL(k_utf8_decode_chunks, dim3((unsigned)utf8_blocks(total)), dim3(kU8Threads), 0, st, u8, total,
                       block_base, chunk_pref, cps);
    hipLaunchKernelGGL(k_utf8_cp_offsets, dim3((unsigned)((n_str + 1 + 255) / 256)), dim3(256), 0, st, u8, total, byte_off,
                       n_str, block_base, chunk_pref, total_cps, cp_off);
    return hipGetLastError();
}

// ---- PEP 393 code units -> UTF-32 (the entry points that need code points in HBM: featurize, run-time rule tables) ----
template <typename T>
__global__ void k_widen_units(const T* __restrict__ units, int64_t n, uint32_t* __restrict__ cps) {
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 + 4 <= n && ((uintptr_t)units & (4 * sizeof(T) - 1)) == 0) {
        T u[4];
        if (sizeof(T) == 1) *reinterpret_cast<uint32_t*>(u) = *reinterpret_cast<const uint32_t*>(units + i0);
        else *reinterpret_cast<uint2*>(u) = *reinterpret_cast<const uint2*>(units + i0);
        *reinterpret_cast<uint4*>(cps + i0) = make_uint4(u[0], u[1], u[2], u[3]);
    } else {
        for (int64_t i = i0; i < n && i < i0 + 4; ++i) cps[i] = units[i];
    }
}

hipError_t launch_widen_units(const void* units, int kind, int64_t n, uint32_t* cps, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 1023) / 1024);
    if (kind == 1) hipLaunchKernelGGL(k_widen_units<uint8_t>, dim3(blocks), dim3(256), 0, st, (const uint8_t*)units, n, cps);
    else hipLaunchKernelGGL(k_widen_units<uint16_t>, dim3(blocks), dim3(256), 0, st, (const uint16_t*)units, n, cps);
    return hipGetLastError();
}

// ---- synthetic corpus ------------------------------------------------------------------------------------------
__global__ void k_corpus_fill(uint64_t seed, int model, uint64_t sid0, int64_t n_str,
                              const int64_t* __restrict__ row_off, uint32_t* __restrict__ cps) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_str) return;
    const int64_t b = row_off[s];
    latok_corpus_string(seed, model, sid0 + (uint64_t)s, cps + b, row_off[s + 1] - b);
}

hipError_t launch_corpus_fill(uint64_t seed, int model, uint64_t sid0, int64_t n_str, const int64_t* row_off,
                              uint32_t* cps, hipStream_t st) {
    if (n_str <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_corpus_fill, dim3((unsigned)((n_str + 63) / 64)), dim3(64), 0, st, seed, model, sid0, n_str,
                       row_off, cps);
    return hipGetLastError();
}

// ---- streaming-read ceiling (measurement only) -------------------------------------------------------------------
// The tile kernel's own load pattern (16 non-temporal 16-byte loads per lane and step, one wave per 16 KiB) with
// nothing else attached: what a pure HBM read of the same buffer costs on this box.  bench.py reports it beside the
// 8 TB/s spec figure.
typedef uint32_t stream_u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512) k_stream_read(const stream_u32x4* __restrict__ src, int64_t n_steps,
                                                     uint32_t* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (int64_t t = wave; t < n_steps; t += n_waves) {
        const stream_u32x4* p = src + t * 1024 + lane;
        stream_u32x4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(p + 64 * i);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;   // keeps the loads alive; practically never taken
}

hipError_t launch_stream_read(const void* src, int64_t bytes, uint32_t* sink, int n_cu, hipStream_t st) {
    const int64_t n_steps = bytes / 16384;
    if (n_steps <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_stream_read, dim3((unsigned)(n_cu * 2)), dim3(512), 0, st, (const stream_u32x4*)src, n_steps, sink);
    return hipGetLastError();
}

__global__ void k_utf8_bytes(const uint32_t* __restrict__ cps, int64_t n, unsigned long long* __restrict__ total) {
    long long local = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        local += latok_utf8_len(cps[i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) local += __shfl_xor(local, d);
    if ((threadIdx.x & 63) == 0) atomicAdd(total, (unsigned long long)local);
}

hipError_t launch_utf8_bytes(const uint32_t* cps, int64_t n, unsigned long long* total, hipStream_t st) {
    hipError_t e = hipMemsetAsync(total, 0, sizeof(unsigned long long), st);
    if (e != hipSuccess || n <= 0) return e;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_utf8_bytes, dim3((unsigned)blocks), dim3(256), 0, st, cps, n, total);
    return hipGetLastError();
}

}  // namespace latok
