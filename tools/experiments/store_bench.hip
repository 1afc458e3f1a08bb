// Micro-benchmark (round 3): how fast can ONE CU push stores that miss the L2, by access pattern?
// Motivation: gemm_nt256's exposed epilogue is a store drain capped at ~11 B/clk per CU whatever the
// number of storing CUs (profiles/r03d).  One 512-thread workgroup per CU (100 KiB of LDS declared so
// that only one fits) writes `tiles` 256x256 output tiles into its own rows of a big matrix:
//   pat 0  the bf16 epilogue's pattern: a wave instruction = 8 rows x 128 B (lanes 0-7 one row)
//   pat 1  the fp32 epilogue's pattern: a wave instruction = 4 rows x 256 B
//   pat 2  whole rows: a wave instruction = 2 rows x 512 B (bf16 tile width), 8 waves on 16 rows
//   pat 3  1 KiB contiguous per wave instruction (what a streaming copy does)
//   pat 4  pattern 0 with sc1 (write-through) stores;  pat 5  pattern 0 with nt stores
// build: hipcc -O3 --offload-arch=gfx950 store_bench.hip -o store_bench ; run: ./store_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PAT>
__global__ __launch_bounds__(512) void store_kernel(char* base, long long ld_bytes, int tiles, int tiles_per_row) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (int t = 0; t < tiles; ++t) {
        const long long tile = (long long)blockIdx.x * tiles + t;
        char* tb = base + (tile / tiles_per_row) * 256 * ld_bytes + (tile % tiles_per_row) * 512;  // 256 rows x 512 B
        if (PAT == 0 || PAT == 4 || PAT == 5) {
            const int wr = wave >> 2, wc = wave & 3;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    char* p = tb + (long long)(128 * wr + 16 * mt + (lane >> 3) + 8 * hh) * ld_bytes + wc * 128 + (lane & 7) * 16;
                    if (PAT == 0) *(u32x4*)p = v;
                    else if (PAT == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
                    else asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
                }
        } else if (PAT == 1) {   // 256 rows x 512 B as two 256-B halves per row: wave (wr, wc2): rows 128 wr.., half wc&1 ... 2 passes
            const int wr = wave >> 2, wc = wave & 3;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    // 16 rows of this block: 4 rows x 256 B per instruction... the wave owns a 128-B column group in pattern 0;
                    // here it owns 256 B of a row pair: rows (lane >> 4) + 4 q + 8 (wc >> 1), half (wc & 1)
                    char* p = tb + (long long)(128 * wr + 16 * mt + (lane >> 4) + 4 * q + 8 * (wc >> 1)) * ld_bytes + (wc & 1) * 256 + (lane & 15) * 16;
                    *(u32x4*)p = v;
                }
        } else if (PAT == 2) {   // 8 waves x 2 rows x 512 B = 16 rows per step
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                char* p = tb + (long long)(16 * s + 2 * wave + (lane >> 5)) * ld_bytes + (lane & 31) * 16;
                *(u32x4*)p = v;
            }
        } else {                 // contiguous: the tile as a 128 KiB block
            char* cb = base + tile * 131072;
#pragma unroll
            for (int s = 0; s < 16; ++s) *(u32x4*)(cb + (s * 8 + wave) * 1024 + lane * 16) = v;
        }
    }
}

int main() {
    const long long ld = 4096;           // a [M, 2048] bf16 matrix
    const int tiles_per_row = 8;
    const size_t bytes = 8ull << 30;     // 8 GiB: far beyond L2 / Infinity Cache
    char* buf;
    hipMalloc(&buf, bytes);
    hipMemset(buf, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int lds = 100 * 1024;
#define RUN(P)                                                                                          \
    for (int cus : {16, 60, 120, 240}) {                                                                  \
        const int tiles = 64;  /* 8 MiB per workgroup */                                                  \
        hipFuncSetAttribute((const void*)store_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        store_kernel<P><<<cus, 512, lds>>>(buf, ld, 4, tiles_per_row);                                    \
        hipDeviceSynchronize();                                                                           \
        hipEventRecord(e0);                                                                               \
        store_kernel<P><<<cus, 512, lds>>>(buf, ld, tiles, tiles_per_row);                                \
        hipEventRecord(e1);                                                                               \
        hipEventSynchronize(e1);                                                                          \
        float ms;                                                                                         \
        hipEventElapsedTime(&ms, e0, e1);                                                                 \
        const double gb = (double)cus * tiles * 131072 / 1e9;                                             \
        printf("pat %d cus %3d: %7.3f ms  %7.1f GB/s total  %6.1f GB/s per CU  (%.2f us per 128 KiB tile)\n", P, cus, ms, \
               gb / (ms * 1e-3), gb / (ms * 1e-3) / cus, ms * 1e3 / tiles);                               \
    }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    return 0;
}
