// Micro-benchmark (round 4): what limits the store rate of gemm_nt256's epilogue PER CU?
// r04a showed that the epilogue alone (no LDS-DMA, no MFMA: SSL4GIE_NT256_NOEPI=7) moves only 13-17 GB/s per CU
// even on 60 CUs, where store_bench.hip's pure store stream reaches 94 GB/s per CU with the same instruction shape
// (8 rows x 128 B per wave instruction).  This kernel starts from the epilogue's structure and removes / adds one
// ingredient at a time (bit flags in `f`):
//   1  tile order: 0 = the GEMM's (tile = xcd_remap(block) + ti * grid), 1 = store_bench's (block * tiles + ti)
//   2  LDS round trip in front of every store (4 ds_write_b64 + ds_read_b128 + lgkmcnt(0)), as the epilogue does
//   4  one dependent global load + s_waitcnt vmcnt(0) at the start of every tile (the bias load: it sits behind the
//      previous tile's stores in the in-order vmcnt queue)
//   8  a pause of `pause_us` between tiles (the K-loop: no memory traffic)
//  16  two output tensors written alternately (the GELU pair)
//  32  stores carry `nt`
//  64  the GELU-pair arithmetic of the real epilogue on 16 values per lane and row block (exp, rcp, erf polynomial)
// 128  no global stores at all (what the arithmetic + LDS round trips cost alone)
// 256  store shape 16 rows x 64 B per wave instruction (what a lane-exchange transposition with v_permlane16_swap
//      gives for bf16, and the accumulator layout itself for fp32) instead of 8 rows x 128 B; use without 2
// One 512-thread workgroup per CU (160 KiB LDS so that only one fits), `tiles` 256 x 256 bf16 tiles per workgroup.
// Per-tile store-issue time is stamped by wave 0 and wave 4 (s_memrealtime, 100 MHz) and reported as a median.
// build: hipcc -O3 --offload-arch=gfx950 epi_bench.hip -o epi_bench ; run: ./epi_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ inline int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

#define MAXT 64
__global__ __launch_bounds__(512) void epi_kernel(char* c0, char* c1, const float* bias, long long ld_bytes, int tiles,
                                                  int tiles_n, int ntiles_total, int f, int pause_ticks,
                                                  unsigned* stamps /* [grid][2][MAXT] issue time in ticks */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int G = gridDim.x;
    const int pos = xcd_remap(blockIdx.x, G);
    char* stg = smem + wave * 4096;
    const int r16 = lane & 15, g4 = lane >> 4, R0 = lane >> 3, Cc = lane & 7;
    float acc = (float)lane;
    float vals[4][4], gd[4][4], gg[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { vals[i][j] = 0.01f * (lane + 4 * i + j) - 0.5f; gd[i][j] = gg[i][j] = 0.f; }
    for (int ti = 0; ti < tiles; ++ti) {
        int tile = (f & 1) ? blockIdx.x * tiles + ti : pos + ti * G;
        if (tile >= ntiles_total) break;
        const long long m0 = (long long)(tile / tiles_n) * 256, n0 = (long long)(tile % tiles_n) * 256;
        if (f & 8) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)pause_ticks) __builtin_amdgcn_s_sleep(8);
            __syncthreads();
        }
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        if (f & 4) {
            const float b = bias[(n0 + wc * 64 + lane) & 1023];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += b;
        }
        char* tb0 = c0 + (m0 + 128 * wr) * ld_bytes + (n0 + wc * 64) * 2;
        char* tb1 = c1 + (m0 + 128 * wr) * ld_bytes + (n0 + wc * 64) * 2;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                if (rep == 1 && !(f & 16)) break;
                char* tb = rep ? tb1 : tb0;
                if ((f & 64) && rep == 0) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float x = vals[nt][q] * 1.0001f + 0.01f * mt;
                            const float e = __expf(-0.5f * x * x);
                            const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * 0.70710678f * fabsf(x));
                            const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
                            const float cdf = 0.5f + 0.5f * copysignf(1.0f - poly * e, x);
                            gd[nt][q] = cdf + x * e * 0.3989422804f;
                            gg[nt][q] = x * cdf;
                        }
                }
                if (f & 2) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        u32x2 pk = {__float_as_uint(acc + nt), __float_as_uint(acc - nt)};
                        if (f & 64) {
                            typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
                            const float* src = rep ? gg[nt] : gd[nt];
                            bf2 a = {(__bf16)src[0], (__bf16)src[1]}, b = {(__bf16)src[2], (__bf16)src[3]};
                            pk[0] = __builtin_bit_cast(unsigned, a); pk[1] = __builtin_bit_cast(unsigned, b);
                        }
                        const int c = nt * 2 + (g4 >> 1);
                        *(u32x2*)(stg + rep * 2048 + r16 * 128 + ((c ^ (r16 & 7)) << 4) + ((g4 & 1) << 3)) = pk;
                    }
                }
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int R = R0 + 8 * hh;
                    u32x4 w = {(unsigned)lane, (unsigned)mt, (unsigned)ti, 7u};
                    if (f & 2) w = *(const u32x4*)(stg + rep * 2048 + R * 128 + ((Cc ^ (R & 7)) << 4));
                    char* p = tb + (long long)(16 * mt + R) * ld_bytes + Cc * 16;
                    if (f & 256) p = tb + (long long)(16 * mt + r16) * ld_bytes + hh * 64 + g4 * 16;
                    if (f & 128) asm volatile("" ::"v"(w));
                    else if (f & 32) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(w) : "memory");
                    else *(u32x4*)p = w;
                }
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && (wave == 0 || wave == 4) && ti < MAXT) stamps[(blockIdx.x * 2 + (wave >> 2)) * MAXT + ti] = (unsigned)(t1 - t0);
    }
}

int main(int argc, char** argv) {
    const long long M = 50432;                 // 256 x 197 rows
    const int N = argc > 1 ? atoi(argv[1]) : 2048;
    const long long ld = (long long)N * 2;
    const int tiles_n = N / 256, ntiles = (int)(M / 256) * tiles_n;
    const size_t bytes = (size_t)M * ld;
    char *c0, *c1; float* bias; unsigned* stamps;
    hipMalloc(&c0, bytes); hipMalloc(&c1, bytes); hipMalloc(&bias, 4096);
    hipMalloc(&stamps, 256 * 2 * MAXT * 4);
    hipMemset(c0, 0, bytes); hipMemset(c1, 0, bytes); hipMemset(bias, 0, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int lds = 160 * 1024;
    hipFuncSetAttribute((const void*)epi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int flagsets[] = {8, 8 + 256, 10, 24, 24 + 256, 26, 24 + 64, 24 + 64 + 256, 26 + 64, 0, 256, 16, 16 + 256};
    for (int cus : {240}) {
        for (int f : flagsets) {
            const int tiles = (ntiles + cus - 1) / cus;
            const int pause = 1200;  // 12 us
            hipMemset(stamps, 0, 256 * 2 * MAXT * 4);
            for (int it = 0; it < 2; ++it) epi_kernel<<<cus, 512, lds>>>(c0, c1, bias, ld, tiles, tiles_n, ntiles, f, pause, stamps);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            const int iters = 5;
            for (int it = 0; it < iters; ++it) epi_kernel<<<cus, 512, lds>>>(c0, c1, bias, ld, tiles, tiles_n, ntiles, f, pause, stamps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
            std::vector<unsigned> h(256 * 2 * MAXT);
            hipMemcpy(h.data(), stamps, h.size() * 4, hipMemcpyDeviceToHost);
            std::vector<unsigned> a, b;
            for (int wg = 0; wg < cus; ++wg)
                for (int t = 1; t < std::min(tiles - 1, MAXT); ++t) {
                    if (h[(wg * 2) * MAXT + t]) a.push_back(h[(wg * 2) * MAXT + t]);
                    if (h[(wg * 2 + 1) * MAXT + t]) b.push_back(h[(wg * 2 + 1) * MAXT + t]);
                }
            std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
            const double per_tile_kb = ((f & 16) ? 256.0 : 128.0);
            const double tot_gb = (double)ntiles * per_tile_kb * 1024 / 1e9;
            const double pause_ms = (f & 8) ? tiles * pause / 100.0 * 1e-3 : 0.0;
            printf("N %4d cus %3d flags %3d [%s%s%s%s%s%s%s%s%s]: kernel %7.1f us (%6.1f without the pauses) %6.0f GB/s total %5.1f GB/s per CU | "
                   "issue per tile: wave0 %5.2f us, wave4 %5.2f us (median) p90 %5.2f\n",
                   N, cus, f, (f & 1) ? "seq " : "gemm", (f & 2) ? " lds" : "", (f & 4) ? " biaswait" : "", (f & 8) ? " pause" : "",
                   (f & 16) ? " pair" : "", (f & 32) ? " nt" : "", (f & 64) ? " gelu" : "", (f & 128) ? " nostore" : "", (f & 256) ? " 16x64B" : "", ms * 1e3, (ms - pause_ms) * 1e3, tot_gb / ((ms - pause_ms) * 1e-3),
                   tot_gb / ((ms - pause_ms) * 1e-3) / cus, a.empty() ? 0.0 : a[a.size() / 2] / 100.0,
                   b.empty() ? 0.0 : b[b.size() / 2] / 100.0, a.empty() ? 0.0 : a[a.size() * 9 / 10] / 100.0);
        }
    }
    return 0;
}
