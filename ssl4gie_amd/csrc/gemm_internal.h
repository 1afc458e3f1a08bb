// Internal GEMM plumbing shared by gemm.hip and gemm_nt256.hip (not part of the C ABI).
#pragma once
#include "common.h"
#include "ssl4gie_hip.h"

struct EpiArgs {
    float alpha;
    int mode;
    const float* bias;
    const float* residual;
    long long ldr;
    const void* aux;
    void* out2;
    int accumulate;
    float* colstats;  // optional (256x256 NT kernel, bf16 C, EPI_NONE): see ssl4gie_gemm_desc::colstats
    int nt_store;     // 256x256 NT kernel: non-temporal output stores (gemm256.h est)
};

// Implicit 3x3 / pad-1 patch-matrix operand (ssl4gie_gemm_desc::conv): device-side geometry with
// the divisions by Wo and Ho*Wo turned into multiply-high + shift (conv_magic, n < 2^31).
struct ConvK {
    int H, W, C, Wo, HoWo, stride;
    unsigned mg_wo, sh_wo, mg_hw, sh_hw;
    const void* zero;  // >= 16 zero bytes: the source of every out-of-image tap
};
static inline void conv_magic(unsigned d, unsigned* mg, unsigned* sh) {  // d >= 2
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    *mg = (unsigned)((1ull << (31 + s)) / d + 1);
    *sh = s - 1;
}
// pixels of the conv output / whether the gathered 256x256 kernels can run this geometry
static inline long long conv_rows(const ssl4gie_conv3x3_geom* g) {
    const long long Ho = (g->H - 1) / g->stride + 1, Wo = (g->W - 1) / g->stride + 1;
    return (long long)g->B * Ho * Wo;
}
bool ssl4gie_internal_conv_geom_ok(const ssl4gie_conv3x3_geom* g);
int ssl4gie_internal_conv_k(const ssl4gie_conv3x3_geom* g, ConvK* k);  // fills k (zero page incl.)

// CUs the persistent / one-workgroup-per-CU GEMM grids are sized for (ssl4gie_set_compute_cus):
// 256 on a GPU of its own; data-parallel runs leave a few CUs to the RCCL kernels, which cannot
// share a CU with a workgroup that holds all 160 KiB of LDS.
int ssl4gie_internal_compute_cus();

// 256x256x64 ping-pong NT kernel (gemm_nt256.hip): C[M,N] = A[M,K] B[N,K]^T with the fused
// epilogues.  `nt256_ok` says whether the descriptor (already known to satisfy the NT fast-path
// layout rules) is worth / able to run on it; `nt256_launch` enqueues it.
bool ssl4gie_internal_nt256_ok(const ssl4gie_gemm_desc* d);
int ssl4gie_internal_nt256_launch(const ssl4gie_gemm_desc* d, hipStream_t st);

// 256x256x64 ping-pong TN kernel (gemm_tn256.hip): weight-gradient product with split-K slabs.
bool ssl4gie_internal_tn256_ok(const ssl4gie_gemm_desc* d);
int ssl4gie_internal_tn256_splits(const ssl4gie_gemm_desc* d);
// `slabs` (splits > 1) and `colsum_part` ([splits][M], splits > 1 and d->colsum_a) are workspace
int ssl4gie_internal_tn256_launch(const ssl4gie_gemm_desc* d, float* slabs, float* colsum_part,
                                  hipStream_t st);
// Further problems of a grouped launch (TN products with the same contraction length K share one
// grid: with enough output tiles in the group no split-K slabs are needed at all).  Problem 0 is
// passed as plain kernel arguments, problems 1..n as this table; `tile0` is the first tile of the
// problem in the group's tile order, after problem 0's tiles.
struct TnSecond {
    const void* At; long long ldat;
    const void* Bt; long long ldbt;
    float* C; long long ldc;
    float* slabs;
    int M, N, tiles_n, ntiles;
    float* colsum; float* colsum_part;
    float alpha; int accumulate; int tile0;
};
#define TN_GROUP_MAX 16
struct TnExtras {
    int n, total_tiles;
    int m_inner;  // walk the shorter tile dimension innermost (gemm_tn256.hip)
    TnSecond p[TN_GROUP_MAX - 1];
};
int ssl4gie_internal_tn256_group_splits(const ssl4gie_gemm_desc* descs, int n);
// slabs[i] / cs[i]: split-K workspace of problem i (splits > 1), cs[i] only if it has colsum_a
int ssl4gie_internal_tn256_launch_group(const ssl4gie_gemm_desc* descs, int n, int splits,
                                        float* const* slabs, float* const* cs, hipStream_t st);
int ssl4gie_internal_tn256_pair_splits(const ssl4gie_gemm_desc* a, const ssl4gie_gemm_desc* b);
int ssl4gie_internal_tn256_launch_pair(const ssl4gie_gemm_desc* a, const ssl4gie_gemm_desc* b,
                                       int splits, float* slabs_a, float* cs_a, float* slabs_b,
                                       float* cs_b, hipStream_t st);
