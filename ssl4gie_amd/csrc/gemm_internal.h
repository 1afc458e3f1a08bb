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
};

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
