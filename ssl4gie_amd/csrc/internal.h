// Internal (non-exported) helpers shared between translation units of libssl4gie_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// out[c] (+)= sum_p partial[p*stride + c], c < n_out   (deterministic second reduction stage)
int ssl4gie_internal_reduce_partials(const float* partial, float* out, int nparts, int n_out,
                                     size_t stride, int accumulate, hipStream_t st);

// the second stage of ssl4gie_layernorm_bwd alone: [dgamma | dbeta] (+)= sums over the per-block partial rows the
// first stage left in `workspace` (ssl4gie_layernorm_bwd called with dgamma == dbeta == NULL) — the block executor
// enqueues it on its weight-gradient stream, off the data-gradient chain (engine.hip)
int ssl4gie_internal_ln_reduce(const float* workspace, float* dgamma, float* dbeta, int rows, int cols,
                               int accumulate, hipStream_t st);
