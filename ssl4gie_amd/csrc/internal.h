// Internal (non-exported) helpers shared between translation units of libssl4gie_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// out[c] (+)= sum_p partial[p*stride + c], c < n_out   (deterministic second reduction stage)
int ssl4gie_internal_reduce_partials(const float* partial, float* out, int nparts, int n_out,
                                     size_t stride, int accumulate, hipStream_t st);
