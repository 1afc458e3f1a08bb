// MAE glue kernels (integer masking, patch gather, token assembly, loss) and weight casts.
// All HBM-bound byte/row movers: 16-byte vector accesses, one row per wave or per block, no
// atomics (every reduction is two-stage and deterministic).
//
// Reference: Models/mae/models_mae.py:95-107 (patchify), :123-148 (random_masking),
// :150-170 (forward_encoder glue), :172-196 (forward_decoder glue), :198-214 (forward_loss).
#include "common.h"
#include "ssl4gie_hip.h"
#include "internal.h"

// ------------------------------------------------------------------ casts
template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long long n) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        st4(dst + i, ld4(src + i));
    } else {
        for (long long j = i; j < n; ++j) Elem<T>::st(dst + j, src[j]);
    }
}
// dst[c][r] = src[r][c] through a 32x33 LDS tile
template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src,
                                                             T* __restrict__ dst, int rows,
                                                             int cols) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + i * 8, c = c0 + tx;
        tile[ty + i * 8][tx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8, r = r0 + tx;
        if (c < cols && r < rows) Elem<T>::st(dst + (size_t)c * rows + r, tile[tx][ty + i * 8]);
    }
}

extern "C" int ssl4gie_cast(const float* src, void* dst, int dst_dtype, long long n,
                            void* stream) {
    REQUIRE(src && dst && n >= 0);
    REQUIRE(dst_dtype == SSL4GIE_F32 || dst_dtype == SSL4GIE_BF16);
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((n + 1023) / 1024);
    if (dst_dtype == SSL4GIE_BF16)
        hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, src, (bf16_t*)dst, n);
    else
        hipLaunchKernelGGL(cast_kernel<float>, dim3(blocks), dim3(256), 0, st, src, (float*)dst, n);
    LAUNCH_CHECK();
    return 0;
}
extern "C" int ssl4gie_cast_transpose(const float* src, void* dst, int dst_dtype, int rows,
                                      int cols, void* stream) {
    REQUIRE(src && dst && rows > 0 && cols > 0);
    REQUIRE(dst_dtype == SSL4GIE_F32 || dst_dtype == SSL4GIE_BF16);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((cols + 31) / 32, (rows + 31) / 32), block(256);
    if (dst_dtype == SSL4GIE_BF16)
        hipLaunchKernelGGL(cast_transpose_kernel<bf16_t>, grid, block, 0, st, src, (bf16_t*)dst, rows, cols);
    else
        hipLaunchKernelGGL(cast_transpose_kernel<float>, grid, block, 0, st, src, (float*)dst, rows, cols);
    LAUNCH_CHECK();
    return 0;
}

// Transposed bf16 copies of MANY matrices of one fp32 arena in one launch (the per-step refresh of the
// pre-transposed weight operands: 83 matrices for MAE ViT-B, which as 83 launches of 5 us each cost
// more than the bytes).  Device tables: mat_off[m] = element offset of matrix m in `src` AND in
// `dst`, rows / cols, tile_start[m] = index of its first 64 x 64 tile (tile_start[S] = total).
// dst[off + c * rows + r] = bf16(src[off + r * cols + c]).
__global__ __launch_bounds__(256) void cast_transpose_batch_kernel(
    const float* __restrict__ src, bf16_t* __restrict__ dst, const long long* __restrict__ mat_off,
    const int* __restrict__ mat_rows, const int* __restrict__ mat_cols, const int* __restrict__ tile_start,
    int S) {
    __shared__ float tile[64][65];
    const int t = blockIdx.x;
    int lo = 0, hi = S;  // last m with tile_start[m] <= t
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tile_start[mid] <= t) lo = mid; else hi = mid;
    }
    const int R = mat_rows[lo], Cc = mat_cols[lo], lt = t - tile_start[lo];
    const int tiles_c = (Cc + 63) >> 6;
    const int r0 = (lt / tiles_c) * 64, c0 = (lt % tiles_c) * 64;
    const float* sp = src + mat_off[lo];
    bf16_t* dp = dst + mat_off[lo];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 x 16 threads, 4 elements each way
    const bool vec = (R % 4 == 0) && (Cc % 4 == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 16 * i, c = c0 + 4 * tx;
        f32x4 v = {0, 0, 0, 0};
        if (r < R) {
            if (vec && c + 3 < Cc) v = ld4(sp + (size_t)r * Cc + c);
            else
#pragma unroll
                for (int j = 0; j < 4; ++j) if (c + j < Cc) v[j] = sp[(size_t)r * Cc + c + j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[ty + 16 * i][4 * tx + j] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 16 * i, r = r0 + 4 * tx;
        if (c < Cc) {
            const f32x4 v = {tile[4 * tx][ty + 16 * i], tile[4 * tx + 1][ty + 16 * i],
                             tile[4 * tx + 2][ty + 16 * i], tile[4 * tx + 3][ty + 16 * i]};
            if (vec && r + 3 < R) st4(dp + (size_t)c * R + r, v);
            else
#pragma unroll
                for (int j = 0; j < 4; ++j) if (r + j < R) dp[(size_t)c * R + r + j] = f2bf(v[j]);
        }
    }
}
extern "C" int ssl4gie_cast_transpose_batch(const float* src, void* dst, const long long* mat_off,
                                            const int* mat_rows, const int* mat_cols,
                                            const int* tile_start, int S, int total_tiles, void* stream) {
    REQUIRE(src && dst && mat_off && mat_rows && mat_cols && tile_start && S > 0 && total_tiles > 0);
    hipLaunchKernelGGL(cast_transpose_batch_kernel, dim3((unsigned)total_tiles), dim3(256), 0,
                       (hipStream_t)stream, src, (bf16_t*)dst, mat_off, mat_rows, mat_cols, tile_start, S);
    LAUNCH_CHECK();
    return 0;
}

// out = a (+ b); optional operand-type copy (residual-gradient stream plumbing)
template <typename T>
__global__ void add_cast_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                float* __restrict__ out, T* __restrict__ out_lp, long long n) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 v = ld4(a + i);
    if (b) v += ld4(b + i);
    if (out) st4(out + i, v);
    if (out_lp) st4(out_lp + i, v);
}
extern "C" int ssl4gie_add_cast(const float* a, const float* b, float* out, void* out_lp,
                                int lp_dtype, long long n, void* stream) {
    REQUIRE(a && n >= 0 && n % 4 == 0 && (out || out_lp));
    REQUIRE(!out_lp || lp_dtype == SSL4GIE_F32 || lp_dtype == SSL4GIE_BF16);
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((n / 4 + 255) / 256);
    if (out_lp && lp_dtype == SSL4GIE_BF16)
        hipLaunchKernelGGL(add_cast_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, a, b, out, (bf16_t*)out_lp, n);
    else
        hipLaunchKernelGGL(add_cast_kernel<float>, dim3(blocks), dim3(256), 0, st, a, b, out, (float*)out_lp, n);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ random masking (integer)
// Stable ascending argsort by counting: rank(i) = #{j : x_j < x_i  or (x_j == x_i and j < i)}.
// ids_restore[i] = rank(i), ids_shuffle[rank(i)] = i, mask[i] = rank(i) >= len_keep.
__global__ __launch_bounds__(256) void mask_argsort_kernel(const float* __restrict__ noise,
                                                           long long* __restrict__ ids_shuffle,
                                                           long long* __restrict__ ids_restore,
                                                           float* __restrict__ mask, int L,
                                                           int len_keep) {
    extern __shared__ float row[];
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < L; i += blockDim.x) row[i] = noise[(size_t)b * L + i];
    __syncthreads();
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const float x = row[i];
        int rank = 0;
        for (int j = 0; j < L; ++j) {
            const float y = row[j];
            rank += (y < x) || (y == x && j < i);
        }
        if (ids_restore) ids_restore[(size_t)b * L + i] = rank;
        if (ids_shuffle) ids_shuffle[(size_t)b * L + rank] = i;
        if (mask) mask[(size_t)b * L + i] = rank >= len_keep ? 1.f : 0.f;
    }
}
extern "C" int ssl4gie_mask_argsort(const float* noise, long long* ids_shuffle,
                                    long long* ids_restore, float* mask, int B, int L,
                                    int len_keep, void* stream) {
    REQUIRE(B >= 0 && L > 0 && L <= 16384 && len_keep >= 0 && len_keep <= L);
    if (B == 0) return 0;
    REQUIRE(noise);
    hipLaunchKernelGGL(mask_argsort_kernel, dim3(B), dim3(256), L * sizeof(float),
                       (hipStream_t)stream, noise, ids_shuffle, ids_restore, mask, L, len_keep);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ patch gather (im2col, k=s=p)
// one block per output row (= one patch); threads sweep the C*p*p columns 4 pixels at a time.
template <typename T>
__global__ __launch_bounds__(256) void patch_gather_kernel(const float* __restrict__ img,
                                                           const long long* __restrict__ ids,
                                                           T* __restrict__ out, int C, int H, int W,
                                                           int p, int nsel, long long ids_stride,
                                                           int order) {
    const int rowi = blockIdx.x;  // b*nsel + j
    const int b = rowi / nsel, j = rowi % nsel;
    const int gw = W / p;
    const int patch = ids ? (int)ids[(size_t)b * ids_stride + j] : j;
    const int gy = patch / gw, gx = patch % gw;
    const int P = C * p * p;
    const float* ib = img + (size_t)b * C * H * W;
    T* orow = out + (size_t)rowi * P;
    for (int k4 = threadIdx.x * 4; k4 < P; k4 += blockDim.x * 4) {
        // enumerate in (c, py, px) order so that 4 consecutive px are one 16-B read
        const int c = k4 / (p * p), rem = k4 % (p * p), py = rem / p, px = rem % p;
        const f32x4 v = ld4(ib + ((size_t)c * H + gy * p + py) * W + gx * p + px);
        if (order == 0) {
            st4(orow + k4, v);
        } else {  // 'nhwpqc': col = (py*p + px)*C + c
#pragma unroll
            for (int i = 0; i < 4; ++i) Elem<T>::st(orow + (py * p + px + i) * C + c, v[i]);
        }
    }
}
extern "C" int ssl4gie_patch_gather(const float* img, const long long* ids, void* out,
                                    int out_dtype, int B, int C, int H, int W, int p, int nsel,
                                    long long ids_stride, int order, void* stream) {
    REQUIRE(img && out && B >= 0 && C > 0 && p > 0 && p % 4 == 0 && H % p == 0 && W % p == 0);
    REQUIRE(W % 4 == 0 && nsel > 0 && nsel <= (H / p) * (W / p) && (order == 0 || order == 1));
    REQUIRE(out_dtype == SSL4GIE_F32 || out_dtype == SSL4GIE_BF16);
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(B * nsel), block(256);
    if (out_dtype == SSL4GIE_BF16)
        hipLaunchKernelGGL(patch_gather_kernel<bf16_t>, grid, block, 0, st, img, ids, (bf16_t*)out,
                           C, H, W, p, nsel, ids_stride, order);
    else
        hipLaunchKernelGGL(patch_gather_kernel<float>, grid, block, 0, st, img, ids, (float*)out,
                           C, H, W, p, nsel, ids_stride, order);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ encoder token assembly
template <typename T>
__global__ void tokens_assemble_kernel(const T* __restrict__ y, const float* __restrict__ cls,
                                       const float* __restrict__ pos,
                                       const long long* __restrict__ ids, long long ids_stride,
                                       float* __restrict__ x, int nsel, int D) {
    const int row = blockIdx.x;  // b*(nsel+1) + t
    const int b = row / (nsel + 1), t = row % (nsel + 1);
    float* xr = x + (size_t)row * D;
    if (t == 0) {
        for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4)
            st4(xr + d, ld4(cls + d) + ld4(pos + d));
    } else {
        const int j = t - 1;
        const int patch = ids ? (int)ids[(size_t)b * ids_stride + j] : j;
        const T* yr = y + ((size_t)b * nsel + j) * D;
        const float* pr = pos + (size_t)(1 + patch) * D;
        for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4)
            st4(xr + d, ld4(yr + d) + ld4(pr + d));
    }
}
extern "C" int ssl4gie_tokens_assemble(const void* y, int y_dtype, const float* cls,
                                       const float* pos, const long long* ids,
                                       long long ids_stride, float* x, int B, int nsel, int D,
                                       void* stream) {
    REQUIRE(y && cls && pos && x && B >= 0 && nsel > 0 && D > 0 && D % 4 == 0);
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(B * (nsel + 1)), block(D / 4 < 256 ? 64 * ((D / 4 + 63) / 64) : 256);
    if (y_dtype == SSL4GIE_BF16)
        hipLaunchKernelGGL(tokens_assemble_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)y,
                           cls, pos, ids, ids_stride, x, nsel, D);
    else if (y_dtype == SSL4GIE_F32)
        hipLaunchKernelGGL(tokens_assemble_kernel<float>, grid, block, 0, st, (const float*)y, cls,
                           pos, ids, ids_stride, x, nsel, D);
    else
        return ARG_ERR;
    LAUNCH_CHECK();
    return 0;
}

template <typename T>
__global__ void tokens_assemble_bwd_kernel(const float* __restrict__ dx, T* __restrict__ dy,
                                           int nsel, int D) {
    const int row = blockIdx.x;  // b*nsel + j
    const int b = row / nsel, j = row % nsel;
    const float* s = dx + ((size_t)b * (nsel + 1) + 1 + j) * D;
    T* d = dy + (size_t)row * D;
    for (int k = threadIdx.x * 4; k < D; k += blockDim.x * 4) st4(d + k, ld4(s + k));
}
// out[d] (+)= sum_b src[b*stride + d].  Block = 64 columns x 16 row groups: a thread sums every 16th
// row (fixed order), the 16 partials of a column are folded in LDS in a fixed order (deterministic);
// the old one-thread-per-column loop over all rows took 75 us for 256 rows (3 blocks, latency-bound).
__global__ __launch_bounds__(1024) void strided_rowsum_kernel(const float* __restrict__ src, float* __restrict__ out,
                                                              int nrows, size_t stride, int D, int accumulate) {
    __shared__ float part[16][64];
    const int dl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + dl;
    float s = 0.f;
    if (d < D)
        for (int b = rg; b < nrows; b += 16) s += src[(size_t)b * stride + d];
    part[rg][dl] = s;
    __syncthreads();
    if (rg == 0 && d < D) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += part[r][dl];
        out[d] = accumulate ? out[d] + t : t;
    }
}
extern "C" int ssl4gie_tokens_assemble_bwd(const float* dx, void* dy, int dy_dtype, float* dcls,
                                           int accumulate, int B, int nsel, int D, void* stream) {
    REQUIRE(dx && B >= 0 && nsel > 0 && D > 0 && D % 4 == 0);
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (dy) {
        dim3 grid(B * nsel), block(D / 4 < 256 ? 64 * ((D / 4 + 63) / 64) : 256);
        if (dy_dtype == SSL4GIE_BF16)
            hipLaunchKernelGGL(tokens_assemble_bwd_kernel<bf16_t>, grid, block, 0, st, dx, (bf16_t*)dy, nsel, D);
        else if (dy_dtype == SSL4GIE_F32)
            hipLaunchKernelGGL(tokens_assemble_bwd_kernel<float>, grid, block, 0, st, dx, (float*)dy, nsel, D);
        else
            return ARG_ERR;
        LAUNCH_CHECK();
    }
    if (dcls) {
        hipLaunchKernelGGL(strided_rowsum_kernel, dim3((D + 63) / 64), dim3(1024), 0, st, dx, dcls,
                           B, (size_t)(nsel + 1) * D, D, accumulate);
        LAUNCH_CHECK();
    }
    return 0;
}

// ------------------------------------------------------------------ decoder token assembly
template <typename T>
__global__ void decoder_assemble_kernel(const T* __restrict__ y, const float* __restrict__ mtok,
                                        const float* __restrict__ dpos,
                                        const long long* __restrict__ ids_restore,
                                        float* __restrict__ xd, int L, int nkeep, int D) {
    const int row = blockIdx.x;  // b*(L+1) + t
    const int b = row / (L + 1), t = row % (L + 1);
    float* xr = xd + (size_t)row * D;
    const float* pr = dpos + (size_t)t * D;
    const T* src = nullptr;
    if (t == 0) {
        src = y + (size_t)b * (nkeep + 1) * D;
    } else {
        const int r = (int)ids_restore[(size_t)b * L + t - 1];
        if (r < nkeep) src = y + ((size_t)b * (nkeep + 1) + 1 + r) * D;
    }
    for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) {
        const f32x4 v = src ? ld4(src + d) : ld4(mtok + d);
        st4(xr + d, v + ld4(pr + d));
    }
}
extern "C" int ssl4gie_decoder_assemble(const void* y, int y_dtype, const float* mask_token,
                                        const float* dpos, const long long* ids_restore,
                                        float* xd, int B, int L, int nkeep, int D, void* stream) {
    REQUIRE(y && mask_token && dpos && ids_restore && xd && B >= 0 && L > 0 && nkeep >= 0);
    REQUIRE(nkeep <= L && D > 0 && D % 4 == 0);
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(B * (L + 1)), block(D / 4 < 256 ? 64 * ((D / 4 + 63) / 64) : 256);
    if (y_dtype == SSL4GIE_BF16)
        hipLaunchKernelGGL(decoder_assemble_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)y,
                           mask_token, dpos, ids_restore, xd, L, nkeep, D);
    else if (y_dtype == SSL4GIE_F32)
        hipLaunchKernelGGL(decoder_assemble_kernel<float>, grid, block, 0, st, (const float*)y,
                           mask_token, dpos, ids_restore, xd, L, nkeep, D);
    else
        return ARG_ERR;
    LAUNCH_CHECK();
    return 0;
}

// dy rows (kept tokens) + partial sums of the removed rows (mask-token gradient).  Grid (B, DAB_Y): block (b, y)
// takes the token rows j = y, y + DAB_Y, ... of sample b — with one block per sample (round 1-4) a thread walked all
// 197 rows through dependent index loads: 109 us for 130 MB; the row sums of a block go to partial[b * DAB_Y + y]
// and the fixed-order second stage adds the B * DAB_Y rows (deterministic)
#define DAB_Y 8
template <typename T>
__global__ void decoder_assemble_bwd_kernel(const float* __restrict__ dxd,
                                            const long long* __restrict__ ids_shuffle,
                                            T* __restrict__ dy, float* __restrict__ partial,
                                            int L, int nkeep, int D) {
    const int b = blockIdx.x, y = blockIdx.y;
    const float* base = dxd + (size_t)b * (L + 1) * D;
    T* dyb = dy + (size_t)b * (nkeep + 1) * D;
    for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) {
        if (y == 0) st4(dyb + d, ld4(base + d));
        for (int j = y; j < nkeep; j += DAB_Y) {
            const int pos = (int)ids_shuffle[(size_t)b * L + j];
            st4(dyb + (size_t)(1 + j) * D + d, ld4(base + (size_t)(1 + pos) * D + d));
        }
        f32x4 acc = {0, 0, 0, 0};
        for (int j = nkeep + y; j < L; j += DAB_Y) {
            const int pos = (int)ids_shuffle[(size_t)b * L + j];
            acc += ld4(base + (size_t)(1 + pos) * D + d);
        }
        st4(partial + ((size_t)b * DAB_Y + y) * D + d, acc);
    }
}
extern "C" size_t ssl4gie_decoder_assemble_bwd_workspace_bytes(int B, int L, int D) {
    (void)L;
    return (size_t)B * DAB_Y * D * sizeof(float);
}
extern "C" int ssl4gie_decoder_assemble_bwd(const float* dxd, const long long* ids_shuffle,
                                            void* dy, int dy_dtype, float* dmask_token,
                                            int accumulate, float* workspace, int B, int L,
                                            int nkeep, int D, void* stream) {
    REQUIRE(dxd && ids_shuffle && dy && dmask_token && workspace && B >= 0 && L > 0);
    REQUIRE(nkeep >= 0 && nkeep <= L && D > 0 && D % 4 == 0);
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(B, DAB_Y), block(D / 4 < 256 ? 64 * ((D / 4 + 63) / 64) : 256);
    if (dy_dtype == SSL4GIE_BF16)
        hipLaunchKernelGGL(decoder_assemble_bwd_kernel<bf16_t>, grid, block, 0, st, dxd,
                           ids_shuffle, (bf16_t*)dy, workspace, L, nkeep, D);
    else if (dy_dtype == SSL4GIE_F32)
        hipLaunchKernelGGL(decoder_assemble_bwd_kernel<float>, grid, block, 0, st, dxd,
                           ids_shuffle, (float*)dy, workspace, L, nkeep, D);
    else
        return ARG_ERR;
    LAUNCH_CHECK();
    return ssl4gie_internal_reduce_partials(workspace, dmask_token, B * DAB_Y, D, (size_t)D, accumulate, st);
}

// ------------------------------------------------------------------ MAE loss (+ its gradient)
// one wave per patch; P = C*p*p target values gathered in 'nhwpqc' order.  A lane owns NV groups of
// 4 consecutive values (k = 4 lane + 256 i): pred / dpred move as 16-byte accesses, 1 KiB per wave
// instruction; the matching target pixels are 4 scalar reads of the image (served by L1: a patch
// row of one channel is 64 contiguous bytes shared by neighbouring lanes).
template <int NV>
__global__ __launch_bounds__(256) void mae_loss_kernel(
    const float* __restrict__ pred, const float* __restrict__ img, const float* __restrict__ mask,
    float* __restrict__ per_patch, float* __restrict__ dpred, const float* __restrict__ gpp,
    float gscale_host, int norm_pix, int has_cls, int B, int C, int H, int W, int p) {
    const int lane = threadIdx.x & 63;
    const int gw = W / p, L = (H / p) * gw, P = C * p * p, R = L + has_cls;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);  // over B*R
    if (row >= (long long)B * R) return;
    const int b = (int)(row / R), t = (int)(row % R);
    float* dr = dpred ? dpred + (size_t)row * P : nullptr;
    if (has_cls && t == 0) {  // cls row: no loss, zero gradient
        if (dr)
            for (int k = 4 * lane; k < P; k += 256) st4(dr + k, f32x4{0.f, 0.f, 0.f, 0.f});
        return;
    }
    const int l = t - has_cls, gy = l / gw, gx = l % gw;
    const float* ib = img + (size_t)b * C * H * W + (size_t)(gy * p) * W + gx * p;
    const float* pr = pred + (size_t)row * P;
    f32x4 tv[NV], pv[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k = 4 * lane + 256 * i;
        tv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        pv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (k < P) {
            pv[i] = ld4(pr + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kk = k + j, c = kk % C, pix = kk / C, py = pix / p, px = pix - py * p;
                tv[i][j] = ib[((size_t)c * H + py) * W + px];
            }
            s += (tv[i][0] + tv[i][1]) + (tv[i][2] + tv[i][3]);
        }
    }
    if (norm_pix) {
        const float mean = wave_sum(s) / (float)P;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (4 * lane + 256 * i < P) {
                const f32x4 d = tv[i] - mean;
                q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
        const float inv = rsqrtf(wave_sum(q) / (float)(P - 1) + 1.0e-6f);  // unbiased var
#pragma unroll
        for (int i = 0; i < NV; ++i) tv[i] = (tv[i] - mean) * inv;
    }
    const float m = mask[(size_t)b * L + l];
    float e = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (4 * lane + 256 * i < P) {
            const f32x4 d = pv[i] - tv[i];
            e += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    e = wave_sum(e) / (float)P;
    if (per_patch && lane == 0) per_patch[(size_t)b * L + l] = e * m;
    if (dr) {
        const float gs = gscale_host * (gpp ? gpp[(size_t)b * L + l] : 1.f) * m * 2.0f / (float)P;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (4 * lane + 256 * i < P) st4(dr + 4 * lane + 256 * i, (pv[i] - tv[i]) * gs);
    }
}
extern "C" int ssl4gie_mae_loss(const float* pred, const float* img, const float* mask,
                                float* per_patch, float* dpred, const float* gpp,
                                float gscale_host, int norm_pix, int has_cls, int B, int C, int H,
                                int W, int p, void* stream) {
    REQUIRE(pred && img && mask && B >= 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0);
    REQUIRE(has_cls == 0 || has_cls == 1);
    const int P = C * p * p;
    REQUIRE(P <= 64 * 16 && P % 4 == 0);
    if (B == 0) return 0;
    const long long rows = (long long)B * ((H / p) * (W / p) + has_cls);
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (P <= 64 * 4)
        hipLaunchKernelGGL(mae_loss_kernel<1>, grid, block, 0, st, pred, img, mask, per_patch, dpred,
                           gpp, gscale_host, norm_pix, has_cls, B, C, H, W, p);
    else if (P <= 64 * 12)
        hipLaunchKernelGGL(mae_loss_kernel<3>, grid, block, 0, st, pred, img, mask, per_patch,
                           dpred, gpp, gscale_host, norm_pix, has_cls, B, C, H, W, p);
    else
        hipLaunchKernelGGL(mae_loss_kernel<4>, grid, block, 0, st, pred, img, mask, per_patch,
                           dpred, gpp, gscale_host, norm_pix, has_cls, B, C, H, W, p);
    LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------ input pipeline
// transforms.ToTensor() + transforms.Normalize(mean, std) (Depth_estimation/Data/dataloaders.py:
// 55-63) on the device: uint8 HWC images [B, H, W, 3] -> fp32 NCHW [B, 3, H, W],
// out = (x / 255 - mean[c]) / std[c].  One thread per 4 pixels of a row (12 bytes in, 3 x 16 B out).
__global__ void normalize_u8_kernel(const unsigned char* __restrict__ img, float* __restrict__ out,
                                    f32x4 scale, f32x4 shift, int HW, long long total4) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total4) return;
    const long long b = idx / (HW / 4);
    const int p = (int)(idx % (HW / 4)) * 4;
    const unsigned* src = (const unsigned*)(img + ((size_t)b * HW + p) * 3);  // 12 bytes, 4-B aligned
    const unsigned w0 = src[0], w1 = src[1], w2 = src[2];
    unsigned char px[12];
#pragma unroll
    for (int j = 0; j < 4; ++j) { px[j] = (w0 >> (8 * j)) & 0xff; px[4 + j] = (w1 >> (8 * j)) & 0xff; px[8 + j] = (w2 >> (8 * j)) & 0xff; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (float)px[3 * j + c] * scale[c] + shift[c];
        st4(out + ((size_t)b * 3 + c) * HW + p, v);
    }
}
extern "C" int ssl4gie_normalize_u8(const unsigned char* img, float* out, const float* mean,
                                    const float* std, int B, int H, int W, void* stream) {
    REQUIRE(img && out && mean && std && B > 0 && H > 0 && W > 0 && ((long long)H * W) % 4 == 0);
    for (int c = 0; c < 3; ++c) REQUIRE(std[c] > 0.f);
    const int HW = H * W;
    const long long total4 = (long long)B * (HW / 4);
    f32x4 scale = {1.f / (255.f * std[0]), 1.f / (255.f * std[1]), 1.f / (255.f * std[2]), 0.f};
    f32x4 shift = {-mean[0] / std[0], -mean[1] / std[1], -mean[2] / std[2], 0.f};
    hipLaunchKernelGGL(normalize_u8_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, img, out, scale, shift, HW, total4);
    LAUNCH_CHECK();
    return 0;
}
