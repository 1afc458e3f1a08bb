// LayerNorm forward/backward and column reductions (bias gradients) for gfx950.
//
// HBM-bound row kernels: one 64-lane wave owns one row, 16-byte vector accesses, statistics in
// fp32 via wavefront shuffles (no LDS in the forward).  The residual stream is always fp32
// (reference autocast semantics); the normalised output feeds MFMA GEMMs and
// is written directly in the GEMM operand type (bf16 or f32) — the cast is fused here.
//
// Replaces: torch.nn.LayerNorm(eps=1e-6) at Models/mae/models_mae.py:227, Models/models.py:384,498
#include "common.h"
#include "prof.h"
#include "ssl4gie_hip.h"
#include "internal.h"

#define LN_ROWS_PER_BLOCK 4  // 4 waves / 256 threads
#define LN_MAX_VEC 8         // up to 64*4*8 = 2048 columns held in registers

template <typename TY, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta,
                                                     TY* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int rows, int cols,
                                                     float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * LN_ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * cols;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) {
            v[i] = ld4(xr + c);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        } else {
            v[i] = f32x4{0, 0, 0, 0};
        }
    }
    const float mu = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[i][j] - mu;
                q += d * d;
            }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)cols + eps);
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
    TY* yr = y + (size_t)row * cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) {
            const f32x4 g = ld4(gamma + c), b = ld4(beta + c);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mu) * rs * g[j] + b[j];
            st4(yr + c, o);
        }
    }
}

// Backward.  dx = dres + rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)).
// Column sums for dgamma/dbeta: every lane owns fixed columns, accumulates in registers over
// the rows its wave visits, the 4 waves of a block combine through LDS and write one partial
// row per block; ln_bwd_finalize reduces the partial rows (deterministic, no atomics).
template <typename TDY, typename TLP, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(
    const TDY* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ dres, float* __restrict__ dx, TLP* __restrict__ dx_lp,
    float* __restrict__ partial /*[gridDim.x][2][cols]*/, int rows, int cols) {
    __shared__ float red[3][2 * 64 * 4 * NV];  // waves 1..3 -> wave 0
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 g[NV], ag[NV], ab[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        g[i] = (c < cols) ? ld4(gamma + c) : f32x4{0, 0, 0, 0};
        ag[i] = f32x4{0, 0, 0, 0};
        ab[i] = f32x4{0, 0, 0, 0};
    }
    for (int row = blockIdx.x * LN_ROWS_PER_BLOCK + wave; row < rows;
         row += gridDim.x * LN_ROWS_PER_BLOCK) {
        const float mu = mean[row], rs = rstd[row];
        const size_t off = (size_t)row * cols;
        f32x4 d[NV], xh[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < cols) {
                d[i] = ld4(dy + off + c);
                const f32x4 xv = ld4(x + off + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh[i][j] = (xv[j] - mu) * rs;
                    ag[i][j] += d[i][j] * xh[i][j];
                    ab[i][j] += d[i][j];
                    const float gd = g[i][j] * d[i][j];
                    s1 += gd;
                    s2 += gd * xh[i][j];
                }
            }
        }
        s1 = wave_sum(s1) / (float)cols;
        s2 = wave_sum(s2) / (float)cols;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < cols) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = rs * (g[i][j] * d[i][j] - s1 - xh[i][j] * s2);
                if (dres) {
                    const f32x4 r = ld4(dres + off + c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] += r[j];
                }
                if (dx) st4(dx + off + c, o);
                if (dx_lp) st4(dx_lp + off + c, o);
            }
        }
    }
    // combine the 4 waves' column sums
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                red[wave - 1][(i * 4 + j) * 64 + lane] = ag[i][j];
                red[wave - 1][(NV * 4 + i * 4 + j) * 64 + lane] = ab[i][j];
            }
    }
    __syncthreads();
    if (wave == 0) {
        float* pg = partial + (size_t)blockIdx.x * 2 * cols;
        float* pb = pg + cols;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < cols) {
                f32x4 og = ag[i], ob = ab[i];
#pragma unroll
                for (int w = 0; w < 3; ++w)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        og[j] += red[w][(i * 4 + j) * 64 + lane];
                        ob[j] += red[w][(NV * 4 + i * 4 + j) * 64 + lane];
                    }
                st4(pg + c, og);
                st4(pb + c, ob);
            }
        }
    }
}

// out[c] (+)= sum_p partial[p*stride + c].  One 64-column strip per block, 16 waves: wave w sums
// parts w, w+16, ... (independent 256-B coalesced loads kept in flight), LDS combines the 16 partial
// strips.  Columns >= split go to out1 (LayerNorm: [dgamma | dbeta] in one launch).
#define RP_WAVES 16
__global__ __launch_bounds__(64 * RP_WAVES) void reduce_partials_kernel(
    const float* __restrict__ partial, float* __restrict__ out0, float* __restrict__ out1,
    int split, int nparts, int n_out, size_t stride, int accumulate) {
    __shared__ float red[RP_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < n_out) {
        int p = wave;
        // sixteen rows' loads in flight (24 blocks cover the 1536 LayerNorm columns: the pass is latency-, not
        // bandwidth-bound), added to the four accumulators in the order of the four-row loop below
        for (; p + 15 * RP_WAVES < nparts; p += 16 * RP_WAVES) {
            float a[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) a[u] = partial[(size_t)(p + u * RP_WAVES) * stride + c];
#pragma unroll
            for (int u = 0; u < 16; u += 4) { s0 += a[u]; s1 += a[u + 1]; s2 += a[u + 2]; s3 += a[u + 3]; }
        }
        for (; p + 3 * RP_WAVES < nparts; p += 4 * RP_WAVES) {
            s0 += partial[(size_t)p * stride + c];
            s1 += partial[(size_t)(p + RP_WAVES) * stride + c];
            s2 += partial[(size_t)(p + 2 * RP_WAVES) * stride + c];
            s3 += partial[(size_t)(p + 3 * RP_WAVES) * stride + c];
        }
        for (; p < nparts; p += RP_WAVES) s0 += partial[(size_t)p * stride + c];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && c < n_out) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < RP_WAVES; ++w) s += red[w][lane];
        float* o = (c < split) ? out0 + c : out1 + (c - split);
        *o = accumulate ? *o + s : s;
    }
}

// column sums of a [rows, cols] matrix (bias gradient): partial[blockIdx.y][cols]
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x,
                                                             float* __restrict__ partial,
                                                             int rows, int cols, size_t ld) {
    __shared__ f32x4 red[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 4;
    f32x4 acc = {0, 0, 0, 0};
    if (c < cols) {
        for (int r = blockIdx.y * 4 + wave; r < rows; r += gridDim.y * 4) {
            const f32x4 v = ld4(x + (size_t)r * ld + c);
            acc += v;
        }
    }
    if (wave > 0) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < cols) {
        acc += red[0][lane];
        acc += red[1][lane];
        acc += red[2][lane];
        st4(partial + (size_t)blockIdx.y * cols + c, acc);
    }
}

// tall and narrow (cols in {4, 8, .., 128}: the 32-channel map in front of the depth head has millions
// of rows): a wave covers 64 / lpr consecutive rows per load (lpr = cols / 4 lanes per row), so its
// loads stay 256-1024 B contiguous, and folds its row groups with shuffles before the LDS stage
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_narrow_kernel(const T* __restrict__ x,
                                                                    float* __restrict__ partial,
                                                                    int rows, int cols, size_t ld,
                                                                    int lpr) {
    __shared__ f32x4 red[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rpw = 64 / lpr, rl = lane / lpr, c = (lane % lpr) * 4;
    f32x4 acc = {0, 0, 0, 0};
    const int step = gridDim.y * 4 * rpw;
    int r = (blockIdx.y * 4 + wave) * rpw + rl;
    for (; r + 3 * step < rows; r += 4 * step) {  // four independent loads in flight
        const f32x4 v0 = ld4(x + (size_t)r * ld + c), v1 = ld4(x + (size_t)(r + step) * ld + c);
        const f32x4 v2 = ld4(x + (size_t)(r + 2 * step) * ld + c), v3 = ld4(x + (size_t)(r + 3 * step) * ld + c);
        acc += (v0 + v1) + (v2 + v3);
    }
    for (; r < rows; r += step) acc += ld4(x + (size_t)r * ld + c);
    for (int o = lpr; o < 64; o <<= 1)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += __shfl_xor(acc[i], o, 64);
    if (wave > 0) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave == 0 && lane < lpr) {
        acc += red[0][lane];
        acc += red[1][lane];
        acc += red[2][lane];
        st4(partial + (size_t)blockIdx.y * cols + c, acc);
    }
}

int ssl4gie_internal_reduce_partials(const float* partial, float* out, int nparts, int n_out,
                                     size_t stride, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((n_out + 63) / 64), dim3(64 * RP_WAVES), 0, st,
                       partial, out, out, n_out, nparts, n_out, stride, accumulate);
    LAUNCH_CHECK();
    return 0;
}

// any column count (e.g. the 6/12-class linear head): one column per thread
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_scalar_kernel(const T* __restrict__ x,
                                                                    float* __restrict__ partial,
                                                                    int rows, int cols, size_t ld) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float acc = 0.f;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) acc += Elem<T>::ld(x + (size_t)r * ld + c);
    partial[(size_t)blockIdx.y * cols + c] = acc;
}

static int ln_nv(int cols) { return (cols + 255) / 256; }
static int ln_bwd_blocks(int rows) {
    int b = (rows + LN_ROWS_PER_BLOCK - 1) / LN_ROWS_PER_BLOCK;
    return b < 1024 ? b : 1024;
}

extern "C" int ssl4gie_layernorm_fwd(const float* x, const float* gamma, const float* beta,
                                     void* y, int y_dtype, float* mean, float* rstd, int rows,
                                     int cols, float eps, void* stream) {
    REQUIRE(x && gamma && beta && y && rows >= 0 && cols > 0 && cols % 4 == 0);
    REQUIRE(cols <= 256 * LN_MAX_VEC);
    REQUIRE(y_dtype == SSL4GIE_F32 || y_dtype == SSL4GIE_BF16);
    if (rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_LN, (double)rows * cols * (4 + (y_dtype == SSL4GIE_BF16 ? 2 : 4)), st);  // x fp32 -> y
    dim3 grid((rows + LN_ROWS_PER_BLOCK - 1) / LN_ROWS_PER_BLOCK), block(256);
#define LN_FWD(NV)                                                                              \
    if (y_dtype == SSL4GIE_BF16)                                                                \
        hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, NV>), grid, block, 0, st, x, gamma, beta,     \
                           (bf16_t*)y, mean, rstd, rows, cols, eps);                            \
    else                                                                                        \
        hipLaunchKernelGGL((ln_fwd_kernel<float, NV>), grid, block, 0, st, x, gamma, beta,      \
                           (float*)y, mean, rstd, rows, cols, eps);
    switch (ln_nv(cols)) {
        case 1: LN_FWD(1) break;
        case 2: LN_FWD(2) break;
        case 3: LN_FWD(3) break;
        case 4: LN_FWD(4) break;
        default: LN_FWD(8) break;
    }
#undef LN_FWD
    LAUNCH_CHECK();
    return 0;
}

extern "C" size_t ssl4gie_layernorm_bwd_workspace_bytes(int rows, int cols) {
    return (size_t)ln_bwd_blocks(rows) * 2 * cols * sizeof(float);
}

extern "C" int ssl4gie_layernorm_bwd(const void* dy, int dy_dtype, const float* x,
                                     const float* gamma, const float* mean, const float* rstd,
                                     const float* dres, float* dx, void* dx_lp, int lp_dtype,
                                     float* dgamma, float* dbeta, int accumulate,
                                     float* workspace, int rows, int cols, void* stream) {
    REQUIRE(dy && x && gamma && mean && rstd && workspace && rows >= 0 && cols > 0);
    REQUIRE(cols % 4 == 0 && cols <= 256 * 4);
    REQUIRE(dy_dtype == SSL4GIE_F32 || dy_dtype == SSL4GIE_BF16);
    REQUIRE(!dx_lp || lp_dtype == dy_dtype);
    if (rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    // dy, x fp32 (+ the residual gradient fp32) -> dx fp32 (+ its operand-type copy)
    const double ln_es = dy_dtype == SSL4GIE_BF16 ? 2 : 4;
    ProfScope prof(PROF_LN, (double)rows * cols * (ln_es + 4 + (dres ? 4 : 0) + 4 + (dx_lp ? ln_es : 0)), st);
    const int nb = ln_bwd_blocks(rows);
    dim3 grid(nb), block(256);
#define LN_BWD(NV)                                                                             \
    if (dy_dtype == SSL4GIE_BF16)                                                              \
        hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, bf16_t, NV>), grid, block, 0, st,            \
                           (const bf16_t*)dy, x, gamma, mean, rstd, dres, dx, (bf16_t*)dx_lp,  \
                           workspace, rows, cols);                                             \
    else                                                                                       \
        hipLaunchKernelGGL((ln_bwd_kernel<float, float, NV>), grid, block, 0, st,              \
                           (const float*)dy, x, gamma, mean, rstd, dres, dx, (float*)dx_lp,    \
                           workspace, rows, cols);
    switch (ln_nv(cols)) {
        case 1: LN_BWD(1) break;
        case 2: LN_BWD(2) break;
        case 3: LN_BWD(3) break;
        default: LN_BWD(4) break;
    }
#undef LN_BWD
    LAUNCH_CHECK();
    if (dgamma && dbeta) {
        // partial rows are [dgamma | dbeta] of width 2*cols: one launch, two destinations
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((2 * cols + 63) / 64), dim3(64 * RP_WAVES),
                           0, st, workspace, dgamma, dbeta, cols, nb, 2 * cols, (size_t)2 * cols,
                           accumulate);
        LAUNCH_CHECK();
    }
    return 0;
}

int ssl4gie_internal_ln_reduce(const float* workspace, float* dgamma, float* dbeta, int rows, int cols,
                               int accumulate, hipStream_t st) {
    if (rows == 0 || !dgamma || !dbeta) return 0;
    const int nb = ln_bwd_blocks(rows);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((2 * cols + 63) / 64), dim3(64 * RP_WAVES), 0, st, workspace,
                       dgamma, dbeta, cols, nb, 2 * cols, (size_t)2 * cols, accumulate);
    LAUNCH_CHECK();
    return 0;
}

static bool colsum_narrow(int cols) { return cols <= 128 && cols % 4 == 0 && ((cols / 4) & (cols / 4 - 1)) == 0; }
static int colsum_parts(int rows, int cols) {
    const int cap = colsum_narrow(cols) ? 2048 : 256;  // narrow rows: partials are tiny, so more blocks
    int p = (rows + 63) / 64;
    return p < cap ? (p < 1 ? 1 : p) : cap;
}
extern "C" size_t ssl4gie_colsum_workspace_bytes(int rows, int cols) {
    return (size_t)colsum_parts(rows, cols) * cols * sizeof(float);
}
extern "C" int ssl4gie_colsum(const void* x, int dtype, float* out, int accumulate,
                              float* workspace, int rows, int cols, long long ld, void* stream) {
    REQUIRE(x && out && workspace && rows >= 0 && cols > 0 && ld >= cols);
    REQUIRE(dtype == SSL4GIE_F32 || dtype == SSL4GIE_BF16);
    hipStream_t st = (hipStream_t)stream;
    const int parts = colsum_parts(rows, cols);
    dim3 grid((cols + 255) / 256, parts), block(256);
    const bool vec = (cols % 4 == 0) && (ld % 4 == 0) && ((((uintptr_t)x) & 15) == 0);
    if (vec && colsum_narrow(cols)) {
        if (dtype == SSL4GIE_BF16)
            hipLaunchKernelGGL(colsum_partial_narrow_kernel<bf16_t>, dim3(1, parts), block, 0, st,
                               (const bf16_t*)x, workspace, rows, cols, (size_t)ld, cols / 4);
        else
            hipLaunchKernelGGL(colsum_partial_narrow_kernel<float>, dim3(1, parts), block, 0, st,
                               (const float*)x, workspace, rows, cols, (size_t)ld, cols / 4);
    } else if (dtype == SSL4GIE_BF16) {
        if (vec)
            hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x,
                               workspace, rows, cols, (size_t)ld);
        else
            hipLaunchKernelGGL(colsum_partial_scalar_kernel<bf16_t>, grid, block, 0, st,
                               (const bf16_t*)x, workspace, rows, cols, (size_t)ld);
    } else {
        if (vec)
            hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, block, 0, st, (const float*)x,
                               workspace, rows, cols, (size_t)ld);
        else
            hipLaunchKernelGGL(colsum_partial_scalar_kernel<float>, grid, block, 0, st,
                               (const float*)x, workspace, rows, cols, (size_t)ld);
    }
    LAUNCH_CHECK();
    return ssl4gie_internal_reduce_partials(workspace, out, parts, cols, (size_t)cols, accumulate, st);
}
