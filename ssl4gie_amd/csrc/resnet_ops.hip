// ResNet50 glue for gfx950: everything around the convolution GEMMs of
// torchvision's ResNet(Bottleneck, [3, 4, 6, 3]) as the reference uses it (Models/models.py:63-152,
// Models/moco_v3/main_moco.py:185-187).  Maps are channels-last [B, H, W, C] in the operand type, so
// a 1x1 convolution is a token-major GEMM and BatchNorm2d statistics are column sums over M = B*H*W
// rows.  All HBM-bound: 16-byte accesses, two-stage deterministic reductions (no atomics).
//
//   stem_im2col7x7   fp32 NCHW image -> patch matrix of conv1 (7x7, stride 2, pad 3), K = 147 (+pad)
//   subsample2       rows of a stride-2 1x1 convolution (downsample.0) and its gradient
//   bn_stats         per-channel sum / sum of squares partials  (+ finalize: mean, rstd)
//   bn_apply         y = act((x - mean) rstd gamma + beta (+ residual))     act = ReLU | id
//   bn_bwd_reduce    partials of sum(dy) and sum(dy * xhat) over the rows (dy already ReLU-masked)
//   bn_bwd_apply     dx = gamma rstd (g - mean(g) - xhat mean(g xhat)); optional dres = g
//   maxpool3x3s2     forward with argmax byte, backward in gather form
//   avgpool          global average per image and its gradient
#include "common.h"
#include "prof.h"
#include "ssl4gie_hip.h"
#include "internal.h"

namespace {
template <typename T> struct V16;
template <> struct V16<bf16_t> { static constexpr int N = 8; typedef u32x4 raw; };
template <> struct V16<float> { static constexpr int N = 4; typedef f32x4 raw; };
template <typename T> DEVI void un(const typename V16<T>::raw& r, float (&f)[V16<T>::N]);
template <> DEVI void un<bf16_t>(const u32x4& r, float (&f)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = __uint_as_float(r[j] << 16);
        f[2 * j + 1] = __uint_as_float(r[j] & 0xffff0000u);
    }
}
template <> DEVI void un<float>(const f32x4& r, float (&f)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = r[j];
}
template <typename T> DEVI typename V16<T>::raw pk(const float (&f)[V16<T>::N]);
template <> DEVI u32x4 pk<bf16_t>(const float (&f)[8]) {
    u32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = pack_bf2(f[2 * j], f[2 * j + 1]);
    return r;
}
template <> DEVI f32x4 pk<float>(const float (&f)[4]) { return f32x4{f[0], f[1], f[2], f[3]}; }
}  // namespace

#define RN_GRID(total) dim3((unsigned)(((total) + 255) / 256)), dim3(256)
#define RN_LAUNCH(dtype, KERNEL, total, ...)                                                       \
    do {                                                                                           \
        if ((dtype) == SSL4GIE_BF16) {                                                             \
            typedef bf16_t T;                                                                      \
            hipLaunchKernelGGL(KERNEL<T>, RN_GRID(total), 0, st, __VA_ARGS__);                     \
        } else {                                                                                   \
            typedef float T;                                                                       \
            hipLaunchKernelGGL(KERNEL<T>, RN_GRID(total), 0, st, __VA_ARGS__);                     \
        }                                                                                          \
        LAUNCH_CHECK();                                                                            \
    } while (0)

// ------------------------------------------------------------------ stem patch matrix
// cols[(b, oy, ox), (dy*7 + dx)*3 + c] = img[b, c, 2 oy + dy - 3, 2 ox + dx - 3]; [147, ld) zero
template <typename T>
__global__ void stem_im2col_kernel(const float* __restrict__ img, T* __restrict__ cols, int B, int H,
                                   int W, int Ho, int Wo, long long ld, long long total) {
    // one thread = 8 consecutive columns of one output row (ld % 8 == 0): a 16-B (bf16) store
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cpr = (int)(ld >> 3);
    const long long m = idx / cpr;
    const int col0 = (int)(idx % cpr) * 8;
    const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), b = (int)(m / ((long long)Wo * Ho));
    const float* base = img + (size_t)b * 3 * H * W;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int col = col0 + j;
        v[j] = 0.f;
        if (col < 147) {
            const int tap = col / 3, c = col - tap * 3, dy = tap / 7, dx = tap - dy * 7;
            const int iy = 2 * oy + dy - 3, ix = 2 * ox + dx - 3;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v[j] = base[((size_t)c * H + iy) * W + ix];
        }
    }
    T* dst = cols + (size_t)m * ld + col0;
#pragma unroll
    for (int j = 0; j < 8; j += 4) st4(dst + j, f32x4{v[j], v[j + 1], v[j + 2], v[j + 3]});
}

// ------------------------------------------------------------------ stride-2 row subsampling
template <typename T>
__global__ void subsample2_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C,
                                  int Ho, int Wo, long long total) {
    constexpr int V = V16<T>::N;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cpr = C / V;
    const long long p = idx / cpr;
    const int c = (int)(idx % cpr) * V;
    const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho);
    const long long b = p / ((long long)Wo * Ho);
    *(typename V16<T>::raw*)(y + (size_t)p * C + c) =
        *(const typename V16<T>::raw*)(x + ((b * H + 2 * oy) * W + 2 * ox) * C + c);
}
template <typename T>
__global__ void subsample2_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int H, int W,
                                      int C, int Ho, int Wo, long long total) {
    constexpr int V = V16<T>::N;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cpr = C / V;
    const long long p = idx / cpr;
    const int c = (int)(idx % cpr) * V;
    const int x = (int)(p % W), y = (int)((p / W) % H);
    const long long b = p / ((long long)W * H);
    typename V16<T>::raw v = {};
    if (!(x & 1) && !(y & 1))
        v = *(const typename V16<T>::raw*)(dy + ((b * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c);
    *(typename V16<T>::raw*)(dx + (size_t)p * C + c) = v;
}

// ------------------------------------------------------------------ BatchNorm (training mode)
// partial[blockIdx.y][0][c] = sum_r x[r, c], partial[..][1][c] = sum_r x^2 over this block's rows
// (sums are taken about the pivot x[0, c] — written to `pivot` — so that E[d^2] - E[d]^2 does not
// cancel catastrophically when |mean| >> std).
// Lane mapping (both reductions): a lane owns V = 16 B / sizeof(T) channels (one dwordx4 load).  With
// C / V >= 64 a wave spans 64 V channels of one row and gridDim.x strips cover C; with fewer
// (power-of-two) channel groups a wave folds 64 / (C / V) rows per step and lanes of equal channel
// group are combined by shuffles at the end.  A wave step therefore reads 1 KiB of contiguous
// memory, and four steps are in flight per lane (the pass is pure HBM streaming: ~4 KiB per wave
// outstanding x 16 waves per CU is what it takes to cover the latency).
struct BnLanes {
    int lpr, rpw, c, rsub;
};
template <int V>
DEVI BnLanes bn_lanes(int C, int lane) {
    BnLanes m;
    const int cg = C / V;
    m.lpr = (cg < 64 && !(cg & (cg - 1))) ? cg : 64;  // fold only powers of two
    m.rpw = 64 / m.lpr;
    m.c = (blockIdx.x * 64 + (lane % m.lpr)) * V;
    m.rsub = lane / m.lpr;
    return m;
}
// combine the folded rows of a wave, then the 4 waves of the block; result in wave 0, rsub == 0
template <int V>
DEVI void bn_block_reduce(float (&s)[V], float (&q)[V], float (*red)[2][64][V], int lpr, int lane,
                          int wave) {
    for (int o = lpr; o < 64; o <<= 1) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            s[j] += __shfl_xor(s[j], o, 64);
            q[j] += __shfl_xor(q[j], o, 64);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < V; ++j) { red[wave - 1][0][lane][j] = s[j]; red[wave - 1][1][lane][j] = q[j]; }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int j = 0; j < V; ++j) { s[j] += red[w][0][lane][j]; q[j] += red[w][1][lane][j]; }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x,
                                                       float* __restrict__ partial,
                                                       float* __restrict__ pivot, long long rows,
                                                       int C) {
    constexpr int V = V16<T>::N;
    typedef typename V16<T>::raw raw_t;
    __shared__ float red[3][2][64][V];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const BnLanes m = bn_lanes<V>(C, lane);
    float s[V], q[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s[j] = q[j] = 0.f;
    if (m.c < C) {
        float piv[V];
        un<T>(*(const raw_t*)(x + m.c), piv);
        if (blockIdx.y == 0 && wave == 0 && m.rsub == 0) {
#pragma unroll
            for (int j = 0; j < V; ++j) pivot[m.c + j] = piv[j];
        }
        auto acc = [&](const raw_t& v) {
            float f[V];
            un<T>(v, f);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float d = f[j] - piv[j];
                s[j] += d;
                q[j] += d * d;
            }
        };
        const long long step = (long long)gridDim.y * 4 * m.rpw;
        long long r = ((long long)blockIdx.y * 4 + wave) * m.rpw + m.rsub;
        const T* px = x + m.c;
        for (; r + 3 * step < rows; r += 4 * step) {
            const raw_t v0 = *(const raw_t*)(px + (size_t)r * C);
            const raw_t v1 = *(const raw_t*)(px + (size_t)(r + step) * C);
            const raw_t v2 = *(const raw_t*)(px + (size_t)(r + 2 * step) * C);
            const raw_t v3 = *(const raw_t*)(px + (size_t)(r + 3 * step) * C);
            acc(v0); acc(v1); acc(v2); acc(v3);
        }
        for (; r < rows; r += step) acc(*(const raw_t*)(px + (size_t)r * C));
    }
    bn_block_reduce<V>(s, q, red, m.lpr, lane, wave);
    if (wave == 0 && m.c < C && m.rsub == 0) {
        float* p = partial + (size_t)blockIdx.y * 2 * C;
#pragma unroll
        for (int j = 0; j < V; j += 4) {
            st4(p + m.c + j, f32x4{s[j], s[j + 1], s[j + 2], s[j + 3]});
            st4(p + C + m.c + j, f32x4{q[j], q[j + 1], q[j + 2], q[j + 3]});
        }
    }
}
// sums [2][C] -> mean, rstd (biased variance, as F.batch_norm normalises with), and the running
// statistics update of nn.BatchNorm2d (momentum, unbiased variance) when running_* are given
__global__ void bn_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ pivot,
                                   float* __restrict__ mean, float* __restrict__ rstd,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float count, float eps, float momentum, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float d = sums[c] / count;
    float var = sums[C + c] / count - d * d;
    var = var > 0.f ? var : 0.f;
    const float m = pivot[c] + d;
    mean[c] = m;
    rstd[c] = rsqrtf(var + eps);
    if (running_mean) {
        const float unbiased = count > 1.f ? var * count / (count - 1.f) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}
// per-channel affine of the normalisation: y = x * coef[c] + coef[C + c]
// the forward's normalisation coefficients y = x a + b; ONE definition, because the backward kernels that rebuild
// the ReLU mask from x (xmask) must reproduce the forward's sign decisions
DEVI void bn_affine(float gamma, float beta, float mean, float rstd, float& a, float& b) {
    a = rstd * gamma;
    b = beta - mean * a;
}
__global__ void bn_fwd_coef_kernel(const float* __restrict__ mean, const float* __restrict__ rstd,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ coef, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float a, b;
    bn_affine(gamma ? gamma[c] : 1.f, beta ? beta[c] : 0.f, mean[c], rstd[c], a, b);
    coef[c] = a;
    coef[C + c] = b;
}
// dx = coef[c] g + coef[C+c] x + coef[2C+c]   (expansion of gamma rstd (g - s1/n - xhat s2/n))
__global__ void bn_bwd_coef_kernel(const float* __restrict__ mean, const float* __restrict__ rstd,
                                   const float* __restrict__ gamma, const float* __restrict__ sums,
                                   float inv_n, float* __restrict__ coef, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float r = rstd[c], a = r * (gamma ? gamma[c] : 1.f);
    const float s1 = sums[c] * inv_n, s2 = sums[C + c] * inv_n;
    coef[c] = a;
    coef[C + c] = -a * r * s2;
    coef[2 * C + c] = a * (mean[c] * r * s2 - s1);
}
// Tails of the single-GPU BatchNorm: the partition partials are summed by 4 waves x 64 channels per
// block (a fixed order: deterministic), and the per-channel results are finished in the same
// launch — forward: mean / rstd / running statistics / normalisation coefficients; backward:
// dx coefficients and dgamma / dbeta.  One launch instead of three (four) ~5 us ones per layer.
DEVI bool bn_sum_partials(const float* __restrict__ partial, int parts, int C, float& s, float& q,
                          float (*red)[2][64]) {
    // 4 waves per block, or 16 (BN_TAIL_WIDE: a few hundred to two thousand partial rows summed in ONE launch,
    // without the 64-way fold in front — two ~5 us launches become one)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    s = 0.f; q = 0.f;
    if (c < C) {
        // eight rows' loads in flight, then the adds in row order (the sums are those of the plain loop; a
        // dependent load per row cost up to 64 L2 round trips in this kernel's one wave per 64 channels)
        int p = wave;
        for (; p + 7 * nw < parts; p += 8 * nw) {
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = partial[(size_t)(p + nw * u) * 2 * C + c];
                b[u] = partial[(size_t)(p + nw * u) * 2 * C + C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += a[u]; q += b[u]; }
        }
        for (; p < parts; p += nw) {
            s += partial[(size_t)p * 2 * C + c];
            q += partial[(size_t)p * 2 * C + C + c];
        }
    }
    if (wave > 0) { red[wave - 1][0][lane] = s; red[wave - 1][1][lane] = q; }
    __syncthreads();
    if (wave != 0 || c >= C) return false;
    for (int w = 0; w < nw - 1; ++w) { s += red[w][0][lane]; q += red[w][1][lane]; }
    return true;
}
__global__ __launch_bounds__(1024) void bn_fwd_tail_kernel(
    const float* __restrict__ partial, int parts, const float* __restrict__ pivot,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ mean,
    float* __restrict__ rstd, float* __restrict__ running_mean, float* __restrict__ running_var,
    float* __restrict__ coef, float count, float eps, float momentum, int C) {
    __shared__ float red[15][2][64];
    float s, q;
    if (!bn_sum_partials(partial, parts, C, s, q, red)) return;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const float d = s / count;
    float var = q / count - d * d;
    var = var > 0.f ? var : 0.f;
    const float m = (pivot ? pivot[c] : 0.f) + d, r = rsqrtf(var + eps);
    mean[c] = m;
    rstd[c] = r;
    if (running_mean) {
        const float unbiased = count > 1.f ? var * count / (count - 1.f) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
    float a, b;
    bn_affine(gamma ? gamma[c] : 1.f, beta ? beta[c] : 0.f, m, r, a, b);
    coef[c] = a;
    coef[C + c] = b;
}
__global__ __launch_bounds__(1024) void bn_bwd_tail_kernel(
    const float* __restrict__ partial, int parts, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, float inv_n,
    float* __restrict__ coef, float* __restrict__ dgamma, float* __restrict__ dbeta,
    int accumulate, int C, const float* __restrict__ beta, float* __restrict__ mcoef) {
    __shared__ float red[15][2][64];
    float s, q;
    if (!bn_sum_partials(partial, parts, C, s, q, red)) return;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    if (mcoef) {  // the forward's y = x a + b, for the apply pass's ReLU mask
        float ma, mb;
        bn_affine(gamma ? gamma[c] : 1.f, beta ? beta[c] : 0.f, mean[c], rstd[c], ma, mb);
        mcoef[c] = ma;
        mcoef[C + c] = mb;
    }
    const float r = rstd[c], a = r * (gamma ? gamma[c] : 1.f);
    const float s1 = s * inv_n, s2 = q * inv_n;
    coef[c] = a;
    coef[C + c] = -a * r * s2;
    coef[2 * C + c] = a * (mean[c] * r * s2 - s1);
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + s : s;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + q : q;
}
// y = act(x coef[c] + coef[C + c] (+ res));  bits (optional, bf16 only: 8 elements per thread = one byte):
// bit j of bits[idx / 8] = (y[idx + j] > 0) — the ReLU mask the backward needs, at 1/16 of y's bytes
template <typename T>
__global__ void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ coef,
                                const T* __restrict__ res, T* __restrict__ y, int relu, int C,
                                long long total, unsigned char* __restrict__ bits) {
    constexpr int V = V16<T>::N;
    const long long idx = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    float f[V], r[V], a[V], b[V];
    un<T>(*(const typename V16<T>::raw*)(x + idx), f);
    if (res) un<T>(*(const typename V16<T>::raw*)(res + idx), r);
#pragma unroll
    for (int j = 0; j < V; j += 4) {
        *(f32x4*)(a + j) = ld4(coef + c + j);
        *(f32x4*)(b + j) = ld4(coef + C + c + j);
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
        float v = f[j] * a[j] + b[j];
        if (res) v += r[j];
        f[j] = (relu && v < 0.f) ? 0.f : v;
    }
    const typename V16<T>::raw out = pk<T>(f);
    *(typename V16<T>::raw*)(y + idx) = out;
    if constexpr (V == 8) {
        if (bits) {  // of the ROUNDED outputs: what `y > 0` reads back
            float r2[V];
            un<T>(out, r2);
            unsigned m = 0;
#pragma unroll
            for (int j = 0; j < V; ++j) m |= (r2[j] > 0.f ? 1u : 0u) << j;
            bits[idx >> 3] = (unsigned char)m;
        }
    }
}
// g = relu ? (y > 0 ? dy : 0) : dy;  partial[blk][0][c] = sum g, [1][c] = sum g * xhat; when the block
// had a residual input its gradient is g itself (dres, optional).  Lane mapping as bn_stats_kernel,
// two rows (x 2-3 streams) in flight per lane.
// XMASK (BatchNorm + ReLU without a residual input): the mask is rebuilt as x a + b > 0 from the forward's own
// coefficients instead of read from the ReLU output — two streams instead of three.
// XMASK == 2 (BITS): the mask comes from the forward's bit map (`y` points at it): one byte per 8 elements.
template <typename T, int XMASK>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dres,
    float* __restrict__ partial, int relu, long long rows, int C,
    const float* __restrict__ gamma, const float* __restrict__ beta) {
    constexpr int V = V16<T>::N;
    typedef typename V16<T>::raw raw_t;
    __shared__ float red[3][2][64][V];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const BnLanes m = bn_lanes<V>(C, lane);
    float s[V], q[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s[j] = q[j] = 0.f;
    if (m.c < C) {
        float mu[V], rs[V];
        [[maybe_unused]] float ma[V], mb[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { mu[j] = mean[m.c + j]; rs[j] = rstd[m.c + j]; }
        if constexpr (XMASK == 1) {
#pragma unroll
            for (int j = 0; j < V; ++j)
                bn_affine(gamma ? gamma[m.c + j] : 1.f, beta ? beta[m.c + j] : 0.f, mu[j], rs[j], ma[j], mb[j]);
        }
        [[maybe_unused]] const unsigned char* bits = (const unsigned char*)y;
        auto acc = [&](const raw_t& gv, const raw_t& yv, const raw_t& xv, size_t o) {
            float g[V], xx[V];
            un<T>(gv, g);
            un<T>(xv, xx);
            if constexpr (XMASK == 1) {
#pragma unroll
                for (int j = 0; j < V; ++j) g[j] = xx[j] * ma[j] + mb[j] > 0.f ? g[j] : 0.f;
            } else if constexpr (XMASK == 2) {
                const unsigned mk = (unsigned)yv[0];  // the byte, carried in the first word of `yv`
#pragma unroll
                for (int j = 0; j < V; ++j) g[j] = ((mk >> j) & 1u) ? g[j] : 0.f;
            } else if (relu) {
                float yy[V];
                un<T>(yv, yy);
#pragma unroll
                for (int j = 0; j < V; ++j) g[j] = yy[j] > 0.f ? g[j] : 0.f;
            }
            if (dres) *(raw_t*)(dres + o) = relu ? pk<T>(g) : gv;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                s[j] += g[j];
                q[j] += g[j] * ((xx[j] - mu[j]) * rs[j]);
            }
        };
        const long long step = (long long)gridDim.y * 4 * m.rpw;
        long long r = ((long long)blockIdx.y * 4 + wave) * m.rpw + m.rsub;
        constexpr int NR = XMASK ? 4 : 2;  // rows in flight per lane: ~8 16-B loads outstanding either way
        auto mask_of = [&](size_t o) -> raw_t {  // XMASK == 2: this row's byte of the bit map, in word 0
            raw_t r = {};
            if constexpr (XMASK == 2 && V == 8) r[0] = bits[o >> 3];
            return r;
        };
        for (; r + (NR - 1) * step < rows; r += NR * step) {
            size_t o[NR];
            raw_t gv[NR], xv[NR], yv[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                o[i] = (size_t)(r + i * step) * C + m.c;
                gv[i] = *(const raw_t*)(dy + o[i]);
                xv[i] = *(const raw_t*)(x + o[i]);
                yv[i] = gv[i];
                if (XMASK == 2) yv[i] = mask_of(o[i]);
                if (!XMASK && relu) yv[i] = *(const raw_t*)(y + o[i]);
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) acc(gv[i], yv[i], xv[i], o[i]);
        }
        for (; r < rows; r += step) {
            const size_t o0 = (size_t)r * C + m.c;
            const raw_t g0 = *(const raw_t*)(dy + o0), x0 = *(const raw_t*)(x + o0);
            raw_t y0 = g0;
            if (XMASK == 2) y0 = mask_of(o0);
            if (!XMASK && relu) y0 = *(const raw_t*)(y + o0);
            acc(g0, y0, x0, o0);
        }
    }
    bn_block_reduce<V>(s, q, red, m.lpr, lane, wave);
    if (wave == 0 && m.c < C && m.rsub == 0) {
        float* p = partial + (size_t)blockIdx.y * 2 * C;
#pragma unroll
        for (int j = 0; j < V; j += 4) {
            st4(p + m.c + j, f32x4{s[j], s[j + 1], s[j + 2], s[j + 3]});
            st4(p + C + m.c + j, f32x4{q[j], q[j + 1], q[j + 2], q[j + 3]});
        }
    }
}
// dx = coef[c] g + coef[C + c] x + coef[2C + c];  mcoef (optional, [2][C]): g = x mcoef[c] + mcoef[C + c] > 0 ? dy : 0
template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                    const T* __restrict__ x, const float* __restrict__ coef,
                                    T* __restrict__ dx, int relu, int C, long long total,
                                    const float* __restrict__ mcoef) {
    constexpr int V = V16<T>::N;
    const long long idx = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    float g[V], xv[V], yy[V], a[V], b[V], d[V];
    un<T>(*(const typename V16<T>::raw*)(dy + idx), g);
    un<T>(*(const typename V16<T>::raw*)(x + idx), xv);
    if (mcoef) {  // the forward's pre-activation, sign only
#pragma unroll
        for (int j = 0; j < V; j += 4) {
            const f32x4 ma = ld4(mcoef + c + j), mb = ld4(mcoef + C + c + j);
#pragma unroll
            for (int i = 0; i < 4; ++i) yy[j + i] = xv[j + i] * ma[i] + mb[i];
        }
    } else if (relu) {
        un<T>(*(const typename V16<T>::raw*)(y + idx), yy);
    }
#pragma unroll
    for (int j = 0; j < V; j += 4) {
        *(f32x4*)(a + j) = ld4(coef + c + j);
        *(f32x4*)(b + j) = ld4(coef + C + c + j);
        *(f32x4*)(d + j) = ld4(coef + 2 * C + c + j);
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const float gg = (relu && !(yy[j] > 0.f)) ? 0.f : g[j];
        g[j] = a[j] * gg + b[j] * xv[j] + d[j];
    }
    *(typename V16<T>::raw*)(dx + idx) = pk<T>(g);
}

// ------------------------------------------------------------------ MaxPool2d(3, 2, 1)
// arg = window position (dy*3+dx) of the first maximum in row-major scan order (ATen's choice)
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                   unsigned char* __restrict__ arg, int H, int W, int C, int Ho,
                                   int Wo, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long p = idx / C;
    const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho);
    const long long b = p / ((long long)Wo * Ho);
    float best = -INFINITY;
    int bi = 0;
    bool found = false;
    for (int dy = 0; dy < 3; ++dy) {
        const int iy = 2 * oy + dy - 1;
        if (iy < 0 || iy >= H) continue;
        for (int dx = 0; dx < 3; ++dx) {
            const int ix = 2 * ox + dx - 1;
            if (ix < 0 || ix >= W) continue;
            const float v = Elem<T>::ld(x + ((b * H + iy) * W + ix) * C + c);
            if (!found || v > best) {  // first maximum in scan order
                best = v;
                bi = dy * 3 + dx;
                found = true;
            }
        }
    }
    Elem<T>::st(y + idx, best);
    arg[idx] = (unsigned char)bi;
}
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ arg,
                                   T* __restrict__ dx, int H, int W, int C, int Ho, int Wo,
                                   long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long p = idx / C;
    const int x = (int)(p % W), y = (int)((p / W) % H);
    const long long b = p / ((long long)W * H);
    float acc = 0.f;
    for (int dy_ = 0; dy_ < 3; ++dy_) {
        const int ny = y + 1 - dy_;
        if (ny < 0 || (ny & 1)) continue;
        const int oy = ny >> 1;
        if (oy >= Ho) continue;
        for (int dx_ = 0; dx_ < 3; ++dx_) {
            const int nx = x + 1 - dx_;
            if (nx < 0 || (nx & 1)) continue;
            const int ox = nx >> 1;
            if (ox >= Wo) continue;
            const size_t o = (((size_t)b * Ho + oy) * Wo + ox) * C + c;
            if (arg[o] == dy_ * 3 + dx_) acc += Elem<T>::ld(dy + o);
        }
    }
    Elem<T>::st(dx + idx, acc);
}

// the same with a thread owning V = 16 B / sizeof(T) channels of a pixel (C % V == 0): 16-byte
// loads / stores, V argmax bytes per store
// coef (optional, [2][C]): the window runs over act(x coef[0][c] + coef[1][c]) rounded to T — the training-mode
// BatchNorm (+ ReLU) between the stem convolution and the pool, applied on the way in: its output map is never
// written or re-read (values and argmax equal the two-kernel form bit for bit: same arithmetic, same rounding
// before the comparisons, so the many ties at 0 behind the ReLU resolve the same way)
template <typename T>
__global__ void maxpool_fwd_vec_kernel(const T* __restrict__ x, T* __restrict__ y,
                                       unsigned char* __restrict__ arg, int H, int W, int C, int Ho,
                                       int Wo, long long total, const float* __restrict__ coef, int relu) {
    constexpr int V = V16<T>::N;
    typedef typename V16<T>::raw raw_t;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const MapIdx mi = map_index(idx, total, C / V, Wo, Ho);
    const int c = mi.cv * V, ox = mi.x, oy = mi.y;
    const long long p = mi.p, b = mi.b;
    float best[V];
    unsigned char bi[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { best[j] = -INFINITY; bi[j] = 0; }
    float ca[V], cb[V];
    if (coef) {
#pragma unroll
        for (int j = 0; j < V; j += 4) {
            *(f32x4*)(ca + j) = ld4(coef + c + j);
            *(f32x4*)(cb + j) = ld4(coef + C + c + j);
        }
    }
    bool found = false;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int iy = 2 * oy + dy - 1;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ix = 2 * ox + dx - 1;
            if (ix < 0 || ix >= W) continue;
            float v[V];
            un<T>(*(const raw_t*)(x + ((b * H + iy) * W + ix) * C + c), v);
            if (coef) {  // as bn_apply_kernel, then through T's rounding
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float u = v[j] * ca[j] + cb[j];
                    v[j] = (relu && u < 0.f) ? 0.f : u;
                }
                un<T>(pk<T>(v), v);
            }
#pragma unroll
            for (int j = 0; j < V; ++j) {
                if (!found || v[j] > best[j]) {  // first maximum in scan order
                    best[j] = v[j];
                    bi[j] = (unsigned char)(dy * 3 + dx);
                }
            }
            found = true;
        }
    }
    *(raw_t*)(y + (size_t)p * C + c) = pk<T>(best);
#pragma unroll
    for (int j = 0; j < V; j += 4)
        *(unsigned*)(arg + (size_t)p * C + c + j) =
            bi[j] | (bi[j + 1] << 8) | (bi[j + 2] << 16) | ((unsigned)bi[j + 3] << 24);
}
template <typename T>
__global__ void maxpool_bwd_vec_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ arg,
                                       T* __restrict__ dx, int H, int W, int C, int Ho, int Wo,
                                       long long total) {
    constexpr int V = V16<T>::N;
    typedef typename V16<T>::raw raw_t;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const MapIdx mi = map_index(idx, total, C / V, W, H);
    const int c = mi.cv * V, x = mi.x, y = mi.y;
    const long long p = mi.p, b = mi.b;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
    for (int dy_ = 0; dy_ < 3; ++dy_) {
        const int ny = y + 1 - dy_;
        if (ny < 0 || (ny & 1)) continue;
        const int oy = ny >> 1;
        if (oy >= Ho) continue;
#pragma unroll
        for (int dx_ = 0; dx_ < 3; ++dx_) {
            const int nx = x + 1 - dx_;
            if (nx < 0 || (nx & 1)) continue;
            const int ox = nx >> 1;
            if (ox >= Wo) continue;
            const size_t o = (((size_t)b * Ho + oy) * Wo + ox) * C + c;
            float g[V];
            un<T>(*(const raw_t*)(dy + o), g);
#pragma unroll
            for (int j = 0; j < V; j += 4) {
                const unsigned a = *(const unsigned*)(arg + o + j);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (((a >> (8 * k)) & 0xff) == (unsigned)(dy_ * 3 + dx_)) acc[j + k] += g[j + k];
            }
        }
    }
    *(raw_t*)(dx + (size_t)p * C + c) = pk<T>(acc);
}

// ------------------------------------------------------------------ global average pool
// y[b, c] = mean over the HW rows of image b (fp32 out); one block per (image, 256-channel strip)
template <typename T>
__global__ void avgpool_fwd_kernel(const T* __restrict__ x, float* __restrict__ y, int HW, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (c >= C) return;
    float s = 0.f;
    for (int r = 0; r < HW; ++r) s += Elem<T>::ld(x + ((size_t)b * HW + r) * C + c);
    y[(size_t)b * C + c] = s / (float)HW;
}
template <typename T>
__global__ void avgpool_bwd_kernel(const float* __restrict__ dy, T* __restrict__ dx, int HW, int C,
                                   long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long b = idx / ((long long)HW * C);
    Elem<T>::st(dx + idx, dy[b * C + c] / (float)HW);
}

// =====================================================================================
// C ABI
// =====================================================================================
static bool rdt(int dt) { return dt == SSL4GIE_F32 || dt == SSL4GIE_BF16; }
static int rvn(int dt) { return dt == SSL4GIE_BF16 ? 8 : 4; }
// grid of the two BatchNorm reductions: strips of 64 lanes x V channels over C (y: row partitions).
// ~64K elements per workgroup, at most 1024 partitions (the partials are parts x 2C floats)
static int bn_strips(int C, int dtype) {
    const int v = rvn(dtype);
    return (C / v + 63) / 64;
}
static int bn_parts(long long rows, int C) {
    long long p = ((rows * C) >> 16) / ((C + 511) / 512);
    return (int)(p < 1 ? 1 : (p > 1024 ? 1024 : p));
}

extern "C" int ssl4gie_stem_im2col7x7(const float* img, void* cols, int dtype, int B, int H, int W,
                                      long long ld, void* stream) {
    REQUIRE(img && cols && rdt(dtype) && B > 0 && H > 0 && W > 0 && ld >= 147 && ld % 8 == 0);
    hipStream_t st = (hipStream_t)stream;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)B * Ho * Wo * (ld / 8);
    RN_LAUNCH(dtype, stem_im2col_kernel, total, img, (T*)cols, B, H, W, Ho, Wo, ld, total);
    return 0;
}
extern "C" int ssl4gie_subsample2(const void* x, void* y, int dtype, int B, int H, int W, int C,
                                  int backward, void* stream) {
    REQUIRE(x && y && rdt(dtype) && B > 0 && H > 0 && W > 0 && C > 0 && C % rvn(dtype) == 0);
    hipStream_t st = (hipStream_t)stream;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if (!backward) {
        const long long total = (long long)B * Ho * Wo * (C / rvn(dtype));
        RN_LAUNCH(dtype, subsample2_kernel, total, (const T*)x, (T*)y, H, W, C, Ho, Wo, total);
    } else {  // x = dy [B, Ho, Wo, C], y = dx [B, H, W, C]
        const long long total = (long long)B * H * W * (C / rvn(dtype));
        RN_LAUNCH(dtype, subsample2_bwd_kernel, total, (const T*)x, (T*)y, H, W, C, Ho, Wo, total);
    }
    return 0;
}
// ---- statistics that arrive as per-128-row partials from the producing GEMM's epilogue
// (ssl4gie_gemm_desc::colstats: [parts][2][C], sums about 0): many partials (12 544 for the stem
// map of a 512-image batch) are first folded to 64 by a wide grid, then finished by the tail kernel
#define BN_FOLD 64
__global__ __launch_bounds__(256) void bn_fold_partials_kernel(const float* __restrict__ pin, int parts,
                                                               float* __restrict__ pout, int C) {
    __shared__ float red[3][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, k = blockIdx.y;
    float s = 0.f, q = 0.f;
    if (c < C) {
        int p = k + BN_FOLD * wave;  // eight rows' loads in flight, adds in row order (as bn_sum_partials)
        for (; p + 7 * BN_FOLD * 4 < parts; p += 8 * BN_FOLD * 4) {
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = pin[(size_t)(p + u * BN_FOLD * 4) * 2 * C + c];
                b[u] = pin[(size_t)(p + u * BN_FOLD * 4) * 2 * C + C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += a[u]; q += b[u]; }
        }
        for (; p < parts; p += BN_FOLD * 4) {
            s += pin[(size_t)p * 2 * C + c];
            q += pin[(size_t)p * 2 * C + C + c];
        }
    }
    if (wave > 0) { red[wave - 1][0][lane] = s; red[wave - 1][1][lane] = q; }
    __syncthreads();
    if (wave == 0 && c < C) {
#pragma unroll
        for (int w = 0; w < 3; ++w) { s += red[w][0][lane]; q += red[w][1][lane]; }
        pout[(size_t)k * 2 * C + c] = s;
        pout[(size_t)k * 2 * C + C + c] = q;
    }
}
// -> (partials to hand to the tail kernels, their count); `scratch` holds BN_FOLD x 2C floats
#define BN_TAIL_WIDE 2048
// threads of the tail kernel for `np` partial rows (see bn_sum_partials)
static dim3 bn_tail_block(int np) { return dim3(np > 4 * BN_FOLD ? 1024 : 256); }
static int bn_fold(const float* partial, int parts, float* scratch, int C, hipStream_t st,
                   const float** out, int* nout) {
    if (parts <= BN_TAIL_WIDE) { *out = partial; *nout = parts; return 0; }
    hipLaunchKernelGGL(bn_fold_partials_kernel, dim3((C + 63) / 64, BN_FOLD), dim3(256), 0, st, partial,
                       parts, scratch, C);
    LAUNCH_CHECK();
    *out = scratch; *nout = BN_FOLD;
    return 0;
}
extern "C" size_t ssl4gie_bn_workspace_bytes(long long rows, int C) {
    size_t parts = (size_t)bn_parts(rows, C);
    if (parts < 64) parts = 64;  // the partials paths fold into 64 x 2C floats of this workspace
    // coef 3C, partials parts x 2C, sums 2C, pivot C, fold scratch 64 x 2C (backward tail)
    return ((parts + 1) * 2 + 1 + 3 + 2 * BN_FOLD) * C * sizeof(float);
}
// forward: statistics over the rows of x [rows, C] (biased variance), optional running-stat update,
// y = act(xhat gamma + beta (+ res)); mean / rstd [C] are kept for backward
extern "C" int ssl4gie_bn_fwd(const void* x, const float* gamma, const float* beta, const void* res,
                              void* y, float* mean, float* rstd, float* running_mean,
                              float* running_var, float momentum, float eps, int relu,
                              int training, float* workspace, int dtype, long long rows, int C,
                              void* stream) {
    REQUIRE(x && y && mean && rstd && workspace && rdt(dtype) && rows > 0 && C > 0 && C % 8 == 0);
    hipStream_t st = (hipStream_t)stream;
    float* coef = workspace;  // workspace: [coef 3C][partials parts x 2C][sums 2C][pivot C]
    const long long total = rows * C;
    // algorithmic bytes: (statistics pass: x) + apply: x (+ res) -> y
    ProfScope prof(PROF_BN, (double)total * (dtype == SSL4GIE_BF16 ? 2 : 4) * ((training ? 1 : 0) + 2 + (res ? 1 : 0)), st);
    if (!training) {  // evaluation: mean / rstd are INPUTS (running statistics prepared by the caller)
        hipLaunchKernelGGL(bn_fwd_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, st, mean, rstd,
                           gamma, beta, coef, C);
        LAUNCH_CHECK();
        RN_LAUNCH(dtype, bn_apply_kernel, total / rvn(dtype), (const T*)x, coef, (const T*)res, (T*)y,
                  relu, C, total, (unsigned char*)nullptr);
        return 0;
    }
    const int parts = bn_parts(rows, C);
    dim3 grid(bn_strips(C, dtype), parts), block(256);
    float* partial = workspace + 3 * (size_t)C;
    float* sums = partial + (size_t)parts * 2 * C;
    float* pivot = sums + 2 * C;
    if (dtype == SSL4GIE_BF16)
        hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, partial,
                           pivot, rows, C);
    else
        hipLaunchKernelGGL(bn_stats_kernel<float>, grid, block, 0, st, (const float*)x, partial,
                           pivot, rows, C);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_fwd_tail_kernel, dim3((C + 63) / 64), bn_tail_block(parts), 0, st, partial, parts, pivot,
                       gamma, beta, mean, rstd, running_mean, running_var, coef, (float)rows, eps,
                       momentum, C);
    LAUNCH_CHECK();
    RN_LAUNCH(dtype, bn_apply_kernel, total / rvn(dtype), (const T*)x, coef, (const T*)res, (T*)y, relu,
              C, total, (unsigned char*)nullptr);
    return 0;
}
// backward: dgamma / dbeta (overwritten or accumulated), dx, and (optional) the residual gradient.
// xmask: BatchNorm + ReLU without a residual input — the ReLU mask is rebuilt from x and the forward's
// coefficients (gamma, beta, mean, rstd) instead of read from the ReLU output: 5 instead of 7 tensor passes.
static int bn_bwd_impl(const void* dy, const void* y, const void* x, const float* gamma, const float* beta,
                       const float* mean, const float* rstd, void* dx, void* dres, float* dgamma,
                       float* dbeta, int accumulate, int relu, int xmask /* 0: mask from y, 1: from x, 2: y is the forward's bit map (bf16) */,
                       float* workspace, int dtype, long long rows, int C, hipStream_t st) {
    const int parts = bn_parts(rows, C);
    dim3 grid(bn_strips(C, dtype), parts), block(256);
    // algorithmic bytes: reduce reads dy, x (+ the ReLU output when the mask comes from it; the bit map is 1/16 of
    // a tensor) and writes dres; apply reads the gradient, x (+ the ReLU output) and writes dx
    const double bn_es = dtype == SSL4GIE_BF16 ? 2 : 4;
    const bool bn_masked = relu && dres;
    ProfScope prof(PROF_BN, (double)rows * C * bn_es * ((2 + ((relu && xmask == 0) ? 1 : 0) + (xmask == 2 ? 0.0625 : 0) + (dres ? 1 : 0)) +
                                                       (2 + ((relu && xmask == 0 && !bn_masked) ? 1 : 0) + 1)), st);
    float* coef = workspace;
    float* partial = workspace + 3 * (size_t)C;
    float* mcoef = xmask == 1 ? partial + (size_t)parts * 2 * C : nullptr;  // the forward's `sums` slot: free here
#define BN_REDUCE(T_, XM_)                                                                              \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T_, XM_>), grid, block, 0, st, (const T_*)dy, (const T_*)y, \
                       (const T_*)x, mean, rstd, (T_*)dres, partial, relu, rows, C, gamma, beta)
    if (dtype == SSL4GIE_BF16) {
        if (xmask == 1) BN_REDUCE(bf16_t, 1); else if (xmask == 2) BN_REDUCE(bf16_t, 2); else BN_REDUCE(bf16_t, 0);
    } else {
        if (xmask == 1) BN_REDUCE(float, 1); else BN_REDUCE(float, 0);
    }
#undef BN_REDUCE
    LAUNCH_CHECK();
    // dbeta = sum g, dgamma = sum g xhat, and the dx coefficients, in one launch (after a 64-way
    // fold when the reduction left hundreds of partials: the tail's blocks are few and sequential)
    const float* pp; int np;
    float* fold_scratch = partial + ((size_t)parts * 2 + 3) * C;
    int rc = bn_fold(partial, parts, fold_scratch, C, st, &pp, &np);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd_tail_kernel, dim3((C + 63) / 64), bn_tail_block(np), 0, st, pp, np, mean,
                       rstd, gamma, 1.0f / (float)rows, coef, dgamma, dbeta, accumulate, C, beta, mcoef);
    LAUNCH_CHECK();
    const long long total = rows * C;
    // with a residual branch the reduction pass has just written dres = the MASKED gradient (dy or 0, no
    // rounding): the apply pass reads that one tensor instead of dy and the ReLU output (6 -> 4 B/element
    // read on the widest BatchNorm of a bottleneck)
    const bool masked = relu && dres;
    const void* gsrc = masked ? dres : dy;
    const void* ysrc = masked ? nullptr : y;
    const int relu_apply = masked ? 0 : relu;
    RN_LAUNCH(dtype, bn_bwd_apply_kernel, total / rvn(dtype), (const T*)gsrc, (const T*)ysrc, (const T*)x,
              coef, (T*)dx, relu_apply, C, total, (const float*)mcoef);
    return 0;
}
extern "C" int ssl4gie_bn_bwd(const void* dy, const void* y, const void* x, const float* gamma,
                              const float* mean, const float* rstd, void* dx, void* dres,
                              float* dgamma, float* dbeta, int accumulate, int relu,
                              float* workspace, int dtype, long long rows, int C, void* stream) {
    REQUIRE(dy && x && mean && rstd && dx && workspace && rdt(dtype) && rows > 0 && C > 0 && C % 8 == 0);
    REQUIRE(!relu || y);
    return bn_bwd_impl(dy, y, x, gamma, nullptr, mean, rstd, dx, dres, dgamma, dbeta, accumulate, relu, 0,
                       workspace, dtype, rows, C, (hipStream_t)stream);
}
// BatchNorm (+ residual) + ReLU backward with the ReLU mask from the bit map ssl4gie_bn_fwd_partials_bits wrote
// (one byte per 8 elements) instead of the ReLU output: the reduction pass streams dy, x and 1/16 of a tensor
// (bf16 maps only; `dres` is required: the apply pass reads the masked gradient the reduction wrote)
extern "C" int ssl4gie_bn_bwd_bits(const void* dy, const unsigned char* relu_bits, const void* x,
                                   const float* gamma, const float* mean, const float* rstd, void* dx,
                                   void* dres, float* dgamma, float* dbeta, int accumulate, float* workspace,
                                   int dtype, long long rows, int C, void* stream) {
    REQUIRE(dy && relu_bits && x && mean && rstd && dx && dres && workspace && dtype == SSL4GIE_BF16 && rows > 0 &&
            C > 0 && C % 8 == 0);
    return bn_bwd_impl(dy, relu_bits, x, gamma, nullptr, mean, rstd, dx, dres, dgamma, dbeta, accumulate, 1, 2,
                       workspace, dtype, rows, C, (hipStream_t)stream);
}
extern "C" int ssl4gie_bn_bwd_xmask(const void* dy, const void* x, const float* gamma, const float* beta,
                                    const float* mean, const float* rstd, void* dx, float* dgamma,
                                    float* dbeta, int accumulate, float* workspace, int dtype,
                                    long long rows, int C, void* stream) {
    REQUIRE(dy && x && mean && rstd && dx && workspace && rdt(dtype) && rows > 0 && C > 0 && C % 8 == 0);
    return bn_bwd_impl(dy, nullptr, x, gamma, beta, mean, rstd, dx, nullptr, dgamma, dbeta, accumulate, 1, 1,
                       workspace, dtype, rows, C, (hipStream_t)stream);
}
extern "C" int ssl4gie_maxpool3x3s2_fwd(const void* x, void* y, unsigned char* arg, int dtype, int B,
                                        int H, int W, int C, void* stream) {
    REQUIRE(x && y && arg && rdt(dtype) && B > 0 && H > 0 && W > 0 && C > 0);
    hipStream_t st = (hipStream_t)stream;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if (C % rvn(dtype) == 0) {
        const long long total = (long long)B * Ho * Wo * (C / rvn(dtype));
        RN_LAUNCH(dtype, maxpool_fwd_vec_kernel, total, (const T*)x, (T*)y, arg, H, W, C, Ho, Wo, total,
                  (const float*)nullptr, 0);
        return 0;
    }
    const long long total = (long long)B * Ho * Wo * C;
    RN_LAUNCH(dtype, maxpool_fwd_kernel, total, (const T*)x, (T*)y, arg, H, W, C, Ho, Wo, total);
    return 0;
}
// MaxPool2d(3, 2, 1) over act(x coef[0][c] + coef[1][c]) (coef [2][C] from ssl4gie_bn_coef_partials): BatchNorm
// + ReLU + pool of the ResNet stem in one pass over the convolution's output
extern "C" int ssl4gie_bn_maxpool3x3s2_fwd(const void* x, const float* coef, int relu, void* y,
                                           unsigned char* arg, int dtype, int B, int H, int W, int C,
                                           void* stream) {
    REQUIRE(x && coef && y && arg && rdt(dtype) && B > 0 && H > 0 && W > 0 && C > 0 && C % rvn(dtype) == 0);
    hipStream_t st = (hipStream_t)stream;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)B * Ho * Wo * (C / rvn(dtype));
    RN_LAUNCH(dtype, maxpool_fwd_vec_kernel, total, (const T*)x, (T*)y, arg, H, W, C, Ho, Wo, total, coef, relu);
    return 0;
}
extern "C" int ssl4gie_maxpool3x3s2_bwd(const void* dy, const unsigned char* arg, void* dx, int dtype,
                                        int B, int H, int W, int C, void* stream) {
    REQUIRE(dy && dx && arg && rdt(dtype) && B > 0 && H > 0 && W > 0 && C > 0);
    hipStream_t st = (hipStream_t)stream;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if (C % rvn(dtype) == 0) {
        const long long total = (long long)B * H * W * (C / rvn(dtype));
        RN_LAUNCH(dtype, maxpool_bwd_vec_kernel, total, (const T*)dy, arg, (T*)dx, H, W, C, Ho, Wo, total);
        return 0;
    }
    const long long total = (long long)B * H * W * C;
    RN_LAUNCH(dtype, maxpool_bwd_kernel, total, (const T*)dy, arg, (T*)dx, H, W, C, Ho, Wo, total);
    return 0;
}
extern "C" int ssl4gie_avgpool_fwd(const void* x, float* y, int dtype, int B, int HW, int C,
                                   void* stream) {
    REQUIRE(x && y && rdt(dtype) && B > 0 && HW > 0 && C > 0);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((C + 255) / 256, B), block(256);
    if (dtype == SSL4GIE_BF16)
        hipLaunchKernelGGL(avgpool_fwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, y, HW, C);
    else
        hipLaunchKernelGGL(avgpool_fwd_kernel<float>, grid, block, 0, st, (const float*)x, y, HW, C);
    LAUNCH_CHECK();
    return 0;
}
extern "C" int ssl4gie_avgpool_bwd(const float* dy, void* dx, int dtype, int B, int HW, int C,
                                   void* stream) {
    REQUIRE(dy && dx && rdt(dtype) && B > 0 && HW > 0 && C > 0);
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)B * HW * C;
    RN_LAUNCH(dtype, avgpool_bwd_kernel, total, dy, (T*)dx, HW, C, total);
    return 0;
}

// ------------------------------------------------------------------ pieces for SyncBatchNorm / MoCo
// dst = dst * m + src * (1 - m): MoCo._update_momentum_encoder (moco/builder.py:57-61) over a whole
// arena slice (base and momentum encoders are laid out identically)
__global__ void ema_update_kernel(float* __restrict__ dst, const float* __restrict__ src, float m,
                                  long long n) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4 d = ld4(dst + i), v = ld4(src + i);
        st4(dst + i, d * m + v * (1.f - m));
    } else {
        for (long long j = i; j < n; ++j) dst[j] = dst[j] * m + src[j] * (1.f - m);
    }
}
extern "C" int ssl4gie_ema_update(float* dst, const float* src, float m, long long n, void* stream) {
    REQUIRE(dst && src && n >= 0);
    if (n == 0) return 0;
    hipLaunchKernelGGL(ema_update_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0,
                       (hipStream_t)stream, dst, src, m, n);
    LAUNCH_CHECK();
    return 0;
}
// SyncBatchNorm, forward half: LOCAL batch statistics only (mean, biased var [C]); the caller
// combines them across ranks (counts may differ) and then calls ssl4gie_bn_fwd(training = 0) with
// the global mean / rstd.
__global__ void bn_local_stats_kernel(const float* __restrict__ sums, const float* __restrict__ pivot,
                                      float* __restrict__ mean, float* __restrict__ var, float count,
                                      int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float d = sums[c] / count;
    const float v = sums[C + c] / count - d * d;
    mean[c] = (pivot ? pivot[c] : 0.f) + d;
    var[c] = v > 0.f ? v : 0.f;
}
extern "C" int ssl4gie_bn_stats(const void* x, float* mean, float* var, float* workspace, int dtype,
                                long long rows, int C, void* stream) {
    REQUIRE(x && mean && var && workspace && rdt(dtype) && rows > 0 && C > 0 && C % 8 == 0);
    hipStream_t st = (hipStream_t)stream;
    const int parts = bn_parts(rows, C);
    dim3 grid(bn_strips(C, dtype), parts), block(256);
    float* partial = workspace + 3 * (size_t)C;
    float* sums = partial + (size_t)parts * 2 * C;
    float* pivot = sums + 2 * C;
    if (dtype == SSL4GIE_BF16)
        hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, partial,
                           pivot, rows, C);
    else
        hipLaunchKernelGGL(bn_stats_kernel<float>, grid, block, 0, st, (const float*)x, partial,
                           pivot, rows, C);
    LAUNCH_CHECK();
    int rc = ssl4gie_internal_reduce_partials(partial, sums, parts, 2 * C, (size_t)2 * C, 0, st);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_local_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, pivot,
                       mean, var, (float)rows, C);
    LAUNCH_CHECK();
    return 0;
}
// training-mode forward with the statistics taken from `partial` [parts][2][C] instead of a pass
// over x; everything else as ssl4gie_bn_fwd.  workspace: ssl4gie_bn_workspace_bytes(rows, C).
static int bn_fwd_partials_impl(const void* x, const float* partial, int parts, const float* gamma,
                                const float* beta, const void* res, void* y, float* mean,
                                float* rstd, float* running_mean, float* running_var,
                                float momentum, float eps, int relu, float* workspace, int dtype,
                                long long rows, int C, void* stream, unsigned char* relu_bits);
extern "C" int ssl4gie_bn_fwd_partials(const void* x, const float* partial, int parts, const float* gamma,
                                       const float* beta, const void* res, void* y, float* mean,
                                       float* rstd, float* running_mean, float* running_var,
                                       float momentum, float eps, int relu, float* workspace, int dtype,
                                       long long rows, int C, void* stream) {
    return bn_fwd_partials_impl(x, partial, parts, gamma, beta, res, y, mean, rstd, running_mean, running_var,
                                momentum, eps, relu, workspace, dtype, rows, C, stream, nullptr);
}
// ... and relu_bits [rows * C / 8] bytes: bit j of byte i = (y[8 i + j] > 0), for ssl4gie_bn_bwd_bits (bf16 maps)
extern "C" int ssl4gie_bn_fwd_partials_bits(const void* x, const float* partial, int parts, const float* gamma,
                                            const float* beta, const void* res, void* y,
                                            unsigned char* relu_bits, float* mean, float* rstd,
                                            float* running_mean, float* running_var, float momentum, float eps,
                                            float* workspace, int dtype, long long rows, int C, void* stream) {
    REQUIRE(relu_bits && dtype == SSL4GIE_BF16);
    return bn_fwd_partials_impl(x, partial, parts, gamma, beta, res, y, mean, rstd, running_mean, running_var,
                                momentum, eps, 1, workspace, dtype, rows, C, stream, relu_bits);
}
static int bn_fwd_partials_impl(const void* x, const float* partial, int parts, const float* gamma,
                                const float* beta, const void* res, void* y, float* mean,
                                float* rstd, float* running_mean, float* running_var,
                                float momentum, float eps, int relu, float* workspace, int dtype,
                                long long rows, int C, void* stream, unsigned char* relu_bits) {
    REQUIRE(x && partial && parts > 0 && y && mean && rstd && workspace && rdt(dtype) && rows > 0 &&
            C > 0 && C % 8 == 0);
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_BN, (double)rows * C * (dtype == SSL4GIE_BF16 ? 2 : 4) * (2 + (res ? 1 : 0)), st);  // x (+ res) -> y
    float* coef = workspace;
    float* scratch = workspace + 3 * (size_t)C;  // >= BN_FOLD x 2C floats by construction of the size
    const float* pp; int np;
    int rc = bn_fold(partial, parts, scratch, C, st, &pp, &np);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_fwd_tail_kernel, dim3((C + 63) / 64), bn_tail_block(np), 0, st, pp, np,
                       (const float*)nullptr, gamma, beta, mean, rstd, running_mean, running_var, coef,
                       (float)rows, eps, momentum, C);
    LAUNCH_CHECK();
    const long long total = rows * C;
    RN_LAUNCH(dtype, bn_apply_kernel, total / rvn(dtype), (const T*)x, coef, (const T*)res, (T*)y, relu,
              C, total, relu_bits);
    return 0;
}
// the statistics half of ssl4gie_bn_fwd_partials alone: mean / rstd / running statistics and the normalisation
// coefficients coef [2][C] (y = x coef[0][c] + coef[1][c]) for a consumer that applies them itself — the
// SSL4GIE_EPI_AFFINE_AUX_RELU epilogue of the 1x1 convolution recomputed after its statistics-only product
extern "C" int ssl4gie_bn_coef_partials(const float* partial, int parts, const float* gamma, const float* beta,
                                        float* mean, float* rstd, float* running_mean, float* running_var,
                                        float momentum, float eps, float* coef, float* workspace,
                                        long long rows, int C, void* stream) {
    REQUIRE(partial && parts > 0 && mean && rstd && coef && workspace && rows > 0 && C > 0 && C % 8 == 0);
    hipStream_t st = (hipStream_t)stream;
    float* scratch = workspace + 3 * (size_t)C;
    const float* pp; int np;
    int rc = bn_fold(partial, parts, scratch, C, st, &pp, &np);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_fwd_tail_kernel, dim3((C + 63) / 64), bn_tail_block(np), 0, st, pp, np,
                       (const float*)nullptr, gamma, beta, mean, rstd, running_mean, running_var, coef,
                       (float)rows, eps, momentum, C);
    LAUNCH_CHECK();
    return 0;
}
// SyncBatchNorm's local (mean, biased var) from the same partials
extern "C" int ssl4gie_bn_stats_partials(const float* partial, int parts, float* mean, float* var,
                                         float* workspace, long long rows, int C, void* stream) {
    REQUIRE(partial && parts > 0 && mean && var && workspace && rows > 0 && C > 0 && C % 8 == 0);
    hipStream_t st = (hipStream_t)stream;
    float* scratch = workspace + 3 * (size_t)C;
    float* sums = scratch + (size_t)BN_FOLD * 2 * C;
    const float* pp; int np;
    int rc = bn_fold(partial, parts, scratch, C, st, &pp, &np);
    if (rc) return rc;
    rc = ssl4gie_internal_reduce_partials(pp, sums, np, 2 * C, (size_t)2 * C, 0, st);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_local_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums,
                       (const float*)nullptr, mean, var, (float)rows, C);
    LAUNCH_CHECK();
    return 0;
}
// SyncBatchNorm, backward halves: sums[0][c] = sum g, sums[1][c] = sum g xhat over the LOCAL rows
// (the caller all-reduces them), then dx with the GLOBAL sums and 1 / (global row count).
extern "C" int ssl4gie_bn_bwd_reduce(const void* dy, const void* y, const void* x, const float* mean,
                                     const float* rstd, void* dres, float* sums, int relu,
                                     float* workspace, int dtype, long long rows, int C,
                                     void* stream) {
    REQUIRE(dy && x && mean && rstd && sums && workspace && rdt(dtype) && rows > 0 && C > 0 && C % 8 == 0);
    REQUIRE(!relu || y);
    hipStream_t st = (hipStream_t)stream;
    const int parts = bn_parts(rows, C);
    dim3 grid(bn_strips(C, dtype), parts), block(256);
    float* partial = workspace + 3 * (size_t)C;
    if (dtype == SSL4GIE_BF16)
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16_t, 0>), grid, block, 0, st, (const bf16_t*)dy,
                           (const bf16_t*)y, (const bf16_t*)x, mean, rstd, (bf16_t*)dres, partial,
                           relu, rows, C, (const float*)nullptr, (const float*)nullptr);
    else
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, 0>), grid, block, 0, st, (const float*)dy,
                           (const float*)y, (const float*)x, mean, rstd, (float*)dres, partial, relu,
                           rows, C, (const float*)nullptr, (const float*)nullptr);
    LAUNCH_CHECK();
    return ssl4gie_internal_reduce_partials(partial, sums, parts, 2 * C, (size_t)2 * C, 0, st);
}
// ... and for BatchNorm + ReLU without a residual input, with the mask rebuilt from x and the forward's
// coefficients (as ssl4gie_bn_bwd_xmask; mean / rstd are the GLOBAL statistics the forward normalised with)
extern "C" int ssl4gie_bn_bwd_reduce_xmask(const void* dy, const void* x, const float* gamma, const float* beta,
                                           const float* mean, const float* rstd, float* sums,
                                           float* workspace, int dtype, long long rows, int C, void* stream) {
    REQUIRE(dy && x && mean && rstd && sums && workspace && rdt(dtype) && rows > 0 && C > 0 && C % 8 == 0);
    hipStream_t st = (hipStream_t)stream;
    const int parts = bn_parts(rows, C);
    dim3 grid(bn_strips(C, dtype), parts), block(256);
    float* partial = workspace + 3 * (size_t)C;
    if (dtype == SSL4GIE_BF16)
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16_t, 1>), grid, block, 0, st, (const bf16_t*)dy,
                           (const bf16_t*)nullptr, (const bf16_t*)x, mean, rstd, (bf16_t*)nullptr, partial, 1,
                           rows, C, gamma, beta);
    else
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, 1>), grid, block, 0, st, (const float*)dy,
                           (const float*)nullptr, (const float*)x, mean, rstd, (float*)nullptr, partial, 1, rows,
                           C, gamma, beta);
    LAUNCH_CHECK();
    return ssl4gie_internal_reduce_partials(partial, sums, parts, 2 * C, (size_t)2 * C, 0, st);
}
extern "C" int ssl4gie_bn_bwd_apply_xmask(const void* dy, const void* x, const float* gamma, const float* beta,
                                          const float* mean, const float* rstd, const float* sums,
                                          float inv_count, void* dx, float* workspace, int dtype,
                                          long long rows, int C, void* stream) {
    REQUIRE(dy && x && mean && rstd && sums && dx && workspace && rdt(dtype) && rows > 0 && C > 0 &&
            C % 8 == 0);
    hipStream_t st = (hipStream_t)stream;
    float* mcoef = workspace + 3 * (size_t)C;  // [2][C] behind the dx coefficients
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, st, mean, rstd, gamma,
                       sums, inv_count, workspace, C);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_fwd_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, st, mean, rstd, gamma, beta,
                       mcoef, C);
    LAUNCH_CHECK();
    const long long total = rows * C;
    RN_LAUNCH(dtype, bn_bwd_apply_kernel, total / rvn(dtype), (const T*)dy, (const T*)nullptr, (const T*)x,
              workspace, (T*)dx, 1, C, total, (const float*)mcoef);
    return 0;
}
extern "C" int ssl4gie_bn_bwd_apply(const void* dy, const void* y, const void* x, const float* gamma,
                                    const float* mean, const float* rstd, const float* sums,
                                    float inv_count, void* dx, int relu, float* workspace, int dtype,
                                    long long rows, int C, void* stream) {
    REQUIRE(dy && x && mean && rstd && sums && dx && workspace && rdt(dtype) && rows > 0 && C > 0 &&
            C % 8 == 0);
    REQUIRE(!relu || y);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, st, mean, rstd, gamma,
                       sums, inv_count, workspace, C);
    LAUNCH_CHECK();
    const long long total = rows * C;
    RN_LAUNCH(dtype, bn_bwd_apply_kernel, total / rvn(dtype), (const T*)dy, (const T*)y, (const T*)x,
              workspace, (T*)dx, relu, C, total, (const float*)nullptr);
    return 0;
}

// ---- SyncBatchNorm with the single-process fusions (round 6) ----------------------------------------------------
// With world_size > 1 the statistics cross an exchange between the producing GEMM's partials and the apply, so the
// fused forms take the GLOBAL (mean, rstd) the exchange returned instead of computing their own:
//   ssl4gie_bn_coef_stats      (mean, rstd, gamma, beta) -> coef [2][C] for a consumer that applies it itself: the
//                              bn1 -> relu -> maxpool pass of the stem (ssl4gie_bn_maxpool3x3s2_fwd) and the
//                              SSL4GIE_EPI_AFFINE_AUX_RELU epilogue of the momentum encoder's widening 1x1 products;
//   ssl4gie_bn_apply_bits      y = relu(x coef[0] + coef[1] (+ res)) and the ReLU bit map of ssl4gie_bn_fwd_partials_bits;
//   ssl4gie_bn_bwd_reduce_bits the first backward half of ssl4gie_bn_bwd_reduce with the mask from that bit map
//                              (dres = the masked gradient; the second half is ssl4gie_bn_bwd_apply on dres, relu 0).
// Reference: torch.nn.SyncBatchNorm via convert_sync_batchnorm, Models/moco_v3/main_moco.py:196,
// Depth_estimation/train_depth.py:225.
extern "C" int ssl4gie_bn_coef_stats(const float* mean, const float* rstd, const float* gamma, const float* beta,
                                     float* coef, int C, void* stream) {
    REQUIRE(mean && rstd && coef && C > 0);
    hipLaunchKernelGGL(bn_fwd_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, rstd,
                       gamma, beta, coef, C);
    LAUNCH_CHECK();
    return 0;
}
extern "C" int ssl4gie_bn_apply_bits(const void* x, const float* coef, const void* res, void* y,
                                     unsigned char* relu_bits, int dtype, long long rows, int C, void* stream) {
    REQUIRE(x && coef && y && relu_bits && dtype == SSL4GIE_BF16 && rows > 0 && C > 0 && C % 8 == 0);
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_BN, (double)rows * C * 2 * (2 + (res ? 1 : 0)), st);
    const long long total = rows * C;
    RN_LAUNCH(dtype, bn_apply_kernel, total / rvn(dtype), (const T*)x, coef, (const T*)res, (T*)y, 1, C, total,
              relu_bits);
    return 0;
}
extern "C" int ssl4gie_bn_bwd_reduce_bits(const void* dy, const unsigned char* relu_bits, const void* x,
                                          const float* mean, const float* rstd, void* dres, float* sums,
                                          float* workspace, int dtype, long long rows, int C, void* stream) {
    REQUIRE(dy && relu_bits && x && mean && rstd && dres && sums && workspace && dtype == SSL4GIE_BF16 && rows > 0 &&
            C > 0 && C % 8 == 0);
    hipStream_t st = (hipStream_t)stream;
    const int parts = bn_parts(rows, C);
    dim3 grid(bn_strips(C, dtype), parts), block(256);
    float* partial = workspace + 3 * (size_t)C;
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16_t, 2>), grid, block, 0, st, (const bf16_t*)dy,
                       (const bf16_t*)relu_bits, (const bf16_t*)x, mean, rstd, (bf16_t*)dres, partial, 1, rows, C,
                       (const float*)nullptr, (const float*)nullptr);
    LAUNCH_CHECK();
    return ssl4gie_internal_reduce_partials(partial, sums, parts, 2 * C, (size_t)2 * C, 0, st);
}

// ---- SyncBatchNorm: pooled statistics out of the gathered per-rank records ----------------------------
// gathered [W][2C + 1] = (mean_w[C], biased var_w[C], rows_w) of every rank (ranks may hold different
// row counts): total = sum rows_w, mean = sum rows_w/total mean_w, var = sum rows_w/total (var_w +
// (mean_w - mean)^2); rstd = rsqrt(var + eps); running statistics with the unbiased factor total/(total-1).
// One launch instead of ~10 torch kernels on [W, 2C+1] floats per layer; everything stays on the device.
__global__ void bn_combine_stats_kernel(const float* __restrict__ g, int W, int C, float eps, float momentum,
                                        float* __restrict__ running_mean, float* __restrict__ running_var,
                                        float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                        float* __restrict__ total_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int rec = 2 * C + 1;
    float total = 0.f;
    for (int w = 0; w < W; ++w) total += g[(size_t)w * rec + 2 * C];
    if (c == 0) *total_out = total;
    if (c >= C) return;
    const float inv = 1.f / total;
    float mean = 0.f;
    for (int w = 0; w < W; ++w) mean += g[(size_t)w * rec + c] * (g[(size_t)w * rec + 2 * C] * inv);
    float var = 0.f;
    for (int w = 0; w < W; ++w) {
        const float d = g[(size_t)w * rec + c] - mean;
        var += (g[(size_t)w * rec + C + c] + d * d) * (g[(size_t)w * rec + 2 * C] * inv);
    }
    mean_out[c] = mean;
    rstd_out[c] = rsqrtf(var + eps);
    if (running_mean) {
        const float unb = total / fmaxf(total - 1.f, 1.f);
        running_mean[c] = running_mean[c] * (1.f - momentum) + mean * momentum;
        running_var[c] = running_var[c] * (1.f - momentum) + var * unb * momentum;
    }
}

extern "C" int ssl4gie_bn_combine_stats(const float* gathered, int world, int C, float eps, float momentum,
                                        float* running_mean, float* running_var, float* mean, float* rstd,
                                        float* total, void* stream) {
    REQUIRE(gathered && mean && rstd && total && world >= 1 && C >= 1 && (!running_mean == !running_var));
    hipLaunchKernelGGL(bn_combine_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gathered,
                       world, C, eps, momentum, running_mean, running_var, mean, rstd, total);
    LAUNCH_CHECK();
    return 0;
}
