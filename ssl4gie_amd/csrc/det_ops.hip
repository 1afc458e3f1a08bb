// Channels-last glue of the detection backbone's feature pyramid (reference
// ViTDet_FPN, Models/models.py:213-259): MaxPool2d(2), nn.LayerNorm((C, H, W)) — a per-image
// normalisation over the WHOLE map with a per-element affine — and nn.GELU on a map.  The 1x1 / 3x3
// convolutions and ConvTranspose2d(2, 2) of the pyramid are the GEMM paths of gemm*.hip.
//
// All of it is HBM-bound streaming: 16-byte accesses, two-stage deterministic reductions.
#include "common.h"
#include "internal.h"
#include "ssl4gie_hip.h"

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
bool okdt(int dt) { return dt == SSL4GIE_F32 || dt == SSL4GIE_BF16; }
int vn(int dt) { return dt == SSL4GIE_BF16 ? 8 : 4; }
}  // namespace

#define DET_LAUNCH(dtype, KERNEL, total, ...)                                                       \
    do {                                                                                            \
        const dim3 grid_((unsigned)(((total) + 255) / 256)), block_(256);                           \
        if ((dtype) == SSL4GIE_BF16) {                                                              \
            typedef bf16_t T;                                                                       \
            hipLaunchKernelGGL(KERNEL<T>, grid_, block_, 0, st, __VA_ARGS__);                       \
        } else {                                                                                    \
            typedef float T;                                                                        \
            hipLaunchKernelGGL(KERNEL<T>, grid_, block_, 0, st, __VA_ARGS__);                       \
        }                                                                                           \
        LAUNCH_CHECK();                                                                             \
    } while (0)

// ------------------------------------------------------------------ MaxPool2d(kernel 2, stride 2)
// x [B, H, W, C] -> y [B, H/2, W/2, C]; backward routes dy to the FIRST maximum of the window in
// scan order (ATen's tie rule), recomputed from x (no index tensor)
template <typename T>
__global__ void maxpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C,
                                    long long total) {
    constexpr int V = V16<T>::N;
    typedef typename V16<T>::raw raw_t;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cpr = C / V, Ho = H / 2, Wo = W / 2;
    const int c = (int)(idx % cpr) * V;
    const long long p = idx / cpr;
    const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho);
    const long long b = p / ((long long)Wo * Ho);
    float best[V];
#pragma unroll
    for (int j = 0; j < V; ++j) best[j] = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float v[V];
        un<T>(*(const raw_t*)(x + ((b * H + 2 * oy + (t >> 1)) * W + 2 * ox + (t & 1)) * C + c), v);
#pragma unroll
        for (int j = 0; j < V; ++j) best[j] = v[j] > best[j] ? v[j] : best[j];
    }
    *(raw_t*)(y + (size_t)p * C + c) = pk<T>(best);
}
template <typename T>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                    T* __restrict__ dx, int H, int W, int C, long long total) {
    constexpr int V = V16<T>::N;
    typedef typename V16<T>::raw raw_t;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // one output window
    if (idx >= total) return;
    const int cpr = C / V, Ho = H / 2, Wo = W / 2;
    const int c = (int)(idx % cpr) * V;
    const long long p = idx / cpr;
    const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho);
    const long long b = p / ((long long)Wo * Ho);
    float v[4][V], g[V];
    un<T>(*(const raw_t*)(dy + (size_t)p * C + c), g);
#pragma unroll
    for (int t = 0; t < 4; ++t)
        un<T>(*(const raw_t*)(x + ((b * H + 2 * oy + (t >> 1)) * W + 2 * ox + (t & 1)) * C + c), v[t]);
    int win[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        win[j] = 0;
        float best = v[0][j];
#pragma unroll
        for (int t = 1; t < 4; ++t)
            if (v[t][j] > best) { best = v[t][j]; win[j] = t; }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float o[V];
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = win[j] == t ? g[j] : 0.f;
        *(raw_t*)(dx + ((b * H + 2 * oy + (t >> 1)) * W + 2 * ox + (t & 1)) * C + c) = pk<T>(o);
    }
}

// ------------------------------------------------------------------ GELU on a map
template <typename T>
__global__ void gelu_map_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out,
                                long long total) {
    constexpr int V = V16<T>::N;
    typedef typename V16<T>::raw raw_t;
    const long long idx = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (idx >= total) return;
    float f[V];
    un<T>(*(const raw_t*)(x + idx), f);
    if (dy) {
        float g[V];
        un<T>(*(const raw_t*)(dy + idx), g);
#pragma unroll
        for (int j = 0; j < V; ++j) f[j] = g[j] * dgelu_f(f[j]);
    } else {
#pragma unroll
        for (int j = 0; j < V; ++j) f[j] = gelu_f(f[j]);
    }
    *(raw_t*)(out + idx) = pk<T>(f);
}

// ------------------------------------------------------------------ LayerNorm over whole maps
// x [B, M] (M = H*W*C elements of one image, channels-last), weight / bias fp32 [M] in the SAME
// element order.  partial[(b * parts + p) * 2 + {0, 1}] = sum, sum of squares about the image's
// first element (pivot: no catastrophic cancellation when |mean| >> std).
#define MLN_PARTS 256
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void mapln_reduce_kernel(
    const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ w,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ partial,
    long long M) {
    constexpr int V = V16<T>::N;
    typedef typename V16<T>::raw raw_t;
    __shared__ float red[2][4];
    const int b = blockIdx.y, p = blockIdx.x;
    const T* xb = x + (size_t)b * M;
    float piv = 0.f, mu = 0.f, rs = 0.f;
    if (BWD) { mu = mean[b]; rs = rstd[b]; }
    else { float f[V]; un<T>(*(const raw_t*)xb, f); piv = f[0]; }
    float s = 0.f, q = 0.f;
    for (long long i = ((long long)p * 256 + threadIdx.x) * V; i < M; i += (long long)MLN_PARTS * 256 * V) {
        float f[V];
        un<T>(*(const raw_t*)(xb + i), f);
        if (BWD) {  // s = sum g, q = sum g xhat with g = dy * w
            float g[V];
            un<T>(*(const raw_t*)(dy + (size_t)b * M + i), g);
#pragma unroll
            for (int j = 0; j < V; j += 4) {
                const f32x4 ww = ld4(w + i + j);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float gg = g[j + k] * ww[k];
                    s += gg;
                    q += gg * ((f[j + k] - mu) * rs);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float d = f[j] - piv;
                s += d;
                q += d * d;
            }
        }
    }
    s = wave_sum(s);
    q = wave_sum(q);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = s; red[1][wave] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = partial + ((size_t)b * MLN_PARTS + p) * 2;
        o[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        o[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}
// one wave per image: fold the partials; forward: mean / rstd; backward: the two means of dx
template <typename T>
__global__ __launch_bounds__(64) void mapln_finalize_kernel(const float* __restrict__ partial,
                                                            const T* __restrict__ x, float* __restrict__ o0,
                                                            float* __restrict__ o1, long long M, float eps,
                                                            int bwd) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float s = 0.f, q = 0.f;
    for (int p = lane; p < MLN_PARTS; p += 64) {
        s += partial[((size_t)b * MLN_PARTS + p) * 2];
        q += partial[((size_t)b * MLN_PARTS + p) * 2 + 1];
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (lane != 0) return;
    const float n = (float)M;
    if (bwd) {
        o0[b] = s / n;  // mean(g)
        o1[b] = q / n;  // mean(g xhat)
    } else {
        constexpr int V = V16<T>::N;
        float f[V];
        un<T>(*(const typename V16<T>::raw*)(x + (size_t)b * M), f);
        const float d = s / n;
        float var = q / n - d * d;
        var = var > 0.f ? var : 0.f;
        o0[b] = f[0] + d;
        o1[b] = rsqrtf(var + eps);
    }
}
template <typename T>
__global__ void mapln_apply_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                   const float* __restrict__ bias, const float* __restrict__ mean,
                                   const float* __restrict__ rstd, T* __restrict__ y, long long M,
                                   long long total) {
    constexpr int V = V16<T>::N;
    typedef typename V16<T>::raw raw_t;
    const long long idx = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (idx >= total) return;
    const int b = (int)(idx / M);
    const long long i = idx - (long long)b * M;
    const float mu = mean[b], rs = rstd[b];
    float f[V];
    un<T>(*(const raw_t*)(x + idx), f);
#pragma unroll
    for (int j = 0; j < V; j += 4) {
        const f32x4 ww = ld4(w + i + j), bb = ld4(bias + i + j);
#pragma unroll
        for (int k = 0; k < 4; ++k) f[j + k] = (f[j + k] - mu) * rs * ww[k] + bb[k];
    }
    *(raw_t*)(y + idx) = pk<T>(f);
}
// a thread owns V elements of the map for ALL images: dx for each image and the per-element
// dweight = sum_b dy xhat, dbias = sum_b dy in one pass
template <typename T>
__global__ void mapln_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                       const float* __restrict__ w, const float* __restrict__ mean,
                                       const float* __restrict__ rstd, const float* __restrict__ mg,
                                       const float* __restrict__ mgx, T* __restrict__ dx,
                                       float* __restrict__ dw, float* __restrict__ db, int accumulate,
                                       int B, long long M) {
    constexpr int V = V16<T>::N;
    typedef typename V16<T>::raw raw_t;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (i >= M) return;
    float ww[V], aw[V], ab[V];
#pragma unroll
    for (int j = 0; j < V; j += 4) {
        const f32x4 t = ld4(w + i + j);
#pragma unroll
        for (int k = 0; k < 4; ++k) { ww[j + k] = t[k]; aw[j + k] = 0.f; ab[j + k] = 0.f; }
    }
    for (int b = 0; b < B; ++b) {
        const float mu = mean[b], rs = rstd[b], m1 = mg[b], m2 = mgx[b];
        float f[V], g[V], o[V];
        un<T>(*(const raw_t*)(x + (size_t)b * M + i), f);
        un<T>(*(const raw_t*)(dy + (size_t)b * M + i), g);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float xh = (f[j] - mu) * rs;
            aw[j] += g[j] * xh;
            ab[j] += g[j];
            o[j] = rs * (g[j] * ww[j] - m1 - xh * m2);
        }
        *(raw_t*)(dx + (size_t)b * M + i) = pk<T>(o);
    }
#pragma unroll
    for (int j = 0; j < V; j += 4) {
        f32x4 tw = {aw[j], aw[j + 1], aw[j + 2], aw[j + 3]}, tb = {ab[j], ab[j + 1], ab[j + 2], ab[j + 3]};
        if (dw) { if (accumulate) tw += ld4(dw + i + j); st4(dw + i + j, tw); }
        if (db) { if (accumulate) tb += ld4(db + i + j); st4(db + i + j, tb); }
    }
}

// =====================================================================================
extern "C" int ssl4gie_maxpool2x2_fwd(const void* x, void* y, int dtype, int B, int H, int W, int C,
                                      void* stream) {
    REQUIRE(x && y && okdt(dtype) && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 &&
            C % vn(dtype) == 0);
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / vn(dtype));
    DET_LAUNCH(dtype, maxpool2_fwd_kernel, total, (const T*)x, (T*)y, H, W, C, total);
    return 0;
}
extern "C" int ssl4gie_maxpool2x2_bwd(const void* x, const void* dy, void* dx, int dtype, int B, int H,
                                      int W, int C, void* stream) {
    REQUIRE(x && dy && dx && okdt(dtype) && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 &&
            C % vn(dtype) == 0);
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / vn(dtype));
    DET_LAUNCH(dtype, maxpool2_bwd_kernel, total, (const T*)x, (const T*)dy, (T*)dx, H, W, C, total);
    return 0;
}
// dy == NULL: out = gelu(x) (exact erf form, nn.GELU);  dy != NULL: out = dy * gelu'(x)
extern "C" int ssl4gie_gelu_map(const void* x, const void* dy, void* out, int dtype, long long n,
                                void* stream) {
    REQUIRE(x && out && okdt(dtype) && n >= 0 && n % vn(dtype) == 0);
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    DET_LAUNCH(dtype, gelu_map_kernel, n / vn(dtype), (const T*)x, (const T*)dy, (T*)out, n);
    return 0;
}
extern "C" size_t ssl4gie_map_layernorm_workspace_bytes(int B) {
    return ((size_t)B * MLN_PARTS * 2 + 2 * (size_t)B) * sizeof(float);  // partials, two means
}
// y[b, i] = (x[b, i] - mean_b) rstd_b w[i] + bias[i]; mean / rstd [B] are outputs kept for backward
extern "C" int ssl4gie_map_layernorm_fwd(const void* x, const float* w, const float* bias, void* y,
                                         float* mean, float* rstd, float eps, float* workspace,
                                         int dtype, int B, long long M, void* stream) {
    REQUIRE(x && w && bias && y && mean && rstd && workspace && okdt(dtype) && B > 0 && M > 0 &&
            M % 8 == 0 && B <= 65535);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(MLN_PARTS, B), block(256);
    if (dtype == SSL4GIE_BF16) {
        hipLaunchKernelGGL((mapln_reduce_kernel<bf16_t, false>), grid, block, 0, st, (const bf16_t*)x,
                           (const bf16_t*)nullptr, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, workspace, M);
        hipLaunchKernelGGL(mapln_finalize_kernel<bf16_t>, dim3(B), dim3(64), 0, st, workspace,
                           (const bf16_t*)x, mean, rstd, M, eps, 0);
    } else {
        hipLaunchKernelGGL((mapln_reduce_kernel<float, false>), grid, block, 0, st, (const float*)x,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, workspace, M);
        hipLaunchKernelGGL(mapln_finalize_kernel<float>, dim3(B), dim3(64), 0, st, workspace,
                           (const float*)x, mean, rstd, M, eps, 0);
    }
    LAUNCH_CHECK();
    const long long total = (long long)B * M;
    DET_LAUNCH(dtype, mapln_apply_kernel, total / vn(dtype), (const T*)x, w, bias, mean, rstd, (T*)y, M,
               total);
    return 0;
}
// dx, and dw / db [M] overwritten (accumulate = 0) or accumulated (either may be NULL)
extern "C" int ssl4gie_map_layernorm_bwd(const void* x, const void* dy, const float* w,
                                         const float* mean, const float* rstd, void* dx, float* dw,
                                         float* db, int accumulate, float* workspace, int dtype, int B,
                                         long long M, void* stream) {
    REQUIRE(x && dy && w && mean && rstd && dx && workspace && okdt(dtype) && B > 0 && M > 0 &&
            M % 8 == 0 && B <= 65535);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(MLN_PARTS, B), block(256);
    float* mg = workspace + (size_t)B * MLN_PARTS * 2;
    float* mgx = mg + B;
    if (dtype == SSL4GIE_BF16) {
        hipLaunchKernelGGL((mapln_reduce_kernel<bf16_t, true>), grid, block, 0, st, (const bf16_t*)x,
                           (const bf16_t*)dy, w, mean, rstd, workspace, M);
        hipLaunchKernelGGL(mapln_finalize_kernel<bf16_t>, dim3(B), dim3(64), 0, st, workspace,
                           (const bf16_t*)x, mg, mgx, M, 0.f, 1);
    } else {
        hipLaunchKernelGGL((mapln_reduce_kernel<float, true>), grid, block, 0, st, (const float*)x,
                           (const float*)dy, w, mean, rstd, workspace, M);
        hipLaunchKernelGGL(mapln_finalize_kernel<float>, dim3(B), dim3(64), 0, st, workspace,
                           (const float*)x, mg, mgx, M, 0.f, 1);
    }
    LAUNCH_CHECK();
    DET_LAUNCH(dtype, mapln_bwd_apply_kernel, M / vn(dtype), (const T*)x, (const T*)dy, w, mean, rstd, mg,
               mgx, (T*)dx, dw, db, accumulate, B, M);
    return 0;
}
