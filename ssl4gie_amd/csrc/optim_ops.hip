// Optimizer steps over the parameter arena: AdamW as the reference drives it
// (torch.optim.AdamW, Models/mae/main_pretrain.py:179-180, train_depth.py:280) and LARS
// (Models/moco_v3/moco/optimizer.py:18-43) as ONE or THREE launches over the flat fp32 buffers of
// engine.ParamArena instead of a multi-tensor launch chain per step.
//
// The arena is a sequence of segments (one per parameter, 64-element aligned, padding belongs to the
// preceding parameter and stays zero); per-segment hyper-parameters come from small device tables:
//   seg_start [S + 1]  element offsets (int64, ascending, seg_start[S] = arena length)
//   seg_lr    [S]      learning rate of the segment's group; < 0: skip the segment (frozen
//                      parameter, or no gradient this step — torch skips p.grad is None too)
//   seg_wd    [S]      weight decay
//   seg_mat   [S]      LARS only: 1 for p.ndim > 1 (trust-ratio scaling + weight decay), else 0
// HBM-bound streaming: 16-byte accesses, no atomics, fixed summation order.
#include "common.h"
#include "ssl4gie_hip.h"

namespace {
DEVI int find_seg(const long long* __restrict__ seg_start, int S, long long i) {
    int lo = 0, hi = S;  // invariant: seg_start[lo] <= i < seg_start[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (seg_start[mid] <= i) lo = mid;
        else hi = mid;
    }
    return lo;
}
// segment of element i for a whole 256-thread block: ONE binary search (thread 0), then every
// thread walks forward from there — a block's 1024 elements almost always lie in one or two segments
DEVI int block_seg(const long long* __restrict__ seg_start, int S, long long i, long long block_first) {
    __shared__ int s0;
    if (threadIdx.x == 0) s0 = find_seg(seg_start, S, block_first);
    __syncthreads();
    int s = s0;
    while (s + 1 < S && seg_start[s + 1] <= i) ++s;
    return s;
}
}  // namespace

// torch.optim.AdamW (decoupled decay, bias-corrected): p *= 1 - lr wd; m, v moments;
// p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adamw_arena_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
    const long long* __restrict__ seg_start, const float* __restrict__ seg_lr,
    const float* __restrict__ seg_wd, int S, float b1, float b2, float eps, float bc1, float rsqrt_bc2,
    long long n, bf16_t* __restrict__ lp, long long lo) {
    const long long first = lo + (long long)blockIdx.x * blockDim.x * 4;  // elements [lo, n)
    const long long i = first + threadIdx.x * 4;
    const int s = block_seg(seg_start, S, i < n ? i : n - 4, first);  // slices are 64-element aligned
    if (i >= n) return;
    const float lr = seg_lr[s];
    if (lr < 0.f) return;
    const float wd = seg_wd[s];
    f32x4 pp = ld4(p + i), mm = ld4(m + i), vv = ld4(v + i);
    const f32x4 gg = ld4(g + i);
    const float step = lr / bc1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x = pp[j] * (1.f - lr * wd);
        const float mj = b1 * mm[j] + (1.f - b1) * gg[j];
        const float vj = b2 * vv[j] + (1.f - b2) * gg[j] * gg[j];
        x -= step * mj / (sqrtf(vj) * rsqrt_bc2 + eps);
        pp[j] = x; mm[j] = mj; vv[j] = vj;
    }
    st4(p + i, pp);
    st4(m + i, mm);
    st4(v + i, vv);
    if (lp) st4(lp + i, pp);  // the bf16 operand copy of the updated weights, while they are in registers
}

// LARS stage 1: partial[(s * parts + part) * 2 + {0, 1}] = sum p^2, sum (g + wd p)^2 over the part
#define LARS_PARTS 32
__global__ __launch_bounds__(256) void lars_norm_partial_kernel(
    const float* __restrict__ p, const float* __restrict__ g, const long long* __restrict__ seg_start,
    const float* __restrict__ seg_lr, const float* __restrict__ seg_wd, const float* __restrict__ seg_mat,
    float* __restrict__ partial) {
    __shared__ float red[2][4];
    const int s = blockIdx.y, part = blockIdx.x;
    float a = 0.f, b = 0.f;
    if (seg_lr[s] >= 0.f && seg_mat[s] != 0.f) {
        const long long lo = seg_start[s], hi = seg_start[s + 1];
        const float wd = seg_wd[s];
        for (long long i = lo + ((long long)part * 256 + threadIdx.x) * 4; i < hi; i += (long long)LARS_PARTS * 1024) {
            const f32x4 pp = ld4(p + i), gg = ld4(g + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = gg[j] + wd * pp[j];
                a += pp[j] * pp[j];
                b += d * d;
            }
        }
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[((size_t)s * LARS_PARTS + part) * 2] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        partial[((size_t)s * LARS_PARTS + part) * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}
// stage 2: q[s] = trust |p| / |dp| where both norms are > 0, else 1 (matrices); 1 for vectors
__global__ void lars_trust_kernel(const float* __restrict__ partial, const float* __restrict__ seg_mat,
                                  float* __restrict__ q, float trust, int S) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    float a = 0.f, b = 0.f;
    for (int k = 0; k < LARS_PARTS; ++k) {
        a += partial[((size_t)s * LARS_PARTS + k) * 2];
        b += partial[((size_t)s * LARS_PARTS + k) * 2 + 1];
    }
    const float pn = sqrtf(a), un = sqrtf(b);
    q[s] = (seg_mat[s] != 0.f && pn > 0.f && un > 0.f) ? trust * pn / un : 1.f;
}
// stage 3: dp = (g + wd p) q (matrices) or g (vectors); mu = mom mu + dp; p -= lr mu
__global__ __launch_bounds__(256) void lars_apply_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mu,
    const long long* __restrict__ seg_start, const float* __restrict__ seg_lr,
    const float* __restrict__ seg_wd, const float* __restrict__ seg_mat, const float* __restrict__ q,
    int S, float momentum, long long n) {
    const long long first = (long long)blockIdx.x * blockDim.x * 4;
    const long long i = first + threadIdx.x * 4;
    const int s = block_seg(seg_start, S, i < n ? i : n - 4, first);
    if (i >= n) return;
    const float lr = seg_lr[s];
    if (lr < 0.f) return;
    const bool mat = seg_mat[s] != 0.f;
    const float wd = mat ? seg_wd[s] : 0.f, qq = mat ? q[s] : 1.f;
    f32x4 pp = ld4(p + i), mm = ld4(mu + i);
    const f32x4 gg = ld4(g + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float dp = (gg[j] + wd * pp[j]) * qq;
        mm[j] = momentum * mm[j] + dp;
        pp[j] -= lr * mm[j];
    }
    st4(p + i, pp);
    st4(mu + i, mm);
}

extern "C" int ssl4gie_adamw_arena_range(float* p, const float* g, float* m, float* v,
                                         const long long* seg_start, const float* seg_lr,
                                         const float* seg_wd, int S, float beta1, float beta2, float eps,
                                         int step, long long lo, long long hi, void* lp_bf16,
                                         void* stream) {
    REQUIRE(p && g && m && v && seg_start && seg_lr && seg_wd && S > 0 && step > 0 && lo >= 0 && hi >= lo &&
            lo % 4 == 0 && hi % 4 == 0);
    if (hi == lo) return 0;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_arena_kernel, dim3((unsigned)(((hi - lo) / 4 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, p, g, m, v, seg_start, seg_lr, seg_wd, S, beta1, beta2, eps,
                       bc1, 1.f / sqrtf(bc2), hi, (bf16_t*)lp_bf16, lo);
    LAUNCH_CHECK();
    return 0;
}
extern "C" int ssl4gie_adamw_arena_lp(float* p, const float* g, float* m, float* v,
                                      const long long* seg_start, const float* seg_lr,
                                      const float* seg_wd, int S, float beta1, float beta2, float eps,
                                      int step, long long n, void* lp_bf16, void* stream) {
    REQUIRE(n >= 0);
    return ssl4gie_adamw_arena_range(p, g, m, v, seg_start, seg_lr, seg_wd, S, beta1, beta2, eps, step, 0, n,
                                     lp_bf16, stream);
}
extern "C" int ssl4gie_adamw_arena(float* p, const float* g, float* m, float* v,
                                   const long long* seg_start, const float* seg_lr,
                                   const float* seg_wd, int S, float beta1, float beta2, float eps,
                                   int step, long long n, void* stream) {
    return ssl4gie_adamw_arena_lp(p, g, m, v, seg_start, seg_lr, seg_wd, S, beta1, beta2, eps, step, n,
                                  nullptr, stream);
}
extern "C" size_t ssl4gie_lars_workspace_bytes(int S) {
    return ((size_t)S * LARS_PARTS * 2 + S) * sizeof(float);
}
extern "C" int ssl4gie_lars_arena(float* p, const float* g, float* mu, const long long* seg_start,
                                  const float* seg_lr, const float* seg_wd, const float* seg_mat, int S,
                                  float momentum, float trust, float* workspace, long long n,
                                  void* stream) {
    REQUIRE(p && g && mu && seg_start && seg_lr && seg_wd && seg_mat && workspace && S > 0 && S <= 65535 &&
            n >= 0 && n % 4 == 0);
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    float* partial = workspace;
    float* q = workspace + (size_t)S * LARS_PARTS * 2;
    hipLaunchKernelGGL(lars_norm_partial_kernel, dim3(LARS_PARTS, S), dim3(256), 0, st, p, g, seg_start,
                       seg_lr, seg_wd, seg_mat, partial);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(lars_trust_kernel, dim3((S + 255) / 256), dim3(256), 0, st, partial, seg_mat, q,
                       trust, S);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(lars_apply_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, p, g, mu,
                       seg_start, seg_lr, seg_wd, seg_mat, q, S, momentum, n);
    LAUNCH_CHECK();
    return 0;
}
