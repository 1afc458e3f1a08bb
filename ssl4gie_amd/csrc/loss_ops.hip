// Finetune losses as device kernels: value AND gradient w.r.t. the prediction in
// three passes over the map instead of the ~100 small elementwise / reduction launches of the
// host-side torch formulation (which stays available and is the parity reference).
//
//  * ScaleAndShiftInvariantLoss (Depth_estimation/Metrics/losses.py:120-146): per image the closed-form
//    2x2 least squares (scale s, shift h) on the valid pixels m = target > 0 (:5-25), masked MSE /
//    (2 sum M) (:51-57, batch-based reduction :28-38) + alpha x sum over 4 scales of the masked
//    gradient L1 on the subsampled grids (:60-77, :104-117).
//      pass A  per image sums a00 = sum m p^2, a01 = sum m p, a11 = sum m, b0 = sum m p t, b1 = sum m t
//              and the mask counts of the subsampled grids; -> s, h, batch totals M_k
//      pass B  g = dL/d(ssi) per pixel (MSE part + sign terms of the up to 4 x 4 neighbour pairs),
//              loss partials, per image G0 = sum g, G1 = sum g p
//      pass C  dL/dp = g s + G1 ds/dp + G0 dh/dp   (s and h depend on every pixel of the image)
//    Two-stage deterministic reductions (block partials, then one block per image / per batch).
//  * SoftDiceLoss (Binary_segmentation/Metrics/losses.py:5-24): per image sums of sigmoid(l) t,
//    sigmoid(l)^2, t^2; loss = 1 - mean score; gradient in a second pass.
#include "common.h"
#include "internal.h"
#include "ssl4gie_hip.h"

namespace {

constexpr int NB = 32;      // blocks per image
constexpr int NSUM_A = 8;   // a00 a01 a11 b0 b1 M1 M2 M3
constexpr int NSUM_B = 7;   // mse reg0 reg1 reg2 reg3 G0 G1

template <int K>
DEVI void block_reduce(float (&v)[K], float* sh /* [4][K] */) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < K; ++k) sh[wave * K + k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = sh[k] + sh[K + k] + sh[2 * K + k] + sh[3 * K + k];
}

// layout of the fp32 workspace (floats): partA [B][NB][8] | img [B][12] | tot [8] | partB [B][NB][7] | g [B*H*W]
struct SsiWs {
    float *partA, *img, *tot, *partB, *g;
};
DEVI SsiWs ssi_ws(float* ws, int B, long long HW) {
    SsiWs w;
    w.partA = ws;
    w.img = w.partA + (size_t)B * NB * NSUM_A;
    w.tot = w.img + (size_t)B * 12;
    w.partB = w.tot + 8;
    w.g = w.partB + (size_t)B * NB * NSUM_B;
    return w;
}

__global__ __launch_bounds__(256) void ssi_sums_kernel(const float* __restrict__ pred,
                                                       const float* __restrict__ target, float* __restrict__ ws,
                                                       int B, int H, int W) {
    const int b = blockIdx.y;
    const long long HW = (long long)H * W;
    const float* p = pred + (size_t)b * HW;
    const float* t = target + (size_t)b * HW;
    float v[NSUM_A] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long long)NB * 256) {
        const float tv = t[i], pv = p[i];
        if (tv > 0.f) {
            v[0] += pv * pv; v[1] += pv; v[2] += 1.f; v[3] += pv * tv; v[4] += tv;
            const int y = (int)(i / W), x = (int)(i % W);
            if (!((y | x) & 1)) v[5] += 1.f;
            if (!((y | x) & 3)) v[6] += 1.f;
            if (!((y | x) & 7)) v[7] += 1.f;
        }
    }
    __shared__ float sh[4 * NSUM_A];
    block_reduce<NSUM_A>(v, sh);
    if (threadIdx.x == 0) {
        float* o = ssi_ws(ws, B, HW).partA + ((size_t)b * NB + blockIdx.x) * NSUM_A;
#pragma unroll
        for (int k = 0; k < NSUM_A; ++k) o[k] = v[k];
    }
}

// one block; thread b finishes image b (NB partials in a fixed order), then thread 0 the batch totals
__global__ void ssi_finalize_a_kernel(float* __restrict__ ws, int B, long long HW) {
    const SsiWs w = ssi_ws(ws, B, HW);
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        float s[NSUM_A] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < NB; ++j)
            for (int k = 0; k < NSUM_A; ++k) s[k] += w.partA[((size_t)b * NB + j) * NSUM_A + k];
        const float a00 = s[0], a01 = s[1], a11 = s[2], b0 = s[3], b1 = s[4];
        const float det = a00 * a11 - a01 * a01;
        float sc = 0.f, shf = 0.f;
        if (det != 0.f) {  // losses.py:18-23: images with a singular system keep scale = shift = 0
            sc = (a11 * b0 - a01 * b1) / det;
            shf = (-a01 * b0 + a00 * b1) / det;
        }
        float* o = w.img + (size_t)b * 12;
        o[0] = a00; o[1] = a01; o[2] = a11; o[3] = b0; o[4] = b1; o[5] = det; o[6] = sc; o[7] = shf;
        o[8] = s[5]; o[9] = s[6]; o[10] = s[7]; o[11] = 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float m[4] = {0, 0, 0, 0};
        for (int b = 0; b < B; ++b) {
            const float* o = w.img + (size_t)b * 12;
            m[0] += o[2]; m[1] += o[8]; m[2] += o[9]; m[3] += o[10];
        }
        for (int k = 0; k < 4; ++k) w.tot[k] = m[k];
    }
}

DEVI float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void ssi_grad_kernel(const float* __restrict__ pred,
                                                       const float* __restrict__ target, float* __restrict__ ws,
                                                       int B, int H, int W, float alpha, int scales) {
    const int b = blockIdx.y;
    const long long HW = (long long)H * W;
    const SsiWs w = ssi_ws(ws, B, HW);
    const float* p = pred + (size_t)b * HW;
    const float* t = target + (size_t)b * HW;
    const float sc = w.img[(size_t)b * 12 + 6], shf = w.img[(size_t)b * 12 + 7];
    float inv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) inv[k] = w.tot[k] > 0.f ? 1.f / w.tot[k] : 0.f;
    auto D = [&](int y, int x, float& m) -> float {  // mask * (ssi - target) at (y, x)
        const long long i = (long long)y * W + x;
        const float tv = t[i];
        m = tv > 0.f ? 1.f : 0.f;
        return m * (sc * p[i] + shf - tv);
    };
    float v[NSUM_B] = {0, 0, 0, 0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long long)NB * 256) {
        const int y = (int)(i / W), x = (int)(i % W);
        float m;
        const float d = D(y, x, m);
        float g = d * inv[0];          // d/dssi of sum m (ssi - t)^2 / (2 sum M)
        v[0] += d * d;
        if (alpha > 0.f && m > 0.f) {
            for (int k = 0; k < scales && k < 4; ++k) {
                const int s = 1 << k;
                if ((y | x) & (s - 1)) break;  // not on this (or any coarser) grid
                float acc = 0.f, m2;
                if (x + s < W) { const float e = D(y, x + s, m2) - d; v[1 + k] += fabsf(e) * m2; acc -= sgn(e) * m2; }
                if (y + s < H) { const float e = D(y + s, x, m2) - d; v[1 + k] += fabsf(e) * m2; acc -= sgn(e) * m2; }
                if (x - s >= 0) { const float e = d - D(y, x - s, m2); acc += sgn(e) * m2; }
                if (y - s >= 0) { const float e = d - D(y - s, x, m2); acc += sgn(e) * m2; }
                g += alpha * inv[k] * acc;
            }
        }
        w.g[(size_t)b * HW + i] = g;
        v[5] += g;
        v[6] += g * p[i];
    }
    __shared__ float sh[4 * NSUM_B];
    block_reduce<NSUM_B>(v, sh);
    if (threadIdx.x == 0) {
        float* o = w.partB + ((size_t)b * NB + blockIdx.x) * NSUM_B;
#pragma unroll
        for (int k = 0; k < NSUM_B; ++k) o[k] = v[k];
    }
}

__global__ void ssi_finalize_b_kernel(float* __restrict__ ws, float* __restrict__ loss, int B, long long HW,
                                      float alpha, int scales) {
    const SsiWs w = ssi_ws(ws, B, HW);
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        float s[NSUM_B] = {0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < NB; ++j)
            for (int k = 0; k < NSUM_B; ++k) s[k] += w.partB[((size_t)b * NB + j) * NSUM_B + k];
        float* o = w.partB + (size_t)b * NB * NSUM_B;  // image totals overwrite the image's first partial
        for (int k = 0; k < NSUM_B; ++k) o[k] = s[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot[5] = {0, 0, 0, 0, 0};
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < 5; ++k) tot[k] += w.partB[(size_t)b * NB * NSUM_B + k];
        float l = w.tot[0] > 0.f ? tot[0] / (2.f * w.tot[0]) : 0.f;
        if (alpha > 0.f)
            for (int k = 0; k < scales && k < 4; ++k)
                if (w.tot[k] > 0.f) l += alpha * tot[1 + k] / w.tot[k];
        *loss = l;
    }
}

__global__ __launch_bounds__(256) void ssi_apply_kernel(const float* __restrict__ pred,
                                                        const float* __restrict__ target,
                                                        const float* __restrict__ ws_c, float* __restrict__ dpred,
                                                        int B, long long HW) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * HW) return;
    const int b = (int)(idx / HW);
    const SsiWs w = ssi_ws(const_cast<float*>(ws_c), B, HW);
    const float* o = w.img + (size_t)b * 12;
    const float a00 = o[0], a01 = o[1], a11 = o[2], b0 = o[3], b1 = o[4], det = o[5], sc = o[6], shf = o[7];
    const float G0 = w.partB[(size_t)b * NB * NSUM_B + 5], G1 = w.partB[(size_t)b * NB * NSUM_B + 6];
    const float pv = pred[idx], tv = target[idx];
    float d = w.g[idx] * sc;
    if (tv > 0.f && det != 0.f) {
        const float dd = 2.f * a11 * pv - 2.f * a01;  // d det / d p
        const float ds = (a11 * tv - b1 - sc * dd) / det;
        const float dh = (-b0 - a01 * tv + 2.f * pv * b1 - shf * dd) / det;
        d += G1 * ds + G0 * dh;
    }
    (void)a00;
    dpred[idx] = d;
}

// ------------------------------------------------------------------ soft Dice
__global__ __launch_bounds__(256) void dice_sums_kernel(const float* __restrict__ logits,
                                                        const float* __restrict__ target, float* __restrict__ ws,
                                                        long long n) {
    const int b = blockIdx.y;
    const float* l = logits + (size_t)b * n;
    const float* t = target + (size_t)b * n;
    float v[3] = {0, 0, 0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)NB * 256) {
        const float m1 = 1.f / (1.f + __expf(-l[i])), m2 = t[i];
        v[0] += m1 * m2; v[1] += m1 * m1; v[2] += m2 * m2;
    }
    __shared__ float sh[4 * 3];
    block_reduce<3>(v, sh);
    if (threadIdx.x == 0)
        for (int k = 0; k < 3; ++k) ws[((size_t)b * NB + blockIdx.x) * 3 + k] = v[k];
}
__global__ void dice_finalize_kernel(float* __restrict__ ws, float* __restrict__ loss, int B, float smooth) {
    float* img = ws + (size_t)B * NB * 3;  // [B][3]: inter, den, score
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        float s[3] = {0, 0, 0};
        for (int j = 0; j < NB; ++j)
            for (int k = 0; k < 3; ++k) s[k] += ws[((size_t)b * NB + j) * 3 + k];
        const float num = s[0] + smooth, den = s[1] + s[2] + smooth;
        img[b * 3] = num; img[b * 3 + 1] = den; img[b * 3 + 2] = 2.f * num / den;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float sum = 0.f;
        for (int b = 0; b < B; ++b) sum += img[b * 3 + 2];
        *loss = 1.f - sum / (float)B;
    }
}
__global__ __launch_bounds__(256) void dice_apply_kernel(const float* __restrict__ logits,
                                                         const float* __restrict__ target,
                                                         const float* __restrict__ ws, float* __restrict__ dlogits,
                                                         int B, long long n) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * n) return;
    const int b = (int)(idx / n);
    const float* img = ws + (size_t)B * NB * 3 + (size_t)b * 3;
    const float num = img[0], den = img[1];
    const float m1 = 1.f / (1.f + __expf(-logits[idx])), m2 = target[idx];
    // score = 2 num / den: d score / d m1 = 2 m2 / den - 2 num 2 m1 / den^2; loss = 1 - mean score
    const float dscore = 2.f * m2 / den - 4.f * num * m1 / (den * den);
    dlogits[idx] = -(dscore / (float)B) * m1 * (1.f - m1);
}

}  // namespace

extern "C" size_t ssl4gie_ssi_loss_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t n = (size_t)B * NB * NSUM_A + (size_t)B * 12 + 8 + (size_t)B * NB * NSUM_B + (size_t)B * H * W;
    return n * sizeof(float);
}

extern "C" int ssl4gie_ssi_loss(const float* pred, const float* target, float* loss, float* dpred, int B, int H,
                                int W, float alpha, int scales, void* workspace, void* stream) {
    REQUIRE(pred && target && loss && dpred && workspace && B > 0 && H > 0 && W > 0 && scales >= 1 && scales <= 4);
    REQUIRE(B <= 65535);
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const long long HW = (long long)H * W;
    hipLaunchKernelGGL(ssi_sums_kernel, dim3(NB, B), dim3(256), 0, st, pred, target, ws, B, H, W);
    hipLaunchKernelGGL(ssi_finalize_a_kernel, dim3(1), dim3(256), 0, st, ws, B, HW);
    hipLaunchKernelGGL(ssi_grad_kernel, dim3(NB, B), dim3(256), 0, st, pred, target, ws, B, H, W, alpha, scales);
    hipLaunchKernelGGL(ssi_finalize_b_kernel, dim3(1), dim3(256), 0, st, ws, loss, B, HW, alpha, scales);
    const long long total = (long long)B * HW;
    hipLaunchKernelGGL(ssi_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pred, target,
                       (const float*)ws, dpred, B, HW);
    LAUNCH_CHECK();
    return 0;
}

extern "C" size_t ssl4gie_dice_loss_workspace_bytes(int B) {
    return B > 0 ? ((size_t)B * NB * 3 + (size_t)B * 3) * sizeof(float) : 0;
}

extern "C" int ssl4gie_dice_loss(const float* logits, const float* target, float* loss, float* dlogits, int B,
                                 long long n, float smooth, void* workspace, void* stream) {
    REQUIRE(logits && target && loss && dlogits && workspace && B > 0 && B <= 65535 && n > 0);
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    hipLaunchKernelGGL(dice_sums_kernel, dim3(NB, B), dim3(256), 0, st, logits, target, ws, n);
    hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(256), 0, st, ws, loss, B, smooth);
    const long long total = (long long)B * n;
    hipLaunchKernelGGL(dice_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, logits, target,
                       (const float*)ws, dlogits, B, n);
    LAUNCH_CHECK();
    return 0;
}
