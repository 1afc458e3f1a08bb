// DPT dense-prediction decoder glue for gfx950: everything around the
// GEMMs of `Models/DPT_decoder.py`.  Activations are channels-last ([B, H, W, C], i.e. the ViT's
// token-major layout extended to the upsampled maps), so 1x1 convolutions and the k = s transposed
// convolutions are plain token-major GEMMs (gemm*.hip) and a 3x3 convolution is a GEMM over a
// (dy, dx, c)-ordered patch matrix.  HBM-bound byte movers: 16-byte accesses, one (pixel, 8- or
// 4-channel chunk) per thread, no atomics (every reduction is two-stage and deterministic).
//
// Reference: Models/DPT_decoder.py — reassemble `act_postprocess*` :333-410, `layer*_rn` :412-447,
// `ResidualConvUnit_custom.forward` :212-233, `FeatureFusionBlock_custom.forward` :281-301 (bilinear
// x2, align_corners=True), depth head `output_conv` :468-482, `Slice` :5-11 (drop the cls token).
#include "common.h"
#include "ssl4gie_hip.h"
#include "internal.h"
#include <stdlib.h>

template <typename T> struct Vec;  // 16-byte vector of T
template <> struct Vec<bf16_t> {
    static constexpr int N = 8;
    typedef u32x4 raw;
};
template <> struct Vec<float> {
    static constexpr int N = 4;
    typedef f32x4 raw;
};
DEVI u32x4 relu_raw(u32x4 v) {  // bf16 pairs: clear negative halves (sign bit set), -0 -> +0
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t lo = (v[j] & 0x8000u) ? 0u : (v[j] & 0xffffu);
        const uint32_t hi = (v[j] & 0x80000000u) ? 0u : (v[j] & 0xffff0000u);
        v[j] = lo | hi;
    }
    return v;
}
DEVI f32x4 relu_raw(f32x4 v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
    return v;
}
template <typename T> DEVI void unpack(const typename Vec<T>::raw& r, float (&f)[Vec<T>::N]);
template <> DEVI void unpack<bf16_t>(const u32x4& r, float (&f)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = __uint_as_float(r[j] << 16);
        f[2 * j + 1] = __uint_as_float(r[j] & 0xffff0000u);
    }
}
template <> DEVI void unpack<float>(const f32x4& r, float (&f)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = r[j];
}
template <typename T> DEVI typename Vec<T>::raw pack(const float (&f)[Vec<T>::N]);
template <> DEVI u32x4 pack<bf16_t>(const float (&f)[8]) {
    u32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = pack_bf2(f[2 * j], f[2 * j + 1]);
    return r;
}
template <> DEVI f32x4 pack<float>(const float (&f)[4]) { return f32x4{f[0], f[1], f[2], f[3]}; }

#define DPT_GRID(total) dim3((unsigned)(((total) + 255) / 256)), dim3(256)
// ------------------------------------------------------------------ 3x3 patch matrix (pad 1)
// cols[(b, oy, ox), (dy*3 + dx)*C + c] = act(x[b, oy*s + dy - 1, ox*s + dx - 1, c]) (0 outside);
// columns [9C, ld) are zero-filled (K padding for the MFMA GEMM)
template <typename T>
__global__ void im2col3x3_kernel(const T* __restrict__ x, T* __restrict__ cols, int B, int H, int W,
                                 int C, int Ho, int Wo, int stride, int relu, long long ld,
                                 long long total) {
    constexpr int V = Vec<T>::N;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cpr = (int)(ld / V);  // chunks per output row
    const long long m = idx / cpr;
    const int ch = (int)(idx % cpr);
    const int col = ch * V;
    typename Vec<T>::raw v = {};
    if (col < 9 * C) {
        const int tap = col / C, c = col % C;
        const int dy = tap / 3, dx = tap % 3;
        const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), b = (int)(m / ((long long)Wo * Ho));
        const int iy = oy * stride + dy - 1, ix = ox * stride + dx - 1;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            v = *(const typename Vec<T>::raw*)(x + (((size_t)b * H + iy) * W + ix) * C + c);
            if (relu) v = relu_raw(v);
        }
    }
    *(typename Vec<T>::raw*)(cols + (size_t)m * ld + col) = v;
}

// gather-form transpose of the above (data gradient of a strided conv):
// dx[b, y, x, c] = sum over taps with (y + 1 - dy) % s == 0, ... of dcols[(b, oy, ox), tap*C + c]
template <typename T>
__global__ void col2im3x3_kernel(const T* __restrict__ dcols, T* __restrict__ dx, int B, int H,
                                 int W, int C, int Ho, int Wo, int stride, long long ld,
                                 long long total) {
    constexpr int V = Vec<T>::N;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cpr = C / V;
    const long long p = idx / cpr;
    const int c = (int)(idx % cpr) * V;
    const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((long long)W * H));
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int dy = 0; dy < 3; ++dy) {
        const int ny = y + 1 - dy;
        if (ny < 0 || ny % stride) continue;
        const int oy = ny / stride;
        if (oy >= Ho) continue;
        for (int dxx = 0; dxx < 3; ++dxx) {
            const int nx = x + 1 - dxx;
            if (nx < 0 || nx % stride) continue;
            const int ox = nx / stride;
            if (ox >= Wo) continue;
            const size_t m = ((size_t)b * Ho + oy) * Wo + ox;
            float f[V];
            unpack<T>(*(const typename Vec<T>::raw*)(dcols + m * ld + (dy * 3 + dxx) * C + c), f);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += f[j];
        }
    }
    *(typename Vec<T>::raw*)(dx + (size_t)p * C + c) = pack<T>(acc);
}

// ------------------------------------------------------------------ bilinear x2, align_corners=True
// src = dst * (H - 1) / (2H - 1)   (torch.nn.functional.interpolate semantics)
DEVI void bl_src(int d, int n_in, int n_out, int& i0, int& i1, float& w1) {
    const float scale = n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f;
    const float s = scale * (float)d;
    i0 = (int)s;
    i0 = i0 < n_in - 1 ? i0 : n_in - 1;
    i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
    w1 = s - (float)i0;
}
// one output element from its four taps: rows first, then the vertical lerp (ATen's upsample_bilinear2d order), with
// the multiply-adds written out — under -ffp-contract=fast the compiler is otherwise free to contract the two
// forward kernels differently, and they must agree bit for bit
// (the form is the one the compiler had chosen for the round 1-5 kernel — product of the second tap, multiply-add of
// the first — so that the committed 60-step fp32 depth curve, G13, keeps its rounding)
DEVI float bl_lerp(float f00, float f01, float f10, float f11, float wx, float wy) {
    const float top = __builtin_fmaf(1.f - wx, f00, f01 * wx);
    const float bot = __builtin_fmaf(1.f - wx, f10, f11 * wx);
    return __builtin_fmaf(1.f - wy, top, bot * wy);
}
// Grid (x chunks, output / input row, image): the row and the image come from the block index, so a thread's
// index arithmetic is one 32-bit division (the flat form spent four 64-bit divisions per 16-byte store and ran at
// half of the HBM rate: profiles/r04dw) and the vertical taps / weights are uniform over the block.
template <typename T>
__global__ __launch_bounds__(256) void bilinear2x_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int H,
                                                             int W, int C) {
    constexpr int V = Vec<T>::N;
    const unsigned cpr = (unsigned)(C / V), Ho = 2 * H, Wo = 2 * W;
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    const unsigned ox = t / cpr;
    if (ox >= Wo) return;
    const int c = (int)(t - ox * cpr) * V;
    const int oy = blockIdx.y, b = blockIdx.z;
    int y0, y1, x0, x1;
    float wy, wx;
    bl_src(oy, H, Ho, y0, y1, wy);
    bl_src((int)ox, W, Wo, x0, x1, wx);
    const T* base = x + (size_t)b * H * W * C + c;
    float f00[V], f01[V], f10[V], f11[V], o[V];
    unpack<T>(*(const typename Vec<T>::raw*)(base + ((size_t)y0 * W + x0) * C), f00);
    unpack<T>(*(const typename Vec<T>::raw*)(base + ((size_t)y0 * W + x1) * C), f01);
    unpack<T>(*(const typename Vec<T>::raw*)(base + ((size_t)y1 * W + x0) * C), f10);
    unpack<T>(*(const typename Vec<T>::raw*)(base + ((size_t)y1 * W + x1) * C), f11);
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = bl_lerp(f00[j], f01[j], f10[j], f11[j], wx, wy);
    *(typename Vec<T>::raw*)(y + (((size_t)b * Ho + oy) * Wo + ox) * C + c) = pack<T>(o);
}
// backward in gather form: an input pixel collects from every output pixel whose two taps include it, in the
// order (output row, output column) of the candidates.  The column coefficients are computed once per thread (not
// once per candidate row), the row coefficients are uniform over the block.
template <typename T>
__global__ __launch_bounds__(256) void bilinear2x_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int H,
                                                             int W, int C) {
    constexpr int V = Vec<T>::N;
    const unsigned cpr = (unsigned)(C / V);
    const int Ho = 2 * H, Wo = 2 * W;
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    const unsigned uix = t / cpr;
    if (uix >= (unsigned)W) return;
    const int ix = (int)uix, c = (int)(t - uix * cpr) * V;
    const int iy = blockIdx.y, b = blockIdx.z;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    // candidate output rows / columns: src within (i - 1, i + 1)  ->  o in about [2i - 2, 2i + 3]
    const int oy_lo = 2 * iy - 3 > 0 ? 2 * iy - 3 : 0, oy_hi = 2 * iy + 3 < Ho - 1 ? 2 * iy + 3 : Ho - 1;
    const int ox_lo = 2 * ix - 3 > 0 ? 2 * ix - 3 : 0, ox_hi = 2 * ix + 3 < Wo - 1 ? 2 * ix + 3 : Wo - 1;
    float cxs[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int ox = ox_lo + k;
        int x0, x1;
        float wx;
        bl_src(ox <= ox_hi ? ox : ox_hi, W, Wo, x0, x1, wx);
        float cx = 0.f;
        if (x0 == ix) cx += 1.f - wx;
        if (x1 == ix) cx += wx;
        cxs[k] = ox <= ox_hi ? cx : 0.f;
    }
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        int y0, y1;
        float wy;
        bl_src(oy, H, Ho, y0, y1, wy);
        float cy = 0.f;
        if (y0 == iy) cy += 1.f - wy;
        if (y1 == iy) cy += wy;
        if (cy == 0.f) continue;
        const T* row = dy + (((size_t)b * Ho + oy) * Wo + ox_lo) * C + c;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const float cx = cxs[k];
            if (cx == 0.f) continue;
            float f[V];
            unpack<T>(*(const typename Vec<T>::raw*)(row + (size_t)k * C), f);
            const float w = cy * cx;
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += w * f[j];
        }
    }
    *(typename Vec<T>::raw*)(dx + (((size_t)b * H + iy) * W + ix) * C + c) = pack<T>(acc);
}

// ---- 2 x 2 outputs per thread (round 6).  The forward above issues four 16-byte tap loads per 16-byte store and runs
// at 3.2-3.5 TB/s on the depth step's maps: it is bound by the tap loads through the CU's texture path, not by HBM.
// The outputs (2k, 2k + 1) x (2j, 2j + 1) take their taps from at most 3 x 3 input pixels: with
// (y0a, y1a) = taps of row 2k and (y0b, y1b) of row 2k + 1, y0b is y0a or y1a and y1b is y1a or one further (the
// source coordinate advances by less than 1/2 per output row; clamping at the border only merges indices), and the
// same for columns.  Nine loads per four stores instead of sixteen; the selection below is by index comparison, so
// no assumption about the rounding of the source coordinate is baked in, and every output is computed with exactly
// the expressions of bilinear2x_fwd_kernel (bit-identical: tests/test_gpu_dpt.py).
template <typename T>
__global__ __launch_bounds__(256) void bilinear2x_fwd22_kernel(const T* __restrict__ x, T* __restrict__ y, int H,
                                                               int W, int C) {
    constexpr int V = Vec<T>::N;
    typedef typename Vec<T>::raw raw;
    const unsigned cpr = (unsigned)(C / V), Wo = 2 * W;
    const int Ho = 2 * H;
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    const unsigned j = t / cpr;  // output columns 2j, 2j + 1
    if (j >= (unsigned)W) return;
    const int c = (int)(t - j * cpr) * V;
    const int k = blockIdx.y, b = blockIdx.z;
    int y0a, y1a, y0b, y1b, x0a, x1a, x0b, x1b;
    float wya, wyb, wxa, wxb;
    bl_src(2 * k, H, Ho, y0a, y1a, wya);
    bl_src(2 * k + 1, H, Ho, y0b, y1b, wyb);
    bl_src(2 * (int)j, W, (int)Wo, x0a, x1a, wxa);
    bl_src(2 * (int)j + 1, W, (int)Wo, x0b, x1b, wxb);
    const T* base = x + (size_t)b * H * W * C + c;
    const int rows[3] = {y0a, y1a, y1b}, cols[3] = {x0a, x1a, x1b};
    raw tp[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) tp[r][q] = *(const raw*)(base + ((size_t)rows[r] * W + cols[q]) * C);
    // rows of output row b: (y0b == y1a ? row 1 : row 0 [then y0b == y0a]), (y1b == y1a ? row 1 : row 2); uniform
    const bool rb0 = y0b == y1a, rb1 = y1b == y1a;
    const bool cb0 = x0b == x1a, cb1 = x1b == x1a;   // per lane
    auto lerp_store = [&](const raw& r00, const raw& r01, const raw& r10, const raw& r11, float wx, float wy, int oy,
                          unsigned ox) {
        float f00[V], f01[V], f10[V], f11[V], o[V];
        unpack<T>(r00, f00); unpack<T>(r01, f01); unpack<T>(r10, f10); unpack<T>(r11, f11);
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = bl_lerp(f00[i], f01[i], f10[i], f11[i], wx, wy);
        *(raw*)(y + (((size_t)b * Ho + oy) * Wo + ox) * C + c) = pack<T>(o);
    };
    auto sel = [](bool p, const raw& a, const raw& bb) -> raw {
        raw r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = p ? a[i] : bb[i];
        return r;
    };
    // column taps of output column b, for each of the three rows
    raw lb[3], rbv[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        lb[r] = sel(cb0, tp[r][1], tp[r][0]);
        rbv[r] = sel(cb1, tp[r][1], tp[r][2]);
    }
    const unsigned oxa = 2 * j, oxb = 2 * j + 1;
    // output row a: rows (0, 1)
    lerp_store(tp[0][0], tp[0][1], tp[1][0], tp[1][1], wxa, wya, 2 * k, oxa);
    lerp_store(lb[0], rbv[0], lb[1], rbv[1], wxb, wya, 2 * k, oxb);
    // output row b
    const raw t0a = sel(rb0, tp[1][0], tp[0][0]), t1a = sel(rb0, tp[1][1], tp[0][1]);
    const raw b0a = sel(rb1, tp[1][0], tp[2][0]), b1a = sel(rb1, tp[1][1], tp[2][1]);
    lerp_store(t0a, t1a, b0a, b1a, wxa, wyb, 2 * k + 1, oxa);
    const raw t0b = sel(rb0, lb[1], lb[0]), t1b = sel(rb0, rbv[1], rbv[0]);
    const raw b0b = sel(rb1, lb[1], lb[2]), b1b = sel(rb1, rbv[1], rbv[2]);
    lerp_store(t0b, t1b, b0b, b1b, wxb, wyb, 2 * k + 1, oxb);
}
// (The matching backward — 2 x 2 input pixels per thread, 36 taps instead of 64 — measured slower and is not built:
// tools/experiments/bilinear_bwd22_kernel.hip.txt, profiles/r06d_bilinear_ab.log.)
static bool bilinear22() {  // SSL4GIE_BILINEAR22=0: one output pixel per thread in the forward again (A/B; same bits)
    static int on = -1;
    if (on < 0) { const char* e = getenv("SSL4GIE_BILINEAR22"); on = (e && e[0] == '0') ? 0 : 1; }
    return on != 0;
}

// ------------------------------------------------------------------ ConvTranspose2d(k = s) scatter
// g [B*H*W, k*k*C] with column order (i, j, c)  ->  y[b, k*y + i, k*x + j, c] = g + bias[c]
template <typename T>
__global__ void pixel_shuffle_kernel(const T* __restrict__ g, const float* __restrict__ bias,
                                     T* __restrict__ y, int B, int H, int W, int k, int C,
                                     long long total) {
    constexpr int V = Vec<T>::N;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cpr = C / V, Ho = k * H, Wo = k * W;
    const long long p = idx / cpr;  // output pixel
    const int c = (int)(idx % cpr) * V;
    const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), b = (int)(p / ((long long)Wo * Ho));
    const int yy = oy / k, i = oy % k, xx = ox / k, j = ox % k;
    const size_t m = ((size_t)b * H + yy) * W + xx;
    float f[V];
    unpack<T>(*(const typename Vec<T>::raw*)(g + m * ((size_t)k * k * C) + (size_t)(i * k + j) * C + c), f);
    if (bias) {
#pragma unroll
        for (int q = 0; q < V; ++q) f[q] += bias[c + q];
    }
    *(typename Vec<T>::raw*)(y + (size_t)p * C + c) = pack<T>(f);
}
template <typename T>
__global__ void pixel_unshuffle_kernel(const T* __restrict__ dy, T* __restrict__ dg, int B, int H,
                                       int W, int k, int C, long long total) {
    constexpr int V = Vec<T>::N;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cpr = C / V, Ho = k * H, Wo = k * W;
    const long long p = idx / cpr;
    const int c = (int)(idx % cpr) * V;
    const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), b = (int)(p / ((long long)Wo * Ho));
    const int yy = oy / k, i = oy % k, xx = ox / k, j = ox % k;
    const size_t m = ((size_t)b * H + yy) * W + xx;
    *(typename Vec<T>::raw*)(dg + m * ((size_t)k * k * C) + (size_t)(i * k + j) * C + c) =
        *(const typename Vec<T>::raw*)(dy + (size_t)p * C + c);
}

// ------------------------------------------------------------------ tokens <-> map
// z fp32 [B, 1 + L, D] (residual stream tap) -> x [B*L, D] operand type, cls row dropped (Slice(1))
template <typename T>
__global__ void tokens_to_map_kernel(const float* __restrict__ z, T* __restrict__ x, int L, int D,
                                     long long total) {
    const long long idx = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (idx >= total) return;
    const long long row = idx / D;
    const int c = (int)(idx % D);
    const long long b = row / L, l = row % L;
    st4(x + idx, ld4(z + ((size_t)b * (L + 1) + 1 + l) * D + c));
}
// dz[b, 0, :] = 0; dz[b, 1 + l, :] = dx[b*L + l, :]
template <typename T>
__global__ void map_to_tokens_kernel(const T* __restrict__ dx, float* __restrict__ dz, int L, int D,
                                     long long total) {
    const long long idx = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (idx >= total) return;
    const long long row = idx / D;
    const int c = (int)(idx % D);
    const long long b = row / (L + 1), t = row % (L + 1);
    f32x4 v = {0, 0, 0, 0};
    if (t > 0) v = ld4(dx + ((size_t)b * L + t - 1) * D + c);
    st4(dz + idx, v);
}

// ------------------------------------------------------------------ element-wise
// op 0: out = a + b          op 1: out = (a > 0 ? b : 0) + (c ? c : 0)   (ReLU backward + skip)
template <typename T>
__global__ void eltwise_kernel(int op, const T* __restrict__ a, const T* __restrict__ b,
                               const T* __restrict__ c, T* __restrict__ out, long long total) {
    constexpr int V = Vec<T>::N;
    const long long idx = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (idx >= total) return;
    float fa[V], fb[V], fo[V];
    unpack<T>(*(const typename Vec<T>::raw*)(a + idx), fa);
    unpack<T>(*(const typename Vec<T>::raw*)(b + idx), fb);
    if (op == 0) {
#pragma unroll
        for (int j = 0; j < V; ++j) fo[j] = fa[j] + fb[j];
    } else {
#pragma unroll
        for (int j = 0; j < V; ++j) fo[j] = fa[j] > 0.f ? fb[j] : 0.f;
        if (c) {
            float fc[V];
            unpack<T>(*(const typename Vec<T>::raw*)(c + idx), fc);
#pragma unroll
            for (int j = 0; j < V; ++j) fo[j] += fc[j];
        }
    }
    *(typename Vec<T>::raw*)(out + idx) = pack<T>(fo);
}

// ------------------------------------------------------------------ depth head
// y[m] = sigmoid(sum_c relu(x[m, c]) w[c] + bias)   (ReLU -> Conv2d(32, 1, 1) -> Sigmoid, :479-481)
template <typename T>
__global__ void depth_head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                      const float* __restrict__ bias, float* __restrict__ y,
                                      long long M, int C) {
    constexpr int V = Vec<T>::N;
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float s = bias[0];
    for (int c = 0; c < C; c += V) {
        float f[V];
        unpack<T>(*(const typename Vec<T>::raw*)(x + (size_t)m * C + c), f);
#pragma unroll
        for (int j = 0; j < V; ++j) s += (f[j] > 0.f ? f[j] : 0.f) * w[c + j];
    }
    y[m] = 1.f / (1.f + __expf(-s));
}
// ds = dy * y (1 - y); dx[m, c] = x > 0 ? ds w[c] : 0; partial[blk][c] = sum ds relu(x), [C] = sum ds
template <typename T>
__global__ __launch_bounds__(256) void depth_head_bwd_kernel(
    const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ y,
    const float* __restrict__ dy, T* __restrict__ dx, float* __restrict__ partial, long long M,
    int C) {
    constexpr int V = Vec<T>::N;
    __shared__ float red[4][65];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float aw[64];  // C <= 64
#pragma unroll
    for (int c = 0; c < 64; ++c) aw[c] = 0.f;
    float ab = 0.f;
    for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < M;
         m += (long long)gridDim.x * blockDim.x) {
        const float yy = y[m];
        const float ds = dy[m] * yy * (1.f - yy);
        ab += ds;
#pragma unroll
        for (int c0 = 0; c0 < 64; c0 += V) {
            if (c0 < C) {
                float f[V], o[V];
                unpack<T>(*(const typename Vec<T>::raw*)(x + (size_t)m * C + c0), f);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const bool on = f[j] > 0.f;
                    o[j] = on ? ds * w[c0 + j] : 0.f;
                    aw[c0 + j] += on ? ds * f[j] : 0.f;
                }
                *(typename Vec<T>::raw*)(dx + (size_t)m * C + c0) = pack<T>(o);
            }
        }
    }
    // block reduction of the C + 1 sums: wave shuffle, then the 4 waves through LDS
#pragma unroll
    for (int c = 0; c < 64; ++c) {
        if (c < C) {
            const float s = wave_sum(aw[c]);
            if (lane == 0) red[wave][c] = s;
        }
    }
    {
        const float s = wave_sum(ab);
        if (lane == 0) red[wave][64] = s;
    }
    __syncthreads();
    if (threadIdx.x <= C) {
        const int c = threadIdx.x == C ? 64 : threadIdx.x;
        partial[(size_t)blockIdx.x * (C + 1) + threadIdx.x] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    }
}

// =====================================================================================
// C ABI
// =====================================================================================
static bool dt_ok(int dt) { return dt == SSL4GIE_F32 || dt == SSL4GIE_BF16; }
static int vecn(int dt) { return dt == SSL4GIE_BF16 ? 8 : 4; }
#define DPT_LAUNCH(dtype, KERNEL, total, ...)                                                      \
    do {                                                                                           \
        if ((dtype) == SSL4GIE_BF16) {                                                             \
            typedef bf16_t T;                                                                      \
            hipLaunchKernelGGL(KERNEL<T>, DPT_GRID(total), 0, st, __VA_ARGS__);                    \
        } else {                                                                                   \
            typedef float T;                                                                       \
            hipLaunchKernelGGL(KERNEL<T>, DPT_GRID(total), 0, st, __VA_ARGS__);                    \
        }                                                                                          \
        LAUNCH_CHECK();                                                                            \
    } while (0)

extern "C" int ssl4gie_im2col3x3(const void* x, void* cols, int dtype, int B, int H, int W, int C,
                                 int stride, int relu, long long ld, void* stream) {
    REQUIRE(x && cols && dt_ok(dtype) && B > 0 && H > 0 && W > 0 && C > 0);
    REQUIRE((stride == 1 || stride == 2) && C % vecn(dtype) == 0 && ld >= 9LL * C &&
            ld % vecn(dtype) == 0);
    hipStream_t st = (hipStream_t)stream;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long long total = (long long)B * Ho * Wo * (ld / vecn(dtype));
    DPT_LAUNCH(dtype, im2col3x3_kernel, total, (const T*)x, (T*)cols, B, H, W, C, Ho, Wo, stride,
               relu, ld, total);
    return 0;
}
extern "C" int ssl4gie_col2im3x3(const void* dcols, void* dx, int dtype, int B, int H, int W, int C,
                                 int stride, long long ld, void* stream) {
    REQUIRE(dcols && dx && dt_ok(dtype) && B > 0 && H > 0 && W > 0 && C > 0);
    REQUIRE((stride == 1 || stride == 2) && C % vecn(dtype) == 0 && ld >= 9LL * C);
    hipStream_t st = (hipStream_t)stream;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long long total = (long long)B * H * W * (C / vecn(dtype));
    DPT_LAUNCH(dtype, col2im3x3_kernel, total, (const T*)dcols, (T*)dx, B, H, W, C, Ho, Wo, stride,
               ld, total);
    return 0;
}
extern "C" int ssl4gie_bilinear2x_fwd(const void* x, void* y, int dtype, int B, int H, int W, int C,
                                      void* stream) {
    REQUIRE(x && y && dt_ok(dtype) && B > 0 && H > 0 && W > 0 && C > 0 && C % vecn(dtype) == 0);
    hipStream_t st = (hipStream_t)stream;
    REQUIRE(2 * H <= 65535 && B <= 65535);
    if (bilinear22()) {
        const unsigned per = (unsigned)W * (unsigned)(C / vecn(dtype));
        const dim3 grid((per + 255) / 256, H, B), block(256);
        if (dtype == SSL4GIE_BF16)
            hipLaunchKernelGGL(bilinear2x_fwd22_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, (bf16_t*)y, H, W, C);
        else
            hipLaunchKernelGGL(bilinear2x_fwd22_kernel<float>, grid, block, 0, st, (const float*)x, (float*)y, H, W, C);
        LAUNCH_CHECK();
        return 0;
    }
    const unsigned per_row = (unsigned)(2 * W) * (unsigned)(C / vecn(dtype));
    const dim3 grid((per_row + 255) / 256, 2 * H, B), block(256);
    if (dtype == SSL4GIE_BF16)
        hipLaunchKernelGGL(bilinear2x_fwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, (bf16_t*)y, H, W, C);
    else
        hipLaunchKernelGGL(bilinear2x_fwd_kernel<float>, grid, block, 0, st, (const float*)x, (float*)y, H, W, C);
    LAUNCH_CHECK();
    return 0;
}
extern "C" int ssl4gie_bilinear2x_bwd(const void* dy, void* dx, int dtype, int B, int H, int W,
                                      int C, void* stream) {
    REQUIRE(dy && dx && dt_ok(dtype) && B > 0 && H > 0 && W > 0 && C > 0 && C % vecn(dtype) == 0);
    hipStream_t st = (hipStream_t)stream;
    REQUIRE(H <= 65535 && B <= 65535);
    const unsigned per_row = (unsigned)W * (unsigned)(C / vecn(dtype));
    const dim3 grid((per_row + 255) / 256, H, B), block(256);
    if (dtype == SSL4GIE_BF16)
        hipLaunchKernelGGL(bilinear2x_bwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)dy, (bf16_t*)dx, H, W, C);
    else
        hipLaunchKernelGGL(bilinear2x_bwd_kernel<float>, grid, block, 0, st, (const float*)dy, (float*)dx, H, W, C);
    LAUNCH_CHECK();
    return 0;
}
extern "C" int ssl4gie_pixel_shuffle(const void* g, const float* bias, void* y, int dtype, int B,
                                     int H, int W, int k, int C, void* stream) {
    REQUIRE(g && y && dt_ok(dtype) && B > 0 && H > 0 && W > 0 && k > 0 && C > 0 &&
            C % vecn(dtype) == 0);
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)B * k * H * k * W * (C / vecn(dtype));
    DPT_LAUNCH(dtype, pixel_shuffle_kernel, total, (const T*)g, bias, (T*)y, B, H, W, k, C, total);
    return 0;
}
extern "C" int ssl4gie_pixel_unshuffle(const void* dy, void* dg, int dtype, int B, int H, int W,
                                       int k, int C, void* stream) {
    REQUIRE(dy && dg && dt_ok(dtype) && B > 0 && H > 0 && W > 0 && k > 0 && C > 0 &&
            C % vecn(dtype) == 0);
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)B * k * H * k * W * (C / vecn(dtype));
    DPT_LAUNCH(dtype, pixel_unshuffle_kernel, total, (const T*)dy, (T*)dg, B, H, W, k, C, total);
    return 0;
}
extern "C" int ssl4gie_tokens_to_map(const float* z, void* x, int dtype, int B, int L, int D,
                                     void* stream) {
    REQUIRE(z && x && dt_ok(dtype) && B > 0 && L > 0 && D > 0 && D % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)B * L * D;
    DPT_LAUNCH(dtype, tokens_to_map_kernel, total / 4, z, (T*)x, L, D, total);
    return 0;
}
extern "C" int ssl4gie_map_to_tokens(const void* dx, float* dz, int dtype, int B, int L, int D,
                                     void* stream) {
    REQUIRE(dx && dz && dt_ok(dtype) && B > 0 && L > 0 && D > 0 && D % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)B * (L + 1) * D;
    DPT_LAUNCH(dtype, map_to_tokens_kernel, total / 4, (const T*)dx, dz, L, D, total);
    return 0;
}
extern "C" int ssl4gie_eltwise(int op, const void* a, const void* b, const void* c, void* out,
                               int dtype, long long n, void* stream) {
    REQUIRE(a && b && out && dt_ok(dtype) && n >= 0 && n % vecn(dtype) == 0 && (op == 0 || op == 1));
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    DPT_LAUNCH(dtype, eltwise_kernel, n / vecn(dtype), op, (const T*)a, (const T*)b, (const T*)c,
               (T*)out, n);
    return 0;
}
extern "C" int ssl4gie_depth_head_fwd(const void* x, const float* w, const float* bias, float* y,
                                      int dtype, long long M, int C, void* stream) {
    REQUIRE(x && w && bias && y && dt_ok(dtype) && M > 0 && C > 0 && C <= 64 && C % vecn(dtype) == 0);
    hipStream_t st = (hipStream_t)stream;
    DPT_LAUNCH(dtype, depth_head_fwd_kernel, M, (const T*)x, w, bias, y, M, C);
    return 0;
}
static int head_blocks(long long M) {
    long long b = (M + 255) / 256;
    return (int)(b < 1024 ? b : 1024);
}
extern "C" size_t ssl4gie_depth_head_bwd_workspace_bytes(long long M, int C) {
    return (size_t)head_blocks(M) * (C + 1) * sizeof(float);
}
// dw [C] and db [1] are overwritten (accumulate = 0) or accumulated
extern "C" int ssl4gie_depth_head_bwd(const void* x, const float* w, const float* y,
                                      const float* dy, void* dx, float* dw, float* db,
                                      int accumulate, float* workspace, int dtype, long long M,
                                      int C, void* stream) {
    REQUIRE(x && w && y && dy && dx && dw && db && workspace && dt_ok(dtype) && M > 0 && C > 0 &&
            C <= 64 && C % vecn(dtype) == 0);
    hipStream_t st = (hipStream_t)stream;
    const int nb = head_blocks(M);
    if (dtype == SSL4GIE_BF16)
        hipLaunchKernelGGL(depth_head_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, st,
                           (const bf16_t*)x, w, y, dy, (bf16_t*)dx, workspace, M, C);
    else
        hipLaunchKernelGGL(depth_head_bwd_kernel<float>, dim3(nb), dim3(256), 0, st,
                           (const float*)x, w, y, dy, (float*)dx, workspace, M, C);
    LAUNCH_CHECK();
    int rc = ssl4gie_internal_reduce_partials(workspace, dw, nb, C, (size_t)(C + 1), accumulate, st);
    if (rc) return rc;
    return ssl4gie_internal_reduce_partials(workspace + C, db, nb, 1, (size_t)(C + 1), accumulate, st);
}

// ---- operand images of a Conv2d(k = 3) weight ---------------------------------------------------------
// w [Cout][Cin][3][3] fp32 (the parameter) -> the GEMM / direct-kernel operand in ONE launch (cast included):
//   mode 0  out[co][tap * Cin + ci]            row stride ld >= 9 Cin   (forward, weight-gradient layout)
//   mode 1  out[ci][(8 - tap) * Cout + co]     row stride ld >= 9 Cout  (data gradient: flipped kernel)
//   mode 2  out[tap * Cin + ci][co]            rows ld >= 9 Cin         (transpose of mode 0: stride-2 data gradient)
// padding columns / rows are written as zeros.  Replaces a torch permute-copy (+ flip) + cast per operand and
// optimizer step (reference: the cuDNN filter transforms behind nn.Conv2d, Models/DPT_decoder.py:212-233,
// torchvision Bottleneck.conv2).
template <typename TO>
__global__ void conv3x3_weight_pack_kernel(const float* __restrict__ w, TO* __restrict__ out, int Cout, int Cin,
                                           int mode, int ld, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float v = 0.f;
    if (mode == 0) {
        const int co = (int)(i / ld), k = (int)(i % ld);
        if (k < 9 * Cin) { const int tap = k / Cin, ci = k - tap * Cin; v = w[((size_t)co * Cin + ci) * 9 + tap]; }
    } else if (mode == 1) {
        const int ci = (int)(i / ld), k = (int)(i % ld);
        if (k < 9 * Cout) { const int ft = k / Cout, co = k - ft * Cout; v = w[((size_t)co * Cin + ci) * 9 + (8 - ft)]; }
    } else {
        const int k = (int)(i / Cout), co = (int)(i % Cout);
        if (k < 9 * Cin) { const int tap = k / Cin, ci = k - tap * Cin; v = w[((size_t)co * Cin + ci) * 9 + tap]; }
    }
    if constexpr (sizeof(TO) == 2) out[i] = f2bf(v);
    else out[i] = v;
}

extern "C" int ssl4gie_conv3x3_weight_pack(const float* w, void* out, int dtype, int Cout, int Cin, int mode, int ld,
                                           void* stream) {
    REQUIRE(w && out && Cout > 0 && Cin > 0 && mode >= 0 && mode <= 2 && (dtype == SSL4GIE_BF16 || dtype == SSL4GIE_F32));
    REQUIRE(ld >= 9 * (mode == 1 ? Cout : Cin));
    const long long total = mode == 0 ? (long long)Cout * ld : (mode == 1 ? (long long)Cin * ld : (long long)ld * Cout);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (dtype == SSL4GIE_BF16)
        hipLaunchKernelGGL(conv3x3_weight_pack_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                           (bf16_t*)out, Cout, Cin, mode, ld, total);
    else
        hipLaunchKernelGGL(conv3x3_weight_pack_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                           (float*)out, Cout, Cin, mode, ld, total);
    LAUNCH_CHECK();
    return 0;
}

// ... and for MANY weights in one launch (every 3x3 convolution of a ResNet-50 re-packs two or three operand
// images per optimizer step: 61 launches of ~6 us in front of their convolutions): the items ride in the kernel
// arguments, a block finds its item by bisection of the block-count prefix sums
#define PACK_MAX 64
struct PackBatch {
    const float* w[PACK_MAX];
    void* out[PACK_MAX];
    int Cout[PACK_MAX], Cin[PACK_MAX], mode[PACK_MAX], ld[PACK_MAX];
    unsigned first[PACK_MAX + 1];  // first block of item i; first[n] = total blocks
    int n;
};
template <typename TO>
__global__ void conv3x3_weight_pack_batch_kernel(const PackBatch b) {
    int lo = 0, hi = b.n;  // first[lo] <= blockIdx.x < first[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (blockIdx.x >= b.first[mid]) lo = mid; else hi = mid;
    }
    const int Cout = b.Cout[lo], Cin = b.Cin[lo], mode = b.mode[lo], ld = b.ld[lo];
    const float* __restrict__ w = b.w[lo];
    TO* __restrict__ out = (TO*)b.out[lo];
    const long long total = mode == 0 ? (long long)Cout * ld : (mode == 1 ? (long long)Cin * ld : (long long)ld * Cout);
    const long long i = (long long)(blockIdx.x - b.first[lo]) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float v = 0.f;
    if (mode == 0) {
        const int co = (int)(i / ld), k = (int)(i % ld);
        if (k < 9 * Cin) { const int tap = k / Cin, ci = k - tap * Cin; v = w[((size_t)co * Cin + ci) * 9 + tap]; }
    } else if (mode == 1) {
        const int ci = (int)(i / ld), k = (int)(i % ld);
        if (k < 9 * Cout) { const int ft = k / Cout, co = k - ft * Cout; v = w[((size_t)co * Cin + ci) * 9 + (8 - ft)]; }
    } else {
        const int k = (int)(i / Cout), co = (int)(i % Cout);
        if (k < 9 * Cin) { const int tap = k / Cin, ci = k - tap * Cin; v = w[((size_t)co * Cin + ci) * 9 + tap]; }
    }
    if constexpr (sizeof(TO) == 2) out[i] = f2bf(v);
    else out[i] = v;
}
extern "C" int ssl4gie_conv3x3_weight_pack_batch(const void* const* w, void* const* out, const int* Cout,
                                                 const int* Cin, const int* mode, const int* ld, int n, int dtype,
                                                 void* stream) {
    REQUIRE(w && out && Cout && Cin && mode && ld && n >= 0 && (dtype == SSL4GIE_BF16 || dtype == SSL4GIE_F32));
    for (int base = 0; base < n; base += PACK_MAX) {
        PackBatch b;
        b.n = n - base < PACK_MAX ? n - base : PACK_MAX;
        unsigned blocks = 0;
        for (int i = 0; i < b.n; ++i) {
            const int j = base + i;
            REQUIRE(w[j] && out[j] && Cout[j] > 0 && Cin[j] > 0 && mode[j] >= 0 && mode[j] <= 2 &&
                    ld[j] >= 9 * (mode[j] == 1 ? Cout[j] : Cin[j]));
            b.w[i] = (const float*)w[j]; b.out[i] = out[j];
            b.Cout[i] = Cout[j]; b.Cin[i] = Cin[j]; b.mode[i] = mode[j]; b.ld[i] = ld[j];
            const long long total = mode[j] == 0 ? (long long)Cout[j] * ld[j]
                                  : (mode[j] == 1 ? (long long)Cin[j] * ld[j] : (long long)ld[j] * Cout[j]);
            b.first[i] = blocks;
            blocks += (unsigned)((total + 255) / 256);
        }
        b.first[b.n] = blocks;
        if (!blocks) continue;
        if (dtype == SSL4GIE_BF16)
            hipLaunchKernelGGL(conv3x3_weight_pack_batch_kernel<bf16_t>, dim3(blocks), dim3(256), 0,
                               (hipStream_t)stream, b);
        else
            hipLaunchKernelGGL(conv3x3_weight_pack_batch_kernel<float>, dim3(blocks), dim3(256), 0,
                               (hipStream_t)stream, b);
        LAUNCH_CHECK();
    }
    return 0;
}

// the inverse for the weight GRADIENT: dw2 [Cout][ld] fp32 (columns (tap, ci), as the TN product / the direct
// kernels deliver it) -> (+)= dW [Cout][Cin][3][3], the parameter's layout, in one launch
__global__ void conv3x3_wgrad_unpack_kernel(const float* __restrict__ dw2, float* __restrict__ dw, int Cout, int Cin,
                                            int ld, int accumulate, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int tap = (int)(i % 9);
    const long long r = i / 9;
    const int ci = (int)(r % Cin), co = (int)(r / Cin);
    const float v = dw2[(size_t)co * ld + tap * Cin + ci];
    dw[i] = accumulate ? dw[i] + v : v;
}

extern "C" int ssl4gie_conv3x3_wgrad_unpack(const float* dw2, float* dw, int Cout, int Cin, int ld, int accumulate,
                                            void* stream) {
    REQUIRE(dw2 && dw && Cout > 0 && Cin > 0 && ld >= 9 * Cin);
    const long long total = (long long)Cout * Cin * 9;
    hipLaunchKernelGGL(conv3x3_wgrad_unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, dw2, dw, Cout, Cin, ld, accumulate, total);
    LAUNCH_CHECK();
    return 0;
}
