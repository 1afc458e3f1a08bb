// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  Wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define DEVI __device__ __forceinline__

// ---- bf16 <-> f32 -------------------------------------------------------------------
DEVI float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// plain cast -> hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving; MI355X_MICROARCH.md)
DEVI bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
DEVI uint32_t pack_bf2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static DEVI float ld(const float* p) { return *p; }
    static DEVI void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static DEVI float ld(const bf16_t* p) { return bf2f(*p); }
    static DEVI void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// load / store 4 consecutive elements as fp32 (pointer must be 4-element aligned)
DEVI f32x4 ld4(const float* p) { return *(const f32x4*)p; }
DEVI f32x4 ld4(const bf16_t* p) {
    u32x2 r = *(const u32x2*)p;
    f32x4 o;
    o[0] = __uint_as_float(r[0] << 16);
    o[1] = __uint_as_float(r[0] & 0xffff0000u);
    o[2] = __uint_as_float(r[1] << 16);
    o[3] = __uint_as_float(r[1] & 0xffff0000u);
    return o;
}
// flat work-item index -> (channel vector, x, y, image) of an NHWC map with cpr vectors per pixel.  The 64-bit
// divisions of the general form cost more than the rest of a 16-byte-per-thread kernel (the bilinear kernels ran at
// half of the HBM rate because of them: profiles/r04dw); whenever the item count fits 32 bits — every map of the
// bench configurations — three unsigned 32-bit divisions do.  `total` is uniform, so is the branch.
struct MapIdx { int cv, x, y; long long b, p; };
DEVI MapIdx map_index(long long idx, long long total, int cpr, int W, int H) {
    MapIdx m;
    if (total <= 0xffffffffLL) {
        const unsigned i = (unsigned)idx, p = i / (unsigned)cpr, q = p / (unsigned)W;
        m.cv = (int)(i - p * (unsigned)cpr);
        m.x = (int)(p - q * (unsigned)W);
        const unsigned b = q / (unsigned)H;
        m.y = (int)(q - b * (unsigned)H);
        m.b = b;
        m.p = p;
    } else {
        const long long p = idx / cpr, q = p / W;
        m.cv = (int)(idx - p * cpr);
        m.x = (int)(p - q * W);
        m.b = q / H;
        m.y = (int)(q - m.b * H);
        m.p = p;
    }
    return m;
}
DEVI void st4(float* p, f32x4 v) { *(f32x4*)p = v; }
DEVI void st4(bf16_t* p, f32x4 v) {
    u32x2 r;
    r[0] = pack_bf2(v[0], v[1]);
    r[1] = pack_bf2(v[2], v[3]);
    *(u32x2*)p = r;
}

// eight bf16 values through y = act(x a + b) and back to bf16 — EXACTLY bn_apply_kernel's arithmetic (one fused
// multiply-add, the select, one round-to-nearest-even), so that a consumer applying a BatchNorm on the way in
// sees the values the separate pass would have stored
DEVI u32x4 bn_affine_act8(const u32x4 x, const f32x4 (&a)[2], const f32x4 (&b)[2], const int relu) {
    u32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float lo = __uint_as_float(x[j] << 16), hi = __uint_as_float(x[j] & 0xffff0000u);
        float u = lo * a[j >> 1][(2 * j) & 3] + b[j >> 1][(2 * j) & 3];
        float v = hi * a[j >> 1][(2 * j + 1) & 3] + b[j >> 1][(2 * j + 1) & 3];
        u = (relu && u < 0.f) ? 0.f : u;
        v = (relu && v < 0.f) ? 0.f : v;
        r[j] = pack_bf2(u, v);
    }
    return r;
}

// ---- wave reductions (64 lanes) -------------------------------------------------------
DEVI float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
DEVI float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact (erf) GELU and its derivative — nn.GELU() default used by timm Mlp
DEVI float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
DEVI float dgelu_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// bf16-path variants: erf by Abramowitz-Stegun 7.1.26 (|err| < 2e-7 + fast-exp error), sharing
// exp(-x^2/2) between the cdf and the pdf; ~3x fewer VALU ops than erff.  Outputs are rounded to
// bf16 (rel 4e-3) anyway.  The fp32 parity path keeps the exact erff forms above.
DEVI void gelu_parts_fast(float x, float& cdf, float& xpdf) {
    const float e = __expf(-0.5f * x * x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * 0.70710678118654752440f * fabsf(x));
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t
                        + 0.254829592f) * t;
    const float erf_abs = 1.0f - poly * e;
    cdf = 0.5f + 0.5f * copysignf(erf_abs, x);
    xpdf = x * e * 0.39894228040143267794f;
}
// Four elements at once, as two float2 chains evaluated side by side: every multiply / fma of the polynomial and of
// the final combinations is a v_pk_*_f32 on a pair, and the two pairs give the scheduler independent chains to
// interleave (a lone dependent chain of packed ops costs an s_nop after every instruction).  The GELU epilogue of
// the 256 x 256 NT kernel is VALU-issue-bound: 35 -> 24 instructions per pair (profiles/r04*).  Same formulas
// as gelu_parts_fast (A&S 7.1.26 with the 1/2 folded into the coefficients); d = gelu'(x), g = gelu(x).
DEVI void gelu_grad4_fast(const f32x4 x, f32x4& d, f32x4& g) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    const float P = 0.3275911f * 0.70710678118654752440f;
    const f2 x0 = {x[0], x[1]}, x1 = {x[2], x[3]};
    const float C = -0.5f * 1.44269504088896340736f;  // exp(-x^2/2) = exp2(C x^2)
    const f2 a0 = (x0 * x0) * C, a1 = (x1 * x1) * C;
    const f2 e0 = {__builtin_amdgcn_exp2f(a0[0]), __builtin_amdgcn_exp2f(a0[1])};
    const f2 e1 = {__builtin_amdgcn_exp2f(a1[0]), __builtin_amdgcn_exp2f(a1[1])};
    const f2 t0 = {__builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x0[0]), P, 1.0f)),
                   __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x0[1]), P, 1.0f))};
    const f2 t1 = {__builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x1[0]), P, 1.0f)),
                   __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x1[1]), P, 1.0f))};
    const float A5 = 0.5f * 1.061405429f, A4 = -0.5f * 1.453152027f, A3 = 0.5f * 1.421413741f,
                A2 = -0.5f * 0.284496736f, A1 = 0.5f * 0.254829592f;
    f2 p0 = t0 * A5 + A4, p1 = t1 * A5 + A4;
    p0 = p0 * t0 + A3; p1 = p1 * t1 + A3;
    p0 = p0 * t0 + A2; p1 = p1 * t1 + A2;
    p0 = p0 * t0 + A1; p1 = p1 * t1 + A1;
    p0 = p0 * t0;      p1 = p1 * t1;
    f2 h0 = 0.5f - p0 * e0, h1 = 0.5f - p1 * e1;  // Phi(|x|) - 1/2
    h0 = f2{copysignf(h0[0], x0[0]), copysignf(h0[1], x0[1])};
    h1 = f2{copysignf(h1[0], x1[0]), copysignf(h1[1], x1[1])};
    const f2 c0 = h0 + 0.5f, c1 = h1 + 0.5f;      // Phi(x)
    const float S = 0.39894228040143267794f;
    const f2 d0 = (x0 * e0) * S + c0, d1 = (x1 * e1) * S + c1;
    const f2 g0 = x0 * c0, g1 = x1 * c1;
    d = f32x4{d0[0], d0[1], d1[0], d1[1]};
    g = f32x4{g0[0], g0[1], g1[0], g1[1]};
}
DEVI float gelu_fast(float x) { float c, p; gelu_parts_fast(x, c, p); return x * c; }
DEVI float dgelu_fast(float x) { float c, p; gelu_parts_fast(x, c, p); return c + p; }

// XCD-aware bijective block remap (cdna_hip_programming.md §5 / T1): consecutive logical
// tile ids land on the same XCD (blocks b and b+8 share an XCD under round-robin dispatch).
DEVI int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

#define HIP_RET(expr)                                  \
    do {                                               \
        hipError_t e__ = (expr);                       \
        if (e__ != hipSuccess) return (int)e__;        \
    } while (0)
#define LAUNCH_CHECK() HIP_RET(hipGetLastError())
#define ARG_ERR 1000  // invalid-argument code returned by the C ABI
#define REQUIRE(c)                 \
    do {                           \
        if (!(c)) return ARG_ERR;  \
    } while (0)
