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
DEVI void st4(float* p, f32x4 v) { *(f32x4*)p = v; }
DEVI void st4(bf16_t* p, f32x4 v) {
    u32x2 r;
    r[0] = pack_bf2(v[0], v[1]);
    r[1] = pack_bf2(v[2], v[3]);
    *(u32x2*)p = r;
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
