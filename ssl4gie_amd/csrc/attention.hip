// Fused multi-head attention forward / backward for short sequences (N <= 256) on gfx950.
//
// Replaces timm Attention.forward (un-vendored) == Models/models.py:195-209 minus the window
// permutation:  softmax(q k^T * hd^-1/2) v  on the packed activation qkv[B, N, 3, H, hd].
//
// bf16 path — one workgroup (4 waves) per (batch, head); the whole K and V of the head are
// staged once into LDS (N=197, hd=64: 2 x 28 KiB -> 2 workgroups per CU), no N x N score
// matrix ever reaches HBM:
//   * scores are computed TRANSPOSED, S^T = K Q^T (v_mfma_f32_16x16x32_bf16), so that a lane owns
//     one query column: the softmax max/sum are in-register reductions + 2 cross-lane shuffles;
//   * the fp32 S^T accumulators, converted to bf16, ARE the B operand of O^T = V^T P^T (the k-slot
//     permutation that implies is applied to the V fragment, which is read out of LDS with the
//     hardware-transposing ds_read_b64_tr_b16) — P never touches LDS;
//   * one XOR-swizzled LDS image serves both row reads (ds_read_b128) and transposed reads
//     (conflict-free for both, verified with tools/lds_bank_sim.py).
// Backward recomputes P from the saved log-sum-exp in two phases inside one launch: phase A
// (waves own query tiles; LDS = K, V) produces dQ; phase B (waves own key tiles; LDS = Q, dO)
// produces dK and dV — so no gradient is ever summed across workgroups (no atomics).
//
// f32 parity path — scores materialised in a caller workspace and driven through the generic
// strided-batched f32-MFMA GEMM + row softmax kernels below (exact fp32).
#include "common.h"
#include "ssl4gie_hip.h"
#include "prof.h"
int ssl4gie_internal_compute_cus();  // gemm.hip: CUs the persistent grids are sized for (ssl4gie_set_compute_cus)

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

// ------------------------------------------------------------------ LDS image helpers
template <int HD> DEVI int kv_swz(int row) { return HD == 64 ? (row & 6) : ((row >> 1) & 2); }
template <int HD> DEVI int img_off(int row, int chunk) {
    return row * (HD * 2) + ((chunk ^ kv_swz<HD>(row)) << 4);
}

// stage a [n, HD] bf16 slice (row stride rs elements) into an image of npad rows (zero padded)
template <int HD>
DEVI void stage_rows(char* img, const bf16_t* src, long long rs, int n, int npad, int tid,
                     int nth = 256) {
    constexpr int CPR = HD / 8;
    for (int idx = tid; idx < npad * CPR; idx += nth) {
        const int row = idx / CPR, c = idx % CPR;
        u32x4 v = {0, 0, 0, 0};
        if (row < n) v = *(const u32x4*)(src + (size_t)row * rs + c * 8);
        *(u32x4*)(img + img_off<HD>(row, c)) = v;
    }
}
// MFMA operand, 16 rows x 32 k: lane l -> row row0 + (l&15), elements k = ks*32 + 8*(l>>4) .. +7
template <int HD> DEVI bf16x8 row_frag(const char* img, int row0, int ks, int lane) {
    return *(const bf16x8*)(img + img_off<HD>(row0 + (lane & 15), ks * 4 + (lane >> 4)));
}
DEVI bf16x8 row_frag_global(const bf16_t* base, long long rs, int row, int ks, int lane) {
    return *(const bf16x8*)(base + (size_t)row * rs + ks * 32 + 8 * (lane >> 4));
}
// transposed MFMA operand over a 32-row block: lane (g = l>>4, i = l&15) receives
//   element jj: img[row0 + 16*(jj>>2) + 4g + (jj&3)][x0 + i]
// i.e. exactly the k-slot order in which two adjacent 16x16 f32 accumulator tiles, packed to
// bf16, present their rows (see pack8 below).
template <int HD> DEVI bf16x8 tr_frag(const char* img, int row0, int x0, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int r0 = row0 + 4 * g + q, r1 = r0 + 16;
    const int c = (x0 >> 3) + (p >> 1);
    typedef __attribute__((address_space(3))) s16x4* lp_t;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(img + img_off<HD>(r0, c) + ((p & 1) << 3)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(img + img_off<HD>(r1, c) + ((p & 1) << 3)));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
DEVI bf16x8 pack8(f32x4 a, f32x4 b) {
    bf16x8 v;
    v[0] = (__bf16)a[0]; v[1] = (__bf16)a[1]; v[2] = (__bf16)a[2]; v[3] = (__bf16)a[3];
    v[4] = (__bf16)b[0]; v[5] = (__bf16)b[1]; v[6] = (__bf16)b[2]; v[7] = (__bf16)b[3];
    return v;
}
DEVI float group_max(float v) {  // across the 4 lane groups (lanes l, l^16, l^32, l^48)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
DEVI float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
// P = exp2(S c - l) and dS = P dPd for four scores of a lane, in packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 handle
// two values per instruction).  dPd = dP - delta comes out of the MFMA: the dP accumulators START at -delta (del_s
// holds the negated row sums), which costs nothing and saves a packed add per two scores in a VALU-bound kernel.
DEVI void softmax_bwd4(const f32x4 s, const f32x4 dpd, const f32x4 l, const float c, f32x4& p, f32x4& ds) {
    const f32x2 c2 = {c, c};
    f32x2 a = f32x2{s[0], s[1]} * c2 - f32x2{l[0], l[1]};
    f32x2 b = f32x2{s[2], s[3]} * c2 - f32x2{l[2], l[3]};
    a[0] = __builtin_amdgcn_exp2f(a[0]); a[1] = __builtin_amdgcn_exp2f(a[1]);
    b[0] = __builtin_amdgcn_exp2f(b[0]); b[1] = __builtin_amdgcn_exp2f(b[1]);
    const f32x2 da = a * f32x2{dpd[0], dpd[1]};
    const f32x2 db = b * f32x2{dpd[2], dpd[3]};
    p = f32x4{a[0], a[1], b[0], b[1]};
    ds = f32x4{da[0], da[1], db[0], db[1]};
}
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
// Output rows leave as 16-byte pieces: a lane holds 4 consecutive columns (4 g ..) of its row in each 16-column
// accumulator block; v_permlane16_swap between the lane groups g and g ^ 1 of two neighbouring blocks gives it 8
// consecutive columns of one of them (gemm256.h's xpose_swap), so a row's 64 B (hd 32) are 4 stores of 16 B instead
// of 8 of 8 B.  Call with all lanes of the wave active; the partner lanes hold the same row.
DEVI u32x4 pair8(const f32x4 a, const f32x4 b) {
    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pack_bf2(a[0], a[1]), pack_bf2(b[0], b[1]), false, false);
    const u32x2 s2 = __builtin_amdgcn_permlane16_swap(pack_bf2(a[2], a[3]), pack_bf2(b[2], b[3]), false, false);
    return u32x4{s1[0], s2[0], s1[1], s2[1]};
}
DEVI int pair8_col(int g) { return 16 * (g & 1) + 8 * (g >> 1); }  // the piece's first column inside the 32-column pair

// three [n, HD] slices (Q, K, V of one head: same row stride, bases D apart) staged in one go: all
// global loads of a thread are issued before the first LDS write, so the prologue costs one HBM
// round trip instead of one per 16-byte chunk (the per-image loop above waits for each load before
// it stores)
template <int HD, int NPAD, int NTH>
DEVI void stage_qkv(char* img0, char* img1, char* img2, const bf16_t* src0, const bf16_t* src1,
                    const bf16_t* src2, long long rs, int n, int tid) {
    constexpr int CPR = HD / 8, TOTAL = NPAD * CPR, IT = (TOTAL + NTH - 1) / NTH;
    u32x4 v[3][IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int idx = tid + i * NTH, row = idx / CPR, c = idx % CPR;
        const bool ok = idx < TOTAL && row < n;
        const size_t off = (size_t)row * rs + c * 8;
        v[0][i] = ok ? *(const u32x4*)(src0 + off) : u32x4{0, 0, 0, 0};
        v[1][i] = ok ? *(const u32x4*)(src1 + off) : u32x4{0, 0, 0, 0};
        v[2][i] = ok ? *(const u32x4*)(src2 + off) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int idx = tid + i * NTH, row = idx / CPR, c = idx % CPR;
        if (idx < TOTAL) {
            *(u32x4*)(img0 + img_off<HD>(row, c)) = v[0][i];
            *(u32x4*)(img1 + img_off<HD>(row, c)) = v[1][i];
            *(u32x4*)(img2 + img_off<HD>(row, c)) = v[2][i];
        }
    }
}

// two slices (K, V) staged the same way
template <int HD, int NPAD, int NTH>
DEVI void stage_kv(char* img0, char* img1, const bf16_t* src0, const bf16_t* src1, long long rs, int n,
                   int tid) {
    constexpr int CPR = HD / 8, TOTAL = NPAD * CPR, IT = (TOTAL + NTH - 1) / NTH;
    u32x4 v[2][IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int idx = tid + i * NTH, row = idx / CPR, c = idx % CPR;
        const bool ok = idx < TOTAL && row < n;
        const size_t off = (size_t)row * rs + c * 8;
        v[0][i] = ok ? *(const u32x4*)(src0 + off) : u32x4{0, 0, 0, 0};
        v[1][i] = ok ? *(const u32x4*)(src1 + off) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int idx = tid + i * NTH, row = idx / CPR, c = idx % CPR;
        if (idx < TOTAL) {
            *(u32x4*)(img0 + img_off<HD>(row, c)) = v[0][i];
            *(u32x4*)(img1 + img_off<HD>(row, c)) = v[1][i];
        }
    }
}

// ------------------------------------------------------------------ forward
// Three images (K, V, Q) of hd 64 above N = 144 would leave room for ONE workgroup per CU, whose
// staging round trip nothing overlaps: there (AttnFwdQG) only K and V are staged — 57 KiB, two
// 4-wave workgroups per CU — and a wave reads its query fragments straight from global memory, one
// query tile ahead of its use (N = 197, hd 64: 120 -> 98 us).  The same split of the BACKWARD
// kernel into a dQ and a dK/dV launch with two images each was measured and changes nothing (304
// vs 305 us): the backward is issue-bound, not latency-bound.
template <int HD, int NKT> struct AttnFwdQG { static constexpr bool value = HD == 64 && NKT >= 10; };
// HT ("half tail"): N <= 16 (NKT - 1), i.e. the last 16-key tile of the last pair is all padding (N = 197: tiles
// 0..12 hold keys, tile 13 none): its scores, maxima and exponentials are skipped and its probabilities are
// zeros (N = 197 costs what N = 224 costs otherwise: profiles/r04db).  The LDS images keep their NKT tiles of
// zero-padded rows, so the pair-wise transposed reads stay in range.  Used where it was measured to pay
// (profiles/r04dd, same box): the hd-32 forward (N = 197: 60 -> 57 us, N = 208: 58.5 -> 53) and the hd-64
// backward (N = 197: 286 -> 271 us); the hd-64 forward variant spills and the hd-32 backward does not move
// (159 us either way: it follows the bytes, not the tile count), so those keep the plain kernels.
template <int HD, int NKT, int WAVES, bool HT>
__global__ __launch_bounds__(64 * WAVES, (WAVES == 8 ? 1 : (HD == 32 && NKT <= 14 ? 3 : 2))) void attn_fwd_bf16_kernel(
    const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ lse, int N, int H,
    float scale) {
    constexpr int NPAD = NKT * 16, RB = HD * 2, KS = HD / 32, DT = HD / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kimg = smem;
    char* Vimg = smem + NPAD * RB;
    char* Qimg = smem + 2 * NPAD * RB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    // consecutive (b, h) ids share an XCD: with hd = 32 two neighbouring heads share every 128-B
    // line of the packed qkv rows, so the second one hits in that XCD's L2
    const int bh = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bh / H, h = bh % H;
    const int D = H * HD;
    const long long rs = 3LL * D;
    const bf16_t* qb = qkv + (size_t)b * N * rs + h * HD;
    constexpr bool QG = AttnFwdQG<HD, NKT>::value;
    if (QG) stage_kv<HD, NPAD, 64 * WAVES>(Kimg, Vimg, qb + D, qb + 2 * D, rs, N, tid);
    else stage_qkv<HD, NPAD, 64 * WAVES>(Kimg, Vimg, Qimg, qb + D, qb + 2 * D, qb, rs, N, tid);
    // QG: this lane's query fragment of tile qt (rows past N repeat the last one; never stored)
    auto load_q = [&](int qt, bf16x8 (&dst)[KS]) {
        int row = qt * 16 + (lane & 15);
        row = row < N ? row : N - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) dst[ks] = row_frag_global(qb, rs, row, ks, lane);
    };
    bf16x8 qnext[KS];
    if (QG && ((wave + bh) & (WAVES - 1)) < ((N + 15) >> 4)) load_q((wave + bh) & (WAVES - 1), qnext);
    __syncthreads();
    const float c = scale * 1.44269504088896340736f;
    const f32x2 c2 = {c, c};
    const int nqt = (N + 15) >> 4;
    // the wave that takes the extra query tile rotates with the head, so that no SIMD of the CU
    // is always the one with the longer share
    for (int qt = (wave + bh) & (WAVES - 1); qt < nqt; qt += WAVES) {
        const int q = qt * 16 + (lane & 15);
        bf16x8 qf[KS];
        if (QG) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[ks] = qnext[ks];
            if (qt + WAVES < nqt) load_q(qt + WAVES, qnext);  // in flight behind this tile's work
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[ks] = row_frag<HD>(Qimg, qt * 16, ks, lane);
        }
        constexpr int NKC = NKT - (HT ? 1 : 0);  // tiles that hold keys
        f32x4 s[NKT];
#pragma unroll
        for (int kt = 0; kt < NKC; ++kt) {
            s[kt] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                s[kt] = MFMA16(row_frag<HD>(Kimg, kt * 16, ks, lane), qf[ks], s[kt]);
        }
        if (HT) s[NKT - 1] = f32x4{0, 0, 0, 0};  // probabilities of the all-padding tile
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKC; ++kt) {
            if (kt >= NKT - 2) {  // 16 (NKT - 2) < N: only the last two tiles hold padded keys
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kt * 16 + 4 * g + r >= N) s[kt][r] = -INFINITY;
            }
            mx = fmaxf(fmaxf(mx, s[kt][0]), s[kt][1]);  // v_max3_f32
            mx = fmaxf(fmaxf(mx, s[kt][2]), s[kt][3]);
        }
        mx = group_max(mx);
        const f32x2 nmc2 = {-mx * c, -mx * c};
        // hd 32: the row sums come out of the matrix pipe — one more product per key pair against an all-ones operand
        // gives sum_k P[k][q] in every row of a 16 x 16 tile (a lane reads its query's sum from element 0): no packed
        // adds per score, no cross-lane reduction, in a kernel bound by its VALU stream (N = 197: 27 + 7 products
        // against ~160 VALU instructions per query tile; forward 59.8 -> 56.5 us, N = 224 72.5 -> 65.6 same-box,
        // profiles/r04ds).  That sum is the one of the bf16-rounded probabilities, i.e. of exactly the values
        // O^T = V^T P^T is accumulated from.  hd 64 has twice the products per score and no slack for it (N = 256:
        // 29.5 -> 31.0 us): it keeps the packed fp32 adds.
        constexpr bool MSUM = HD == 32;
        f32x2 sum2 = {0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NKC; ++kt) {  // packed fp32: one v_pk_fma (/ v_pk_add) per two scores
            f32x2 a = {s[kt][0], s[kt][1]}, bq = {s[kt][2], s[kt][3]};
            a = a * c2 + nmc2;
            bq = bq * c2 + nmc2;
            a[0] = __builtin_amdgcn_exp2f(a[0]); a[1] = __builtin_amdgcn_exp2f(a[1]);
            bq[0] = __builtin_amdgcn_exp2f(bq[0]); bq[1] = __builtin_amdgcn_exp2f(bq[1]);
            if constexpr (!MSUM) {
                sum2 += a;
                sum2 += bq;
            }
            s[kt] = f32x4{a[0], a[1], bq[0], bq[1]};
        }
        const __bf16 one = (__bf16)1.0f;
        const bf16x8 ones = {one, one, one, one, one, one, one, one};
        f32x4 o[DT], osum = {0, 0, 0, 0};
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int kp = 0; kp < NKT / 2; ++kp) {
            const bf16x8 pf = pack8(s[2 * kp], s[2 * kp + 1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                o[dt] = MFMA16(tr_frag<HD>(Vimg, kp * 32, dt * 16, lane), pf, o[dt]);
            if constexpr (MSUM) osum = MFMA16(ones, pf, osum);
        }
        float sum = osum[0];
        if constexpr (!MSUM) sum = group_sum(sum2[0] + sum2[1]);
        const float inv = 1.0f / sum;
        u32x4 ow[DT / 2];
#pragma unroll
        for (int dp = 0; dp < DT / 2; ++dp) ow[dp] = pair8(o[2 * dp] * inv, o[2 * dp + 1] * inv);
        if (q < N) {
            bf16_t* orow = out + ((size_t)b * N + q) * D + h * HD + pair8_col(g);
#pragma unroll
            for (int dp = 0; dp < DT / 2; ++dp) *(u32x4*)(orow + dp * 32) = ow[dp];
            if (g == 0) lse[((size_t)b * H + h) * N + q] = mx * scale + __logf(sum);
        }
        if (NKT <= WAVES) break;  // at most one tile per wave: not a loop
    }
}

// ------------------------------------------------------------------ backward
// LDS holds all four operand images of the head for the whole kernel — K, V, Q, dO — plus the
// saved log-sum-exp and delta_q = sum_d dO[q,d] O[q,d] (computed while dO is staged): after the
// prologue no wave ever waits on a global load inside its per-tile dependency chain (at 2 waves
// per SIMD an exposed ~2 us HBM round trip per query tile used to dominate this kernel).
// Waves per workgroup: each phase hands out NKT (or NKT - 1) 16-row tiles, so the wave count is
// chosen to divide that evenly — one tile per wave up to 8 tiles (never fewer than 4 waves), two per
// wave above (N = 197: 13 tiles on 7 waves instead of 8 waves of which three would get one).  The
// waves of a workgroup share one set of LDS images; 2-4 workgroups fit a CU.
template <int NKT> struct AttnBwdWaves {
    static constexpr int value = NKT <= 4 ? 4 : (NKT <= 8 ? NKT : NKT / 2);
};
// tiles a wave works on at once (see "U tiles per wave" in the kernel): 2 where the LDS images leave room for one
// workgroup per CU anyway, i.e. hd 64 from 10 tiles on (4 images x 160 rows x 128 B = 80 KB)
template <int HD, int NKT> struct AttnBwdU { static constexpr int value = (HD == 64 && NKT >= 10) ? 2 : 1; };
// The two phases of the whole-head backward on staged LDS images (K, V, Q, dO, then lse and -delta rows), shared by the
// one-head-per-workgroup kernel and the persistent prefetching kernel below.  `hook` lets the caller run code at two
// points of phase B (wave-uniform, every wave of the workgroup reaches them once per head): after the wave's own K / V
// fragments are in registers, and before each query pair.
struct AttnBwdNoHook {
    DEVI void after_kv_frags() const {}
    DEVI void before_query_pair(int) const {}
};
template <int HD, int NKT, bool HT, class Hook>
DEVI void attn_bwd_phases(char* smem, bf16_t* __restrict__ dqb, const int N, const long long rs, const int D,
                          const float c, const float scale, const int tid, Hook&& hook) {
    constexpr int NPAD = NKT * 16, RB = HD * 2, KS = HD / 32, DT = HD / 16;
    constexpr int ATTN_BWD_WAVES = AttnBwdWaves<NKT>::value, BWD_U = AttnBwdU<HD, NKT>::value;
    const char* Kimg = smem;
    const char* Vimg = smem + NPAD * RB;
    const char* Qimg = smem + 2 * NPAD * RB;
    const char* Oimg = smem + 3 * NPAD * RB;  // dO
    const float* lse_s = (const float*)(smem + 4 * NPAD * RB);
    const float* del_s = lse_s + NPAD;  // -delta
    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    const int nqt = (N + 15) >> 4;

    // U tiles per wave AT ONCE (register blocking).  hd 64 above 144 tokens needs > 80 KB of LDS: one workgroup per
    // CU, two waves per SIMD at most — 256 VGPRs are there for the taking, and the kernel is bound by its LDS reads
    // (12 KB of fragments per key pair and query tile in phase A, 20 KB in phase B: an LDS floor of ~117 us at
    // N = 197 against 51 us of MFMA).  With its two tiles in flight together a wave reads every K / V (Q / dO)
    // fragment, transposed fragment and statistics row once for both: half the LDS traffic, and two independent
    // dependency chains per wave where two waves per SIMD hide little.
    constexpr int U = BWD_U;
    // ---------------- phase A: dQ (waves own query tiles)
    for (int qt0 = wave; qt0 < nqt; qt0 += ATTN_BWD_WAVES * U) {
        int qt[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            live[u] = qt0 + u * ATTN_BWD_WAVES < nqt;
            qt[u] = live[u] ? qt0 + u * ATTN_BWD_WAVES : qt0;  // a dead slot repeats tile 0's work and stores nothing
        }
        bf16x8 qf[U][KS], dof[U][KS];
        f32x4 l4[U], dl4[U], dq[U][DT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                qf[u][ks] = row_frag<HD>(Qimg, qt[u] * 16, ks, lane);
                dof[u][ks] = row_frag<HD>(Oimg, qt[u] * 16, ks, lane);
            }
            const float dl = del_s[qt[u] * 16 + (lane & 15)];
            const float l2 = lse_s[qt[u] * 16 + (lane & 15)];
            // dS^T needs only the saved row statistics, so key tiles are consumed pair by pair
            l4[u] = f32x4{l2, l2, l2, l2};
            dl4[u] = f32x4{dl, dl, dl, dl};
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dq[u][dt] = f32x4{0, 0, 0, 0};
        }
        // MODE 0: a pair of full tiles; 1: the last pair, padded keys masked; 2: the last pair when its second
        // tile is all padding (HT): one tile's worth of scores, the other half of dS^T is zero
        auto pairA = [&](const int kp, auto modec) {
            constexpr int MODE = decltype(modec)::value;
            f32x4 s0[U], s1[U], p0[U], p1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                s0[u] = s1[u] = f32x4{0, 0, 0, 0};
                p0[u] = p1[u] = dl4[u];  // dP - delta (softmax_bwd4)
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 k0 = row_frag<HD>(Kimg, kp * 32, ks, lane), v0 = row_frag<HD>(Vimg, kp * 32, ks, lane);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    s0[u] = MFMA16(k0, qf[u][ks], s0[u]);
                    p0[u] = MFMA16(v0, dof[u][ks], p0[u]);
                }
                if constexpr (MODE != 2) {
                    const bf16x8 k1 = row_frag<HD>(Kimg, kp * 32 + 16, ks, lane),
                                 v1 = row_frag<HD>(Vimg, kp * 32 + 16, ks, lane);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        s1[u] = MFMA16(k1, qf[u][ks], s1[u]);
                        p1[u] = MFMA16(v1, dof[u][ks], p1[u]);
                    }
                }
            }
            bf16x8 dsf[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                f32x4 pa, pb, dsa, dsb = {0, 0, 0, 0};
                softmax_bwd4(s0[u], p0[u], l4[u], c, pa, dsa);
                if constexpr (MODE != 2) softmax_bwd4(s1[u], p1[u], l4[u], c, pb, dsb);
                if constexpr (MODE != 0) {
                    const int ka = kp * 32 + 4 * g;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dsa[r] = (ka + r < N) ? dsa[r] : 0.f;
                        if constexpr (MODE == 1) dsb[r] = (ka + 16 + r < N) ? dsb[r] : 0.f;
                    }
                }
                dsf[u] = pack8(dsa, dsb);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 kt_ = tr_frag<HD>(Kimg, kp * 32, dt * 16, lane);
#pragma unroll
                for (int u = 0; u < U; ++u) dq[u][dt] = MFMA16(kt_, dsf[u], dq[u][dt]);
            }
        };
        // 16 (NKT - 2) < N: padded keys only in the last pair, which is peeled
#pragma unroll 1
        for (int kp = 0; kp < NKT / 2 - 1; ++kp) pairA(kp, std::integral_constant<int, 0>{});
        pairA(NKT / 2 - 1, std::integral_constant<int, HT ? 2 : 1>{});
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = qt[u] * 16 + (lane & 15);
            u32x4 qw[DT / 2];
#pragma unroll
            for (int dp = 0; dp < DT / 2; ++dp) qw[dp] = pair8(dq[u][2 * dp] * scale, dq[u][2 * dp + 1] * scale);
            if (live[u] && q < N) {
                bf16_t* r = dqb + (size_t)q * rs + pair8_col(g);
#pragma unroll
                for (int dp = 0; dp < DT / 2; ++dp) *(u32x4*)(r + dp * 32) = qw[dp];
            }
        }
        if (NKT <= ATTN_BWD_WAVES * U) break;  // at most U tiles per wave: not a loop
    }

    // ---------------- phase B: dK, dV (waves own key tiles); same LDS images, no restaging
    for (int kt0 = wave; kt0 < nqt; kt0 += ATTN_BWD_WAVES * U) {
        int kt[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            live[u] = kt0 + u * ATTN_BWD_WAVES < nqt;
            kt[u] = live[u] ? kt0 + u * ATTN_BWD_WAVES : kt0;
        }
        bf16x8 kf[U][KS], vf[U][KS];
        f32x4 dk[U][DT], dv[U][DT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                kf[u][ks] = row_frag<HD>(Kimg, kt[u] * 16, ks, lane);
                vf[u][ks] = row_frag<HD>(Vimg, kt[u] * 16, ks, lane);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                dk[u][dt] = f32x4{0, 0, 0, 0};
                dv[u][dt] = f32x4{0, 0, 0, 0};
            }
        }
        hook.after_kv_frags();  // the wave's own K / V fragments are in registers: it no longer reads the K / V images
        // HT: the last pair's second query tile is all padding — its scores are skipped (a wave-uniform branch, not
        // a peeled copy of the body: the copy costs registers and spills at hd 64) and its P / dS are zeros
#pragma unroll 1
        for (int qp = 0; qp < NKT / 2; ++qp) {
            hook.before_query_pair(qp);
            const bool full = !HT || qp < NKT / 2 - 1;
            const int qa = qp * 32 + 4 * g;
            // the dP accumulators start at -delta of their query rows (softmax_bwd4)
            const f32x4 da = *(const f32x4*)(del_s + qa), db = *(const f32x4*)(del_s + qa + 16);
            const f32x4 la = *(const f32x4*)(lse_s + qa), lb = *(const f32x4*)(lse_s + qa + 16);
            f32x4 s0[U], s1[U], p0[U], p1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                s0[u] = s1[u] = f32x4{0, 0, 0, 0};
                p0[u] = da;
                p1[u] = db;
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 q0 = row_frag<HD>(Qimg, qp * 32, ks, lane), o0 = row_frag<HD>(Oimg, qp * 32, ks, lane);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    s0[u] = MFMA16(q0, kf[u][ks], s0[u]);
                    p0[u] = MFMA16(o0, vf[u][ks], p0[u]);
                }
            }
            if (full) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 q1 = row_frag<HD>(Qimg, qp * 32 + 16, ks, lane),
                                 o1 = row_frag<HD>(Oimg, qp * 32 + 16, ks, lane);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        s1[u] = MFMA16(q1, kf[u][ks], s1[u]);
                        p1[u] = MFMA16(o1, vf[u][ks], p1[u]);
                    }
                }
            }
            // lane holds S[q = qp*32 (+16) + 4g + r][key]; p0/p1 hold dP.  Padded queries carry
            // lse = +inf (P = 0) and zero dO / delta rows, so they need no mask here.
            bf16x8 pf[U], dsf[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                f32x4 pa, pb = {0, 0, 0, 0}, dsa, dsb = {0, 0, 0, 0};
                softmax_bwd4(s0[u], p0[u], la, c, pa, dsa);
                if (full) softmax_bwd4(s1[u], p1[u], lb, c, pb, dsb);
                pf[u] = pack8(pa, pb);
                dsf[u] = pack8(dsa, dsb);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 ot = tr_frag<HD>(Oimg, qp * 32, dt * 16, lane), qt_ = tr_frag<HD>(Qimg, qp * 32, dt * 16, lane);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    dv[u][dt] = MFMA16(ot, pf[u], dv[u][dt]);
                    dk[u][dt] = MFMA16(qt_, dsf[u], dk[u][dt]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = kt[u] * 16 + (lane & 15);
            u32x4 kw[DT / 2], vw[DT / 2];
#pragma unroll
            for (int dp = 0; dp < DT / 2; ++dp) {
                kw[dp] = pair8(dk[u][2 * dp] * scale, dk[u][2 * dp + 1] * scale);
                vw[dp] = pair8(dv[u][2 * dp], dv[u][2 * dp + 1]);
            }
            if (live[u] && key < N) {
                bf16_t* r = dqb + (size_t)key * rs + pair8_col(g);
#pragma unroll
                for (int dp = 0; dp < DT / 2; ++dp) {
                    *(u32x4*)(r + D + dp * 32) = kw[dp];
                    *(u32x4*)(r + 2 * D + dp * 32) = vw[dp];
                }
            }
        }
        if (NKT <= ATTN_BWD_WAVES * U) break;
    }
}

// HT as in the forward kernel: the last pair of key tiles (phase A) / query tiles (phase B) is a single tile
template <int HD, int NKT, bool HT>
__global__ __launch_bounds__(64 * AttnBwdWaves<NKT>::value, (AttnBwdU<HD, NKT>::value == 2 ? 2 : 4)) void attn_bwd_bf16_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
    const bf16_t* __restrict__ dout, const float* __restrict__ lse, bf16_t* __restrict__ dqkv,
    int N, int H, float scale) {
    constexpr int NPAD = NKT * 16, RB = HD * 2, CPR = HD / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kimg = smem;
    char* Vimg = smem + NPAD * RB;
    char* Qimg = smem + 2 * NPAD * RB;
    char* Oimg = smem + 3 * NPAD * RB;  // dO
    float* lse_s = (float*)(smem + 4 * NPAD * RB);
    float* del_s = lse_s + NPAD;  // -delta
    const int tid = threadIdx.x;
    // consecutive (b, h) ids share an XCD: with hd = 32 two neighbouring heads share every 128-B
    // line of the packed qkv rows, so the second one hits in that XCD's L2
    const int bh = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bh / H, h = bh % H;
    const int D = H * HD;
    const long long rs = 3LL * D;
    const bf16_t* qb = qkv + (size_t)b * N * rs + h * HD;
    const bf16_t* ob = out + (size_t)b * N * D + h * HD;
    const bf16_t* dob = dout + (size_t)b * N * D + h * HD;
    bf16_t* dqb = dqkv + (size_t)b * N * rs + h * HD;
    const float* lrow = lse + ((size_t)b * H + h) * N;
    const float LOG2E = 1.44269504088896340736f;
    const float c = scale * LOG2E;

    constexpr int ATTN_BWD_WAVES = AttnBwdWaves<NKT>::value, NTH = 64 * ATTN_BWD_WAVES;
    // ---------------- prologue: stage K, V, Q, dO; delta and lse rows.  EVERY global load of the prologue is issued
    // before the first LDS write — one HBM round trip per workgroup instead of three (Q/K/V, then dO/O, then lse:
    // with two workgroups per CU nothing hides them, and this kernel's time follows its bytes: profiles/r04dd)
    {
        constexpr int TOTAL = NPAD * CPR, IT = (TOTAL + NTH - 1) / NTH;  // TOTAL % 64 == 0: wave-uniform guards
        static_assert(NTH >= NPAD, "one lse row per thread");
        u32x4 kk[IT], vk[IT], qq[IT], vv[IT], oo[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = tid + i * NTH, row = idx / CPR, ch = idx % CPR;
            const bool ok = idx < TOTAL && row < N;
            const size_t off = (size_t)row * rs + ch * 8, offo = (size_t)row * D + ch * 8;
            const u32x4 z = {0, 0, 0, 0};
            kk[i] = ok ? *(const u32x4*)(qb + D + off) : z;
            vk[i] = ok ? *(const u32x4*)(qb + 2 * D + off) : z;
            qq[i] = ok ? *(const u32x4*)(qb + off) : z;
            vv[i] = ok ? *(const u32x4*)(dob + offo) : z;
            oo[i] = ok ? *(const u32x4*)(ob + offo) : z;
        }
        const float lv = tid < N ? lrow[tid] : 0.f;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = tid + i * NTH, row = idx / CPR, ch = idx % CPR;
            if (idx < TOTAL) {
                *(u32x4*)(Kimg + img_off<HD>(row, ch)) = kk[i];
                *(u32x4*)(Vimg + img_off<HD>(row, ch)) = vk[i];
                *(u32x4*)(Qimg + img_off<HD>(row, ch)) = qq[i];
                float dot = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dot += __uint_as_float(vv[i][j] << 16) * __uint_as_float(oo[i][j] << 16);
                    dot += __uint_as_float(vv[i][j] & 0xffff0000u) * __uint_as_float(oo[i][j] & 0xffff0000u);
                }
                *(u32x4*)(Oimg + img_off<HD>(row, ch)) = vv[i];
#pragma unroll
                for (int o = CPR / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);  // the row's CPR lanes
                if (ch == 0) del_s[row] = -dot;  // kept negated: see softmax_bwd4
            }
        }
        if (tid < NPAD) lse_s[tid] = tid < N ? lv * LOG2E : INFINITY;  // padded queries: P = 0
    }
    __syncthreads();
    attn_bwd_phases<HD, NKT, HT>(smem, dqb, N, rs, D, c, scale, tid, AttnBwdNoHook{});
}

// PERSISTENT, PREFETCHING form for the shapes that leave room for ONE workgroup per CU (hd 64 from 10 tiles on: the four
// images are 80-128 KB).  With one workgroup per CU nothing overlaps a head's operand round trip and its stores with
// arithmetic: at N = 197 the staging + stores alone take 144 us and the phases alone 186 us of the kernel's 263
// (profiles/r05o).  Here a workgroup walks heads first, first + grid, ... and fetches the NEXT head while phase B of the
// current one runs, in two batches through 3 IT staging registers (48 VGPRs at N = 197):
//   - after every wave holds its own K / V fragments (barrier 1) the K / V images are dead: the next head's K / V rows are
//     loaded into registers, and written into the K / V images a few query pairs later;
//   - from then on the same registers carry the next head's Q / dO / O rows (and its lse), which are written (with the new
//     -delta rows) when the last wave has left phase B (barrier 2); barrier 3 opens the next head's phase A.
// At hd 32 (two workgroups per CU, which overlap each other's round trips on their own) the same kernel is SLOWER
// (N = 197: 176 -> 206 us; profiles/r05z): those shapes keep one workgroup per head.
// The head id and the thread id enter the loaders as OPAQUE values: left to itself the compiler hoists the loop-invariant
// lane offsets as 64-bit pairs and strength-reduces the addresses into per-lane induction variables — 20+ VGPRs held
// across the phases, spills, and a vmcnt(0) behind each scratch reload, i.e. a wait on the prefetch itself.
template <class F1, class F2> struct AttnBwdHook {
    F1 f1; F2 f2;
    DEVI void after_kv_frags() { f1(); }
    DEVI void before_query_pair(int qp) { f2(qp); }
};
template <class F1, class F2> DEVI AttnBwdHook<F1, F2> attn_bwd_hook(F1 f1, F2 f2) { return AttnBwdHook<F1, F2>{f1, f2}; }
template <int HD, int NKT, bool HT>
__global__ __launch_bounds__(64 * AttnBwdWaves<NKT>::value, 2) void attn_bwd_pf_bf16_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
    const bf16_t* __restrict__ dout, const float* __restrict__ lse, bf16_t* __restrict__ dqkv,
    int N, int H, float scale, int nheads) {
    constexpr int NPAD = NKT * 16, RB = HD * 2, CPR = HD / 8;
    constexpr int NTH = 64 * AttnBwdWaves<NKT>::value, TOTAL = NPAD * CPR, IT = (TOTAL + NTH - 1) / NTH;
    constexpr int QSW = NKT / 4;  // query pair of phase B before which the staging registers change hands
    static_assert(NTH >= NPAD, "one lse row per thread");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kimg = smem;
    char* Vimg = smem + NPAD * RB;
    char* Qimg = smem + 2 * NPAD * RB;
    char* Oimg = smem + 3 * NPAD * RB;  // dO
    float* lse_s = (float*)(smem + 4 * NPAD * RB);
    float* del_s = lse_s + NPAD;  // -delta
    const int tid = threadIdx.x;
    const int D = H * HD;
    const long long rs = 3LL * D;
    const float LOG2E = 1.44269504088896340736f;
    const float c = scale * LOG2E;
    const int first = xcd_remap(blockIdx.x, gridDim.x), stride = (int)gridDim.x;
    u32x4 ra[IT], rb[IT], rc[IT];
    float lv = 0.f;
    // rows of the packed qkv tensor (which = 0 Q, 1 K, 2 V) and of out / dout; uniform base + 32-bit lane offset
    auto load_kv = [&](const int bh_) {
        int bh = __builtin_amdgcn_readfirstlane(bh_), t = tid;
        asm volatile("" : "+s"(bh), "+v"(t));
        const char* qb = (const char*)(qkv + (size_t)(bh / H) * N * rs + (bh % H) * HD);
        const char* kb = qb + 2 * D, *vb = qb + 4 * D;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = t + i * NTH, row = idx / CPR, ch = idx % CPR;
            const bool ok = idx < TOTAL && row < N;
            const unsigned off = (unsigned)(row * (int)rs + ch * 8) * 2u;
            const u32x4 z = {0, 0, 0, 0};
            ra[i] = ok ? *(const u32x4*)(kb + off) : z;
            rb[i] = ok ? *(const u32x4*)(vb + off) : z;
        }
    };
    auto load_qoo = [&](const int bh_) {
        int bh = __builtin_amdgcn_readfirstlane(bh_), t = tid;
        asm volatile("" : "+s"(bh), "+v"(t));
        const int b = bh / H, h = bh % H;
        const char* qb = (const char*)(qkv + (size_t)b * N * rs + h * HD);
        const char* ob = (const char*)(out + (size_t)b * N * D + h * HD);
        const char* dob = (const char*)(dout + (size_t)b * N * D + h * HD);
        const char* lb = (const char*)(lse + ((size_t)b * H + h) * N);
        lv = t < N ? *(const float*)(lb + (unsigned)t * 4u) : 0.f;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = t + i * NTH, row = idx / CPR, ch = idx % CPR;
            const bool ok = idx < TOTAL && row < N;
            const unsigned off = (unsigned)(row * (int)rs + ch * 8) * 2u, offo = (unsigned)(row * D + ch * 8) * 2u;
            const u32x4 z = {0, 0, 0, 0};
            ra[i] = ok ? *(const u32x4*)(qb + off) : z;
            rb[i] = ok ? *(const u32x4*)(dob + offo) : z;
            rc[i] = ok ? *(const u32x4*)(ob + offo) : z;
        }
    };
    auto store_kv = [&]() {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = tid + i * NTH, row = idx / CPR, ch = idx % CPR;
            if (idx < TOTAL) {
                *(u32x4*)(Kimg + img_off<HD>(row, ch)) = ra[i];
                *(u32x4*)(Vimg + img_off<HD>(row, ch)) = rb[i];
            }
        }
    };
    auto store_qoo = [&]() {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = tid + i * NTH, row = idx / CPR, ch = idx % CPR;
            if (idx < TOTAL) {
                *(u32x4*)(Qimg + img_off<HD>(row, ch)) = ra[i];
                float dot = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dot += __uint_as_float(rb[i][j] << 16) * __uint_as_float(rc[i][j] << 16);
                    dot += __uint_as_float(rb[i][j] & 0xffff0000u) * __uint_as_float(rc[i][j] & 0xffff0000u);
                }
                *(u32x4*)(Oimg + img_off<HD>(row, ch)) = rb[i];
#pragma unroll
                for (int o = CPR / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);  // the row's CPR lanes
                if (ch == 0) del_s[row] = -dot;  // kept negated: see softmax_bwd4
            }
        }
        if (tid < NPAD) lse_s[tid] = tid < N ? lv * LOG2E : INFINITY;  // padded queries: P = 0
    };
    // the first head: nothing to hide its operands behind
    load_kv(first);
    store_kv();
    load_qoo(first);
    store_qoo();
    __syncthreads();
#pragma unroll 1
    for (int bh = first; bh < nheads; bh += stride) {
        const bool more = bh + stride < nheads;
        bf16_t* dqb = dqkv + (size_t)(bh / H) * N * rs + (bh % H) * HD;
        attn_bwd_phases<HD, NKT, HT>(smem, dqb, N, rs, D, c, scale, tid, attn_bwd_hook(
            [&]() {
                __syncthreads();  // barrier 1: no wave reads the K / V images of this head any more
                if (more) load_kv(bh + stride);
            },
            [&](const int qp) {
                if (qp == QSW && more) {
                    store_kv();
                    load_qoo(bh + stride);
                }
            }));
        if (!more) break;
        __syncthreads();  // barrier 2: every wave has left this head's Q / dO images and statistics rows
        store_qoo();
        __syncthreads();  // barrier 3
    }
}

// ------------------------------------------------------------------ streaming kernels (N > 128)
// Streaming ("flash") variants for sequences that do not fit the whole-head kernels above (N > 256):
// the detection backbone's global blocks (N = 4096; reference models.py:281-285,310-336) and other
// fixed_size grids.  A workgroup of 8 waves owns 128 queries
// (forward, dQ) or 128 keys (dK/dV) — their MFMA fragments stay in registers — and streams the
// other side through LDS in blocks of 128 rows: two images (K,V resp. Q,dO) x two buffers, the next
// block prefetched into registers while the current one is consumed, one __syncthreads per block.
// With 32 KiB (hd 32) / 64 KiB (hd 64) of LDS two or more workgroups share a CU, so one
// workgroup's prologue (an HBM round trip) hides behind another's MFMAs.  Forward keeps the
// running row maximum / sum per query lane (online softmax, exp2 domain) and rescales the O^T
// accumulators; backward recomputes P from the saved log-sum-exp, with delta_q = sum_d dO O from a
// small pre-pass kernel.  MASK: N is not a multiple of 128 (rows past N are staged as zeros, padded
// keys get probability 0, padded queries are never stored).  No N x N tensor reaches HBM, no
// atomics, fixed summation order.
#define LONG_WAVES 8
#define LONG_BLK 128
template <int HD> struct LongRegs {
    static constexpr int NL = HD / 32;  // 16-byte chunks per thread and image
    u32x4 a[NL], b[NL];
};
// rows row0 .. row0+127 of two [*, HD] bf16 operands (row strides rsa / rsb elements)
template <int HD, bool MASK>
DEVI void long_load(LongRegs<HD>& r, const bf16_t* A, long long rsa, const bf16_t* Bp, long long rsb,
                    int row0, int N, int tid) {
    constexpr int CPR = HD / 8;
#pragma unroll
    for (int i = 0; i < LongRegs<HD>::NL; ++i) {
        const int idx = tid + i * 64 * LONG_WAVES, row = row0 + idx / CPR, c = idx % CPR;
        if (!MASK || row < N) {
            r.a[i] = *(const u32x4*)(A + (size_t)row * rsa + c * 8);
            r.b[i] = *(const u32x4*)(Bp + (size_t)row * rsb + c * 8);
        } else {
            r.a[i] = u32x4{0, 0, 0, 0};
            r.b[i] = u32x4{0, 0, 0, 0};
        }
    }
}
template <int HD>
DEVI void long_store(const LongRegs<HD>& r, char* imgA, char* imgB, int tid) {
    constexpr int CPR = HD / 8;
#pragma unroll
    for (int i = 0; i < LongRegs<HD>::NL; ++i) {
        const int idx = tid + i * 64 * LONG_WAVES, row = idx / CPR, c = idx % CPR;
        *(u32x4*)(imgA + img_off<HD>(row, c)) = r.a[i];
        *(u32x4*)(imgB + img_off<HD>(row, c)) = r.b[i];
    }
}

template <int HD, bool MASK>
__global__ __launch_bounds__(64 * LONG_WAVES, 2) void attn_long_fwd_kernel(
    const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ lse, int N, int H,
    float scale) {
    constexpr int KS = HD / 32, DT = HD / 16, NKT = LONG_BLK / 16, IMG = LONG_BLK * HD * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [buf][K | V]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    const int b = blockIdx.y / H, h = blockIdx.y % H;
    const int D = H * HD;
    const long long rs = 3LL * D;
    const bf16_t* qb = qkv + (size_t)b * N * rs + h * HD;
    const int q = blockIdx.x * LONG_BLK + wave * 16 + (lane & 15);
    const int qc = (MASK && q >= N) ? N - 1 : q;  // padded query lanes compute on a valid row, never store
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = row_frag_global(qb, rs, qc, ks, lane);
    const float c = scale * 1.44269504088896340736f;
    float m = -INFINITY, lsum = 0.f;
    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0, 0, 0, 0};
    const int nkb = (N + LONG_BLK - 1) / LONG_BLK;
    LongRegs<HD> r;
    long_load<HD, MASK>(r, qb + D, rs, qb + 2 * D, rs, 0, N, tid);
    long_store<HD>(r, smem, smem + IMG, tid);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const char* Kimg = smem + (kb & 1) * 2 * IMG;
        const char* Vimg = Kimg + IMG;
        if (kb + 1 < nkb) long_load<HD, MASK>(r, qb + D, rs, qb + 2 * D, rs, (kb + 1) * LONG_BLK, N, tid);
        f32x4 s[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            s[kt] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                s[kt] = MFMA16(row_frag<HD>(Kimg, kt * 16, ks, lane), qf[ks], s[kt]);
        }
        if (MASK && kb == nkb - 1) {  // wave-uniform: padded keys only in the last block
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    if (kb * LONG_BLK + kt * 16 + 4 * g + rr >= N) s[kt][rr] = -INFINITY;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) mx = fmaxf(mx, s[kt][rr]);
        mx = fmaxf(m, group_max(mx));
        const float alpha = __builtin_amdgcn_exp2f((m - mx) * c);  // 0 on the first block (m = -inf)
        const float mc = mx * c;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float pp = __builtin_amdgcn_exp2f(s[kt][rr] * c - mc);
                s[kt][rr] = pp;
                sum += pp;
            }
        lsum = lsum * alpha + sum;
        m = mx;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] *= alpha;
#pragma unroll
        for (int kp = 0; kp < NKT / 2; ++kp) {
            const bf16x8 pf = pack8(s[2 * kp], s[2 * kp + 1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                o[dt] = MFMA16(tr_frag<HD>(Vimg, kp * 32, dt * 16, lane), pf, o[dt]);
        }
        if (kb + 1 < nkb) {
            char* nK = smem + ((kb + 1) & 1) * 2 * IMG;
            long_store<HD>(r, nK, nK + IMG, tid);
        }
        __syncthreads();
    }
    const float l = group_sum(lsum);
    const float inv = 1.0f / l;
    if (!MASK || q < N) {
        bf16_t* orow = out + ((size_t)b * N + q) * D + h * HD + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) st4(orow + dt * 16, o[dt] * inv);
        if (g == 0) lse[((size_t)b * H + h) * N + q] = m * scale + __logf(l);
    }
}

// delta[b, h, q] = sum_d dO[b, q, h, d] O[b, q, h, d]   (HD / 8 lanes per (row, head), 16 B each)
template <int HD>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ out,
                                                         const bf16_t* __restrict__ dout,
                                                         float* __restrict__ delta, long long rows,
                                                         int N, int H) {
    constexpr int CPR = HD / 8;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (row, head, chunk)
    const long long rh = idx / CPR;
    const int ch = (int)(idx % CPR);
    if (rh >= rows * H) return;
    const long long row = rh / H;
    const int h = (int)(rh % H);
    const size_t off = ((size_t)row * H + h) * HD + ch * 8;
    const u32x4 v = *(const u32x4*)(dout + off), o = *(const u32x4*)(out + off);
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        dot += __uint_as_float(v[j] << 16) * __uint_as_float(o[j] << 16);
        dot += __uint_as_float(v[j] & 0xffff0000u) * __uint_as_float(o[j] & 0xffff0000u);
    }
#pragma unroll
    for (int o2 = CPR / 2; o2 > 0; o2 >>= 1) dot += __shfl_xor(dot, o2, 64);
    if (ch == 0) {
        const long long bb = row / N, n = row % N;
        delta[((size_t)bb * H + h) * N + n] = dot;
    }
}

// dQ: the workgroup owns 128 queries and streams K, V
template <int HD, bool MASK>
__global__ __launch_bounds__(64 * LONG_WAVES, 2) void attn_long_bwd_dq_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta, bf16_t* __restrict__ dqkv, int N, int H, float scale) {
    constexpr int KS = HD / 32, DT = HD / 16, IMG = LONG_BLK * HD * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    const int b = blockIdx.y / H, h = blockIdx.y % H;
    const int D = H * HD;
    const long long rs = 3LL * D;
    const bf16_t* qb = qkv + (size_t)b * N * rs + h * HD;
    const bf16_t* dob = dout + (size_t)b * N * D + h * HD;
    const int q = blockIdx.x * LONG_BLK + wave * 16 + (lane & 15);
    const int qc = (MASK && q >= N) ? N - 1 : q;
    const float LOG2E = 1.44269504088896340736f;
    const float c = scale * LOG2E;
    bf16x8 qf[KS], dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = row_frag_global(qb, rs, qc, ks, lane);
        dof[ks] = row_frag_global(dob, D, qc, ks, lane);
    }
    const float l2 = lse[((size_t)b * H + h) * N + qc] * LOG2E;
    const float dl = delta[((size_t)b * H + h) * N + qc];
    f32x4 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[dt] = f32x4{0, 0, 0, 0};
    const int nkb = (N + LONG_BLK - 1) / LONG_BLK;
    LongRegs<HD> r;
    long_load<HD, MASK>(r, qb + D, rs, qb + 2 * D, rs, 0, N, tid);
    long_store<HD>(r, smem, smem + IMG, tid);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const char* Kimg = smem + (kb & 1) * 2 * IMG;
        const char* Vimg = Kimg + IMG;
        if (kb + 1 < nkb) long_load<HD, MASK>(r, qb + D, rs, qb + 2 * D, rs, (kb + 1) * LONG_BLK, N, tid);
        const bool tail = MASK && kb == nkb - 1;  // wave-uniform
#pragma unroll 1
        for (int kp = 0; kp < LONG_BLK / 32; ++kp) {
            f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0}, p0 = {0, 0, 0, 0}, p1 = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s0 = MFMA16(row_frag<HD>(Kimg, kp * 32, ks, lane), qf[ks], s0);
                s1 = MFMA16(row_frag<HD>(Kimg, kp * 32 + 16, ks, lane), qf[ks], s1);
                p0 = MFMA16(row_frag<HD>(Vimg, kp * 32, ks, lane), dof[ks], p0);
                p1 = MFMA16(row_frag<HD>(Vimg, kp * 32 + 16, ks, lane), dof[ks], p1);
            }
            const int ka = kb * LONG_BLK + kp * 32 + 4 * g;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float pa = __builtin_amdgcn_exp2f(s0[rr] * c - l2);
                float pb = __builtin_amdgcn_exp2f(s1[rr] * c - l2);
                if (tail) {
                    pa = (ka + rr < N) ? pa : 0.f;
                    pb = (ka + 16 + rr < N) ? pb : 0.f;
                }
                s0[rr] = pa * (p0[rr] - dl);  // dS^T
                s1[rr] = pb * (p1[rr] - dl);
            }
            const bf16x8 dsf = pack8(s0, s1);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                dq[dt] = MFMA16(tr_frag<HD>(Kimg, kp * 32, dt * 16, lane), dsf, dq[dt]);
        }
        if (kb + 1 < nkb) {
            char* nK = smem + ((kb + 1) & 1) * 2 * IMG;
            long_store<HD>(r, nK, nK + IMG, tid);
        }
        __syncthreads();
    }
    if (!MASK || q < N) {
        bf16_t* dr = dqkv + (size_t)b * N * rs + h * HD + (size_t)q * rs + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) st4(dr + dt * 16, dq[dt] * scale);
    }
}

// dK, dV: the workgroup owns 128 keys and streams Q, dO (+ the lse / delta rows of the block)
template <int HD, bool MASK>
__global__ __launch_bounds__(64 * LONG_WAVES, 2) void attn_long_bwd_dkv_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta, bf16_t* __restrict__ dqkv, int N, int H, float scale) {
    constexpr int KS = HD / 32, DT = HD / 16, IMG = LONG_BLK * HD * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [buf][Q | dO], then [buf][lse | delta]
    float* stat = (float*)(smem + 4 * IMG);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    const int b = blockIdx.y / H, h = blockIdx.y % H;
    const int D = H * HD;
    const long long rs = 3LL * D;
    const bf16_t* qb = qkv + (size_t)b * N * rs + h * HD;
    const bf16_t* dob = dout + (size_t)b * N * D + h * HD;
    const float* lrow = lse + ((size_t)b * H + h) * N;
    const float* drow = delta + ((size_t)b * H + h) * N;
    const int key = blockIdx.x * LONG_BLK + wave * 16 + (lane & 15);
    const int kc = (MASK && key >= N) ? N - 1 : key;
    const float LOG2E = 1.44269504088896340736f;
    const float c = scale * LOG2E;
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = row_frag_global(qb + D, rs, kc, ks, lane);
        vf[ks] = row_frag_global(qb + 2 * D, rs, kc, ks, lane);
    }
    f32x4 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        dk[dt] = f32x4{0, 0, 0, 0};
        dv[dt] = f32x4{0, 0, 0, 0};
    }
    const int nqb = (N + LONG_BLK - 1) / LONG_BLK;
    LongRegs<HD> r;
    float st_next = 0.f;  // threads 0..127: lse * log2e, 128..255: delta of the next block
    auto stat_load = [&](int blk) {
        const int i = blk * LONG_BLK + (tid & (LONG_BLK - 1));
        st_next = 0.f;
        if (tid < 2 * LONG_BLK && (!MASK || i < N)) st_next = tid < LONG_BLK ? lrow[i] * LOG2E : drow[i];
    };
    long_load<HD, MASK>(r, qb, rs, dob, D, 0, N, tid);
    stat_load(0);
    long_store<HD>(r, smem, smem + IMG, tid);
    if (tid < 2 * LONG_BLK) stat[tid] = st_next;
    __syncthreads();
    for (int qbk = 0; qbk < nqb; ++qbk) {
        const char* Qimg = smem + (qbk & 1) * 2 * IMG;
        const char* Oimg = Qimg + IMG;
        const float* lse_s = stat + (qbk & 1) * 2 * LONG_BLK;
        const float* del_s = lse_s + LONG_BLK;
        if (qbk + 1 < nqb) {
            long_load<HD, MASK>(r, qb, rs, dob, D, (qbk + 1) * LONG_BLK, N, tid);
            stat_load(qbk + 1);
        }
        const bool tail = MASK && qbk == nqb - 1;  // wave-uniform: padded queries only in the last block
#pragma unroll 1
        for (int qp = 0; qp < LONG_BLK / 32; ++qp) {
            f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0}, p0 = {0, 0, 0, 0}, p1 = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s0 = MFMA16(row_frag<HD>(Qimg, qp * 32, ks, lane), kf[ks], s0);
                s1 = MFMA16(row_frag<HD>(Qimg, qp * 32 + 16, ks, lane), kf[ks], s1);
                p0 = MFMA16(row_frag<HD>(Oimg, qp * 32, ks, lane), vf[ks], p0);
                p1 = MFMA16(row_frag<HD>(Oimg, qp * 32 + 16, ks, lane), vf[ks], p1);
            }
            const int qa = qp * 32 + 4 * g;
            const f32x4 la = *(const f32x4*)(lse_s + qa), lb = *(const f32x4*)(lse_s + qa + 16);
            const f32x4 da = *(const f32x4*)(del_s + qa), db = *(const f32x4*)(del_s + qa + 16);
            f32x4 pa, pb, dsa, dsb;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                pa[rr] = __builtin_amdgcn_exp2f(s0[rr] * c - la[rr]);
                pb[rr] = __builtin_amdgcn_exp2f(s1[rr] * c - lb[rr]);
                if (tail) {
                    pa[rr] = (qbk * LONG_BLK + qa + rr < N) ? pa[rr] : 0.f;
                    pb[rr] = (qbk * LONG_BLK + qa + 16 + rr < N) ? pb[rr] : 0.f;
                }
                dsa[rr] = pa[rr] * (p0[rr] - da[rr]);
                dsb[rr] = pb[rr] * (p1[rr] - db[rr]);
            }
            const bf16x8 pf = pack8(pa, pb), dsf = pack8(dsa, dsb);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                dv[dt] = MFMA16(tr_frag<HD>(Oimg, qp * 32, dt * 16, lane), pf, dv[dt]);
                dk[dt] = MFMA16(tr_frag<HD>(Qimg, qp * 32, dt * 16, lane), dsf, dk[dt]);
            }
        }
        if (qbk + 1 < nqb) {
            char* nQ = smem + ((qbk + 1) & 1) * 2 * IMG;
            long_store<HD>(r, nQ, nQ + IMG, tid);
            if (tid < 2 * LONG_BLK) stat[((qbk + 1) & 1) * 2 * LONG_BLK + tid] = st_next;
        }
        __syncthreads();
    }
    if (!MASK || key < N) {
        bf16_t* dr = dqkv + (size_t)b * N * rs + h * HD + (size_t)key * rs + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            st4(dr + D + dt * 16, dk[dt] * scale);
            st4(dr + 2 * D + dt * 16, dv[dt]);
        }
    }
}

// ------------------------------------------------------------------ f32 row kernels
// one wave per row; rows of `cols` floats (cols <= a few thousand)
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s,
                                                           float* __restrict__ lse,
                                                           long long rows, int cols) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* p = s + row * cols;
    float mx = -INFINITY;
    for (int i = lane; i < cols; i += 64) mx = fmaxf(mx, p[i]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int i = lane; i < cols; i += 64) {
        const float e = expf(p[i] - mx);
        p[i] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int i = lane; i < cols; i += 64) p[i] *= inv;
    if (lse && lane == 0) lse[row] = mx + logf(sum);
}
__global__ __launch_bounds__(256) void exp_rows_kernel(float* __restrict__ s,
                                                       const float* __restrict__ lse,
                                                       long long rows, int cols) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* p = s + row * cols;
    const float l = lse[row];
    for (int i = lane; i < cols; i += 64) p[i] = expf(p[i] - l);
}
// dP <- P * (dP - sum(P*dP))
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ p,
                                                               float* __restrict__ dp,
                                                               long long rows, int cols) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* pr = p + row * cols;
    float* dr = dp + row * cols;
    float d = 0.f;
    for (int i = lane; i < cols; i += 64) d += pr[i] * dr[i];
    d = wave_sum(d);
    for (int i = lane; i < cols; i += 64) dr[i] = pr[i] * (dr[i] - d);
}

// ------------------------------------------------------------------ host side
static ssl4gie_gemm_desc bdesc(int M, int N, int K, int B, int H) {
    ssl4gie_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.M = M; d.N = N; d.K = K; d.batch1 = B; d.batch2 = H;
    d.dtype_ab = SSL4GIE_F32; d.dtype_c = SSL4GIE_F32; d.alpha = 1.f;
    d.epilogue = SSL4GIE_EPI_NONE;
    return d;
}

// SSL4GIE_ATTN_STREAM_MIN: smallest N that takes the streaming kernels.  Default 257: measured
// (profiles/r01q_attn_stream_vs_whole.log), the whole-head-in-LDS kernels win up to N = 256 —
// at N = 197 the 128-row blocks pad keys to 256 and queries to 256, K/V are staged twice per head
// and the backward recomputes the scores in two kernels.  Lower values are for A/B measurements;
// the streaming kernels handle any N (MASK variant) for hd 32 / 64.
static int attn_stream_min() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("SSL4GIE_ATTN_STREAM_MIN");
        v = e ? atoi(e) : 257;
        if (v < 1) v = 1;
    }
    return v;
}
static bool attn_long(int N, int hd) {
    return (hd == 64 || hd == 32) && (N > 256 || N >= attn_stream_min());
}

extern "C" size_t ssl4gie_attn_workspace_bytes(int dtype, int B, int N, int H, int hd) {
    if (dtype == SSL4GIE_BF16)  // long sequences: delta[b, h, q] of the backward pass
        return attn_long(N, hd) ? (size_t)B * H * N * sizeof(float) : 0;
    return (size_t)2 * B * H * N * N * sizeof(float);  // scores + dscores
}

// SSL4GIE_ATTN_HALF_TAIL=0: the kernels without the half-tail variants (A/B)
static bool attn_half_tail() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("SSL4GIE_ATTN_HALF_TAIL"); v = e ? atoi(e) != 0 : 1; }
    return v != 0;
}
template <int HD, int NKT, bool HT>
static int launch_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, float scale,
                      hipStream_t st) {
    constexpr bool QG = AttnFwdQG<HD, NKT>::value;
    const size_t lds = (size_t)(QG ? 2 : 3) * NKT * 16 * HD * 2;
    constexpr int WAVES = (!QG && 3 * NKT * 16 * HD * 2 > 80 * 1024) ? 8 : 4;
    auto k = attn_fwd_bf16_kernel<HD, NKT, WAVES, HT>;
    if (lds > 65536)
        HIP_RET(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope prof(PROF_ATTN_FWD, 4.0 * B * H * (double)N * N * HD, st);
    hipLaunchKernelGGL(k, dim3(B * H), dim3(64 * WAVES), lds, st, (const bf16_t*)qkv, (bf16_t*)out, lse,
                       N, H, scale);
    LAUNCH_CHECK();
    return 0;
}
static bool attn_prefetch() {  // SSL4GIE_ATTN_PREFETCH=0: the one-head-per-workgroup kernel everywhere (A/B; same results)
    static int on = -1;
    if (on < 0) { const char* e = getenv("SSL4GIE_ATTN_PREFETCH"); on = (e && e[0] == '0') ? 0 : 1; }
    return on != 0;
}
template <int HD, int NKT, bool HT>
static int launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                      void* dqkv, int B, int N, int H, float scale, hipStream_t st);
template <int HD, int NKT, bool HT>
static int launch_bwd_pf(const void* qkv, const void* out, const void* dout, const float* lse,
                         void* dqkv, int B, int N, int H, float scale, hipStream_t st) {
    constexpr int W = AttnBwdWaves<NKT>::value;
    const size_t lds = (size_t)4 * NKT * 16 * HD * 2 + 2 * NKT * 16 * sizeof(float);
    auto k = attn_bwd_pf_bf16_kernel<HD, NKT, HT>;
    // workgroups one CU holds at once: a pure function of the kernel and the (single) architecture, so the benign
    // race of two host threads computing it twice writes the same value.  The persistent grid covers the
    // CUs the compute kernels may use (ssl4gie_set_compute_cus: data-parallel runs keep a few for the
    // communication kernels, which these long-lived workgroups would otherwise occupy).
    static int per_cu = 0;
    if (!per_cu) {
        HIP_RET(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int v = 0;
        HIP_RET(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, (const void*)k, 64 * W, lds));
        per_cu = v < 1 ? 1 : v;
    }
    if (((N + 15) >> 4) < W)  // a wave without a tile would miss the hooks' barriers: the one-head kernel handles it
        return launch_bwd<HD, NKT, HT>(qkv, out, dout, lse, dqkv, B, N, H, scale, st);
    const int slots = per_cu * ssl4gie_internal_compute_cus();
    const int heads = B * H;
    int grid = heads;
    if (heads > slots) {  // every workgroup walks the same number of heads (+- 1)
        const int rounds = (heads + slots - 1) / slots;
        grid = (heads + rounds - 1) / rounds;
    }
    ProfScope prof(PROF_ATTN_BWD, 10.0 * B * H * (double)N * N * HD, st);
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * W), lds, st, (const bf16_t*)qkv, (const bf16_t*)out,
                       (const bf16_t*)dout, lse, (bf16_t*)dqkv, N, H, scale, heads);
    LAUNCH_CHECK();
    return 0;
}
template <int HD, int NKT, bool HT>
static int launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                      void* dqkv, int B, int N, int H, float scale, hipStream_t st) {
    const size_t lds = (size_t)4 * NKT * 16 * HD * 2 + 2 * NKT * 16 * sizeof(float);
    auto k = attn_bwd_bf16_kernel<HD, NKT, HT>;
    if (lds > 65536)
        HIP_RET(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope prof(PROF_ATTN_BWD, 10.0 * B * H * (double)N * N * HD, st);
    hipLaunchKernelGGL(k, dim3(B * H), dim3(64 * AttnBwdWaves<NKT>::value), lds, st, (const bf16_t*)qkv,
                       (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, N, H, scale);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int ssl4gie_attn_fwd(const void* qkv, void* out, float* lse, int dtype, int B, int N,
                                int H, int hd, void* workspace, void* stream) {
    REQUIRE(qkv && out && lse && B > 0 && N > 0 && H > 0 && hd > 0);
    hipStream_t st = (hipStream_t)stream;
    const float scale = 1.0f / sqrtf((float)hd);
    const int D = H * hd;
    if (dtype == SSL4GIE_BF16) {
        REQUIRE(hd == 32 || hd == 64);
        if (attn_long(N, hd)) {
            const int lds = 4 * LONG_BLK * hd * 2;
            const bool mask = N % LONG_BLK != 0;
            const dim3 grid((N + LONG_BLK - 1) / LONG_BLK, B * H), block(64 * LONG_WAVES);
            ProfScope prof(PROF_ATTN_FWD, 4.0 * B * H * (double)N * N * hd, st);
#define LFWD(HD_, MASK_)                                                                           \
    do {                                                                                           \
        auto kfn = attn_long_fwd_kernel<HD_, MASK_>;                                               \
        static bool attr = false;                                                                  \
        if (!attr) {                                                                               \
            HIP_RET(hipFuncSetAttribute((const void*)kfn,                                          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 65536));      \
            attr = true;                                                                           \
        }                                                                                          \
        hipLaunchKernelGGL(kfn, grid, block, lds, st, (const bf16_t*)qkv, (bf16_t*)out, lse, N, H, \
                           scale);                                                                 \
    } while (0)
            if (hd == 64) { if (mask) LFWD(64, true); else LFWD(64, false); }
            else { if (mask) LFWD(32, true); else LFWD(32, false); }
#undef LFWD
            LAUNCH_CHECK();
            return 0;
        }
        REQUIRE(N <= 256);
// the kernels are instantiated for every even number of 16-key tiles: 16 (NKT - 2) < N <= 16 NKT,
// so that only the last tile pair can hold padded keys (the kernels rely on it)
#define FWD_HT(HD_, NKT_)                                                                 \
    return (HD_ == 32 && N <= 16 * (NKT_ - 1) && attn_half_tail()) /* hd 64: the variant spills */ \
               ? launch_fwd<HD_, NKT_, HD_ == 32>(qkv, out, lse, B, N, H, scale, st)           \
               : launch_fwd<HD_, NKT_, false>(qkv, out, lse, B, N, H, scale, st)
#define FWD(HD_)                                                                          \
    switch ((N + 31) >> 5) {                                                              \
    case 1: FWD_HT(HD_, 2); \
    case 2: FWD_HT(HD_, 4); \
    case 3: FWD_HT(HD_, 6); \
    case 4: FWD_HT(HD_, 8); \
    case 5: FWD_HT(HD_, 10); \
    case 6: FWD_HT(HD_, 12); \
    case 7: FWD_HT(HD_, 14); \
    default: FWD_HT(HD_, 16); \
    }
        if (hd == 64) { FWD(64) } else { FWD(32) }
#undef FWD
    }
    REQUIRE(dtype == SSL4GIE_F32 && workspace);
    float* S = (float*)workspace;
    const float* q = (const float*)qkv;
    // S = scale * Q K^T
    ssl4gie_gemm_desc d = bdesc(N, N, hd, B, H);
    d.A = q; d.sAm = 3LL * D; d.sAk = 1; d.sAb1 = (long long)N * 3 * D; d.sAb2 = hd;
    d.B = q + D; d.sBk = 1; d.sBn = 3LL * D; d.sBb1 = (long long)N * 3 * D; d.sBb2 = hd;
    d.C = S; d.ldc = N; d.sCb1 = (long long)H * N * N; d.sCb2 = (long long)N * N;
    d.alpha = scale;
    int rc = ssl4gie_gemm(&d, nullptr, 0, stream);
    if (rc) return rc;
    const long long rows = (long long)B * H * N;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, S,
                       lse, rows, N);
    LAUNCH_CHECK();
    // O = P V
    d = bdesc(N, hd, N, B, H);
    d.A = S; d.sAm = N; d.sAk = 1; d.sAb1 = (long long)H * N * N; d.sAb2 = (long long)N * N;
    d.B = q + 2 * D; d.sBk = 3LL * D; d.sBn = 1; d.sBb1 = (long long)N * 3 * D; d.sBb2 = hd;
    d.C = out; d.ldc = D; d.sCb1 = (long long)N * D; d.sCb2 = hd;
    return ssl4gie_gemm(&d, nullptr, 0, stream);
}

extern "C" int ssl4gie_attn_bwd(const void* qkv, const void* out, const void* dout,
                                const float* lse, void* dqkv, int dtype, int B, int N, int H,
                                int hd, void* workspace, void* stream) {
    REQUIRE(qkv && out && dout && lse && dqkv && B > 0 && N > 0 && H > 0 && hd > 0);
    hipStream_t st = (hipStream_t)stream;
    const float scale = 1.0f / sqrtf((float)hd);
    const int D = H * hd;
    if (dtype == SSL4GIE_BF16) {
        REQUIRE(hd == 32 || hd == 64);
        if (attn_long(N, hd)) {
            REQUIRE(workspace);
            float* delta = (float*)workspace;
            const long long rows = (long long)B * N;
            const long long items = rows * H * (hd / 8);
            const dim3 dgrid((unsigned)((items + 255) / 256));
            if (hd == 64)
                hipLaunchKernelGGL(attn_delta_kernel<64>, dgrid, dim3(256), 0, st, (const bf16_t*)out,
                                   (const bf16_t*)dout, delta, rows, N, H);
            else
                hipLaunchKernelGGL(attn_delta_kernel<32>, dgrid, dim3(256), 0, st, (const bf16_t*)out,
                                   (const bf16_t*)dout, delta, rows, N, H);
            LAUNCH_CHECK();
            const int lds_q = 4 * LONG_BLK * hd * 2, lds_kv = lds_q + 4 * LONG_BLK * (int)sizeof(float);
            const bool mask = N % LONG_BLK != 0;
            ProfScope prof(PROF_ATTN_BWD, 10.0 * B * H * (double)N * N * hd, st);
            const dim3 grid((N + LONG_BLK - 1) / LONG_BLK, B * H), block(64 * LONG_WAVES);
#define LBWD(HD_, MASK_)                                                                           \
    do {                                                                                           \
        auto kq = attn_long_bwd_dq_kernel<HD_, MASK_>;                                             \
        auto kkv = attn_long_bwd_dkv_kernel<HD_, MASK_>;                                           \
        static bool attr = false;                                                                  \
        if (!attr) {                                                                               \
            HIP_RET(hipFuncSetAttribute((const void*)kq,                                           \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 65536));      \
            HIP_RET(hipFuncSetAttribute((const void*)kkv,                                          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 2048)); \
            attr = true;                                                                           \
        }                                                                                          \
        hipLaunchKernelGGL(kq, grid, block, lds_q, st, (const bf16_t*)qkv, (const bf16_t*)dout,    \
                           lse, delta, (bf16_t*)dqkv, N, H, scale);                                \
        hipLaunchKernelGGL(kkv, grid, block, lds_kv, st, (const bf16_t*)qkv, (const bf16_t*)dout,  \
                           lse, delta, (bf16_t*)dqkv, N, H, scale);                                \
    } while (0)
            if (hd == 64) { if (mask) LBWD(64, true); else LBWD(64, false); }
            else { if (mask) LBWD(32, true); else LBWD(32, false); }
#undef LBWD
            LAUNCH_CHECK();
            return 0;
        }
        REQUIRE(N <= 256);
#define BWD_HT(HD_, NKT_)                                                                          \
    if (HD_ == 64 && NKT_ >= 10 && attn_prefetch()) /* one workgroup per CU: persistent + prefetch */ \
        return (N <= 16 * (NKT_ - 1) && attn_half_tail())                                              \
                   ? launch_bwd_pf<64, (NKT_ >= 10 ? NKT_ : 10), true>(qkv, out, dout, lse, dqkv, B, N, H, scale, st) \
                   : launch_bwd_pf<64, (NKT_ >= 10 ? NKT_ : 10), false>(qkv, out, dout, lse, dqkv, B, N, H, scale, st);     \
    return (HD_ == 64 && N <= 16 * (NKT_ - 1) && attn_half_tail()) /* hd 32: measured no gain */   \
               ? launch_bwd<HD_, NKT_, HD_ == 64>(qkv, out, dout, lse, dqkv, B, N, H, scale, st)   \
               : launch_bwd<HD_, NKT_, false>(qkv, out, dout, lse, dqkv, B, N, H, scale, st)
#define BWD(HD_)                                                                                   \
    switch ((N + 31) >> 5) {                                                                       \
    case 1: BWD_HT(HD_, 2); \
    case 2: BWD_HT(HD_, 4); \
    case 3: BWD_HT(HD_, 6); \
    case 4: BWD_HT(HD_, 8); \
    case 5: BWD_HT(HD_, 10); \
    case 6: BWD_HT(HD_, 12); \
    case 7: BWD_HT(HD_, 14); \
    default: BWD_HT(HD_, 16); \
    }
        if (hd == 64) { BWD(64) } else { BWD(32) }
#undef BWD
    }
    REQUIRE(dtype == SSL4GIE_F32 && workspace);
    float* P = (float*)workspace;
    float* dP = P + (size_t)B * H * N * N;
    const float* q = (const float*)qkv;
    const float* dO = (const float*)dout;
    float* dq = (float*)dqkv;
    const long long sS1 = (long long)H * N * N, sS2 = (long long)N * N;
    const long long sQ1 = (long long)N * 3 * D, sO1 = (long long)N * D;
    const long long rows = (long long)B * H * N;
    int rc;
    // P = exp(scale Q K^T - lse)
    ssl4gie_gemm_desc d = bdesc(N, N, hd, B, H);
    d.A = q; d.sAm = 3LL * D; d.sAk = 1; d.sAb1 = sQ1; d.sAb2 = hd;
    d.B = q + D; d.sBk = 1; d.sBn = 3LL * D; d.sBb1 = sQ1; d.sBb2 = hd;
    d.C = P; d.ldc = N; d.sCb1 = sS1; d.sCb2 = sS2; d.alpha = scale;
    if ((rc = ssl4gie_gemm(&d, nullptr, 0, stream))) return rc;
    hipLaunchKernelGGL(exp_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, P, lse,
                       rows, N);
    LAUNCH_CHECK();
    // dV[key, d] = sum_q P[q, key] dO[q, d]
    d = bdesc(N, hd, N, B, H);
    d.A = P; d.sAm = 1; d.sAk = N; d.sAb1 = sS1; d.sAb2 = sS2;
    d.B = dO; d.sBk = D; d.sBn = 1; d.sBb1 = sO1; d.sBb2 = hd;
    d.C = dq + 2 * D; d.ldc = 3LL * D; d.sCb1 = sQ1; d.sCb2 = hd;
    if ((rc = ssl4gie_gemm(&d, nullptr, 0, stream))) return rc;
    // dP = dO V^T
    d = bdesc(N, N, hd, B, H);
    d.A = dO; d.sAm = D; d.sAk = 1; d.sAb1 = sO1; d.sAb2 = hd;
    d.B = q + 2 * D; d.sBk = 1; d.sBn = 3LL * D; d.sBb1 = sQ1; d.sBb2 = hd;
    d.C = dP; d.ldc = N; d.sCb1 = sS1; d.sCb2 = sS2;
    if ((rc = ssl4gie_gemm(&d, nullptr, 0, stream))) return rc;
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st,
                       P, dP, rows, N);
    LAUNCH_CHECK();
    // dQ = scale dS K
    d = bdesc(N, hd, N, B, H);
    d.A = dP; d.sAm = N; d.sAk = 1; d.sAb1 = sS1; d.sAb2 = sS2;
    d.B = q + D; d.sBk = 3LL * D; d.sBn = 1; d.sBb1 = sQ1; d.sBb2 = hd;
    d.C = dq; d.ldc = 3LL * D; d.sCb1 = sQ1; d.sCb2 = hd; d.alpha = scale;
    if ((rc = ssl4gie_gemm(&d, nullptr, 0, stream))) return rc;
    // dK = scale dS^T Q
    d = bdesc(N, hd, N, B, H);
    d.A = dP; d.sAm = 1; d.sAk = N; d.sAb1 = sS1; d.sAb2 = sS2;
    d.B = q; d.sBk = 3LL * D; d.sBn = 1; d.sBb1 = sQ1; d.sBb2 = hd;
    d.C = dq + D; d.ldc = 3LL * D; d.sCb1 = sQ1; d.sCb2 = hd; d.alpha = scale;
    return ssl4gie_gemm(&d, nullptr, 0, stream);
}
