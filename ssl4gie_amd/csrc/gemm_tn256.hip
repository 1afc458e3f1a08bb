// 256x256x64 ping-pong bf16 TN GEMM for gfx950:  C[M,N] = alpha * At[K,M]^T Bt[K,N]  (fp32 out).
//
// The weight-gradient product dW[out,in] = dY[T,out]^T X[T,in]: the contraction runs over tokens
// (K = T = 12 800 / 50 432 in the MAE step) while M, N are the small layer widths, so the work is
// split along K across workgroups (fp32 slabs + deterministic slab reduction, no atomics).
//
// The four-phase schedule gemm_nt256.hip had until round 4 (4 phases per K-tile of 16 MFMAs, two staggered wave
// rows, LDS-DMA half-tile stream 7 ahead behind a counted vmcnt; hazard bookkeeping in that file's header).  The
// two-phase form that made the NT K-loop 8 % faster was measured here too and is 4-5 % SLOWER (dec.dWfc1 104 ->
// 116 us, profiles/r04t_tn_two_phase.log): a phase would issue 32 transposed reads against the 4-bit lgkmcnt.  What
// differs is the operand image: both operands are k-major in memory, so a half-tile is 64 k-rows x
// 256 B (128 m- or n-values), staged by 4-row LDS-DMA pieces, and MFMA fragments are gathered
// with ds_read_b64_tr_b16 (hardware transpose, two reads per fragment).  Image: 16-B chunk c of
// k-row r at position c ^ tn_swz(r) (same swizzle as the 128x128 TN kernel, conflict-free for the
// transposed reads; tools/lds_bank_sim.py).  P0 issues 8 + 16 transposed reads; lgkmcnt is a
// 4-bit counter, so the early retire of the B_h0 reads is s_waitcnt lgkmcnt(15) (oldest 9 done).
//
// Replaces the autograd weight-gradient GEMMs of nn.Linear in timm Block / MAE decoder (SURVEY
// §2.2 "bwd adds dW = X^T dY"; reference modules Models/mae/models_mae.py:39-41,47,53-55,59).
#include "gemm256.h"
#include "prof.h"

#include <stdlib.h>
#include <type_traits>

#ifndef TN256_ABL
#define TN256_ABL 0
#endif

DEVI int q_swz(int k) { return ((k & 3) | ((k >> 1) & 4)) << 1; }

// MFMA operand (16 x-values x 32 k) out of a k-major half-tile image: lane l gets, for
// x = x0 + (l & 15), the 8 k-values krow0 .. krow0+7 (krow0 = 32 ks + 8 (l >> 4))
DEVI bf16x8 q_frag(const char* img, int krow0, int x0, int lane) {
    const int q = (lane & 15) >> 2, p = lane & 3;
    const int c = (x0 >> 3) + (p >> 1);
    const int r0 = krow0 + q, r1 = krow0 + 4 + q;
    const int o0 = r0 * 256 + ((c ^ q_swz(r0)) << 4) + ((p & 1) << 3);
    const int o1 = r1 * 256 + ((c ^ q_swz(r1)) << 4) + ((p & 1) << 3);
    typedef __attribute__((address_space(3))) s16x4* lp_t;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(img + o0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(img + o1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// grid.x = ntiles * splits; split s of tile t covers K-tiles [nkt*s/splits, nkt*(s+1)/splits).
// COLSUM: the workgroups of the first tile column (n0 == 0) also produce colsum[m] = sum_k At[k][m]
// (the bias gradient): wave (wr, wc) multiplies its wc-th A fragment of each quadrant against a
// ones fragment — 2 extra MFMAs in P0 and in P2 — and writes 2 x 16 sums.
// CONV: 0 = Bt is a matrix; 1 = Bt is the implicit 3x3 patch matrix of the map at `Bt` (k-row =
// output pixel, column = (tap, channel); ssl4gie_conv3x3_geom); 2 = with ReLU on the B fragments.
// PARTIAL: the launch has tiles that M x N does not fill (see `mmask` below); a separate instantiation, so that
// the full-tile kernels keep their register allocation.
template <bool COLSUM, int CONV, bool PARTIAL = false>
__global__ __launch_bounds__(512, 2) void gemm_bf16_tn256_kernel(
    const bf16_t* __restrict__ At, long long ldat, const bf16_t* __restrict__ Bt, long long ldbt,
    float* __restrict__ C, long long ldc, float* __restrict__ slabs, int M, int N, int K,
    int tiles_n, int ntiles, int splits, float alpha, int accumulate, float* __restrict__ colsum,
    float* __restrict__ colsum_part, ConvK cg, TnExtras ex) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // -DTN256_ABL=<bits> (tools/build_variant.sh experiment libraries; timing only, garbage outputs): bit 0 no output
    // stores, bit 1 no fragment reads, bit 2 no LDS-DMA, bit 3 no MFMA — which of the K-loop's streams sets its time
    constexpr int dbg = TN256_ABL;
    const bool tn_m_inner_ok = ex.m_inner != 0;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int all_tiles = ntiles + ex.total_tiles;
    const int split = bid / all_tiles;
    int tile = bid % all_tiles;
    if (tile >= ntiles) {  // grouped launch: this workgroup belongs to a later product (uniform)
        const int rel = tile - ntiles;
#pragma unroll
        for (int i = 0; i < TN_GROUP_MAX - 1; ++i) {
            if (i < ex.n && rel >= ex.p[i].tile0 && rel < ex.p[i].tile0 + ex.p[i].ntiles) {
                tile = rel - ex.p[i].tile0;
                At = (const bf16_t*)ex.p[i].At; ldat = ex.p[i].ldat;
                Bt = (const bf16_t*)ex.p[i].Bt; ldbt = ex.p[i].ldbt;
                C = ex.p[i].C; ldc = ex.p[i].ldc; slabs = ex.p[i].slabs;
                M = ex.p[i].M; N = ex.p[i].N; tiles_n = ex.p[i].tiles_n;
                colsum = ex.p[i].colsum; colsum_part = ex.p[i].colsum_part;
                alpha = ex.p[i].alpha; accumulate = ex.p[i].accumulate;
            }
        }
    }
    // Consecutive tiles of one K-split sit on one XCD (xcd_remap) and share operand slabs through its
    // L2: a run of r x c tiles reads r + c slabs, so the run should be as square as possible — the
    // SHORTER tile dimension is the inner one (a 3 x 12 product walked row by row makes a 27-tile XCD
    // chunk read 3 + 12 slabs and its 9-tile neighbour 1 + 9; column by column they read 3 + 9 and 3 + 3:
    // 18 slab reads instead of 25 against a minimum of 15).  SSL4GIE_TN_INNER=n restores the old walk.
    const int tiles_m = (M + P_BM - 1) / P_BM;
    const bool m_inner = tiles_m < tiles_n && tn_m_inner_ok;
    const int m0 = (m_inner ? tile % tiles_m : tile / tiles_n) * P_BM;
    const int n0 = (m_inner ? tile / tiles_m : tile % tiles_n) * P_BN;
    const int nkt = K / P_BK;
    const int kt0 = (int)((long long)nkt * split / splits);
    const int kt1 = (int)((long long)nkt * (split + 1) / splits);
    const int total_kt = kt1 - kt0;

    // ---- LDS-DMA stream: this lane's source element offsets (without the k-row term of the
    // K-tile, which advances the wave-uniform base): piece i covers k-rows 4 (2 wave + i) .. +3
    unsigned v0_0, v0_1, v1_0, v1_1, v2_0, v2_1, v3_0, v3_1;
    // CONV: v0_* / v2_* hold the lane's tap as ((dy*W + dx)*C + ci)*2 | dy << 2 | dx (the byte
    // offset is a multiple of 16); the pixel of k-row piece i of the cursor's K-tile is kept as
    // o_i (byte offset of its tap (0,0)) and yx_i (y0 << 16 | x0 & 0xffff)
    const int c_H = cg.H, c_W = cg.W, c_C = cg.C, c_Wo = cg.Wo, c_HoWo = cg.HoWo, c_s = cg.stride;
    const unsigned c_mgw = cg.mg_wo, c_shw = cg.sh_wo, c_mgh = cg.mg_hw, c_shh = cg.sh_hw;
    const char* c_zero = (const char*)cg.zero;
    int o_0 = 0, o_1 = 0;
    unsigned yx_0 = 0, yx_1 = 0;
    {
        const int mch = (M >> 3) - 1, nch = (N >> 3) - 1;  // last valid 16-B chunk of a row
        auto offs = [&](int i, int h, bool is_a) -> unsigned {
            const int kr = (wave * 2 + i) * 4 + (lane >> 4);  // k-row inside the K-tile
            const int c = (lane & 15) ^ q_swz(kr);            // logical chunk kept at position l&15
            if (is_a) {
                int ch = ((m0 + (c >> 3) * 128 + h * 64) >> 3) + (c & 7);
                ch = ch < mch ? ch : mch;
                return (unsigned)(((long long)kr * ldat + ch * 8) * 2);
            }
            int ch = ((n0 + (c >> 2) * 64 + h * 32) >> 3) + (c & 3);
            ch = ch < nch ? ch : nch;
            if constexpr (CONV != 0) {
                const int n = ch * 8, tap = n / c_C, ci = n - tap * c_C;
                const int dy = tap / 3, dx = tap - dy * 3;
                return (unsigned)((((dy * c_W + dx) * c_C + ci) * 2) | (dy << 2) | dx);
            }
            return (unsigned)(((long long)kr * ldbt + ch * 8) * 2);
        };
        v0_0 = offs(0, 0, false); v0_1 = offs(1, 0, false);  // B_h0
        v1_0 = offs(0, 0, true);  v1_1 = offs(1, 0, true);   // A_h0
        v2_0 = offs(0, 1, false); v2_1 = offs(1, 1, false);  // B_h1
        v3_0 = offs(0, 1, true);  v3_1 = offs(1, 1, true);   // A_h1
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(p_lds_addr(smem) + wave * 2048);
    int s_kt = 0;  // stream cursor (K-tile relative to kt0)
    auto issue = [&](auto Jc) {
        constexpr int J = decltype(Jc)::value;
        if (s_kt < total_kt && !(dbg & 4)) {
            const unsigned dst = lds0 + (s_kt & 1) * P_BUF + J * P_HALF;
            const size_t krow = (size_t)(kt0 + s_kt) * P_BK;
            const bf16_t* base = (J & 1) ? At + krow * ldat : Bt + krow * ldbt;
            const unsigned va = J == 0 ? v0_0 : J == 1 ? v1_0 : J == 2 ? v2_0 : v3_0;
            const unsigned vb = J == 0 ? v0_1 : J == 1 ? v1_1 : J == 2 ? v2_1 : v3_1;
            if constexpr (CONV != 0 && !(J & 1)) {
                if (J == 0) {  // decompose the two k-rows (output pixels) of this K-tile once
                    auto pix = [&](int i, int& o, unsigned& yx) {
                        const unsigned m = (unsigned)krow + (wave * 2 + i) * 4 + (lane >> 4);
                        const unsigned b = p_fastdiv(m, c_mgh, c_shh);
                        const unsigned r = m - b * (unsigned)c_HoWo;
                        const unsigned oy = p_fastdiv(r, c_mgw, c_shw);
                        const unsigned ox = r - oy * (unsigned)c_Wo;
                        const int y0 = (int)oy * c_s - 1, x0 = (int)ox * c_s - 1;
                        o = ((((int)b * c_H + y0) * c_W + x0) * c_C) * 2;
                        yx = ((unsigned)y0 << 16) | ((unsigned)x0 & 0xffffu);
                    };
                    pix(0, o_0, yx_0);
                    pix(1, o_1, yx_1);
                }
                auto src = [&](unsigned tapv, int o, unsigned yx) -> const char* {
                    const int y = ((int)yx >> 16) + (int)((tapv >> 2) & 3);
                    const int x = (int)(short)(yx & 0xffffu) + (int)(tapv & 3);
                    const bool ok = (unsigned)y < (unsigned)c_H && (unsigned)x < (unsigned)c_W;
                    return ok ? (const char*)Bt + (long long)(o + (int)(tapv & ~15u)) : c_zero;
                };
                p_glds2v(src(va, o_0, yx_0), src(vb, o_1, yx_1), dst, dst + 1024);
            } else {
                p_glds2(base, va, vb, dst, dst + 1024);
            }
        }
        if (J == 3) ++s_kt;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    const int kl = 8 * (lane >> 4);
    bf16x8 dbg_z;
#pragma unroll
    for (int j = 0; j < 8; ++j) dbg_z[j] = (__bf16)(0.001f * (float)(lane + j));
    auto dbg_frag = [&]() -> bf16x8 { return dbg_z; };
    auto ldA = [&](int buf, int h, int mi, int ks) -> bf16x8 {
        if (dbg & 2) return dbg_frag();
        return q_frag(smem + buf * P_BUF + (h ? 3 : 1) * P_HALF, ks * 32 + kl, wr * 64 + mi * 16, lane);
    };
    auto ldB = [&](int buf, int h, int ni, int ks) -> bf16x8 {
        if (dbg & 2) return dbg_frag();
        return q_frag(smem + buf * P_BUF + (h ? 2 : 0) * P_HALF, ks * 32 + kl, wc * 32 + ni * 16, lane);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    bf16x8 a[4][2], b0[2][2], b1[2][2];
    f32x4 accb[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    const bool do_cs = COLSUM && (n0 == 0) && colsum != nullptr;
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    // Partial tiles (the 1x1-convolution weight gradients of ResNet's first stages: dW [64, 256], [128, 512] ...
    // over hundreds of thousands of pixels — one or two tiles, HBM-streaming): 16-row / 16-column fragment blocks
    // that lie wholly outside M x N are skipped (wave-uniform masks), so a 64-row product costs a quarter of the
    // tile's MFMAs instead of all of them and stays HBM-bound.  Full tiles take the unconditional sequence.
    [[maybe_unused]] unsigned mmask = 0xffu, nmask = 0xfu;
    if constexpr (PARTIAL) {
        mmask = nmask = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) mmask |= (m0 + wr * 128 + (i >> 2) * 64 + (i & 3) * 16 < M ? 1u : 0u) << i;
#pragma unroll
        for (int i = 0; i < 4; ++i) nmask |= (n0 + wc * 64 + (i >> 1) * 32 + (i & 1) * 16 < N ? 1u : 0u) << i;
        mmask = __builtin_amdgcn_readfirstlane(mmask);
        nmask = __builtin_amdgcn_readfirstlane(nmask);
    }
    auto mma = [&](auto QMc, auto QNc, bf16x8 (&bb)[2][2]) {
        constexpr int QM = decltype(QMc)::value, QN = decltype(QNc)::value;
        if (dbg & 8) return;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (CONV == 2 && QM == 0) {  // P0 has just loaded b0, P1 b1
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) bb[ni][ks] = p_relu8(bb[ni][ks]);
        }
        if constexpr (COLSUM && QM == QN) {  // P0 and P2: the phases that have just loaded `a`
            if (do_cs) {
                auto cs = [&](bf16x8 (&af)[2]) {
                    accb[QM] = P_MFMA(ones, af[0], accb[QM]);
                    accb[QM] = P_MFMA(ones, af[1], accb[QM]);
                };
                if (wc == 0) cs(a[0]);
                else if (wc == 1) cs(a[1]);
                else if (wc == 2) cs(a[2]);
                else cs(a[3]);
            }
        }
        if constexpr (!PARTIAL) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[QM * 4 + mi][QN * 2 + ni] =
                            P_MFMA(bb[ni][ks], a[mi][ks], acc[QM * 4 + mi][QN * 2 + ni]);
        } else {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                if (!((mmask >> (QM * 4 + mi)) & 1u)) continue;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    if (!((nmask >> (QN * 2 + ni)) & 1u)) continue;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        acc[QM * 4 + mi][QN * 2 + ni] =
                            P_MFMA(bb[ni][ks], a[mi][ks], acc[QM * 4 + mi][QN * 2 + ni]);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    if (total_kt > 0) {
        issue(I0{}); issue(I1{}); issue(I2{}); issue(I3{});
        issue(I0{}); issue(I1{}); issue(I2{});
        if (total_kt >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();

        for (int T = 0; T < total_kt; ++T) {
            const int cb = T & 1;
            // ---------------- P0
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) b0[ni][ks] = ldB(cb, 0, ni, ks);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) a[mi][ks] = ldA(cb, 0, mi, ks);
            __builtin_amdgcn_sched_barrier(0);
            issue(I3{});
            asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");  // the 8 B_h0 reads are retired
            __builtin_amdgcn_s_barrier();
            mma(I0{}, I0{}, b0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- P1
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) b1[ni][ks] = ldB(cb, 1, ni, ks);
            __builtin_amdgcn_sched_barrier(0);
            issue(I0{});
            __builtin_amdgcn_s_barrier();
            mma(I0{}, I1{}, b1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- P2
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) a[mi][ks] = ldA(cb, 1, mi, ks);
            __builtin_amdgcn_sched_barrier(0);
            issue(I1{});
            __builtin_amdgcn_s_barrier();
            mma(I1{}, I1{}, b1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- P3
            issue(I2{});
            if (T + 2 < total_kt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            mma(I1{}, I0{}, b0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
    }

    // ---- column sums: D[n][m] is the same for every n, so lanes 0..15 hold m = l, register 0
    if constexpr (COLSUM) {
        if (do_cs && lane < 16) {
#pragma unroll
            for (int qm = 0; qm < 2; ++qm) {
                const int m = m0 + wr * 128 + qm * 64 + wc * 16 + lane;
                if (m < M) {
                    if (splits == 1) colsum[m] = accumulate ? colsum[m] + accb[qm][0] : accb[qm][0];
                    else colsum_part[(size_t)split * M + m] = accb[qm][0];
                }
            }
        }
    }
    // ---- epilogue: slab (splits > 1) or C
    const f32x4 zero4 = {0, 0, 0, 0};
    char* stg = smem + 2 * P_BUF + wave * P_STG_WAVE;
    float* out = (splits == 1) ? C : slabs + (size_t)split * M * N;
    const long long ldo = (splits == 1) ? ldc : N;
    const float al = (splits == 1) ? alpha : 1.f;
    const bool accu = (splits == 1) && accumulate;
    if (dbg & 1) {
        asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[7][3][3]), "v"(acc[3][1][2]));
        return;
    }
    if (m0 + P_BM <= M && n0 + P_BN <= N)
        p_store_f32<true, false>(acc, stg, al, zero4, nullptr, 0, accu, out, ldo, m0 + wr * 128,
                                 n0 + wc * 64, M, N, lane);
    else
        p_store_f32<false, false>(acc, stg, al, zero4, nullptr, 0, accu, out, ldo, m0 + wr * 128,
                                  n0 + wc * 64, M, N, lane);
}

// =====================================================================================
// Round 5: the K-SPLIT form of the same product (plain operands, full or clamped tiles).
//
// What the ablations of the kernel above showed (tools/build_variant.sh -DTN256_ABL, profiles/r05a): its K-loop is
// not bound by the transposed reads (reads + skeleton alone: 0.2 us per K-tile) but by two things that add up —
// the four-phase skeleton around the MFMAs (1.67 us per K-tile with MFMAs only: 8 intervals of 16 MFMAs + ~110
// cycles of barrier / wait overhead each, where the NT kernel's two-phase form has 4 of 32) and the LDS-DMA stream
// (1.6 us per K-tile with the DMA only), whose requests were 128-byte (A) and 64-byte (B) pieces of a row: a
// half-tile was "half of the tile's COLUMNS over all 64 k-rows", the NT kernel's split, which suits operands that
// are contiguous along k.  These operands are contiguous along m / n, so here a stage is "32 K-ROWS of all 256
// columns": a row is one contiguous 512-byte run, a DMA piece two whole rows.
//
//  * stage = 32 k-rows of A (16 KiB) + 32 k-rows of B (16 KiB); ring of 4 stages (A stages at 0 .. 64 KiB, B stages
//    at 64 .. 128 KiB, so that every fragment address is a loop-invariant VGPR + a 16-bit immediate);
//  * ONE uniform phase per stage: {LOAD: 24 ds_read_b64_tr_b16 (8 A + 4 B fragments: the k extent of a stage IS the
//    32 of one MFMA) + the DMA of stage s + 3 (two A pieces, two B pieces per wave) + the counted wait for stage
//    s + 1 + lgkmcnt(0); s_barrier; MMA: 32 MFMAs; s_barrier} — the two wave rows one barrier apart as before;
//    48 fragment registers instead of 64, reads balanced 24 / 24 instead of 24 / 8 / 16 / 0;
//  * hazards (interval I_n between barriers n - 1 and n; wr = 0 runs LOAD(s) in I_2s and MMA(s) in I_2s+1, wr = 1
//    one interval later):
//      WAR  LOAD(s) issues stage s + 3 = s - 1 (mod 4) over what LOAD(s - 1) read; every LOAD ends with
//           lgkmcnt(0) before its barrier, so either group's reads of LOAD(s - 1) are complete at least one
//           barrier before either group's LOAD(s);
//      RAW  each wave waits at the end of LOAD(s) until all but its 8 youngest pieces (stages s + 2, s + 3) have
//           landed, i.e. its pieces of stage s + 1; wr = 0 reads stage s + 1 in I_2s+2, after the barrier that
//           ends wr = 1's LOAD(s) in I_2s+1.
// Image of a stage half: k-row r at r * 512 B, 32-byte unit u (16 m- or n-values) at position u ^ k_swz(r): the 32
// lanes of a ds_read_b64_tr_b16 group read 8 rows x 32 B and k_swz takes 8 distinct values on them (conflict-free).
// Same accumulator layout, same per-accumulator summation order (k ascending) as the kernel above: results are
// bit-identical, so SSL4GIE_TN256K=0 (A/B timing) changes nothing else.
DEVI int k_swz(int k) { return (k & 3) | ((k >> 1) & 4); }

template <bool COLSUM, int RING>
__global__ __launch_bounds__(512, 2) void gemm_bf16_tn256k_kernel(
    const bf16_t* __restrict__ At, long long ldat, const bf16_t* __restrict__ Bt, long long ldbt,
    float* __restrict__ C, long long ldc, float* __restrict__ slabs, int M, int N, int K,
    int tiles_n, int ntiles, int splits, float alpha, int accumulate, float* __restrict__ colsum,
    float* __restrict__ colsum_part, TnExtras ex) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bool tn_m_inner_ok = ex.m_inner != 0;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int all_tiles = ntiles + ex.total_tiles;
    const int split = bid / all_tiles;
    int tile = bid % all_tiles;
    if (tile >= ntiles) {  // grouped launch: this workgroup belongs to a later product (uniform)
        const int rel = tile - ntiles;
#pragma unroll
        for (int i = 0; i < TN_GROUP_MAX - 1; ++i) {
            if (i < ex.n && rel >= ex.p[i].tile0 && rel < ex.p[i].tile0 + ex.p[i].ntiles) {
                tile = rel - ex.p[i].tile0;
                At = (const bf16_t*)ex.p[i].At; ldat = ex.p[i].ldat;
                Bt = (const bf16_t*)ex.p[i].Bt; ldbt = ex.p[i].ldbt;
                C = ex.p[i].C; ldc = ex.p[i].ldc; slabs = ex.p[i].slabs;
                M = ex.p[i].M; N = ex.p[i].N; tiles_n = ex.p[i].tiles_n;
                colsum = ex.p[i].colsum; colsum_part = ex.p[i].colsum_part;
                alpha = ex.p[i].alpha; accumulate = ex.p[i].accumulate;
            }
        }
    }
    const int tiles_m = (M + P_BM - 1) / P_BM;
    const bool m_inner = tiles_m < tiles_n && tn_m_inner_ok;  // (tile walk: see the kernel above)
    const int m0 = (m_inner ? tile % tiles_m : tile / tiles_n) * P_BM;
    const int n0 = (m_inner ? tile / tiles_m : tile % tiles_n) * P_BN;
    const int nkt = K / P_BK;
    const int kt0 = (int)((long long)nkt * split / splits);
    const int kt1 = (int)((long long)nkt * (split + 1) / splits);
    const int total_kt = kt1 - kt0;
    const int nph = 2 * total_kt;  // 32-row stages

    // ---- LDS-DMA stream.  Piece j (0, 1) of this wave in a stage half: LDS rows 4 wave + 2 j + (lane >> 5), 16-byte
    // slot lane & 31 of the row; the slot holds the logical chunk whose swizzled position it is.
    unsigned va0, va1, vb0, vb1;
    {
        auto offs = [&](int j, bool is_a) -> unsigned {
            const int kr = wave * 4 + j * 2 + (lane >> 5);
            const int slot = lane & 31;
            int ch = (((slot >> 1) ^ k_swz(kr)) << 1) + (slot & 1) + ((is_a ? m0 : n0) >> 3);
            const int last = ((is_a ? M : N) >> 3) - 1;  // last valid 16-byte chunk of a row
            ch = ch < last ? ch : last;
            return (unsigned)(((long long)kr * (is_a ? ldat : ldbt) + ch * 8) * 2);
        };
        va0 = offs(0, true); va1 = offs(1, true);
        vb0 = offs(0, false); vb1 = offs(1, false);
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(p_lds_addr(smem) + wave * 2048);
    // ring slots: A stages 0-3 at 0 .. 64 KiB, B stages 0-3 at 64 .. 128 KiB; RING == 5: a fifth stage in the 32 KiB the
    // epilogue's staging area occupies AFTER the loop (A at 128 KiB, B at 144 KiB)
    auto a_slot = [](int t) -> unsigned { return t < 4 ? (unsigned)t * 16384u : 131072u; };
    auto b_slot = [](int t) -> unsigned { return t < 4 ? 65536u + (unsigned)t * 16384u : 147456u; };
    const char* baseA = (const char*)(At + (size_t)kt0 * P_BK * ldat);
    const char* baseB = (const char*)(Bt + (size_t)kt0 * P_BK * ldbt);
    const long long stepA = 64LL * ldat, stepB = 64LL * ldbt;  // 32 rows, in bytes
    int s_ph = 0;  // stream cursor (stage index)
    auto issue = [&](auto STc) {
        constexpr int ST = decltype(STc)::value;  // ring slot of the stage being issued
        if (s_ph < nph && !(TN256_ABL & 4)) {
            const unsigned da = lds0 + a_slot(ST), db = lds0 + b_slot(ST);
            p_glds2(baseA, va0, va1, da, da + 1024);
            p_glds2(baseB, vb0, vb1, db, db + 1024);
            if (!(TN256_ABL & 16)) {  // (bit 4: the stream re-reads its first stage — an L2-resident DMA stream)
                baseA += stepA;
                baseB += stepB;
            }
        }
        ++s_ph;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;

    // ---- fragment addresses: lane (g = l >> 4, q = (l & 15) >> 2, p = l & 3) reads row 8 g + q (+ 4 for the second
    // read) of the stage, 8 bytes at p * 8 of the swizzled 32-byte unit
    const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
    const int frow = (8 * fg + fq) * 512 + fp * 8, fsw = k_swz(8 * fg + fq);
    unsigned aoff[8], boff[4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) aoff[mi] = (unsigned)(frow + (((wr * 8 + mi) ^ fsw) << 5));
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) boff[ni] = (unsigned)(65536 + frow + (((wc * 4 + ni) ^ fsw) << 5));
    bf16x8 dbg_z;
#pragma unroll
    for (int j = 0; j < 8; ++j) dbg_z[j] = (__bf16)(0.001f * (float)(lane + j));
    unsigned aoff4[8], boff4[4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) aoff4[mi] = aoff[mi] + 131072u;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) boff4[ni] = boff[ni] + 81920u;   // 147456 = 65536 + 81920
    auto frag = [&](unsigned off, int imm) -> bf16x8 {
        if (TN256_ABL & 2) return dbg_z;
        typedef __attribute__((address_space(3))) s16x4* lp_t;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(smem + off + imm));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(smem + off + imm + 2048));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    bf16x8 a[8], b[4];
    f32x4 accb[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    const bool do_cs = COLSUM && (n0 == 0) && colsum != nullptr;
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    // one phase on ring slot ST (the stage it issues, s + 3, lives in slot (ST + 3) & 3)
    auto phase = [&](auto STc, int s) {
        constexpr int ST = decltype(STc)::value;
        // ---------------- LOAD
        // (slot 4 of the five-stage ring lies beyond a 16-bit immediate from the slot-0 addresses: its own base registers)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) b[ni] = ST < 4 ? frag(boff[ni], ST * 16384) : frag(boff4[ni], 0);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) a[mi] = ST < 4 ? frag(aoff[mi], ST * 16384) : frag(aoff4[mi], 0);
        __builtin_amdgcn_sched_barrier(0);
        issue(std::integral_constant<int, (ST + RING - 1) % RING>{});
        // stage s + 1 must have landed: all but this wave's pieces of the stages after it may stay in flight
        if (s + RING - 1 < nph) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (RING - 2)) : "memory");
        else if (RING == 5 && s + 3 < nph) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (s + 2 < nph) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---------------- MMA
        __builtin_amdgcn_s_setprio(1);
        if constexpr (COLSUM && !(TN256_ABL & 8)) {
            if (do_cs) {
                auto cs = [&](const bf16x8& lo, const bf16x8& hi) {
                    accb[0] = P_MFMA(ones, lo, accb[0]);
                    accb[1] = P_MFMA(ones, hi, accb[1]);
                };
                if (wc == 0) cs(a[0], a[4]);
                else if (wc == 1) cs(a[1], a[5]);
                else if (wc == 2) cs(a[2], a[6]);
                else cs(a[3], a[7]);
            }
        }
        if constexpr (!(TN256_ABL & 8)) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = P_MFMA(b[ni], a[mi], acc[mi][ni]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };

    if (total_kt > 0) {
        issue(I0{}); issue(I1{}); issue(I2{});
        if constexpr (RING == 5) issue(I3{});
        // stage 0 landed: the later ones (nph is even: >= 2) may stay in flight
        if (nph >= RING - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (RING - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // nph == 2: stage 1 may
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();
        int s = 0;
        if constexpr (RING == 4) {
            for (; s + 3 < nph; s += 4) {
                phase(I0{}, s);
                phase(I1{}, s + 1);
                phase(I2{}, s + 2);
                phase(I3{}, s + 3);
            }
            if (s < nph) {  // an odd number of K-tiles: two more stages, slots 0 and 1
                phase(I0{}, s);
                phase(I1{}, s + 1);
            }
        } else {
            // one loop body of five guarded phases (wave-uniform conditions): a separate tail of up to four phases
            // made the register allocator migrate the accumulators between the copies (256 VGPRs + spills)
            for (; s < nph; s += 5) {
                phase(I0{}, s);
                if (s + 1 < nph) phase(I1{}, s + 1);
                if (s + 2 < nph) phase(I2{}, s + 2);
                if (s + 3 < nph) phase(I3{}, s + 3);
                if (s + 4 < nph) phase(I4{}, s + 4);
            }
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
    }

    if constexpr (COLSUM) {  // D[n][m] is the same for every n: lanes 0..15 hold m = l, register 0
        if (do_cs && lane < 16) {
#pragma unroll
            for (int qm = 0; qm < 2; ++qm) {
                const int m = m0 + wr * 128 + qm * 64 + wc * 16 + lane;
                if (m < M) {
                    if (splits == 1) colsum[m] = accumulate ? colsum[m] + accb[qm][0] : accb[qm][0];
                    else colsum_part[(size_t)split * M + m] = accb[qm][0];
                }
            }
        }
    }
    const f32x4 zero4 = {0, 0, 0, 0};
    char* stg = smem + 2 * P_BUF + wave * P_STG_WAVE;
    float* out = (splits == 1) ? C : slabs + (size_t)split * M * N;
    const long long ldo = (splits == 1) ? ldc : N;
    const float al = (splits == 1) ? alpha : 1.f;
    const bool accu = (splits == 1) && accumulate;
    if (m0 + P_BM <= M && n0 + P_BN <= N)
        p_store_f32<true, false>(acc, stg, al, zero4, nullptr, 0, accu, out, ldo, m0 + wr * 128,
                                 n0 + wc * 64, M, N, lane);
    else
        p_store_f32<false, false>(acc, stg, al, zero4, nullptr, 0, accu, out, ldo, m0 + wr * 128,
                                  n0 + wc * 64, M, N, lane);
}

// =====================================================================================
// host side
// =====================================================================================
static int tn256_mode() {  // SSL4GIE_TN256: "0" never, "1" whenever possible, unset = heuristic
    static int v = -2;
    if (v == -2) {
        const char* s = getenv("SSL4GIE_TN256");
        v = !s ? -1 : (s[0] == '0' ? 0 : 1);
    }
    return v;
}

bool ssl4gie_internal_tn256_ok(const ssl4gie_gemm_desc* d) {
    const int mode = tn256_mode();
    if (mode == 0 && !d->conv) return false;
    if (d->K % P_BK != 0 || d->K < P_BK) return false;
    if (d->conv)
        return ssl4gie_internal_conv_geom_ok(d->conv) && d->conv->C % 8 == 0 &&
               d->N == 9 * d->conv->C && (long long)d->K == conv_rows(d->conv) &&
               (long long)d->K * d->sAk * 2 < (1LL << 32);
    if ((long long)d->K * d->sAk * 2 >= (1LL << 32) || (long long)d->K * d->sBk * 2 >= (1LL << 32))
        return false;  // 32-bit per-lane byte offsets are relative to a per-K-tile base: generous
    if (mode == 1) return true;
    // (round 5, measured and not changed: the narrow long-contraction products of ResNet50's layer1 — dW [256, 64],
    // [64, 256], [64, 64], [128, 256] over 802 816 pixels — and the MLP heads' short ones stay on the 128-tile
    // kernel: 98 / 98 / 70 / 121 us there (5.2 TB/s) against 158 / 120 / 108 / 140 us on the partial-tile path
    // here, tools/tn_small_bench.py, profiles/r05_tn_small_routing.log; the "206 us" those launches showed in the
    // round-4 MoCo profile was concurrency with the data-gradient chain, not the kernel)
    return d->K >= 16 * P_BK && (long long)d->M * d->N >= 128 * 128 * 4;
}

int ssl4gie_internal_tn256_splits(const ssl4gie_gemm_desc* d) {
    const int tiles = ((d->M + P_BM - 1) / P_BM) * ((d->N + P_BN - 1) / P_BN);
    const int nkt = d->K / P_BK;
    // SSL4GIE_TN_FILL1 (percent, default 75): share of the CUs a lone product aims to fill.  These launches run
    // beside the data-gradient chain on a weight-gradient stream, like the pairs (whose fill is 75 % too): MoCo-R50
    // step 55.5 / 55.7 -> 54.9 / 55.0 ms in a same-box A/B (profiles/r04cy_tn_fill1.log), depth and MAE unchanged
    static int fill1 = -1;
    if (fill1 < 0) { const char* e = getenv("SSL4GIE_TN_FILL1"); fill1 = e ? atoi(e) : 75; if (fill1 < 10) fill1 = 10; }
    int s = (ssl4gie_internal_compute_cus() * fill1 / 100 + tiles / 2) / tiles;  // one workgroup per CU
    if (s > nkt / 8) s = nkt / 8;       // at least 8 K-tiles per split
    if (s < 1) s = 1;
    if (s > 256) s = 256;               // (a single-tile product over a long contraction: one split per CU)
    return s;
}

// Predicted HBM bytes of a paired launch with `splits` K-splits (tools/tn_traffic_model.py, checked against the PMC
// counters: 405 MB per launch predicted, 397 measured): consecutive logical workgroups share an XCD's L2
// (xcd_remap), so an XCD chunk holding r m-tiles and c n-tiles of one split of one product reads r + c operand
// slabs; plus the fp32 slabs (GEMM write, reduce read + write).
static double tn256_pair_traffic(const ssl4gie_gemm_desc* const* ds, int n, int splits, bool m_inner_ok) {
    int tm[2], tn[2], nt[2], all = 0;
    for (int i = 0; i < n; ++i) {
        tm[i] = (ds[i]->M + P_BM - 1) / P_BM; tn[i] = (ds[i]->N + P_BN - 1) / P_BN;
        if (tm[i] > 64 || tn[i] > 64) return 1e30;  // (bit sets below; no such product in the models)
        nt[i] = tm[i] * tn[i]; all += nt[i];
    }
    const int G = all * splits, q = G / 8, r = G % 8;
    long long slabs = 0;
    int lo = 0;
    for (int x = 0; x < 8; ++x) {  // the 8 contiguous chunks of logical ids xcd_remap deals to the XCDs
        const int hi = lo + q + (x < r ? 1 : 0);
        int cur_s = -1, cur_p = -1;
        unsigned long long rows = 0, cols = 0;
        for (int lid = lo; lid < hi; ++lid) {
            const int sidx = lid / all;
            int t = lid % all, p = 0;
            while (p < n - 1 && t >= nt[p]) t -= nt[p++];
            if (sidx != cur_s || p != cur_p) {
                slabs += __builtin_popcountll(rows) + __builtin_popcountll(cols);
                rows = cols = 0; cur_s = sidx; cur_p = p;
            }
            const bool m_inner = tm[p] < tn[p] && m_inner_ok;
            const int mt = m_inner ? t % tm[p] : t / tn[p], ntile = m_inner ? t / tm[p] : t % tn[p];
            rows |= 1ull << mt; cols |= 1ull << ntile;
        }
        slabs += __builtin_popcountll(rows) + __builtin_popcountll(cols);
        lo = hi;
    }
    double out = 0;
    for (int i = 0; i < n; ++i) out += (double)ds[i]->M * ds[i]->N * 4;
    return (double)slabs * 256 * ((double)ds[0]->K / splits) * 2 + (splits > 1 ? out * (2 * splits + 1) : out);
}

int ssl4gie_internal_tn256_pair_splits(const ssl4gie_gemm_desc* a, const ssl4gie_gemm_desc* b) {
    const int tiles = ((a->M + P_BM - 1) / P_BM) * ((a->N + P_BN - 1) / P_BN) +
                      ((b->M + P_BM - 1) / P_BM) * ((b->N + P_BN - 1) / P_BN);
    const int nkt = a->K / P_BK;
    // SSL4GIE_TN_FILL (percent, default 75): share of the CUs a paired launch aims to fill — on the
    // weight-gradient side stream fewer, longer workgroups mean less slab traffic (measured:
    // profiles/r01w_ab_grid_sizing.log)
    static int fill = -1;
    if (fill < 0) {
        const char* e = getenv("SSL4GIE_TN_FILL");
        fill = e ? atoi(e) : 75;
        if (fill < 10) fill = 10;
        if (fill > 200) fill = 200;
    }
    const int target = ssl4gie_internal_compute_cus() * fill / 100;
    int s = (target + tiles / 2) / tiles;
    if (s > nkt / 8) s = nkt / 8;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    // SSL4GIE_TN_ALIGN=1 (A/B): among the split counts that keep 70 .. 110 % of the target's workgroups, the one
    // with the least predicted HBM traffic (chunks of whole K-splits per XCD read every operand slab once)
    static int align = -1;
    if (align < 0) { const char* e = getenv("SSL4GIE_TN_ALIGN"); align = e ? atoi(e) : 0; }
    if (align) {
        const ssl4gie_gemm_desc* ds[2] = {a, b};
        int best = s;
        double best_t = tn256_pair_traffic(ds, 2, s, true);
        for (int c = 1; c <= 64 && c <= nkt / 8; ++c) {
            const int wgs = c * tiles;
            if (wgs * 10 < target * 7 || wgs * 10 > target * 11) continue;
            const double t = tn256_pair_traffic(ds, 2, c, true);
            if (t < best_t * 0.97) { best_t = t; best = c; }
        }
        s = best;
    }
    return s;
}

int ssl4gie_internal_tn256_group_splits(const ssl4gie_gemm_desc* descs, int n) {
    int tiles = 0;
    for (int i = 0; i < n; ++i)
        tiles += ((descs[i].M + P_BM - 1) / P_BM) * ((descs[i].N + P_BN - 1) / P_BN);
    const int nkt = descs[0].K / P_BK;
    const int cus = ssl4gie_internal_compute_cus();
    if (tiles * 10 >= cus * 7) return 1;  // >= 70 % of the CUs busy with whole-K tiles: no slabs at all
    int s = (cus + tiles / 2) / tiles;
    if (s > nkt / 8) s = nkt / 8;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return s;
}

static int tn256_launch_impl(const ssl4gie_gemm_desc* descs, int n, int splits, float* const* slabs,
                             float* const* cs, hipStream_t st);

int ssl4gie_internal_tn256_launch(const ssl4gie_gemm_desc* d, float* slabs, float* colsum_part,
                                  hipStream_t st) {
    return tn256_launch_impl(d, 1, ssl4gie_internal_tn256_splits(d), &slabs, &colsum_part, st);
}
int ssl4gie_internal_tn256_launch_pair(const ssl4gie_gemm_desc* a, const ssl4gie_gemm_desc* b,
                                       int splits, float* slabs_a, float* cs_a, float* slabs_b,
                                       float* cs_b, hipStream_t st) {
    const ssl4gie_gemm_desc ds[2] = {*a, *b};
    float* sl[2] = {slabs_a, slabs_b};
    float* cp[2] = {cs_a, cs_b};
    return tn256_launch_impl(ds, 2, splits, sl, cp, st);
}
int ssl4gie_internal_tn256_launch_group(const ssl4gie_gemm_desc* descs, int n, int splits,
                                        float* const* slabs, float* const* cs, hipStream_t st) {
    return tn256_launch_impl(descs, n, splits, slabs, cs, st);
}

static int tn256_launch_impl(const ssl4gie_gemm_desc* descs, int n, int splits, float* const* slabs_v,
                             float* const* cs_v, hipStream_t st) {
    if (n < 1 || n > TN_GROUP_MAX) return ARG_ERR;
    const ssl4gie_gemm_desc* d = &descs[0];
    float* slabs = slabs_v[0];
    float* colsum_part = cs_v[0];
    const int tm = (d->M + P_BM - 1) / P_BM, tn = (d->N + P_BN - 1) / P_BN;
    TnExtras sec{};
    bool any_colsum = d->colsum_a != nullptr;
    for (int i = 1; i < n; ++i) {
        const ssl4gie_gemm_desc* d2 = &descs[i];
        if (d2->K != d->K || d2->conv) return ARG_ERR;
        const int tm2 = (d2->M + P_BM - 1) / P_BM, tn2 = (d2->N + P_BN - 1) / P_BN;
        TnSecond& q = sec.p[i - 1];
        q.At = d2->A; q.ldat = d2->sAk; q.Bt = d2->B; q.ldbt = d2->sBk;
        q.C = (float*)d2->C; q.ldc = d2->ldc; q.slabs = slabs_v[i];
        q.M = d2->M; q.N = d2->N; q.tiles_n = tn2; q.ntiles = tm2 * tn2;
        q.colsum = d2->colsum_a; q.colsum_part = cs_v[i];
        q.alpha = d2->alpha; q.accumulate = d2->accumulate; q.tile0 = sec.total_tiles;
        sec.total_tiles += q.ntiles;
        any_colsum = any_colsum || d2->colsum_a != nullptr;
    }
    sec.n = n - 1;
    static int m_inner = -1;  // SSL4GIE_TN_INNER=n: the row-by-row tile walk of rounds 1-2 (A/B)
    if (m_inner < 0) { const char* s = getenv("SSL4GIE_TN_INNER"); m_inner = (s && (s[0] == 'n' || s[0] == 'N')) ? 0 : 1; }
    sec.m_inner = m_inner;
    dim3 grid((tm * tn + sec.total_tiles) * splits), block(512);
    ConvK ck{};
    if (d->conv) {
        const int rc = ssl4gie_internal_conv_k(d->conv, &ck);
        if (rc) return rc;
    }
#define Q_LAUNCH(CS_, CONV_) do { if (partial) Q_LAUNCH_P(CS_, CONV_, true); else Q_LAUNCH_P(CS_, CONV_, false); } while (0)
#define Q_LAUNCH_P(CS_, CONV_, PART_)                                                              \
    do {                                                                                           \
        auto kfn = gemm_bf16_tn256_kernel<CS_, CONV_, PART_>;                                      \
        static bool attr_set = false; /* idempotent; a benign race only repeats the call */        \
        if (!attr_set) {                                                                           \
            HIP_RET(hipFuncSetAttribute((const void*)kfn,                                          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BYTES)); \
            attr_set = true;                                                                       \
        }                                                                                          \
        hipLaunchKernelGGL(kfn, grid, block, P_LDS_BYTES, st, (const bf16_t*)d->A, d->sAk,         \
                           (const bf16_t*)d->B, d->sBk, (float*)d->C, d->ldc, slabs, d->M, d->N,   \
                           d->K, tn, tm * tn, splits, d->alpha, d->accumulate, d->colsum_a,        \
                           colsum_part, ck, sec);                                                  \
    } while (0)
    const int cv = d->conv ? (d->conv->relu ? 2 : 1) : 0;
    // a lone plain product whose tiles M x N does not fill (ResNet's narrow 1x1 weight gradients)
    const bool partial = n == 1 && cv == 0 && (d->M % P_BM != 0 || d->N % P_BN != 0);
    // plain operands, no skipped fragment blocks: the k-split kernel (SSL4GIE_TN256K=0: the column-split one; the
    // results are bit-identical, the knob exists for A/B timing)
    static int ksplit = -1;
    if (ksplit < 0) {  // SSL4GIE_TN256K=0: the column-split kernel of rounds 1-4 (A/B); =5 (debug library only): a five-stage ring, measured null
        const char* s = getenv("SSL4GIE_TN256K");
        ksplit = !s ? 1 : (s[0] == '0' ? 0 : 1);
#ifdef SSL4GIE_DEBUG_KNOBS
        if (s && s[0] == '5') ksplit = 5;
#endif
    }
    if (ksplit && cv == 0 && !partial) {
#define K_LAUNCH(CS_, RING_)                                                                          \
    do {                                                                                           \
        auto kfn = gemm_bf16_tn256k_kernel<CS_, RING_>;                                            \
        static bool attr_set = false;                                                              \
        if (!attr_set) {                                                                           \
            HIP_RET(hipFuncSetAttribute((const void*)kfn,                                          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BYTES)); \
            attr_set = true;                                                                       \
        }                                                                                          \
        hipLaunchKernelGGL(kfn, grid, block, P_LDS_BYTES, st, (const bf16_t*)d->A, d->sAk,         \
                           (const bf16_t*)d->B, d->sBk, (float*)d->C, d->ldc, slabs, d->M, d->N,   \
                           d->K, tn, tm * tn, splits, d->alpha, d->accumulate, d->colsum_a,        \
                           colsum_part, sec);                                                      \
    } while (0)
        // SSL4GIE_TN256K=5: a five-stage ring (the fifth stage in the epilogue's staging area: one more stage in
        // flight, 24 more address registers) — built, bit-identical, measured NULL against the four-stage ring
        // (pair launches 589 vs 586-591 us, MAE step 22.08 vs 22.09-22.13 ms same-box: profiles/r05j): the
        // stream's cost is not latency a deeper ring would cover (an L2-RESIDENT stream makes the kernel 14 % faster,
        // profiles/r05b, so it is the L2 -> LDS path under load)
#ifdef SSL4GIE_DEBUG_KNOBS
        if (ksplit == 5) { if (any_colsum) K_LAUNCH(true, 5); else K_LAUNCH(false, 5); }
        else
#endif
        { if (any_colsum) K_LAUNCH(true, 4); else K_LAUNCH(false, 4); }
#undef K_LAUNCH
        LAUNCH_CHECK();
        return 0;
    }
    if (any_colsum) {
        if (cv == 0) Q_LAUNCH(true, 0);
        else if (cv == 1) Q_LAUNCH(true, 1);
        else Q_LAUNCH(true, 2);
    } else {
        if (cv == 0) Q_LAUNCH(false, 0);
        else if (cv == 1) Q_LAUNCH(false, 1);
        else Q_LAUNCH(false, 2);
    }
#undef Q_LAUNCH
#undef Q_LAUNCH_P
    LAUNCH_CHECK();
    return 0;
}
