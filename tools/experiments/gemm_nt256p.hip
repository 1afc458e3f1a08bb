// EXPERIMENT (not built, kept for the record — see DESIGN.md §3 "what did not work"): measured on
// MI355X at the MAE shapes this two-pass / pipelined-epilogue variant was SLOWER than gemm_nt256.hip
// (enc.qkv 82-92 us against 50 us): its K-loop alone ran at 650-750 TFLOP/s against 1.13 PFLOP/s
// (1.5x the LDS-DMA bytes and 1.33x the fragment reads per MFMA, and a shorter prefetch distance),
// which costs more than the hidden epilogue returns.
// 256x256 bf16 NT GEMM with a PIPELINED epilogue (gfx950):  C[M,N] = epilogue(A[M,K] B[N,K]^T).
//
// gemm_nt256.hip's K-loop runs at 1.1-1.3 PFLOP/s, but its epilogue (bias / GELU math, LDS
// transpose, 128 KiB of stores per tile) is serial with it inside a CU: measured with the K-loop-only
// ablation (SSL4GIE_NT256_NOEPI=1) the epilogues cost 20-50 % of the launch on the short-K shapes of
// this workload (K = 512 / 768).  No second 128-register accumulator set fits next to the first,
// so this variant splits the wave tile instead: a 256x256 workgroup tile is computed in two
// PASSES over K — rows {128 wr + 64 h + 0..63}, h = 0, 1 — each with a 64-register accumulator set
// (4 x 4 MFMA tiles per wave).  While pass p accumulates into set p & 1, the finished set of pass
// p-1 is drained: one 16x16 block per LOAD segment (bias is already in the accumulator's initial
// value), packed to bf16 into the wave-private LDS staging image, and every 4th step a finished
// 16-row block leaves as whole 128-byte row segments.  The partner wave of the SIMD is in its MMA
// segment meanwhile, so the drain's VALU / LDS / store work hides behind MFMAs.
//
// Price: B is streamed twice per tile (48 KiB of LDS-DMA per K-tile and 32 MFMAs per wave instead
// of 64 KiB per 64): 85 FLOP/B against 128 — still well inside what the L2s deliver.
//
// Schedule per K-tile (64 deep), 2 phases of 16 MFMAs per wave, two wave rows staggered by one
// barrier exactly as in gemm_nt256.hip:
//     P0: read B_h0 (4 ds_read_b128) + A (8); issue B_h0(t+2), A(t+2); wait B_h1(t); drain step
//     P1: read B_h1 (4);                      issue B_h1(t+2); wait A(t+1);             drain step
// LDS: 3 K-tile buffers x [B_h0 | A | B_h1] x 16 KiB + 8 x 2 KiB staging = 160 KiB.  A slot is
// refilled two phases after its last read; a half-tile is waited for (counted vmcnt) one phase
// before its first read.  All VMEM traffic of the loop is issued by this code in program order (LDS-DMA and
// bias loads from inline asm, stores through the compiler but never waited on), so the count for
// `s_waitcnt vmcnt(N)` is kept exactly by a software model of the in-order queue: `opc` counts
// issued operations, `mark` remembers the count right after the LDS-DMA that must have landed.
//
// Modes: bf16 outputs with NONE / BIAS / BIAS_GELU_GRAD epilogues and alpha == 1 (the forward and
// data-gradient projections of the transformer blocks); everything else stays on gemm_nt256.hip.
#include "gemm256.h"
#include "prof.h"

#include <stdlib.h>
#include <type_traits>

#define R_BUF 49152
#define R_NBUF 3
#define R_STG (R_NBUF * R_BUF)
#define R_LDS_BYTES (R_STG + 8 * 2048)  // 163840

template <int N> DEVI void r_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most n VMEM operations of this wave are outstanding (n clamped down: stricter = safe)
DEVI void r_wait_vm(int n) {
    switch (n >> 1) {  // every count in this kernel is even
        case 0: r_vmcnt<0>(); break;
        case 1: r_vmcnt<2>(); break;
        case 2: r_vmcnt<4>(); break;
        case 3: r_vmcnt<6>(); break;
        case 4: r_vmcnt<8>(); break;
        case 5: r_vmcnt<10>(); break;
        case 6: r_vmcnt<12>(); break;
        case 7: r_vmcnt<14>(); break;
        case 8: r_vmcnt<16>(); break;
        case 9: r_vmcnt<18>(); break;
        default: r_vmcnt<20>(); break;
    }
}
// 16-byte global load the compiler knows nothing about (no compiler-inserted vmcnt can drain the
// LDS-DMA stream); the caller guarantees the data has landed before the first use
DEVI f32x4 r_asm_load16(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt256p_kernel(
    const bf16_t* __restrict__ A, long long lda, const bf16_t* __restrict__ B, long long ldb,
    bf16_t* __restrict__ C, long long ldc, int M, int N, int K, int tiles_n, int ntiles,
    const float* __restrict__ bias, bf16_t* __restrict__ out2, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int G = gridDim.x;
    const int pos = xcd_remap(blockIdx.x, G);
    const int my_tiles = (ntiles - pos + G - 1) / G;
    const int nk = K / P_BK;
    const int npass = 2 * my_tiles;
    const int total_kt = npass * nk;
    constexpr bool HAS_BIAS = (MODE == SSL4GIE_EPI_BIAS || MODE == SSL4GIE_EPI_BIAS_GELU_GRAD);
    constexpr bool DUAL = (MODE == SSL4GIE_EPI_BIAS_GELU_GRAD);

    // ------------------------------------------------------------------ LDS-DMA stream
    unsigned vb0_0 = 0, vb0_1 = 0, va_0 = 0, va_1 = 0, vb1_0 = 0, vb1_1 = 0;
    auto point_at = [&](int pass) __attribute__((always_inline)) {
        const int tile = pos + (pass >> 1) * G, h = pass & 1;
        const int sm0 = (tile / tiles_n) * P_BM, sn0 = (tile % tiles_n) * P_BN;
        auto offs = [&](int i, int which) -> unsigned {  // which: 0 B_h0, 1 A, 2 B_h1
            const int lr = (wave * 2 + i) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ p_swz(lr);
            if (which == 1) {
                int ra = sm0 + (lr >> 6) * 128 + h * 64 + (lr & 63);
                ra = ra < M ? ra : M - 1;
                return (unsigned)(((long long)ra * lda + c * 8) * 2);
            }
            int rb = sn0 + (lr >> 5) * 64 + (which == 2 ? 32 : 0) + (lr & 31);
            rb = rb < N ? rb : N - 1;
            return (unsigned)(((long long)rb * ldb + c * 8) * 2);
        };
        vb0_0 = offs(0, 0); vb0_1 = offs(1, 0);
        va_0 = offs(0, 1);  va_1 = offs(1, 1);
        vb1_0 = offs(0, 2); vb1_1 = offs(1, 2);
    };
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(p_lds_addr(smem) + wave * 2048);
    int s_T = 0, s_kt = 0, s_pass = 0;  // stream cursor: global K-tile, K-tile in pass, pass
    int opc = 0;                        // VMEM operations issued by this wave so far (queue model)
    auto issue = [&](auto Jc) __attribute__((always_inline)) {  // item J (0 B_h0, 1 A, 2 B_h1) of stream K-tile s_T
        constexpr int J = decltype(Jc)::value;
        if (s_T < total_kt) {
            const unsigned dst = lds0 + (s_T % R_NBUF) * R_BUF + J * P_HALF;
            const bf16_t* base = (J == 1 ? A : B) + (size_t)s_kt * P_BK;
            const unsigned x0 = J == 0 ? vb0_0 : J == 1 ? va_0 : vb1_0;
            const unsigned x1 = J == 0 ? vb0_1 : J == 1 ? va_1 : vb1_1;
            p_glds2(base, x0, x1, dst, dst + 1024);
            opc += 2;
        }
        if (J == 2) {
            ++s_T;
            if (++s_kt == nk) {
                s_kt = 0;
                if (++s_pass < npass) point_at(s_pass);
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    // ------------------------------------------------------------------ fragment reads
    const int l15 = lane & 15, lg = lane >> 4;
    const int sw = p_swz(l15);
    const int offA = (wr * 64 + l15) * 128, offB = (wc * 32 + l15) * 128;
    const int ch0 = ((0 * 4 + lg) ^ sw) << 4, ch1 = ((1 * 4 + lg) ^ sw) << 4;
    auto ldA = [&](int buf, int mi, int ks) __attribute__((always_inline)) -> bf16x8 {
        return *(const bf16x8*)(smem + buf * R_BUF + P_HALF + offA + mi * 2048 + (ks ? ch1 : ch0));
    };
    auto ldB = [&](int buf, int h, int ni, int ks) __attribute__((always_inline)) -> bf16x8 {
        return *(const bf16x8*)(smem + buf * R_BUF + (h ? 2 : 0) * P_HALF + offB + ni * 2048 +
                                (ks ? ch1 : ch0));
    };

    f32x4 acc[2][4][4];
    bf16x8 a[4][2], b[2][2];
    f32x4 bias_next[4];  // bias of the NEXT pass's columns (asm-loaded one pass ahead)
#pragma unroll
    for (int j = 0; j < 4; ++j) bias_next[j] = f32x4{0, 0, 0, 0};
    const int g4 = lane >> 4, r16 = lane & 15;
    char* stg = smem + R_STG + wave * 2048;

    // bias of pass `pass` -> bias_next (4 asm loads, counted in opc)
    auto fetch_bias = [&](int pass) __attribute__((always_inline)) {
        if constexpr (HAS_BIAS) {
            const int tile = pos + (pass >> 1) * G;
            const int n0 = (tile % tiles_n) * P_BN + wc * 64 + 4 * g4;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                int n = n0 + 16 * nt;
                n = n + 4 <= N ? n : N - 4;  // clamped columns are never stored
                bias_next[nt] = r_asm_load16(bias + n);
            }
            opc += 4;
        }
    };

    // ------------------------------------------------------------------ drain machinery
    int d_m0 = 0, d_n0 = 0;  // rows / columns of the sub-tile being drained (wave origin)
    bool d_full = false;
    // one step = one 16x16 block (mt = S >> 2, nt = S & 3) of accumulator set DS
    u32x2 hold[DUAL ? 4 : 1];  // GELU_GRAD: packed gelu(u) blocks wait here until the row block is flushed
    auto put = [&](int nt, u32x2 pk) __attribute__((always_inline)) {
        const int c = nt * 2 + (g4 >> 1);
        *(u32x2*)(stg + r16 * 128 + ((c ^ (r16 & 7)) << 4) + ((g4 & 1) << 3)) = pk;
    };
    auto flush = [&](bf16_t* __restrict__ dst, int mt) __attribute__((always_inline)) {
        const int R0 = lane >> 3, Cc = lane & 7;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int R = R0 + 8 * hh;
            const u32x4 w = *(const u32x4*)(stg + R * 128 + ((Cc ^ (R & 7)) << 4));
            const int gm = d_m0 + 16 * mt + R, gn = d_n0 + 8 * Cc;
            if (dbg & 2) {
                asm volatile("" ::"v"(w));
            } else if (d_full) {
                *(u32x4*)(dst + (size_t)gm * ldc + gn) = w;
            } else if (gm < M && gn < N) {
                *(u32x4*)(dst + (size_t)gm * ldc + gn) = w;
            }
        }
        if (d_full && !(dbg & 2)) opc += 2;  // unconditional stores only: an over-count would under-wait
    };
    auto drain_block = [&](auto DSc, auto Sc) __attribute__((always_inline)) {
        constexpr int DS = decltype(DSc)::value, S = decltype(Sc)::value;
        constexpr int mt = S >> 2, nt = S & 3;
        const f32x4 v = acc[DS][mt][nt];
        u32x2 pk;
        if constexpr (DUAL) {
            f32x4 gq, dq;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float cdf, xpdf;
                gelu_parts_fast(v[q], cdf, xpdf);
                gq[q] = v[q] * cdf;
                dq[q] = cdf + xpdf;
            }
            pk[0] = pack_bf2(dq[0], dq[1]);
            pk[1] = pack_bf2(dq[2], dq[3]);
            hold[nt][0] = pack_bf2(gq[0], gq[1]);
            hold[nt][1] = pack_bf2(gq[2], gq[3]);
        } else {
            pk[0] = pack_bf2(v[0], v[1]);
            pk[1] = pack_bf2(v[2], v[3]);
        }
        put(nt, pk);
        if constexpr (nt == 3) {
            flush(C, mt);
            if constexpr (DUAL) {
#pragma unroll
                for (int j = 0; j < 4; ++j) put(j, hold[j]);
                flush(out2, mt);
            }
        }
    };
    auto drain_step = [&](auto DSc, int step) __attribute__((always_inline)) {
        using D = decltype(DSc);
        switch (step) {
#define R_CASE(S_) case S_: drain_block(D{}, std::integral_constant<int, S_>{}); break;
            R_CASE(0) R_CASE(1) R_CASE(2) R_CASE(3) R_CASE(4) R_CASE(5) R_CASE(6) R_CASE(7)
            R_CASE(8) R_CASE(9) R_CASE(10) R_CASE(11) R_CASE(12) R_CASE(13) R_CASE(14) R_CASE(15)
#undef R_CASE
            default: break;
        }
    };

    auto mma = [&](auto SETc, auto QNc) __attribute__((always_inline)) {
        constexpr int SET = decltype(SETc)::value, QN = decltype(QNc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[SET][mi][QN * 2 + ni] =
                        P_MFMA(b[ni][ks], a[mi][ks], acc[SET][mi][QN * 2 + ni]);
        __builtin_amdgcn_s_setprio(0);
    };

    // ------------------------------------------------------------------ prologue
    // Refill rule (reads of phase g complete when MMA(g) opens, one barrier later for the wr = 1
    // row): a slot is refilled two phases after its last read.  P0(T) issues B_h0(T+2), A(T+2)
    // (slots read in P0(T-1)); P1(T) issues B_h1(T+2) (slot read in P1(T-1)).  Waits (one phase
    // before the first read): P0(T) for B_h1(T), P1(T) for A(T+1).
    point_at(0);
    fetch_bias(0);
    issue(I0{}); issue(I1{});
    const int m_a0 = opc;         // after A(0)
    issue(I2{});
    int mark_b = opc;             // after B_h1(0): waited in P0(0)
    issue(I0{}); issue(I1{});
    int mark_a_cur = opc;         // after A(1): waited in P1(0)
    issue(I2{});
    int mark_b_cur2 = opc;        // after B_h1(1): becomes mark_b for P0(1)
    int mark_a = 0, mark_b_next = 0;
    r_wait_vm(opc - m_a0);        // B_h0(0), A(0) (and the bias of pass 0) have landed
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave row by one barrier

    int T = 0;  // global K-tile counter
    bool have_drain = false;
    // one pass: SET = pass & 1 accumulates, 1 - SET drains
    auto run_pass = [&](auto SETc, int pass) __attribute__((always_inline)) {
        constexpr int SET = decltype(SETc)::value;
        using DS = std::integral_constant<int, 1 - SET>;
        // accumulators start from the bias (alpha == 1): the epilogue never adds it
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[SET][mi][nt] = bias_next[nt];
        __builtin_amdgcn_sched_barrier(0);
        if (pass + 1 < npass) fetch_bias(pass + 1);  // lands long before the next pass starts
        for (int kt = 0; kt < nk; ++kt, ++T) {
            const int cb = T % R_NBUF;
            // ---------------- P0
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) b[ni][ks] = ldB(cb, 0, ni, ks);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) a[mi][ks] = ldA(cb, mi, ks);
            __builtin_amdgcn_sched_barrier(0);
            issue(I0{});  // B_h0(T+2)
            issue(I1{});  // A(T+2)
            mark_a = opc;
            r_wait_vm(opc - mark_b);  // everything up to B_h1(T) has landed (read in P1)
            if (have_drain) drain_step(DS{}, 2 * kt);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            mma(SETc, I0{});
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- P1
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) b[ni][ks] = ldB(cb, 1, ni, ks);
            __builtin_amdgcn_sched_barrier(0);
            issue(I2{});  // B_h1(T+2)
            mark_b_next = opc;
            r_wait_vm(opc - mark_a_cur);  // everything up to A(T+1) has landed (read in the next P0)
            mark_a_cur = mark_a;
            mark_b = mark_b_cur2;
            mark_b_cur2 = mark_b_next;
            if (have_drain) drain_step(DS{}, 2 * kt + 1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            mma(SETc, I1{});
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        // drain steps that did not fit in this pass's 2 nk segments (K < 512)
        if (have_drain) {
            for (int s = 2 * nk; s < 16; ++s) drain_step(DS{}, s);
        }
        // this pass's sub-tile becomes the one to drain
        const int tile = pos + (pass >> 1) * G, h = pass & 1;
        const int m0 = (tile / tiles_n) * P_BM, n0 = (tile % tiles_n) * P_BN;
        d_m0 = m0 + wr * 128 + h * 64;
        d_n0 = n0 + wc * 64;
        d_full = (m0 + P_BM <= M) && (n0 + P_BN <= N);
        have_drain = !(dbg & 1);
    };
    for (int pass = 0; pass < npass; pass += 2) {
        run_pass(I0{}, pass);
        run_pass(I1{}, pass + 1);
    }
    // the last sub-tile (set 1) has nobody to hide behind
    if (have_drain)
        for (int s = 0; s < 16; ++s) drain_step(I1{}, s);
    if (wr == 0) __builtin_amdgcn_s_barrier();  // every wave executes the same number of barriers
}

// =====================================================================================
// host side
// =====================================================================================
static int nt256p_mode() {  // SSL4GIE_NT256P: "0" never, otherwise whenever the shape qualifies
    static int v = -2;
    if (v == -2) {
        const char* s = getenv("SSL4GIE_NT256P");
        v = (s && s[0] == '0') ? 0 : 1;
    }
    return v;
}

bool ssl4gie_internal_nt256p_ok(const ssl4gie_gemm_desc* d) {
    if (!nt256p_mode()) return false;
    if (d->dtype_c != SSL4GIE_BF16 || d->alpha != 1.0f || d->accumulate) return false;
    const int ep = d->epilogue;
    if (ep != SSL4GIE_EPI_NONE && ep != SSL4GIE_EPI_BIAS && ep != SSL4GIE_EPI_BIAS_GELU_GRAD)
        return false;
    if ((ep == SSL4GIE_EPI_BIAS || ep == SSL4GIE_EPI_BIAS_GELU_GRAD) && !d->bias) return false;
    if (d->N < 4 || d->K < P_BK) return false;
    return true;
}

int ssl4gie_internal_nt256p_launch(const ssl4gie_gemm_desc* d, hipStream_t st) {
    const int tm = (d->M + P_BM - 1) / P_BM, tn = (d->N + P_BN - 1) / P_BN;
    const int ntiles = tm * tn;
    dim3 grid(ntiles < 256 ? ntiles : 256), block(512);
    static int dbg = -1;  // SSL4GIE_NT256P_DBG: ablation bits (1 no drain, 2 no stores); wrong outputs
    if (dbg < 0) { const char* s = getenv("SSL4GIE_NT256P_DBG"); dbg = s ? atoi(s) : 0; }
    ProfScope prof(PROF_GEMM_NT, 2.0 * d->M * d->N * d->K, st);
#define R_LAUNCH(MODE_)                                                                            \
    do {                                                                                           \
        auto kfn = gemm_bf16_nt256p_kernel<MODE_>;                                                 \
        static bool attr_set = false; /* idempotent; a benign race only repeats the call */        \
        if (!attr_set) {                                                                           \
            HIP_RET(hipFuncSetAttribute((const void*)kfn,                                          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS_BYTES)); \
            attr_set = true;                                                                       \
        }                                                                                          \
        hipLaunchKernelGGL(kfn, grid, block, R_LDS_BYTES, st, (const bf16_t*)d->A, d->sAm,         \
                           (const bf16_t*)d->B, d->sBn, (bf16_t*)d->C, d->ldc, d->M, d->N, d->K,   \
                           tn, ntiles, d->bias, (bf16_t*)d->out2, dbg);                                 \
    } while (0)
    switch (d->epilogue) {
        case SSL4GIE_EPI_BIAS: R_LAUNCH(SSL4GIE_EPI_BIAS); break;
        case SSL4GIE_EPI_BIAS_GELU_GRAD: R_LAUNCH(SSL4GIE_EPI_BIAS_GELU_GRAD); break;
        case SSL4GIE_EPI_NONE: R_LAUNCH(SSL4GIE_EPI_NONE); break;
        default: return ARG_ERR;
    }
#undef R_LAUNCH
    LAUNCH_CHECK();
    return 0;
}
