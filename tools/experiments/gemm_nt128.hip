// SHELVED EXPERIMENT (round 2) — not built, not product.  Kept with its measurements
// (profiles/r02a_nt128_*.log, r02a_nt256_*.log) because the negative result shaped the design:
//   * K-loop alone 0.85-0.92 PFLOP/s vs 1.10 for gemm_nt256 (64-B LDS rows, no hand ping-pong);
//   * with epilogues 0.61-0.65 vs 0.73 PFLOP/s: a co-resident workgroup does NOT hide the other's
//     epilogue — K-loop-only + epilogue-only times add up (r02a_nt128_kloop_vs_epilogue.log), a
//     per-CU start skew (g_nt128_arrivals) and paced stores change nothing but the added delay
//     (r02a_nt128_skew_pace.log), streaming (nt) stores change nothing, and chip-level start jitter
//     in gemm_nt256 changes nothing either (r02a_nt256_start_jitter.log);
//   * ablations on gemm_nt256 (r02a_nt256_epilogue_ablation.log): the GELU pair's arithmetic is
//     ~45 % of fc1's epilogue, the residual loads ~40 % of the residual epilogue (load / store
//     coupling through the in-order vmcnt); pure-store epilogues cost ~3 us per 256 x 256 tile.
// It compiled against a gemm256.h whose epilogues took a `pace` argument (removed again).
//
// 128x256x32 "two workgroups per CU" bf16 NT GEMM for gfx950:  C[M,N] = epilogue(A[M,K] * B[N,K]^T).
//
// Why a third NT kernel.  The 256x256 ping-pong kernel (gemm_nt256.hip) owns a CU outright (8 waves,
// all 160 KiB of LDS), so while its waves run an epilogue — bias / GELU pair / fp32 residual
// read-modify-write / aux multiply, 256 KiB to 512 KiB of global traffic per tile — the CU's matrix
// pipes idle, and because every CU reaches its epilogue at the same time the chip alternates between
// an MFMA-only phase and an HBM-only phase (r01: K-loop alone 1.06-1.31 PFLOP/s, with epilogues
// 0.45-1.04).  Here a workgroup is HALF a CU's budget: 4 waves (one per SIMD), tile 128 x 256, wave
// tile 128 x 64 (the same 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16, so gemm256.h's epilogues
// are reused unchanged), 72 KiB of LDS.  Two workgroups share a CU and drift apart by themselves:
// while one writes its tile out (VALU + LDS + HBM), the other's K-loop has the matrix pipes to
// itself, and when both are in their K-loops the two waves of a SIMD interleave MFMAs with each
// other's LDS reads — the overlap the ping-pong kernel builds with barriers inside one workgroup,
// obtained from the hardware scheduler across two.  Left alone the two would run in lockstep (same
// start, same work) and meet in their epilogues, so the second arrival on a CU starts half a tile
// period late (g_nt128_arrivals) and the workgroups are persistent, which keeps the offset.
//
// K-loop: stages of BK = 32 (one MFMA K-step): A 128 rows x 64 B + B 256 rows x 64 B = 24 KiB, ring
// of 3.  Per stage and wave: 6 LDS-DMA pieces (global_load_lds_dwordx4, 1 KiB each, issued from
// inline asm so hipcc never drains them), one counted s_waitcnt vmcnt(6) (the newest stage stays in
// flight), ONE s_barrier, 12 ds_read_b128 (B 4, A 8), 32 MFMAs.
//   RAW: a wave waits for its own pieces of stage s, then the barrier: every piece of stage s has
//        landed before any wave reads it.
//   WAR: stage s+2 is issued after that barrier into the slot stage s-1 was read from; every wave
//        finished those reads (they feed its MFMAs of iteration s-1) before it arrived.
// LDS rows are 64 B = 4 chunks of 16 B; chunk c of row r sits at position c ^ ((r >> 1) & 3):
// ds_read_b128 of a 16-row fragment is conflict-free (tools/lds_bank_sim.py).  The DMA writes
// linearly (lane l -> byte 16 l of its KiB), so the permutation is applied on the global side.
//
// Replaces the cuBLAS calls behind nn.Linear in timm Block / MAE decoder (reference
// call sites Models/mae/models_mae.py:39-41,47,53-55,59; Models/models.py:171-173).
#include "gemm256.h"
#include "prof.h"

#include <stdio.h>
#include <stdlib.h>

#define Q_BM 128
#define Q_BN 256
#define Q_BK 32
#define Q_STAGE 24576             // 8 KiB of A + 16 KiB of B
#define Q_LDS_BYTES (3 * Q_STAGE)  // 73728: two workgroups per CU

DEVI int q_swz(int r) { return (r >> 1) & 3; }

// Arrival counters, one per CU (index: XCC_ID | SE_ID | SH_ID | CU_ID), never reset: the workgroup
// that finds an odd count is the SECOND of the two sharing its CU in this launch and starts half a
// tile period late, so that from then on one workgroup's epilogue meets the other's K-loop.  A
// scheduling hint only — results do not depend on it (a CU that got one or three workgroups in some
// launch merely flips the roles).
__device__ unsigned g_nt128_arrivals[4096];

template <typename TC, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt128_kernel(
    const bf16_t* __restrict__ A, long long lda, const bf16_t* __restrict__ B, long long ldb,
    TC* __restrict__ C, long long ldc, int M, int N, int K, int tiles_n, int ntiles, EpiArgs e,
    int dbg_skip_epilogue, int skew_sleeps, int pace) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int G = gridDim.x;
    const int gpos = xcd_remap(blockIdx.x, G);
    // dbg_skip_epilogue: 1 = K-loop only, 2 = epilogue only (ablation knobs; outputs are garbage)
    const int nk = dbg_skip_epilogue == 2 ? 0 : K / Q_BK;
    if (skew_sleeps > 0) {  // second workgroup of this CU: start late (see g_nt128_arrivals)
        if (t == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
            const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
            const unsigned idx = ((xcc & 15u) << 8) | ((hw >> 8) & 255u);
            *(volatile unsigned*)smem = atomicAdd(&g_nt128_arrivals[idx], 1u);
        }
        __builtin_amdgcn_s_barrier();
        const unsigned order = *(volatile unsigned*)smem;
        __builtin_amdgcn_s_barrier();
        if (order & 1u)
            for (int i = 0; i < skew_sleeps; ++i) __builtin_amdgcn_s_sleep(16);  // ~1024 clocks each
    }
    for (int tile = gpos; tile < ntiles; tile += G) {
    const int m0 = (tile / tiles_n) * Q_BM, n0 = (tile % tiles_n) * Q_BN;
    const float e_alpha = e.alpha;
    const float* e_bias = e.bias;
    const float* e_residual = e.residual;
    const long long e_ldr = e.ldr;
    const bf16_t* e_aux = (const bf16_t*)e.aux;
    bf16_t* e_out2 = (bf16_t*)e.out2;
    const int e_accumulate = e.accumulate;

    // ------------------------------------------------------------------ LDS-DMA sources
    // piece j of a stage covers 16 tile rows; lane l -> row 16 j + (l >> 2), LDS position l & 3,
    // i.e. global chunk (l & 3) ^ swz(row).  This wave issues A pieces 2w, 2w+1 and B pieces 4w..4w+3.
    auto offs = [&](int j, bool is_a) -> unsigned {
        const int lr = 16 * j + (lane >> 2);
        const int c = (lane & 3) ^ q_swz(lr);
        if (is_a) {
            int ra = m0 + lr;
            ra = ra < M ? ra : M - 1;
            return (unsigned)(((long long)ra * lda + c * 8) * 2);
        }
        int rb = n0 + lr;
        rb = rb < N ? rb : N - 1;
        return (unsigned)(((long long)rb * ldb + c * 8) * 2);
    };
    const unsigned va0 = offs(2 * wave, true), va1 = offs(2 * wave + 1, true);
    const unsigned vb0 = offs(4 * wave, false), vb1 = offs(4 * wave + 1, false);
    const unsigned vb2 = offs(4 * wave + 2, false), vb3 = offs(4 * wave + 3, false);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(p_lds_addr(smem));
    const unsigned la = lds0 + wave * 2048, lb = lds0 + 8192 + wave * 4096;
    auto issue = [&](int s) {  // all 6 pieces of stage s (s < nk)
        const unsigned slot = (unsigned)(s % 3) * Q_STAGE;
        const bf16_t* ka = A + (size_t)s * Q_BK;
        const bf16_t* kb = B + (size_t)s * Q_BK;
        p_glds2(kb, vb0, vb1, lb + slot, lb + slot + 1024);
        p_glds2(ka, va0, va1, la + slot, la + slot + 1024);
        p_glds2(kb, vb2, vb3, lb + slot + 2048, lb + slot + 3072);
    };

    // ------------------------------------------------------------------ fragment reads
    const int l15 = lane & 15, lg = lane >> 4;
    const int pos = (lg ^ q_swz(l15)) << 4;  // tile-local rows are l15 + multiples of 16
    const int offA = l15 * 64 + pos, offB = 8192 + (wave * 64 + l15) * 64 + pos;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    if (nk > 0) issue(0);
    if (nk > 1) issue(1);
    for (int s = 0; s < nk; ++s) {
        if (s + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + 2 < nk) issue(s + 2);
        const char* st = smem + (s % 3) * Q_STAGE;
        bf16x8 b[4], a[8];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) b[ni] = *(const bf16x8*)(st + offB + ni * 1024);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) a[mi] = *(const bf16x8*)(st + offA + mi * 1024);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = P_MFMA(b[ni], a[mi], acc[mi][ni]);
        __builtin_amdgcn_s_setprio(0);
    }
    // the ring is dead: its first 16 KiB become the four wave-private staging areas
    __builtin_amdgcn_s_barrier();
    char* stg = smem + wave * P_STG_WAVE;
    if (dbg_skip_epilogue == 1) {
        asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[7][3][3]));
    } else if (m0 + Q_BM <= M && n0 + Q_BN <= N) {
        p_epilogue<TC, MODE, true, false>(acc, stg, e_alpha, e_bias, e_residual, e_ldr, e_aux, e_out2,
                                          e_accumulate, C, ldc, m0, n0 + wave * 64, M, N, lane, nullptr, pace);
    } else {
        p_epilogue<TC, MODE, false, false>(acc, stg, e_alpha, e_bias, e_residual, e_ldr, e_aux, e_out2,
                                           e_accumulate, C, ldc, m0, n0 + wave * 64, M, N, lane, nullptr, pace);
    }
    __builtin_amdgcn_s_barrier();  // staging reads done before the next tile's DMA lands on them
    }
}

// =====================================================================================
// host side
// =====================================================================================
static int nt128_mode() {  // SSL4GIE_NT128: "0" never, "1" whenever possible, unset = heuristic
    static int v = -2;
    if (v == -2) {
        const char* s = getenv("SSL4GIE_NT128");
        v = !s ? -1 : (s[0] == '0' ? 0 : 1);
    }
    return v;
}

bool ssl4gie_internal_nt128_ok(const ssl4gie_gemm_desc* d) {
    const int mode = nt128_mode();
    if (mode == 0 || d->conv || d->colstats) return false;
    if ((long long)d->M * d->sAm * 2 >= (1LL << 32) || (long long)d->N * d->sBn * 2 >= (1LL << 32))
        return false;
    if (d->K % Q_BK != 0 || d->K < Q_BK || d->N % 8 != 0 || d->ldc % 8 != 0) return false;
    const int ep = d->epilogue;
    if (d->dtype_c == SSL4GIE_BF16) {
        if (ep == SSL4GIE_EPI_BIAS_RESIDUAL || d->accumulate) return false;
        if (ep == SSL4GIE_EPI_RELU_MASK_AUX) return false;  // implicit convolutions only
    } else {
        if (ep != SSL4GIE_EPI_BIAS && ep != SSL4GIE_EPI_BIAS_RESIDUAL && ep != SSL4GIE_EPI_NONE) return false;
    }
    if (ep == SSL4GIE_EPI_ADD_AUX && !d->aux) return false;
    if (mode == 1) return true;
    // heuristic: enough tiles to give most CUs two workgroups
    const long long tiles = (long long)((d->M + Q_BM - 1) / Q_BM) * ((d->N + Q_BN - 1) / Q_BN);
    return tiles >= 256;
}

int ssl4gie_internal_nt128_launch(const ssl4gie_gemm_desc* d, hipStream_t st) {
    const int tm = (d->M + Q_BM - 1) / Q_BM, tn = (d->N + Q_BN - 1) / Q_BN;
    const int ntiles = tm * tn;
    // persistent: two workgroups per CU (they do not hold a CU's whole LDS, so RCCL's kernels can
    // still be placed: no CU reservation here)
    dim3 grid(ntiles < 512 ? ntiles : 512), block(256);
    // start skew of a CU's second workgroup, in units of ~1024 clocks: half a tile period
    // ~ nk stages x ~1 unit (two workgroups share the matrix pipes) / 2 + half an epilogue
    static int skew_a = -1, skew_b = -1;
    if (skew_a < 0) {
        const char* s = getenv("SSL4GIE_NT128_SKEW");  // "a,b": sleeps = (a * nk) / 16 + b
        int a = 8, b = 6;
        if (s) sscanf(s, "%d,%d", &a, &b);
        skew_a = a; skew_b = b;
    }
    static int pace = -1;
    if (pace < 0) {
        const char* s = getenv("SSL4GIE_NT128_PACE");
        pace = s ? atoi(s) : 0;
        const char* n = getenv("SSL4GIE_NT_STORES");  // "nt": streaming stores in the epilogues
        if (n && n[0] == 'n') pace |= P_NTS_FLAG;
    }
    const int nkk = d->K / Q_BK;
    const int skew = ntiles > 256 ? (skew_a * nkk) / 16 + skew_b : 0;
    EpiArgs e{d->alpha, d->epilogue, d->bias, d->residual, d->ldr, d->aux, d->out2, d->accumulate,
              nullptr};
    static int skip_epi = -1;  // SSL4GIE_NT256_NOEPI=1: ablation (K-loop only; outputs are garbage)
    if (skip_epi < 0) {
        const char* s = getenv("SSL4GIE_NT256_NOEPI");
        skip_epi = (s && s[0] == '1') ? 1 : ((s && s[0] == '2') ? 2 : 0);
    }
    ProfScope prof(PROF_GEMM_NT, 2.0 * d->M * d->N * d->K, st);
#define Q_LAUNCH(TC_, MODE_)                                                                       \
    do {                                                                                           \
        auto kfn = gemm_bf16_nt128_kernel<TC_, MODE_>;                                             \
        static bool attr_set = false; /* idempotent; a benign race only repeats the call */        \
        if (!attr_set) {                                                                           \
            HIP_RET(hipFuncSetAttribute((const void*)kfn,                                          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, Q_LDS_BYTES)); \
            attr_set = true;                                                                       \
        }                                                                                          \
        hipLaunchKernelGGL(kfn, grid, block, Q_LDS_BYTES, st, (const bf16_t*)d->A, d->sAm,         \
                           (const bf16_t*)d->B, d->sBn, (TC_*)d->C, d->ldc, d->M, d->N, d->K, tn,  \
                           ntiles, e, skip_epi, skew, pace);                                          \
    } while (0)
    if (d->dtype_c == SSL4GIE_BF16) {
        switch (d->epilogue) {
            case SSL4GIE_EPI_BIAS: Q_LAUNCH(bf16_t, SSL4GIE_EPI_BIAS); break;
            case SSL4GIE_EPI_BIAS_GELU: Q_LAUNCH(bf16_t, SSL4GIE_EPI_BIAS_GELU); break;
            case SSL4GIE_EPI_DGELU: Q_LAUNCH(bf16_t, SSL4GIE_EPI_DGELU); break;
            case SSL4GIE_EPI_BIAS_GELU_GRAD: Q_LAUNCH(bf16_t, SSL4GIE_EPI_BIAS_GELU_GRAD); break;
            case SSL4GIE_EPI_MUL_AUX: Q_LAUNCH(bf16_t, SSL4GIE_EPI_MUL_AUX); break;
            case SSL4GIE_EPI_ADD_AUX: Q_LAUNCH(bf16_t, SSL4GIE_EPI_ADD_AUX); break;
            case SSL4GIE_EPI_NONE: Q_LAUNCH(bf16_t, SSL4GIE_EPI_NONE); break;
            default: return ARG_ERR;
        }
    } else {
        switch (d->epilogue) {
            case SSL4GIE_EPI_BIAS: Q_LAUNCH(float, SSL4GIE_EPI_BIAS); break;
            case SSL4GIE_EPI_BIAS_RESIDUAL: Q_LAUNCH(float, SSL4GIE_EPI_BIAS_RESIDUAL); break;
            case SSL4GIE_EPI_NONE: Q_LAUNCH(float, SSL4GIE_EPI_NONE); break;
            default: return ARG_ERR;
        }
    }
#undef Q_LAUNCH
    LAUNCH_CHECK();
    return 0;
}
