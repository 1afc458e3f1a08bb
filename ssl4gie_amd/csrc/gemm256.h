// Shared pieces of the 256x256x64 ping-pong GEMM kernels (gemm_nt256.hip, gemm_tn256.hip): tile
// constants, the inline-asm LDS-DMA issue, and the fp32 epilogue that transposes accumulators
// through the wave-private LDS staging area so that global accesses are whole 256-byte rows.
#pragma once
#include "gemm_internal.h"

#define P_BM 256
#define P_BN 256
#define P_BK 64
#define P_HALF 16384
#define P_BUF 65536
#define P_STG_WAVE 4096
#define P_LDS_BYTES (2 * P_BUF + 8 * P_STG_WAVE)  // 163840 = all of the CU's LDS

DEVI int p_swz(int r) { return (r >> 1) & 7; }

DEVI unsigned p_lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// two LDS-DMA pieces (1 KiB each) of one half-tile: wave-uniform 64-bit base in SGPRs, per-lane
// 32-bit byte offsets, wave-uniform LDS destinations l0 and l0 + 1024.  M0 is saved / restored
// inside the statement (cdna_hip_programming.md §5.7); nothing here is visible to hipcc's waitcnt
// bookkeeping — the caller counts vmcnt by hand.
DEVI void p_glds2(const void* sbase, unsigned v0, unsigned v1, unsigned l0, unsigned l1) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %1\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(sbase), "v"(v0), "v"(v1), "s"(l0), "s"(l1)
        : "memory");
}

// a single 1-KiB piece (the 64-row B half-tile of the 256 x 192 variant has one piece per wave)
DEVI void p_glds1(const void* sbase, unsigned v0, unsigned l0) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(sbase), "v"(v0), "s"(l0)
        : "memory");
}

// one LDS-DMA of 4 B per lane (256 B per wave): the bias row of a wave's 64 output columns
DEVI void p_glds1_dword(const void* sbase, unsigned v0, unsigned l0) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %2, %1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(sbase), "v"(v0), "s"(l0)
        : "memory");
}

// the same with per-lane 64-bit addresses (gathered operands: a lane may point at the zero page)
DEVI void p_glds2v(const void* a0, const void* a1, unsigned l0, unsigned l1) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(a0), "v"(a1), "s"(l0), "s"(l1)
        : "memory");
}

// n / d for n < 2^31 with (mg, sh) from conv_magic
DEVI unsigned p_fastdiv(unsigned n, unsigned mg, unsigned sh) { return __umulhi(n, mg) >> sh; }

// ReLU on packed bf16: a negative bf16 is a negative int16
DEVI bf16x8 p_relu8(bf16x8 v) {
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8_t, v), z));
}

#ifndef P_EPI_XPOSE_SWAP
#define P_EPI_XPOSE_SWAP true
#endif
#ifndef P_EPI_PRIO_MODE
#define P_EPI_PRIO_MODE 0  /* measured: no policy changes the tile time (profiles/r04j) */
#endif
#define P_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// fp32 outputs of one wave's 128 x 64 sub-tile.  acc[mt][nt]: rows rbase + 16 mt + (l & 15),
// columns cbase + 16 nt + 4 (l >> 4) + 0..3.  Each 16 x 64 block goes through `stg` (4 KiB,
// wave-private: LDS executes one wave's instructions in order, so no barrier is needed) and comes
// back with a lane holding 16 B of a row and 16 lanes covering 256 contiguous bytes; the residual /
// accumulate operand is read in that same layout.  Staging image: 16-B chunk c of row r at position
// c ^ r (ds_write_b128 and ds_read_b128 conflict-free, tools/lds_bank_sim.py).
// NJ: 16-column accumulator blocks per wave (4: 128 x 64 wave tile; 3: 128 x 48, the 256 x 192 variant — the
// staging image keeps its 64-column geometry, lanes that would own columns 48 .. 63 stay idle).
// Output stores of the epilogues can be NON-TEMPORAL (`nt`, chosen per launch: ssl4gie_internal_nt256_launch).  A round of
// tiles writes 7.8 MB per XCD through a 4 MiB L2 and evicts the weight panels every workgroup re-reads in the next round
// (dec.fc1: 157 MB read per launch against 36 MB algorithmic, profiles/r05_pmc_summary.txt); streamed past the L2 the
// outputs leave them in place: dec.fc1 169 -> 152 us, dec.dfc2 144 -> 128, dec.proj 58 -> 53 (profiles/r05nt).  The
// short encoder products (1-3 rounds of tiles, outputs the next kernel finds in cache) lose 1-4 % and keep plain stores;
// non-temporal LOADS of the residual / aux tiles measured worse and are not used.
// (A compile-time choice: behind a runtime flag the compiler merges the two stores and drops the hint.)
template <bool NT, class V> DEVI void est(V* p, V v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <bool NT> DEVI void est4(float* p, f32x4 v) { est<NT>((f32x4*)p, v); }
template <bool NT> DEVI void est4(bf16_t* p, f32x4 v) {
    u32x2 r;
    r[0] = pack_bf2(v[0], v[1]);
    r[1] = pack_bf2(v[2], v[3]);
    est<NT>((u32x2*)p, r);
}

template <bool FULL, bool RESID, int NJ = 4, bool NT = false>
DEVI void p_store_f32(f32x4 (&acc)[8][4], char* stg, const float alpha, const f32x4 bias_t /* columns gn .. gn + 3 */,
                      const float* __restrict__ residual, const long long ldr, const bool accumulate,
                      float* __restrict__ C, const long long ldc, int rbase, int cbase, int M, int N,
                      int lane) {
    const int r16 = lane & 15, g4 = lane >> 4;
    const int R0 = lane >> 4, Cc = lane & 15;
    const int gn = cbase + 4 * Cc;
    const bool mine = NJ == 4 || Cc < 4 * NJ;  // this lane's 4 columns exist in the wave tile
    // the residual rows are fetched PF blocks ahead (a ring of PF x 4 row segments in the registers
    // that held the operand fragments): one exposed HBM round trip per tile instead of one per block
    constexpr int PF = NJ == 3 ? 3 : 2;  // row blocks ahead (+ the 16 registers of the pipelined block); 3 spills at NJ = 4
    f32x4 rs[PF][4];
    auto fetch = [&](int mt, f32x4 (&dst)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gm = rbase + 16 * mt + R0 + 4 * q;
            dst[q] = (mine && (FULL || (gm < M && gn < N))) ? ld4(residual + (size_t)gm * ldr + gn)
                                                            : f32x4{0, 0, 0, 0};
        }
    };
    if constexpr (RESID) {
#pragma unroll
        for (int mt = 0; mt < PF; ++mt) fetch(mt, rs[mt]);
    }
    // Software pipeline over the 8 row blocks: the transposed reads of block mt are issued, then block mt + 1 is
    // scaled and staged (VALU + ds_writes: LDS executes one wave's instructions in order, so those writes land
    // after the reads without any wait), and only then block mt's stores go out — the reads' latency hides
    // under the next block's arithmetic instead of stalling every block.
    auto put = [&](int mt) {
#pragma unroll
        for (int nt = 0; nt < NJ; ++nt) {
            const f32x4 v = acc[mt][nt] * alpha;
            const int c = nt * 4 + g4;
            *(f32x4*)(stg + r16 * 256 + ((c ^ r16) << 4)) = v;
        }
    };
    put(0);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        f32x4 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int R = R0 + 4 * q;
            w[q] = *(const f32x4*)(stg + R * 256 + ((Cc ^ R) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (mt + 1 < 8) put(mt + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int R = R0 + 4 * q;
            const int gm = rbase + 16 * mt + R;
            w[q] += bias_t;  // in the transposed layout a lane owns 4 columns: one bias quad instead of four
            if constexpr (RESID) w[q] += rs[mt % PF][q];
            if (mine && (FULL || (gm < M && gn < N))) {
                float* c = C + (size_t)gm * ldc + gn;
                if (accumulate) w[q] += ld4(c);
                est4<NT>(c, w[q]);
            }
        }
        if constexpr (RESID) {
            if (mt + PF < 8) fetch(mt + PF, rs[mt % PF]);
        }
    }
}

// bf16 outputs multiplied by a bf16 auxiliary tile (the saved GELU derivative): same transposition,
// fp32 through the staging area, the aux rows requested 6 row blocks ahead (48 VGPRs, the registers of the
// dead operand fragments) in the coalesced row layout.
// AUXF: 1 multiply by aux, 2 multiply by gelu'(aux), 3 zero where aux <= 0 (ReLU mask), 4 add aux,
// 5 per-column affine, optional aux, optional ReLU: act(acc scale[n] + shift[n] (+ aux)) — the BatchNorm that
// follows a 1x1 convolution (+ the bottleneck's residual add and ReLU) applied to the fp32 accumulators.
template <bool FULL, int AUXF, int NJ = 4, bool NT = false>
DEVI void p_store_bf16_aux(f32x4 (&acc)[8][4], char* stg, const float alpha,
                           const bf16_t* __restrict__ aux, bf16_t* __restrict__ C,
                           const long long ldc, int rbase, int cbase, int M, int N, int lane,
                           const float* __restrict__ scale = nullptr, const float* __restrict__ shift = nullptr,
                           const int relu = 0) {
    const int r16 = lane & 15, g4 = lane >> 4;
    const int R0 = lane >> 4, Cc = lane & 15;
    const int gn = cbase + 4 * Cc;
    const bool mine = NJ == 4 || Cc < 4 * NJ;
    [[maybe_unused]] f32x4 sc4 = {1, 1, 1, 1}, sh4 = {0, 0, 0, 0};
    [[maybe_unused]] const bool has_aux = AUXF != 5 || aux != nullptr;
    if constexpr (AUXF == 5) {
        if (mine && (FULL || gn < N)) { sc4 = ld4(scale + gn); sh4 = ld4(shift + gn); }
    }
    constexpr int PF = AUXF == 5 ? 2 : 4;  // ring of 4 of the 8 row blocks (32 VGPRs; 2 beside the affine quads) + the 16 of the pipelined block
    u32x2 ax[PF][4];
    auto fetch = [&](int mt, u32x2 (&dst)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gm = rbase + 16 * mt + R0 + 4 * q;
            dst[q] = (has_aux && mine && (FULL || (gm < M && gn < N))) ? *(const u32x2*)(aux + (size_t)gm * ldc + gn)
                                                                       : u32x2{0, 0};
        }
    };
#pragma unroll
    for (int mt = 0; mt < PF; ++mt) fetch(mt, ax[mt]);
    auto put = [&](int mt) {
#pragma unroll
        for (int nt = 0; nt < NJ; ++nt) {
            const f32x4 v = acc[mt][nt] * alpha;
            const int c = nt * 4 + g4;
            *(f32x4*)(stg + r16 * 256 + ((c ^ r16) << 4)) = v;
        }
    };
    put(0);  // pipelined as p_store_f32: reads of block mt, staging of block mt + 1, stores of block mt
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        f32x4 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int R = R0 + 4 * q;
            w[q] = *(const f32x4*)(stg + R * 256 + ((Cc ^ R) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (mt + 1 < 8) put(mt + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int R = R0 + 4 * q;
            const int gm = rbase + 16 * mt + R;
            const u32x2 r = ax[mt % PF][q];
            f32x4 u = {__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u),
                       __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u)};
            if constexpr (AUXF == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) u[j] = dgelu_fast(u[j]);
            }
            if constexpr (AUXF == 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) w[q][j] = u[j] > 0.f ? w[q][j] : 0.f;
            } else if constexpr (AUXF == 4) {
                w[q] += u;
            } else if constexpr (AUXF == 5) {
                w[q] = w[q] * sc4 + sh4 + u;  // u = 0 without aux
                if (relu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[q][j] = w[q][j] > 0.f ? w[q][j] : 0.f;
                }
            } else {
                w[q] *= u;
            }
            if (mine && (FULL || (gm < M && gn < N))) est4<NT>(C + (size_t)gm * ldc + gn, w[q]);
        }
        if (mt + PF < 8) fetch(mt + PF, ax[mt % PF]);
    }
}

// sum over the 8 lanes l, l ^ 8, l ^ 16, ... of a wave that share (l & 7), in that pairing order (the values equal
// three __shfl_xor rounds bit for bit): a DPP rotation inside the 16-lane rows and the two row / half swaps of
// gfx950 — VALU only, where 48 ds_bpermute round trips with their lgkmcnt waits cost ~1 us per output tile
DEVI float p_fold_lanes8(float v) {
    const unsigned b0 = __float_as_uint(v);
    v += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)b0, 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    const unsigned b1 = __float_as_uint(v);
    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(b1, b1, false, false);
    v = __uint_as_float(s1[0]) + __uint_as_float(s1[1]);
    const unsigned b2 = __float_as_uint(v);
    const u32x2 s2 = __builtin_amdgcn_permlane32_swap(b2, b2, false, false);
    return __uint_as_float(s2[0]) + __uint_as_float(s2[1]);
}

// ---- GELU pair by table (TAB epilogues of the 256 x 256 NT kernel) --------------------------------------------
// The reference applies GELU to the half-precision output of fc1 (autocast; timm Mlp, Models/mae/models_mae.py:39-41,
// 53-55), so Phi and gelu' are functions of the bf16-rounded pre-activation: 2 x 2048 distinct arguments with
// 2^-12 <= |x| < 16 (gelu_table.h, tools/gen_gelu_table.py), 1/2 below, 0 / 1 above — which the clamped index
// delivers.  The 16-KiB table sits in the epilogue staging area (unused by the lane-exchange form of the pair
// epilogue); one ds_read_b32 per element replaces exp2 + rcp + the 5-term polynomial of gelu_grad4_fast
// (24 VALU + 4 quarter-rate ops per pair -> 15): the LDS pipe is idle in this epilogue, the VALU was its bound.
// gelu = u * Phi(bf16(u)) with u the fp32 sum (v_fma_mix reads the fp16 Phi in place); gelu' comes out of the
// table already in bf16.  tests/test_gpu_gelu_table.py checks both against tools/gen_gelu_table.emulate bit for bit.
#define P_TAB_BYTES 16384
struct GeluTabQ { f32x4 u; unsigned e[4]; };
DEVI void p_gelu_tab_read(const f32x4 u, const char* tab, GeluTabQ& q) {
    typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
    q.u = u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const unsigned p = pack_bf2(u[2 * h], u[2 * h + 1]);
        const u16x2_t m = __builtin_bit_cast(u16x2_t, p & 0x7fff7fffu);
        u16x2_t t = __builtin_elementwise_sub_sat(m, (u16x2_t)(unsigned short)0x3980);   // v_pk_sub_u16 clamp
        t = __builtin_elementwise_min(t, (u16x2_t)(unsigned short)2047);                   // v_pk_min_u16
        const unsigned T = (((p >> 4) & 0x08000800u) | __builtin_bit_cast(unsigned, t)) << 2;  // sign -> bit 11; byte offsets
        q.e[2 * h] = *(const unsigned*)(tab + (T & 0xffffu));
        q.e[2 * h + 1] = *(const unsigned*)(tab + (T >> 16));
    }
}
DEVI float p_mix_lo16(float x, unsigned e) {  // x * fp16(e[15:0])
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(x), "v"(e));
    return r;
}
// packed results of the four elements: g[0..1] = gelu (columns 0-1, 2-3), d[0..1] = gelu'
DEVI void p_gelu_tab_finish(const GeluTabQ& q, unsigned (&d)[2], unsigned (&g)[2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        g[h] = pack_bf2(p_mix_lo16(q.u[2 * h], q.e[2 * h]), p_mix_lo16(q.u[2 * h + 1], q.e[2 * h + 1]));
        d[h] = __builtin_amdgcn_perm(q.e[2 * h + 1], q.e[2 * h], 0x07060302u);
    }
}

// ---- epilogue of one wave's 128 x 64 sub-tile ---------------------------------------------------
// acc[mt][nt]: rows rbase + 16 mt + (l & 15), columns cbase + 16 nt + 4 (l >> 4) + 0..3
// STATS (bf16 C, plain epilogue): per-column sums and sums of squares of the STORED (bf16-rounded)
// values over the wave's 128 rows -> colstats[(rbase / 128)][{0,1}][N]: the BatchNorm statistics of
// the layer that follows ride on the producing GEMM instead of costing a pass over the activation.
// BIAS_LDS: the kernel has already brought the wave's 64 bias values (columns cbase .. cbase + 63) into the first
// 256 B of `stg` by LDS-DMA (gemm_nt256.hip: issued in the tile's first K-tile, landed long before): the
// epilogue then starts without a single vector-memory load — four global loads here would queue behind the
// partner wave row's stores in the CU's in-order memory pipe (measured: +3.5 us per tile on the wr = 1 waves).
// (Round 4, measured and dropped: reading the whole residual tile FIRST, in the accumulator layout, and folding it
// into the accumulators before a plain fp32 store phase — dec.proj 57 -> 65 us: two serial phases lose more than
// the loads gain by not queueing behind stores; profiles/r04g_nt_epilogue_residual_first.log.)
template <typename TC, int MODE, bool FULL, bool STATS, bool BIAS_LDS = false, int NJ = 4, bool NT = false, bool TAB = false>
DEVI void p_epilogue(f32x4 (&acc)[8][4], char* stg, const float alpha,
                     const float* __restrict__ bias, const float* __restrict__ residual,
                     const long long ldr, const bf16_t* __restrict__ aux, bf16_t* __restrict__ out2,
                     const int accumulate, TC* __restrict__ C, long long ldc, int rbase, int cbase,
                     int M, int N, int lane, float* __restrict__ colstats,
                     unsigned long long* tstamp = nullptr /* debug library: per-row-block time stamps */,
                     const int prio_mode = P_EPI_PRIO_MODE /* debug library: s_setprio policy of the GELU epilogues */,
                     const bool xpose_swap = P_EPI_XPOSE_SWAP /* GELU pair: lane exchange instead of the LDS transposition */,
                     const char* tab = nullptr /* TAB: the (Phi, gelu') table in LDS (p_gelu_tab_read) */) {
    // EPI_AFFINE_AUX_RELU borrows two slots: `residual` carries scale[N], `accumulate` the ReLU flag (EpiArgs
    // is filled that way by the launcher)
    const int r16 = lane & 15, g4 = lane >> 4;
    constexpr bool HAS_BIAS = MODE == SSL4GIE_EPI_BIAS || MODE == SSL4GIE_EPI_BIAS_GELU ||
                              MODE == SSL4GIE_EPI_BIAS_RESIDUAL || MODE == SSL4GIE_EPI_BIAS_GELU_GRAD;
    // bf16 outputs add the bias in the accumulator layout (before packing): four quads per lane; fp32 outputs
    // add it after the transposition, where a lane owns 4 columns: one quad (12 registers less)
    f32x4 bias4[4], bias_t = {0, 0, 0, 0};
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bias4[nt] = f32x4{0, 0, 0, 0};
    if constexpr (HAS_BIAS && sizeof(TC) == 2) {
#pragma unroll
        for (int nt = 0; nt < NJ; ++nt) {
            const int n = cbase + 16 * nt + 4 * g4;
            if constexpr (BIAS_LDS) {  // unconditional read + select: no branch, one wait for the four reads
                const f32x4 t = *(const f32x4*)(stg + (16 * nt + 4 * g4) * 4);
                bias4[nt] = bias ? t : f32x4{0, 0, 0, 0};
            } else {
                if (bias && (FULL || n < N)) bias4[nt] = ld4(bias + n);
            }
        }
    }
    if constexpr (HAS_BIAS && sizeof(TC) == 4) {
        const int c4 = 4 * (lane & 15);
        if constexpr (BIAS_LDS) {
            const f32x4 t = *(const f32x4*)(stg + c4 * 4);
            bias_t = bias ? t : f32x4{0, 0, 0, 0};
        } else {
            if (bias && (FULL || cbase + c4 < N)) bias_t = ld4(bias + cbase + c4);
        }
    }
    if constexpr (BIAS_LDS) __builtin_amdgcn_sched_barrier(0);  // the reads above come before any write to `stg`
    if constexpr (sizeof(TC) == 2 && MODE == SSL4GIE_EPI_AFFINE_AUX_RELU) {
        p_store_bf16_aux<FULL, 5, NJ, NT>(acc, stg, alpha, aux, (bf16_t*)C, ldc, rbase, cbase, M, N, lane,
                                      residual /* scale */, bias /* shift */, accumulate /* relu */);
    } else if constexpr (sizeof(TC) == 2 && (MODE == SSL4GIE_EPI_DGELU || MODE == SSL4GIE_EPI_MUL_AUX ||
                                      MODE == SSL4GIE_EPI_RELU_MASK_AUX || MODE == SSL4GIE_EPI_ADD_AUX)) {
        constexpr int AUXF = MODE == SSL4GIE_EPI_DGELU ? 2
                           : (MODE == SSL4GIE_EPI_RELU_MASK_AUX ? 3 : (MODE == SSL4GIE_EPI_ADD_AUX ? 4 : 1));
        p_store_bf16_aux<FULL, AUXF, NJ, NT>(acc, stg, alpha, aux, (bf16_t*)C, ldc, rbase, cbase, M, N, lane);
    } else if constexpr (sizeof(TC) == 2) {
        // bf16: stage 16 rows x 64 columns (2 KiB), chunk c of row r at position c ^ (r & 7)
        const int R0 = lane >> 3, Cc = lane & 7;
        // stage one 16 x 64 bf16 block at byte offset `base` of the wave's staging area
        auto put = [&](int base, int nt, const f32x4& x) {
            u32x2 pk;
            pk[0] = pack_bf2(x[0], x[1]);
            pk[1] = pack_bf2(x[2], x[3]);
            const int c = nt * 2 + (g4 >> 1);
            *(u32x2*)(stg + base + r16 * 128 + ((c ^ (r16 & 7)) << 4) + ((g4 & 1) << 3)) = pk;
        };
        f32x2 cs[4], cq[4];  // STATS: this lane's 8 columns (pairs), summed over the rows it flushes
        if constexpr (STATS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) cs[j] = cq[j] = f32x2{0.f, 0.f};
        }
        constexpr bool PAIR = MODE == SSL4GIE_EPI_BIAS_GELU_GRAD || MODE == SSL4GIE_EPI_BIAS_GELU;
        // scale / activate block mt and stage it (PAIR: first output at 0, the GELU output at 2048)
        const bool mine = NJ == 4 || Cc < 2 * NJ;  // this lane's 8 columns exist in the wave tile
        auto stage = [&](int mt) {
#pragma unroll
            for (int nt = 0; nt < NJ; ++nt) {
                const f32x4 v = acc[mt][nt] * alpha + bias4[nt];
                if constexpr (MODE == SSL4GIE_EPI_BIAS_GELU_GRAD) {
                    // gelu(u) and gelu'(u) share exp(-u^2/2) and the erf polynomial
                    f32x4 g, d;
                    gelu_grad4_fast(v, d, g);
                    put(0, nt, d);
                    put(2048, nt, g);
                } else if constexpr (MODE == SSL4GIE_EPI_BIAS_GELU) {
                    f32x4 g;
#pragma unroll
                    for (int q = 0; q < 4; ++q) g[q] = gelu_fast(v[q]);
                    put(0, nt, v);
                    put(2048, nt, g);
                } else {
                    put(0, nt, v);
                }
            }
        };
        auto fetch = [&](int base, u32x4 (&w)[2]) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int R = R0 + 8 * hh;
                w[hh] = *(const u32x4*)(stg + base + R * 128 + ((Cc ^ (R & 7)) << 4));
            }
        };
        auto store = [&](const u32x4 (&w)[2], bf16_t* __restrict__ dst, int mt, bool stats) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int R = R0 + 8 * hh;
                const int gm = rbase + 16 * mt + R, gn = cbase + 8 * Cc;
                if (mine && (FULL || (gm < M && gn < N))) {
                    if (!STATS || dst) est<NT>((u32x4*)(dst + (size_t)gm * ldc + gn), w[hh]);  // STATS with C == NULL: statistics only
                    if constexpr (STATS) {
                        if (stats && !(prio_mode & 8)) {  // (bit 3: debug-library ablation)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {  // one v_pk_add_f32 + one v_pk_fma_f32 per column pair
                                const f32x2 v = {__uint_as_float(w[hh][j] << 16), __uint_as_float(w[hh][j] & 0xffff0000u)};
                                cs[j] += v;
                                cq[j] = v * v + cq[j];
                            }
                        }
                    }
                }
            }
        };
        // GELU pair without LDS (xpose_swap): the accumulator layout gives a lane 4 consecutive columns of a row
        // (8 B of bf16); v_permlane16_swap between the 16-lane groups g4 and g4 ^ 1 of two neighbouring column
        // blocks turns that into 8 consecutive columns (16 B) per lane, and a wave instruction then stores
        // 16 rows x 64 B.  No ds_write / ds_read, no lgkmcnt waits: the LDS round trip (write bandwidth is only
        // ~85 B/clk per CU) is what the pipelined form below still pays per block.
        if constexpr (PAIR && NJ == 4 && TAB) {
            static_assert(!STATS, "the table form exists for the GELU pair only");
            // 16 steps of 8 elements (two 16-column blocks of one 16-row block); the table reads of step i + 1 are
            // in flight while step i is multiplied, packed, exchanged and stored
            GeluTabQ qa[2], qb[2];
            auto rd = [&](int i, GeluTabQ& a, GeluTabQ& b) {
                const int mt = i >> 1, np = i & 1;
                p_gelu_tab_read(acc[mt][2 * np] * alpha + bias4[2 * np], tab, a);
                p_gelu_tab_read(acc[mt][2 * np + 1] * alpha + bias4[2 * np + 1], tab, b);
            };
            rd(0, qa[0], qb[0]);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int mt = i >> 1, np = i & 1;
                if (i + 1 < 16) rd(i + 1, qa[(i + 1) & 1], qb[(i + 1) & 1]);
                unsigned dA[2], gA[2], dB[2], gB[2];
                p_gelu_tab_finish(qa[i & 1], dA, gA);
                p_gelu_tab_finish(qb[i & 1], dB, gB);
                auto xq = [&](const unsigned (&a)[2], const unsigned (&b)[2]) -> u32x4 {
                    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
                    const u32x2 s2 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
                    return u32x4{s1[0], s2[0], s1[1], s2[1]};
                };
                const u32x4 wg = xq(gA, gB);
                const int gm = rbase + 16 * mt + r16;
                const int gn = cbase + 32 * np + 16 * (g4 & 1) + 8 * (g4 >> 1);
                if constexpr (MODE == SSL4GIE_EPI_BIAS_GELU_GRAD) {
                    const u32x4 wd = xq(dA, dB);
                    if (FULL || (gm < M && gn < N)) {
                        est<NT>((u32x4*)((bf16_t*)C + (size_t)gm * ldc + gn), wd);
                        est<NT>((u32x4*)(out2 + (size_t)gm * ldc + gn), wg);
                    }
                } else {  // C = u (bf16), out2 = gelu
                    const f32x4& uA = qa[i & 1].u;
                    const f32x4& uB = qb[i & 1].u;
                    const unsigned ua[2] = {pack_bf2(uA[0], uA[1]), pack_bf2(uA[2], uA[3])};
                    const unsigned ub[2] = {pack_bf2(uB[0], uB[1]), pack_bf2(uB[2], uB[3])};
                    const u32x4 wu = xq(ua, ub);
                    if (FULL || (gm < M && gn < N)) {
                        est<NT>((u32x4*)((bf16_t*)C + (size_t)gm * ldc + gn), wu);
                        est<NT>((u32x4*)(out2 + (size_t)gm * ldc + gn), wg);
                    }
                }
            }
            return;
        }
        if constexpr (PAIR && NJ == 4 && !TAB) {
            if (xpose_swap) {
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    const int gm = rbase + 16 * mt + r16;
#pragma unroll
                    for (int np = 0; np < 2; ++np) {
                        f32x4 dA, gA, dB, gB;
                        const f32x4 vA = acc[mt][2 * np] * alpha + bias4[2 * np];
                        const f32x4 vB = acc[mt][2 * np + 1] * alpha + bias4[2 * np + 1];
                        if constexpr (MODE == SSL4GIE_EPI_BIAS_GELU_GRAD) {
                            gelu_grad4_fast(vA, dA, gA);
                            gelu_grad4_fast(vB, dB, gB);
                        } else {
                            dA = vA; dB = vB;
#pragma unroll
                            for (int q = 0; q < 4; ++q) { gA[q] = gelu_fast(vA[q]); gB[q] = gelu_fast(vB[q]); }
                        }
                        auto xp = [&](const f32x4& a, const f32x4& b) -> u32x4 {
                            const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pack_bf2(a[0], a[1]), pack_bf2(b[0], b[1]), false, false);
                            const u32x2 s2 = __builtin_amdgcn_permlane16_swap(pack_bf2(a[2], a[3]), pack_bf2(b[2], b[3]), false, false);
                            return u32x4{s1[0], s2[0], s1[1], s2[1]};
                        };
                        const u32x4 wd = xp(dA, dB), wg = xp(gA, gB);
                        const int gn = cbase + 32 * np + 16 * (g4 & 1) + 8 * (g4 >> 1);
                        if (FULL || (gm < M && gn < N)) {
                            est<NT>((u32x4*)((bf16_t*)C + (size_t)gm * ldc + gn), wd);
                            est<NT>((u32x4*)(out2 + (size_t)gm * ldc + gn), wg);
                        }
                    }
                    if (tstamp) tstamp[mt] = __builtin_amdgcn_s_memrealtime();
                }
                return;
            }
        }
        // Software pipeline over the 8 row blocks (see p_store_f32): the transposed reads of block mt, then the
        // arithmetic and staging of block mt + 1 (in-order LDS: those writes land after the reads), then block
        // mt's stores — the GELU arithmetic of the next block hides the LDS round trip of this one.
        // PAIR (the GELU epilogues are VALU-bound): the two waves of a SIMD are the tile's two wave rows, and
        // VALU issue goes to the OLDER wave first — left alone the wr = 0 wave finishes at W and its partner,
        // then alone on the SIMD at half the issue rate, at ~3 W.  Alternating the priority per row block
        // (even blocks: wr = 0 ahead, odd: wr = 1) lets both finish at ~2 W.
        const int wrow = (rbase >> 7) & 1;
        // prio_mode: 0 none; 1 the wr = 1 waves ahead throughout; 2 alternate per block, wr = 1 ahead on even
        // blocks; 3 alternate per block, wr = 0 ahead on even blocks
        auto prio = [&](int blk) {
            if constexpr (PAIR) {
                const int pm = prio_mode & 3;
                const bool hi = pm == 1 ? wrow == 1
                              : pm == 2 ? ((blk & 1) == 0) == (wrow == 1)
                              : pm == 3 ? ((blk & 1) == 0) == (wrow == 0) : false;
                if (hi) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
        };
        prio(0);
        stage(0);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            u32x4 w0[2], w1[2];
            fetch(0, w0);
            if constexpr (PAIR) fetch(2048, w1);
            __builtin_amdgcn_sched_barrier(0);
            prio(mt + 1);
            if (mt + 1 < 8) stage(mt + 1);
            __builtin_amdgcn_sched_barrier(0);
            store(w0, (bf16_t*)C, mt, true);
            if constexpr (PAIR) store(w1, out2, mt, false);
            if (tstamp) tstamp[mt] = __builtin_amdgcn_s_memrealtime();
        }
        if constexpr (PAIR) __builtin_amdgcn_s_setprio(0);
        if constexpr (STATS) {
            // lanes with equal (lane & 7) own the same 8 columns: fold the 8 row groups
            if (!(prio_mode & 16))  // (bit 4: debug-library ablation)
            {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cs[j] = f32x2{p_fold_lanes8(cs[j][0]), p_fold_lanes8(cs[j][1])};
                    cq[j] = f32x2{p_fold_lanes8(cq[j][0]), p_fold_lanes8(cq[j][1])};
                }
            }
            const int gn = cbase + 8 * Cc;
            // a 128-row block that starts past M has no row in colstats (ceil(M / 128) blocks)
            if (lane < 8 && mine && !(prio_mode & 32) && (FULL || (gn < N && rbase < M))) {
                float* p = colstats + (size_t)(rbase >> 7) * 2 * N + gn;
                st4(p, f32x4{cs[0][0], cs[0][1], cs[1][0], cs[1][1]});
                st4(p + 4, f32x4{cs[2][0], cs[2][1], cs[3][0], cs[3][1]});
                st4(p + N, f32x4{cq[0][0], cq[0][1], cq[1][0], cq[1][1]});
                st4(p + N + 4, f32x4{cq[2][0], cq[2][1], cq[3][0], cq[3][1]});
            }
        }
    } else {
        // (Round 4, measured and dropped: fp32 outputs straight from the accumulator layout — 16 rows x 64 B per
        // wave instruction, no LDS — are SLOWER than the transposed 4 rows x 256 B stores: dec.proj 101 -> 114 us,
        // epilogue issue 5.1 / 8.4 -> 9.8 / 12.9 us (wave 0 / wave 4); profiles/r04v_nt_f32_direct_vs_lds.log.)
        p_store_f32<FULL, MODE == SSL4GIE_EPI_BIAS_RESIDUAL, NJ, NT>(
            acc, stg, alpha, bias_t, residual, ldr, MODE == SSL4GIE_EPI_NONE && accumulate, (float*)C,
            ldc, rbase, cbase, M, N, lane);
    }
}

