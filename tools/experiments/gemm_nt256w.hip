// EXPERIMENT (round 3; not built — moved here from csrc/ with its measurements, see DESIGN.md "what did not
// work").  Four waves per workgroup (one per SIMD, 512-register budget) so that a finished tile can stay
// behind as packed bf16 registers and leave DURING the next tile's K-loop.  Measured on MI355X
// (profiles/r03f_ntw_kloop_probe.log, r03h_ntw_trickle.log; MAE shapes, sum over the 13 NT products):
//   * K-loop alone, wave tile 128 x 128 (256 x 256 tile, 256 accumulators): 0.870 ms against 0.754 ms of
//     gemm_nt256.hip's 8-wave ping-pong loop, compiler-scheduled, first attempt — promising, but the staged
//     tile (128 registers) does not fit beside 256 accumulators + 76 fragment registers: hipcc spills
//     383-665 VGPRs (the split into 256 VGPR + 256 AGPR leaves no room, and it cannot be told to keep the
//     staged registers in the accumulator half);
//   * wave tile 128 x 64 (256 x 128 tile, this file): fits (no spills), results equal gemm_nt256.hip's to
//     bf16 rounding in all four epilogue kinds (tools/ntw_check.py) — but its K-loop alone takes 0.989 ms
//     (a barrier and a counted wait per 32 MFMAs, 1.5x the LDS-DMA instructions per MFMA), and with the
//     trickled epilogue 1.46 ms against gemm_nt256.hip's 1.20 ms: the stores of the GELU pair / residual
//     shapes exceed the per-CU store rate (~11 B/clk) even spread over the whole K-loop, and every stage's
//     in-order vmcnt wait then stalls behind them.
// Lessons kept: operand rows of a hand-scheduled epilogue must come in by LDS-DMA, not by asm loads into
// registers (a spilled in-flight register gave NaN); a never-switched-off DMA stream removes every branch
// and tail case from the loop body; 64-byte LDS rows are conflict-free with chunk ^ (-(row >> 2) & 3).
//
// 256x256x32 bf16 NT GEMM, FOUR waves per workgroup (one per SIMD, 512-register budget), wave tile
// 128 x 128:  C[M,N] = epilogue(A[M,K] * B[N,K]^T).
//
// Why a third NT kernel (round 3).  gemm_nt256.hip's K-loop runs at 1.1-1.3 PFLOP/s, but measured this
// round (profiles/r03b..r03d) its epilogue is a store DRAIN capped per CU: a CU pushes stores that miss
// the L2 at ~11 B/clk (~24 GB/s) whatever the other CUs do (same per-tile drain on 240, 120 and 60 CUs;
// half-chip start skew, write-through stores and tile order change nothing; stores that hit one
// L2-resident row cost ~nothing).  A 256x256 tile of the K = 512 / 768 projections leaves 128-512 KiB
// per 8-12 us of MFMA work, so the drain is as long as the K-loop — and in gemm_nt256.hip it is serial
// with it: its 8 waves hold 128 accumulator registers of 256 each, nothing of a finished tile can stay
// behind while the next one accumulates, and a wave that has issued stores must wait for them before its
// next LDS-DMA wait (one in-order vmcnt).
// Here a wave has 512 registers: 256 accumulators (8 x 8 tiles of v_mfma_f32_16x16x32_bf16) and room
// for a finished tile's outputs as 128 packed-bf16 registers, which leave through the LDS transposition
// and the stores a few rows per K-stage WHILE the next tile accumulates ("trickle"): the drain runs at
// its ~11 B/clk beside the K-loop instead of after it.  Rounding the product to bf16 before the
// residual add / GELU / gelu' multiply is the reference's autocast data flow (nn.Linear
// returns the operand type, the residual stream and GELU then promote).
//
// K-loop: LDS ring of 4 stages x 32 KiB (32-deep: A image 256 rows x 64 B | B image), filled by
// global_load_lds_dwordx4 from inline asm (8 pieces of 1 KiB per wave and stage), one s_barrier per
// stage.  Per stage a wave reads 8 + 8 fragments (ds_read_b128, chunk c of row r at position
// c ^ (-(r >> 2) & 3): conflict-free, tools/lds_bank_sim.py) and issues 64 MFMAs; the B fragments and the
// first A fragment of stage T+1 are read after barrier B_T, which sits after the 6th of the 8 row
// blocks of stage T, so their latency hides behind the last 16 MFMAs.  Hazards (T = stage index):
//   RAW  stage T+1 is read after B_T; every wave waited (counted vmcnt) for its own pieces of T+1
//        before B_T;
//   WAR  stage T+3 overwrites the slot of T-1, whose last fragment read (row block 7) was issued before
//        B_T's predecessor row block; its pieces are issued after B_T (first half) / B_{T+1} (second).
// The LDS-DMA stream is continuous across output tiles (persistent workgroups, XCD-aware order).
//
// Replaces the cuBLAS calls behind nn.Linear in timm Block / the MAE decoder (reference call sites
// Models/mae/models_mae.py:39-41,47,53-55,59; Models/models.py:171-173).
#include "gemm256.h"
#include "prof.h"

#include <stdlib.h>
#include <type_traits>

#define W_BM 256
#define W_BN 128
#define W_BK 32
#define W_STAGE 24576       // A image 256 rows x 64 B (16 KiB) | B image 128 rows x 64 B (8 KiB)
#define W_BOFF 16384
#define W_NSTAGE 4
#define W_STG (W_NSTAGE * W_STAGE)
#define W_STG_WAVE 16384    // wave-private: 2 x 2 KiB transposition | 2 x 4 KiB operand rows | 1 KiB bias
#define W_OPND 4096
#define W_BIAS 12288
#define W_LDS_BYTES (W_STG + 4 * W_STG_WAVE)  // 163840 = all of the CU's LDS

template <int N> DEVI void w_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One LDS-DMA piece with per-lane 64-bit addresses: lane i's 16 bytes land at l + 16 i.  The epilogue's
// operand rows (residual / saved gelu' / bias) come in this way, NOT through registers: a register that a
// hand-issued load is still filling looks ready to the compiler, which may copy or spill it (seen: a
// spilled in-flight register gave NaN); bytes in LDS are only touched by the ds_read behind the wave's
// own counted vmcnt wait.  No compiler-visible global load exists in this kernel, so hipcc never inserts a
// vmcnt wait of its own (it does not know the pieces in the queue and would drain the stream).
DEVI void w_dma16(const void* a, unsigned l) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(a), "s"(l)
        : "memory");
}

#define WK_PLAIN 0  // bf16 C = acc (+ bias)
#define WK_GELU 1   // bf16 C = gelu'(u), out2 = gelu(u), u = bf16(acc + bias)
#define WK_AUX 2    // bf16 C = bf16(acc) * aux
#define WK_RESID 3  // fp32 C = residual + bf16(acc + bias)
template <typename TC, int MODE> constexpr int w_kind() {
    if (sizeof(TC) == 4) return WK_RESID;
    if (MODE == SSL4GIE_EPI_BIAS_GELU_GRAD) return WK_GELU;
    if (MODE == SSL4GIE_EPI_MUL_AUX) return WK_AUX;
    return WK_PLAIN;
}
template <int I> using IC = std::integral_constant<int, I>;

template <typename TC, int MODE, bool EPI>
__global__ __launch_bounds__(256, 1) void gemm_bf16_nt256w_kernel(
    const bf16_t* __restrict__ A, long long lda, const bf16_t* __restrict__ B, long long ldb,
    TC* __restrict__ C, long long ldc, int M, int N, int K, int tiles_n, int ntiles, EpiArgs e, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KIND = w_kind<TC, MODE>();
    constexpr bool HAS_BIAS = MODE == SSL4GIE_EPI_BIAS || MODE == SSL4GIE_EPI_BIAS_GELU_GRAD ||
                              MODE == SSL4GIE_EPI_BIAS_RESIDUAL;
    // vector-memory instructions of one trickle step (whole tiles): the odd steps store row block rb and
    // (AUX / RESID) request the operand rows of row block rb + 1; even steps only write LDS
    constexpr int N_ST = KIND == WK_PLAIN ? 2 : (KIND == WK_GELU ? 4 : (KIND == WK_AUX ? 2 : 4));
    constexpr int N_LD = KIND == WK_AUX ? 2 : (KIND == WK_RESID ? 4 : 0);
    constexpr int NPC = 6;   // LDS-DMA instructions per wave and stage
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int G = gridDim.x;
    const int pos = xcd_remap(blockIdx.x, G);
    const int my_tiles = (ntiles - pos + G - 1) / G;
    const int nk = K / W_BK;
    const float* e_bias = e.bias;
    const float* e_residual = e.residual;
    const long long e_ldr = e.ldr;
    const bf16_t* e_aux = (const bf16_t*)e.aux;
    bf16_t* e_out2 = (bf16_t*)e.out2;

    // ------------------------------------------------------------------ LDS-DMA stream
    // this wave's 6 pieces per stage: A rows [64 wave + 16 j, +16), j = 0..3, B rows [32 wave + 16 j, +16),
    // j = 0, 1; lane i of a piece fills row (i >> 2), position (i & 3) with global chunk
    // (i & 3) ^ swz(row) (swz = -(row >> 2) & 3 is lane-only here)
    unsigned va0 = 0, va1 = 0, va2 = 0, va3 = 0, vb0 = 0, vb1 = 0;
    const int d_c = (lane & 3) ^ ((-(lane >> 4)) & 3);
    auto point_at = [&](int ti) {
        const int tile = pos + ti * G;
        const int sm0 = (tile / tiles_n) * W_BM, sn0 = (tile % tiles_n) * W_BN;
        auto oa = [&](int j) -> unsigned {
            int r = sm0 + 64 * wave + 16 * j + (lane >> 2);
            r = r < M ? r : M - 1;
            return (unsigned)(((long long)r * lda + d_c * 8) * 2);
        };
        auto ob = [&](int j) -> unsigned {
            int r = sn0 + 32 * wave + 16 * j + (lane >> 2);
            r = r < N ? r : N - 1;
            return (unsigned)(((long long)r * ldb + d_c * 8) * 2);
        };
        va0 = oa(0); va1 = oa(1); va2 = oa(2); va3 = oa(3);
        vb0 = ob(0); vb1 = ob(1);
    };
    const unsigned ldsA = __builtin_amdgcn_readfirstlane(p_lds_addr(smem) + wave * 4096);
    const unsigned ldsB = __builtin_amdgcn_readfirstlane(p_lds_addr(smem) + W_BOFF + wave * 2048);
    int s_st = 0, s_kt = 0, s_ti = 0;  // stream cursor: global stage, stage in tile, tile
    // issue one pair of pieces of the stream's current stage: J = 0, 1 the A rows, J = 2 the B rows (and
    // the cursor advances).  The stream is never switched off: behind the workgroup's last stage it keeps
    // filling ring slots nobody reads any more (3 stages, rows clamped into the matrices) — no branch in
    // the loop body, and every counted wait sees the same number of younger pieces
    auto issue = [&](auto Jc) {
        constexpr int J = decltype(Jc)::value;
        const unsigned slot = __builtin_amdgcn_readfirstlane((s_st & 3) * W_STAGE);
        const int koff = __builtin_amdgcn_readfirstlane(s_kt * W_BK);
        if constexpr (J < 2) {
            p_glds2(A + koff, J == 0 ? va0 : va2, J == 0 ? va1 : va3, ldsA + slot + J * 2048,
                    ldsA + slot + J * 2048 + 1024);
        } else {
            p_glds2(B + koff, vb0, vb1, ldsB + slot, ldsB + slot + 1024);
            ++s_st;
            if (++s_kt == nk) {
                s_kt = 0;
                point_at(++s_ti);
            }
        }
    };

    // ------------------------------------------------------------------ fragment reads
    const int l15 = lane & 15, g4 = lane >> 4;
    const int fchunk = (g4 ^ ((-(l15 >> 2)) & 3)) << 4;
    const int offA = (128 * wr + l15) * 64 + fchunk;
    const int offB = W_BOFF + (64 * wc + l15) * 64 + fchunk;
    auto ldA = [&](int st, int mi) -> bf16x8 { return *(const bf16x8*)(smem + st * W_STAGE + offA + mi * 1024); };
    auto ldB = [&](int st, int ni) -> bf16x8 { return *(const bf16x8*)(smem + st * W_STAGE + offB + ni * 1024); };

    f32x4 acc[8][4];
    bf16x8 bq[4], bn[4], a0, a1, a2, a_n0, a_n1;

    // ------------------------------------------------------------------ the trickled epilogue
    // staged tile: acc[mi][ni] (rows 16 mi + l15, columns 16 ni + 4 g4 + 0..3) as 4 bf16 in 2 registers
    u32x2 sg[8][4];
    int p_m = 0, p_n = 0;     // first row / column of this wave's 128 x 64 part of the staged tile
    bool p_full = true;       // the staged tile lies wholly inside C: the store counts are exact
    // wave-private transposition buffers: 2 x (16 rows x 128 B), chunk c of row r at c ^ ((r >> 1) & 7)
    char* const stg = smem + W_STG + wave * W_STG_WAVE;
    // LDS write of one accumulator tile: row l15, 8 bytes at column 16 ni + 4 g4 (chunk 2 ni + (g4 >> 1))
    const int tw_base = l15 * 128 + ((g4 & 1) << 3), tw_x = (((g4 >> 1) ^ (l15 >> 1)) & 7) << 4;
    // bf16 read-back / global rows: row (lane >> 3) + 8 q, 16 bytes at column 8 (lane & 7)
    const int R0 = lane >> 3, Cc = lane & 7;
    const int tr_base = R0 * 128;
    // fp32 (RESID) read-back / global rows: row 4 i + (lane >> 4), 8 bytes (4 columns) at column 4 (lane & 15)
    const int Dc = (lane & 15) >> 1;
    const int td_base = g4 * 128 + ((lane & 1) << 3);
    u32x4 rd[2];                 // read-back of one row block (bf16 kinds: 2 x 16 B; RESID: 4 x 8 B)
    // operand rows of a row block (AUX: 2 pieces of 8 bf16 per lane; RESID: 4 pieces of 4 fp32 per lane) land
    // in LDS buffer RB & 1, every lane's 16 bytes at piece * 1024 + 16 lane: read back by the same lane
    const unsigned lds_stg = __builtin_amdgcn_readfirstlane(p_lds_addr(smem) + W_STG + wave * W_STG_WAVE);

    auto t_write = [&](auto RBc, auto NIc) {   // LDS write of sg[RB][NI] into transposition buffer RB & 1
        constexpr int RB = decltype(RBc)::value, NI = decltype(NIc)::value;
        *(u32x2*)(stg + (RB & 1) * 2048 + tw_base + ((32 * NI) ^ tw_x)) = sg[RB][NI];
    };
    auto t_load = [&](auto RBc) {              // request the operand rows of row block RB (LDS-DMA)
        constexpr int RB = decltype(RBc)::value;
        const unsigned dst = lds_stg + W_OPND + (RB & 1) * 4096;
        if constexpr (KIND == WK_AUX) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                int gm = p_m + 16 * RB + R0 + 8 * q, gn = p_n + 8 * Cc;
                gm = gm < M ? gm : M - 1;
                gn = gn < N ? gn : N - 8;
                w_dma16(e_aux + (size_t)gm * ldc + gn, dst + q * 1024);
            }
        } else if constexpr (KIND == WK_RESID) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int gm = p_m + 16 * RB + 4 * i + g4, gn = p_n + 4 * (lane & 15);
                gm = gm < M ? gm : M - 1;
                gn = gn < N ? gn : N - 4;
                w_dma16(e_residual + (size_t)gm * e_ldr + gn, dst + i * 1024);
            }
        }
    };
    auto t_opnd = [&](auto RBc, int piece) -> u32x4 {   // this lane's 16 bytes of an operand piece
        constexpr int RB = decltype(RBc)::value;
        return *(const u32x4*)(stg + W_OPND + (RB & 1) * 4096 + piece * 1024 + lane * 16);
    };
    auto t_read = [&](auto RBc, auto Qc) {     // LDS read-back, part Q of 2
        constexpr int RB = decltype(RBc)::value, Q = decltype(Qc)::value;
        const char* b = stg + (RB & 1) * 2048;
        if constexpr (KIND == WK_RESID) {
            u32x2 x[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = 4 * (2 * Q + h) + g4;   // row of the block
                x[h] = *(const u32x2*)(b + td_base + 512 * (2 * Q + h) + (((Dc ^ (r >> 1)) & 7) << 4));
            }
            rd[Q] = u32x4{x[0][0], x[0][1], x[1][0], x[1][1]};
        } else {
            const int r = R0 + 8 * Q;
            rd[Q] = *(const u32x4*)(b + tr_base + 1024 * Q + (((Cc ^ (r >> 1)) & 7) << 4));
        }
    };
    auto unpack = [](unsigned w, float& lo, float& hi) {
        lo = __uint_as_float(w << 16);
        hi = __uint_as_float(w & 0xffff0000u);
    };
    auto t_store = [&](auto RBc, auto Qc) {    // arithmetic + global stores of part Q of row block RB
        constexpr int RB = decltype(RBc)::value, Q = decltype(Qc)::value;
        if constexpr (KIND == WK_RESID) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * Q + h;
                const int gm = p_m + 16 * RB + 4 * i + g4, gn = p_n + 4 * (lane & 15);
                f32x4 r = __builtin_bit_cast(f32x4, t_opnd(RBc, i));
                float v0, v1, v2, v3;
                unpack(rd[Q][2 * h], v0, v1);
                unpack(rd[Q][2 * h + 1], v2, v3);
                r += f32x4{v0, v1, v2, v3};
                if (p_full || (gm < M && gn < N)) *(f32x4*)((float*)C + (size_t)gm * ldc + gn) = r;
            }
        } else {
            const int gm = p_m + 16 * RB + R0 + 8 * Q, gn = p_n + 8 * Cc;
            const bool ok = p_full || (gm < M && gn < N);
            bf16_t* c = (bf16_t*)C + (size_t)gm * ldc + gn;
            if constexpr (KIND == WK_PLAIN) {
                if (ok) *(u32x4*)c = rd[Q];
            } else if constexpr (KIND == WK_AUX) {
                u32x4 o;
                const u32x4 xv = t_opnd(RBc, Q);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float y0, y1, x0, x1;
                    unpack(rd[Q][j], y0, y1);
                    unpack(xv[j], x0, x1);
                    o[j] = pack_bf2(y0 * x0, y1 * x1);
                }
                if (ok) *(u32x4*)c = o;
            } else {  // WK_GELU: gelu(u) and gelu'(u) share exp(-u^2/2) and the erf polynomial
                u32x4 od, og;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float u0, u1, c0, x0, c1, x1;
                    unpack(rd[Q][j], u0, u1);
                    gelu_parts_fast(u0, c0, x0);
                    gelu_parts_fast(u1, c1, x1);
                    od[j] = pack_bf2(c0 + x0, c1 + x1);
                    og[j] = pack_bf2(u0 * c0, u1 * c1);
                }
                if (ok) {
                    *(u32x4*)c = od;
                    *(u32x4*)(e_out2 + (size_t)gm * ldc + gn) = og;
                }
            }
        }
    };
    // one trickle step, interleaved with the row blocks MI = 0..7 of a K-stage: TR = 2 rb writes row
    // block rb into LDS (MI 0..3); TR = 2 rb + 1 reads it back (MI 0, 1) and, behind the stage's counted
    // wait (MI >= 6: every load older than the last NPC pieces has landed), does the arithmetic, the stores
    // and the requests for row block rb + 1.  All vector-memory instructions sit in MI 6, 7.
    auto trickle = [&](auto TRc, auto MIc) {
        constexpr int TR = decltype(TRc)::value, MI = decltype(MIc)::value;
        if constexpr (TR >= 0 && TR < 16) {
            using RB = IC<(TR >> 1)>;
            if constexpr ((TR & 1) == 0) {
                if constexpr (MI < 4) t_write(RB{}, IC<(MI < 4 ? MI : 0)>{});
            } else {
                if constexpr (MI < 2) t_read(RB{}, IC<(MI < 2 ? MI : 0)>{});
                if constexpr (MI == 6) t_store(RB{}, IC<0>{});
                if constexpr (MI == 7) {
                    t_store(RB{}, IC<1>{});
                    if constexpr (N_LD > 0 && (TR >> 1) < 7) t_load(IC<((TR >> 1) < 7 ? (TR >> 1) + 1 : 0)>{});
                }
            }
        }
    };
    // younger vector-memory instructions than the pieces of stage T+1 at the counted wait of a stage that
    // runs trickle step TR (whole staged tile): the NPC pieces of stage T+2, plus what the previous stage
    // issued in its row blocks 6, 7.  prev15: the previous stage ran step 15 (K = 16 stages per tile).
    auto stage_wait = [&](auto TRc, bool prev15) {
        constexpr int TR = decltype(TRc)::value;
        if (!p_full) { w_vmcnt<NPC>(); return; }
        if constexpr (TR == 0) {
            if (prev15) w_vmcnt<NPC + N_LD + N_ST>();   // step 15's stores + the requests for row block 0
            else w_vmcnt<NPC + N_LD>();
        } else if constexpr (TR > 0 && TR < 16 && (TR & 1) == 0) {
            w_vmcnt<NPC + N_ST + N_LD>();
        } else {
            w_vmcnt<NPC>();
        }
    };

    // bias rides in the accumulators' initial value (alpha == 1): one LDS-DMA piece per tile (lanes 0..15
    // fetch the wave's 64 columns, 16 bytes each), read back as column group 4 ni + g4
    auto bias_request = [&](int n0w) {
        if constexpr (HAS_BIAS) {
            int gn = n0w + 4 * (lane & 15);
            gn = gn < N ? gn : N - 4;
            w_dma16(e_bias + gn, lds_stg + W_BIAS);
        }
    };
    auto acc_init = [&]() {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            f32x4 b4 = {0, 0, 0, 0};
            if constexpr (HAS_BIAS) b4 = *(const f32x4*)(stg + W_BIAS + (4 * ni + g4) * 16);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) acc[mi][ni] = b4;
        }
    };

    // ------------------------------------------------------------------ prologue
    point_at(0);
    issue(IC<0>{}); issue(IC<1>{}); issue(IC<2>{});   // stage 0
    issue(IC<0>{}); issue(IC<1>{}); issue(IC<2>{});   // stage 1
    issue(IC<0>{}); issue(IC<1>{});                   // the A rows of stage 2
    bias_request(((pos % tiles_n) * W_BN) + 64 * wc);
    w_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    acc_init();
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) bq[ni] = ldB(0, ni);
    a0 = ldA(0, 0);
    a1 = ldA(0, 1);

    int T = 0;  // global stage index
    auto stage = [&](auto TRc, bool prev15) {
        const int st = T & 3, sn = (T + 1) & 3;
        auto block = [&](auto MIc) {
            constexpr int mi = decltype(MIc)::value;
            if constexpr (mi < 6) a2 = ldA(st, mi + 2);
            if constexpr (mi == 1) issue(IC<2>{});        // the B rows of stage T+2: its last pieces
            if constexpr (mi == 5) {
                stage_wait(TRc, prev15);                  // stage T+1 has landed
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) bn[ni] = ldB(sn, ni);
                a_n0 = ldA(sn, 0);
                a_n1 = ldA(sn, 1);
            }
            if constexpr (mi == 6) issue(IC<0>{});        // behind the barrier: the A rows of stage T+3
            if constexpr (mi == 7) issue(IC<1>{});
            if constexpr (EPI) trickle(TRc, MIc);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = P_MFMA(bq[ni], a0, acc[mi][ni]);
            a0 = a1;
            if constexpr (mi < 6) a1 = a2;
        };
        block(IC<0>{}); block(IC<1>{}); block(IC<2>{}); block(IC<3>{});
        block(IC<4>{}); block(IC<5>{}); block(IC<6>{}); block(IC<7>{});
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bq[ni] = bn[ni];
        a0 = a_n0;
        a1 = a_n1;
        ++T;
    };

    for (int ti = 0; ti < my_tiles; ++ti) {
        int kt = 0;
        if (EPI && dbg != 1 && ti > 0 && nk >= 16) {
            stage(IC<0>{}, nk == 16 && ti > 1);
            stage(IC<1>{}, false); stage(IC<2>{}, false); stage(IC<3>{}, false); stage(IC<4>{}, false);
            stage(IC<5>{}, false); stage(IC<6>{}, false); stage(IC<7>{}, false); stage(IC<8>{}, false);
            stage(IC<9>{}, false); stage(IC<10>{}, false); stage(IC<11>{}, false); stage(IC<12>{}, false);
            stage(IC<13>{}, false); stage(IC<14>{}, false); stage(IC<15>{}, false);
            kt = 16;
        }
        for (; kt < nk; ++kt) stage(IC<-1>{}, false);
        // ---------------------------------------------------------------- tile boundary
        const int tile = pos + ti * G;
        const int m0 = (tile / tiles_n) * W_BM, n0 = (tile % tiles_n) * W_BN;
        if (!EPI || dbg == 1) {
            // K-loop probe: keep every accumulator alive at the price of 128 adds per tile
            f32x4 sum = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) sum += acc[i][j];
            if (dbg == 77) *(f32x4*)((float*)C + (size_t)(m0 + lane) * 4) = sum;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
            continue;
        }
        const bool more = ti + 1 < my_tiles;
        // the next tile's bias is requested first: its latency hides behind the packing below
        if (more) bias_request((((pos + (ti + 1) * G) % tiles_n) * W_BN) + 64 * wc);
        // pack the finished tile: it leaves during the next tile's first 16 stages
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const f32x4 v = acc[mi][ni];
                sg[mi][ni] = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            }
        p_m = m0 + 128 * wr;
        p_n = n0 + 64 * wc;
        p_full = m0 + W_BM <= M && n0 + W_BN <= N;
        if (more) {
            w_vmcnt<0>();   // the bias has landed (and with it every piece issued so far)
            acc_init();
        }
        if (more && nk >= 16) {
            t_load(IC<0>{});   // operand rows of row block 0 (AUX / RESID)
        } else {
            // nothing left to hide behind (last tile of the workgroup, or a K-loop shorter than the
            // trickle): the staged tile leaves serially
            auto flush = [&](auto RBc) {
                t_load(RBc);
                t_write(RBc, IC<0>{}); t_write(RBc, IC<1>{}); t_write(RBc, IC<2>{}); t_write(RBc, IC<3>{});
                t_read(RBc, IC<0>{}); t_read(RBc, IC<1>{});
                w_vmcnt<0>();
                t_store(RBc, IC<0>{}); t_store(RBc, IC<1>{});
            };
            flush(IC<0>{}); flush(IC<1>{}); flush(IC<2>{}); flush(IC<3>{});
            flush(IC<4>{}); flush(IC<5>{}); flush(IC<6>{}); flush(IC<7>{});
        }
    }
    w_vmcnt<0>();  // the stream's surplus pieces land before the workgroup's LDS is released
}

// =====================================================================================
// host side
// =====================================================================================
static int nt256w_mode() {  // SSL4GIE_NT256W: "1" on, "0" / unset off (experimental)
    static int v = -2;
    if (v == -2) {
        const char* s = getenv("SSL4GIE_NT256W");
        v = (s && s[0] == '1') ? 1 : 0;
    }
    return v;
}

bool ssl4gie_internal_nt256w_ok(const ssl4gie_gemm_desc* d) {
    if (!nt256w_mode() || d->conv || d->colstats || d->accumulate || d->alpha != 1.0f) return false;
    if ((long long)d->M * d->sAm * 2 >= (1LL << 32) || (long long)d->N * d->sBn * 2 >= (1LL << 32)) return false;
    if (d->K % W_BK != 0 || d->K < 3 * W_BK || d->N % 8 != 0 || d->ldc % 8 != 0) return false;
    const int ep = d->epilogue;
    if (d->dtype_c == SSL4GIE_BF16) {
        if (ep == SSL4GIE_EPI_NONE) return true;
        if (ep == SSL4GIE_EPI_BIAS) return d->bias != nullptr;
        if (ep == SSL4GIE_EPI_BIAS_GELU_GRAD) return d->bias != nullptr && d->out2 != nullptr;
        if (ep == SSL4GIE_EPI_MUL_AUX) return d->aux != nullptr;
        return false;
    }
    return ep == SSL4GIE_EPI_BIAS_RESIDUAL && d->bias != nullptr && d->residual != nullptr && d->ldr % 4 == 0;
}

int ssl4gie_internal_nt256w_launch(const ssl4gie_gemm_desc* d, hipStream_t st) {
    const int tm = (d->M + W_BM - 1) / W_BM, tn = (d->N + W_BN - 1) / W_BN;
    const int ntiles = tm * tn;
    const int cus = ssl4gie_internal_compute_cus();
    dim3 grid(ntiles < cus ? ntiles : cus), block(256);
    EpiArgs e{d->alpha, d->epilogue, d->bias, d->residual, d->ldr, d->aux, d->out2, d->accumulate, nullptr};
    static int dbg = -1;
    if (dbg < 0) { const char* s = getenv("SSL4GIE_NT256_NOEPI"); dbg = (s && s[0] == '1') ? 1 : 0; }
    ProfScope prof(PROF_GEMM_NT, 2.0 * d->M * d->N * d->K, st);
#define W_LAUNCH(TC_, MODE_)                                                                        \
    do {                                                                                            \
        auto kfn = gemm_bf16_nt256w_kernel<TC_, MODE_, true>;                                             \
        static bool attr_set = false;                                                               \
        if (!attr_set) {                                                                            \
            HIP_RET(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        W_LDS_BYTES));                                              \
            attr_set = true;                                                                        \
        }                                                                                           \
        hipLaunchKernelGGL(kfn, grid, block, W_LDS_BYTES, st, (const bf16_t*)d->A, d->sAm,          \
                           (const bf16_t*)d->B, d->sBn, (TC_*)d->C, d->ldc, d->M, d->N, d->K, tn,   \
                           ntiles, e, dbg);                                                         \
    } while (0)
    if (d->dtype_c == SSL4GIE_BF16) {
        switch (d->epilogue) {
            case SSL4GIE_EPI_NONE: W_LAUNCH(bf16_t, SSL4GIE_EPI_NONE); break;
            case SSL4GIE_EPI_BIAS: W_LAUNCH(bf16_t, SSL4GIE_EPI_BIAS); break;
            case SSL4GIE_EPI_BIAS_GELU_GRAD: W_LAUNCH(bf16_t, SSL4GIE_EPI_BIAS_GELU_GRAD); break;
            case SSL4GIE_EPI_MUL_AUX: W_LAUNCH(bf16_t, SSL4GIE_EPI_MUL_AUX); break;
            default: return ARG_ERR;
        }
    } else {
        switch (d->epilogue) {
            case SSL4GIE_EPI_BIAS_RESIDUAL: W_LAUNCH(float, SSL4GIE_EPI_BIAS_RESIDUAL); break;
            default: return ARG_ERR;
        }
    }
#undef W_LAUNCH
    LAUNCH_CHECK();
    return 0;
}
