// 256x256x64 "ping-pong" bf16 NT GEMM for gfx950:  C[M,N] = epilogue(A[M,K] * B[N,K]^T).
//
// Why a second NT kernel: a 128x128 tile moves 32 KiB of operands per 2.1 MFLOP (64 FLOP/B), so at
// the MFMA rate of one CU it asks the XCD's L2 for ~150 GB/s per CU = 38 TB/s chip-wide — more
// than the L2s deliver (MI355X_MICROARCH.md: ~34 TB/s).  A 256x256 tile halves that.  One
// workgroup of 8 waves per CU (2 waves per SIMD), wave grid 2 (M) x 4 (N), wave tile 128 x 64 =
// 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16 (128 VGPRs).
//
// Schedule (cdna_hip_programming.md §5 "8-phase" structure, own derivation):
//  * a K-tile (64 deep) is split into 4 "half-tiles" of 128 rows x 128 B = 16 KiB, in the order
//    they are needed: B_h0, A_h0, B_h1, A_h1, where A_h{q} holds rows {128 wr + 64 q + 0..63} of
//    both wave rows and B_h{q} holds columns {64 wc + 32 q + 0..31} of all four wave columns.
//    Two K-tile buffers (128 KiB) + 8 x 4 KiB wave-private epilogue staging = all 160 KiB of LDS;
//  * the K-tile is computed in 4 phases of 16 MFMAs per wave (one 64x32 quadrant each):
//        P0: read B_h0 (4 ds_read_b128) then A_h0 (8)  -> quadrant (0,0)
//        P1: read B_h1 (4)                              -> quadrant (0,1)
//        P2: read A_h1 (8)                              -> quadrant (1,1)
//        P3: no LDS reads                               -> quadrant (1,0)
//    Every phase is {LOAD: ds_reads + 2 LDS-DMA of a later half-tile; s_barrier; MMA: 16 MFMAs;
//    s_barrier}.  The wr=1 waves run one barrier behind the wr=0 waves, so on every SIMD one wave
//    is in its MMA segment while its partner is in its LOAD segment (matrix beside memory);
//  * the LDS-DMA stream (global_load_lds_dwordx4 issued from inline asm, so hipcc never drains it)
//    runs 7 half-tiles ahead of consumption and is waited for ONCE per K-tile with a counted
//    s_waitcnt vmcnt(6) (3 half-tiles stay in flight across the barriers).  It is continuous across
//    output tiles (persistent workgroups), so the short K of this workload (8-48 K-tiles) never
//    pays a pipeline fill per tile.
// Hazard bookkeeping (g = global phase index; interval I_n = between barriers n-1 and n; the wr=0
// group runs LOAD(g) in I_2g and MMA(g) in I_2g+1, the wr=1 group one interval later):
//   RAW: half-tile s is first read in phase need(s) >= wait_phase(s)+1 — every wave waited for
//        its own DMA of s before a barrier that precedes the reading interval of either group;
//   WAR: the half-tile issued in phase g overwrites the one read in phase g-2 (reads complete
//        at the lgkmcnt wait that opens MMA(g-2), two barriers earlier for either group), except
//        B_h0, issued in P1 and read in P0: its 4 reads are issued first and retired by
//        s_waitcnt lgkmcnt(8) BEFORE P0's first barrier.
//
// Epilogue: outputs are transposed through the wave-private LDS staging area and leave as whole
// 128-byte (bf16) / 256-byte (fp32) row segments, 16 B per lane; the fp32 residual is read in that
// same coalesced layout.
//
// Replaces the cuBLAS calls behind nn.Linear in timm Block / MAE decoder (reference
// call sites Models/mae/models_mae.py:39-41,47,53-55,59; Models/models.py:171-173).
#include "gemm256.h"
#include "gelu_table.h"
#include "prof.h"

#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#define NT256_DEFAULT_ROLE 0
#define NT256_DEFAULT_PH2 true

// Diagnostics (SSL4GIE_NT256_NOEPI=4): wave 0 of every 16th workgroup stamps s_memrealtime (100 MHz) at five
// points of every output tile — K-loop start, epilogue start, epilogue end (last store issued), end of the
// next tile's first K-tile (its counted vmcnt wait has passed) and of its second — into this buffer, read
// back with ssl4gie_debug_nt256_stamps.  Outputs stay correct; the stamps' own stores perturb a little.
// The stamps, like every ablation mode, exist in the debug library only (make DEBUG_KNOBS=1); the release
// library's ssl4gie_debug_nt256_stamps returns SSL4GIE_ENOTBUILT-style ARG_ERR.
#define NT256_STAMP_WGS 16
#define NT256_STAMP_TILES 16
#ifdef SSL4GIE_DEBUG_KNOBS
__device__ unsigned long long g_nt256_stamps[NT256_STAMP_WGS][NT256_STAMP_TILES][16];  // 0-4: see above; 5-12: row blocks

extern "C" int ssl4gie_debug_nt256_stamps(void* dst, size_t bytes) {
    REQUIRE(dst && bytes <= sizeof(g_nt256_stamps));
    HIP_RET(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_nt256_stamps), bytes));
    return 0;
}
#else
extern "C" int ssl4gie_debug_nt256_stamps(void*, size_t) { return ARG_ERR; }
#endif

// CONV: 0 = A is a matrix; 1 = A is the implicit 3x3 patch matrix of the map at `A` (geometry cg,
// header of ssl4gie_conv3x3_geom); 2 = the same with ReLU applied to the A fragments.
// ROLE: 0 = every wave issues its two LDS-DMA pieces of each half-tile and executes the counted wait;
//       1 = the four wr = 1 waves ("loaders") issue the whole half-tile (four pieces each) and are the only ones
//           that wait on vmcnt in the K-loop: the wr = 0 waves then have nothing but their own output stores in
//           their (in-order) vector-memory queue and never wait for them — the barriers publish the landed tiles.
// NJ:   16-column accumulator blocks per wave: 4 = the 256 x 256 tile (wave tile 128 x 64); 3 = a 256 x 192 tile
//       (wave tile 128 x 48) for products whose 256-wide tiles leave a large part of the chip idle (N = 768 at
//       M = 12800: 150 tiles on 240 CUs -> 200 tiles of 3/4 the work).  Same four phases, barriers and LDS-DMA
//       placement; the second B half-tile shrinks to 64 rows (one piece per wave, columns 48 wc + 32 .. 47) and
//       the phases that use it run 8 MFMAs instead of 16.
// TAB:  the GELU epilogues (SSL4GIE_EPI_BIAS_GELU, _GELU_GRAD) read Phi / gelu' from the 16-KiB table of
//       gelu_table.h, brought into the staging area once per (persistent) workgroup by the first 16 LDS-DMA
//       pieces of the stream; the bias rows then live behind it (256 B per wave).  NJ = 4, ROLE 0 only.
template <typename TC, int MODE, int CONV, bool STATS = false, int ROLE = 0, int NJ = 4, bool PH2 = false,
          bool NTS = false /* non-temporal output stores (gemm256.h est) */, bool TAB = false>
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt256_kernel(
    const bf16_t* __restrict__ A, long long lda, const bf16_t* __restrict__ B, long long ldb,
    TC* __restrict__ C, long long ldc, int M, int N, int K, int tiles_n, int ntiles, EpiArgs e,
    int dbg_arg /* ablation knobs, compiled in with -DSSL4GIE_DEBUG_KNOBS only (timing only, wrong output):
               1 = no epilogue; 2 = epilogue arithmetic and LDS transposition without any global access; 3 = every
               global access of the epilogue lands in ONE row per workgroup (L2-resident: store issue without HBM
               write-back); 4 = time stamps (outputs correct); 5 = the wr = 1 waves skip their epilogue; 6 = they
               skip it and the wr = 0 waves run theirs twice (second time on the partner's rows); 7 = no LDS-DMA,
               no MFMA: the epilogues alone; 8 = the time stamps of 4 taken by wave 4 (a wr = 1 wave) */,
    ConvK cg) {
#ifdef SSL4GIE_DEBUG_KNOBS
    const int dbg = dbg_arg & 15;
    const int dbg_prio = dbg_arg >> 4;  // SSL4GIE_NT256_PRIO: bits 0-1 s_setprio policy of the GELU epilogues, bit 2 = LDS transposition instead of the lane exchange, bits 3 / 4 = STATS ablations (no accumulation / no final fold; gemm256.h)
#else
    constexpr int dbg = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int G = gridDim.x;
    const int pos = xcd_remap(blockIdx.x, G);
    const int my_tiles = (ntiles - pos + G - 1) / G;
    auto tile_of = [&](int ti) { return pos + ti * G; };
    const int nk = K / P_BK;
    const int total_kt = my_tiles * nk;
    // epilogue arguments as plain scalars (a by-reference struct ends up on the stack)
    const float e_alpha = e.alpha;
    const float* e_bias = e.bias;
    const float* e_residual = e.residual;
    const long long e_ldr = e.ldr;
    const bf16_t* e_aux = (const bf16_t*)e.aux;
    bf16_t* e_out2 = (bf16_t*)e.out2;
    const int e_accumulate = e.accumulate;
    float* e_colstats = e.colstats;

    // ------------------------------------------------------------------ LDS-DMA stream state
    // A half-tile is 16 pieces of 1 KiB (8 rows x 128 B, one wave instruction each).  ROLE 0: wave w owns
    // pieces 2w, 2w + 1; ROLE 1: loader wave (wr = 1, wc) owns pieces 4 wc .. 4 wc + 3, the wr = 0 waves none.
    // vo[J][i]: this lane's source byte offset for piece i of half-tile J (0 B_h0, 1 A_h0, 2 B_h1, 3 A_h1);
    // every index below is a compile-time constant after unrolling (a runtime-indexed array would live in
    // scratch, and a scratch reload is a VMEM load whose compiler-inserted wait would drain the stream).
    constexpr int NP = ROLE ? 4 : 2;
    const int pbase = ROLE ? wc * 4 : wave * 2;
    const bool loader = ROLE == 0 || wr == 1;
    // Pieces i and i + 2 of a loader are 16 rows apart and share their swizzled chunk (p_swz(lr) depends on
    // bits 1..3 of the row only), so only pieces 0 and 1 keep per-lane state: vo[J][i] = byte offset of the
    // lane's (clamped) ROW, co[i] = byte offset of its 16-B chunk inside the 128-B K-slice; pieces 2 and 3 are
    // min(vo + 16 rows, last row) + co.  (Four offsets per half-tile cost 8 more VGPRs and spill.)
    unsigned vo[4][2], co[2];
    // CONV (ROLE 0 only): vo[1][*] / vo[3][*] hold the (signed) byte offset of tap (0, 0) of the lane's output
    // pixel, chunk included, and yo[0][*] / yo[1][*] its coordinates (y0 << 16 | x0 & 0xffff)
    unsigned yo[2][2];
    static_assert(CONV == 0 || ROLE == 0, "the gathered operand keeps the two-piece ownership");
    static_assert(NJ == 4 || ((NJ == 3 || NJ == 2) && ROLE == 0 && CONV == 0), "256 x 192 / 256 x 128: plain operands, ROLE 0");
    static_assert(!TAB || (NJ == 4 && ROLE == 0 && CONV == 0 && !STATS && sizeof(TC) == 2 &&
                           (MODE == SSL4GIE_EPI_BIAS_GELU || MODE == SSL4GIE_EPI_BIAS_GELU_GRAD)),
                  "the table form exists for the GELU pair on the 256 x 256 tile");
    constexpr int WN = 16 * NJ;   // columns per wave
    constexpr int BN = 4 * WN;    // columns per tile
#pragma unroll
    for (int J = 0; J < 4; ++J) vo[J][0] = vo[J][1] = 0;
    yo[0][0] = yo[0][1] = yo[1][0] = yo[1][1] = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lr = (pbase + i) * 8 + (lane >> 3);
        co[i] = (unsigned)(((lane & 7) ^ p_swz(lr)) * 16);  // ROLE 0: folded into vo (one add less per issue)
    }
    const unsigned a_step = (unsigned)(16 * lda * 2), b_step = (unsigned)(16 * ldb * 2);
    const unsigned a_last = (unsigned)((long long)(M - 1) * lda * 2), b_last = (unsigned)((long long)(N - 1) * ldb * 2);
    const int c_H = cg.H, c_W = cg.W, c_C = cg.C, c_Wo = cg.Wo, c_HoWo = cg.HoWo, c_s = cg.stride;
    const unsigned c_mgw = cg.mg_wo, c_shw = cg.sh_wo, c_mgh = cg.mg_hw, c_shh = cg.sh_hw;
    const char* c_zero = (const char*)cg.zero;
    const int c_cpt = c_C / P_BK;  // K-tiles per tap
    auto point_at = [&](int ti) {
        const int tile = tile_of(ti);
        const int sm0 = (tile / tiles_n) * P_BM, sn0 = (tile % tiles_n) * BN;
        auto offs = [&](int i, int h, bool is_a) -> unsigned {
            if (NJ == 3 && !is_a && h == 1) {  // 64-row half-tile: piece `wave`, columns 48 (lr >> 4) + 32 + (lr & 15)
                const int lr1 = wave * 8 + (lane >> 3);
                int rb = sn0 + (lr1 >> 4) * WN + 32 + (lr1 & 15);
                rb = rb < N ? rb : N - 1;
                return (unsigned)((long long)rb * ldb * 2) + (unsigned)(((lane & 7) ^ p_swz(lr1)) * 16);
            }
            const int lr = (pbase + i) * 8 + (lane >> 3);  // local row of the half-tile
            if (is_a) {
                int ra = sm0 + (lr >> 6) * 128 + h * 64 + (lr & 63);
                ra = ra < M ? ra : M - 1;
                if constexpr (CONV != 0) return (unsigned)ra;  // decomposed below
                return (unsigned)((long long)ra * lda * 2) + (ROLE == 0 ? co[i] : 0u);
            }
            int rb = sn0 + (lr >> 5) * WN + h * 32 + (lr & 31);
            rb = rb < N ? rb : N - 1;
            return (unsigned)((long long)rb * ldb * 2) + (ROLE == 0 ? co[i] : 0u);
        };
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            vo[0][i] = offs(i, 0, false);  // B_h0
            vo[1][i] = offs(i, 0, true);   // A_h0
            vo[2][i] = offs(i, 1, false);  // B_h1
            vo[3][i] = offs(i, 1, true);   // A_h1
        }
        if constexpr (CONV != 0) {
            auto pix = [&](unsigned& v, unsigned& yx, int i) {
                const int lr = (pbase + i) * 8 + (lane >> 3);
                const int c = (lane & 7) ^ p_swz(lr);
                const unsigned m = v;
                const unsigned b = p_fastdiv(m, c_mgh, c_shh);
                const unsigned r = m - b * (unsigned)c_HoWo;
                const unsigned oy = p_fastdiv(r, c_mgw, c_shw);
                const unsigned ox = r - oy * (unsigned)c_Wo;
                const int y0 = (int)oy * c_s - 1, x0 = (int)ox * c_s - 1;
                v = (unsigned)(((((int)b * c_H + y0) * c_W + x0) * c_C + c * 8) * 2);
                yx = ((unsigned)y0 << 16) | ((unsigned)x0 & 0xffffu);
            };
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                pix(vo[1][i], yo[0][i], i);
                pix(vo[3][i], yo[1][i], i);
            }
        }
    };
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(p_lds_addr(smem) + pbase * 1024);
    int s_ktg = 0, s_kt = 0, s_ti = 0;  // stream cursor: global K-tile, K-tile in tile, tile
    int s_dy = 0, s_dx = 0, s_cib = 0;  // CONV: tap and 64-channel block of the cursor's K-tile
    // issue half-tile J (0 B_h0, 1 A_h0, 2 B_h1, 3 A_h1) of the stream's current K-tile
    auto issue = [&](auto Jc) {
        constexpr int J = decltype(Jc)::value;
        if (!loader) return;
        if (s_ktg < total_kt && dbg != 7) {
            const unsigned dst = lds0 + (s_ktg & 1) * P_BUF + J * P_HALF;
            const bf16_t* base = ((J & 1) ? A : B) + (size_t)(dbg == 11 ? 0 : s_kt) * P_BK;  // (debug library, 11: every K-tile re-reads K-tile 0 — an L2-resident operand stream, timing only)
            if constexpr (CONV != 0 && (J & 1)) {
                const unsigned va = vo[J][0], vb = vo[J][1];
                const unsigned ya = yo[J >> 1][0], yb = yo[J >> 1][1];
                const int tapoff = ((s_dy * c_W + s_dx) * c_C + s_cib * P_BK) * 2;
                auto src = [&](unsigned o, unsigned yx) -> const char* {
                    const int y = ((int)yx >> 16) + s_dy, x = (int)(short)(yx & 0xffffu) + s_dx;
                    const bool ok = (unsigned)y < (unsigned)c_H && (unsigned)x < (unsigned)c_W;
                    return ok ? (const char*)A + (long long)((int)o + tapoff) : c_zero;
                };
                p_glds2v(src(va, ya), src(vb, yb), dst, dst + 1024);
            } else {
                if constexpr (NJ == 2 && J == 2) {
                    // 256 x 128 tile: the wave's 32 columns are all in B_h0 — no second B half-tile
                } else if constexpr (NJ == 3 && J == 2) {
                    p_glds1(base, vo[J][0], dst - pbase * 1024 + wave * 1024);
                } else if constexpr (ROLE == 0) {
                    p_glds2(base, vo[J][0], vo[J][1], dst, dst + 1024);
                } else {
                    p_glds2(base, vo[J][0] + co[0], vo[J][1] + co[1], dst, dst + 1024);
                    const unsigned step = (J & 1) ? a_step : b_step, last = (J & 1) ? a_last : b_last;
                    const unsigned r2 = vo[J][0] + step, r3 = vo[J][1] + step;
                    p_glds2(base, (r2 < last ? r2 : last) + co[0], (r3 < last ? r3 : last) + co[1], dst + 2048,
                            dst + 3072);
                }
            }
        }
        if (J == 3) {
            ++s_ktg;
            if constexpr (CONV != 0) {
                if (++s_cib == c_cpt) {
                    s_cib = 0;
                    if (++s_dx == 3) { s_dx = 0; ++s_dy; }
                }
            }
            if (++s_kt == nk) {
                s_kt = 0;
                s_dy = 0; s_dx = 0; s_cib = 0;
                if (++s_ti < my_tiles) point_at(s_ti);
            }
        }
    };
    // the K-loop's counted wait: K-tile T + 1 has landed, three half-tiles of K-tile T + 2 may stay in flight
    auto stream_wait = [&](bool more) {
        if (!loader) return;
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NJ == 3 ? 5 : (NJ == 2 ? 4 : 3 * NP)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // ------------------------------------------------------------------ fragment reads
    const int l15 = lane & 15, lg = lane >> 4;
    // lane-dependent parts of the LDS addresses (the swizzle only depends on l15: tile-local rows
    // are l15 + multiples of 16)
    const int sw = p_swz(l15);
    const int offA = (wr * 64 + l15) * 128, offB = (wc * 32 + l15) * 128;
    const int offB1 = NJ == 3 ? (wc * 16 + l15) * 128 : offB;  // rows of the wave's columns in B_h1
    const int ch0 = ((0 * 4 + lg) ^ sw) << 4, ch1 = ((1 * 4 + lg) ^ sw) << 4;
    auto ldA = [&](int buf, int h, int mi, int ks) -> bf16x8 {
        return *(const bf16x8*)(smem + buf * P_BUF + (h ? 3 : 1) * P_HALF + offA + mi * 2048 +
                                (ks ? ch1 : ch0));
    };
    auto ldB = [&](int buf, int h, int ni, int ks) -> bf16x8 {
        return *(const bf16x8*)(smem + buf * P_BUF + (h ? 2 : 0) * P_HALF + (h ? offB1 : offB) + ni * 2048 +
                                (ks ? ch1 : ch0));
    };

    f32x4 acc[8][4];  // NJ == 3: column 3 is never touched
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    bf16x8 a[4][2], b0[2][2], b1[2][2];
    constexpr int NI1 = NJ == 3 ? 1 : (NJ == 2 ? 0 : 2);  // 16-column blocks in the second B half-tile

    // row blocks [MI0, MI1) of quadrant (QM, QN): 16-row x 16-column MFMA tiles over both K-steps of the K-tile
    auto mm = [&](auto QMc, auto QNc, auto MI0c, auto MI1c, bf16x8 (&bb)[2][2]) {
        constexpr int QM = decltype(QMc)::value, QN = decltype(QNc)::value;
        constexpr int MI0 = decltype(MI0c)::value, MI1 = decltype(MI1c)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = MI0; mi < MI1; ++mi)
#pragma unroll
                for (int ni = 0; ni < (QN == 1 ? NI1 : 2); ++ni)
                    acc[QM * 4 + mi][QN * 2 + ni] =
                        P_MFMA(bb[ni][ks], a[mi][ks], acc[QM * 4 + mi][QN * 2 + ni]);
    };
    using I4 = std::integral_constant<int, 4>;
    // MMA segment of phase PH.  NJ == 4: one quadrant of 16 MFMAs per phase, P0 (0,0), P1 (0,1), P2 (1,1), P3 (1,0).
    // NJ == 3: the quadrants (.,1) have one column block (8 MFMAs); the 48 MFMAs of a K-tile are dealt 12 per
    // phase so that the two wave rows still trade equal MMA / LOAD intervals: P0 (0,0) rows 0-2; P1 (0,0) row 3 +
    // (0,1); P2 (1,1) + (1,0) row 0; P3 (1,0) rows 1-3 — every fragment is in registers where it is used (b0 lives
    // from P0 to P3, `a` holds A_h0 until P2's reads replace it with A_h1).
    auto mma = [&](auto PHc) {
        constexpr int PH = decltype(PHc)::value;
        if (dbg == 7) return;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (CONV == 2 && (PH == 0 || PH == 2)) {  // P0 / P2 have just loaded `a`
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) a[mi][ks] = p_relu8(a[mi][ks]);
        }
        if constexpr (NJ == 2) {  // quadrants (0,0) and (1,0) only: 16 MFMAs in P0 and in P3 (Pa and Pb of the two-phase form)
            if constexpr (PH == 0) mm(I0{}, I0{}, I0{}, I4{}, b0);
            if constexpr (PH == 3) mm(I1{}, I0{}, I0{}, I4{}, b0);
        } else if constexpr (NJ == 4) {
            if constexpr (PH == 0) mm(I0{}, I0{}, I0{}, I4{}, b0);
            if constexpr (PH == 1) mm(I0{}, I1{}, I0{}, I4{}, b1);
            if constexpr (PH == 2) mm(I1{}, I1{}, I0{}, I4{}, b1);
            if constexpr (PH == 3) mm(I1{}, I0{}, I0{}, I4{}, b0);
        } else {
            if constexpr (PH == 0) mm(I0{}, I0{}, I0{}, I3{}, b0);
            if constexpr (PH == 1) { mm(I0{}, I0{}, I3{}, I4{}, b0); mm(I0{}, I1{}, I0{}, I4{}, b1); }
            if constexpr (PH == 2) { mm(I1{}, I1{}, I0{}, I4{}, b1); mm(I1{}, I0{}, I0{}, I1{}, b0); }
            if constexpr (PH == 3) mm(I1{}, I0{}, I1{}, I4{}, b0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // ------------------------------------------------------------------ prologue
    if constexpr (TAB) {  // 16 pieces of 1 KiB, two per wave, older than every half-tile: the prologue's wait retires them
        const unsigned tl = __builtin_amdgcn_readfirstlane(p_lds_addr(smem) + 2 * P_BUF + wave * 2048);
        p_glds2(g_gelu_tab, (unsigned)(wave * 2048 + lane * 16), (unsigned)(wave * 2048 + 1024 + lane * 16), tl, tl + 1024);
    }
    point_at(0);
    issue(I0{}); issue(I1{}); issue(I2{}); issue(I3{});
    issue(I0{}); issue(I1{}); issue(I2{});
    stream_wait(total_kt >= 2);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave row by one barrier

#ifdef SSL4GIE_DEBUG_KNOBS
    const bool stamping = ((dbg == 4 && wave == 0) || (dbg == 8 && wave == 4)) && lane == 0 && (blockIdx.x & 15) == 0 && (blockIdx.x >> 4) < NT256_STAMP_WGS;
    auto stamp = [&](int ti, int k) {
        if (stamping && ti < NT256_STAMP_TILES) g_nt256_stamps[blockIdx.x >> 4][ti][k] = __builtin_amdgcn_s_memrealtime();
    };
#else
    auto stamp = [](int, int) {};
#endif
    // The tile's bias row (this wave's 64 columns, 256 B) goes into the wave's staging area by ONE LDS-DMA issued
    // in the tile's first K-tile, in front of that K-tile's half-tile issues: the counted wait of the same
    // K-tile (which leaves only the three youngest half-tiles in flight) retires it, the staging area is idle
    // during the K-loop, and the epilogue reads it with four ds_reads (p_epilogue<.., BIAS_LDS>).  With ROLE 1
    // the wr = 0 waves never wait on vmcnt in the K-loop, so they wait for this one DMA at the epilogue instead.
    constexpr bool HAS_BIAS = MODE == SSL4GIE_EPI_BIAS || MODE == SSL4GIE_EPI_BIAS_GELU ||
                              MODE == SSL4GIE_EPI_BIAS_RESIDUAL || MODE == SSL4GIE_EPI_BIAS_GELU_GRAD;
    const unsigned stg_lds = __builtin_amdgcn_readfirstlane(p_lds_addr(smem) + 2 * P_BUF +
                                                            (TAB ? P_TAB_BYTES + wave * 256 : wave * P_STG_WAVE));
    auto issue_bias = [&](int ti) {
        if constexpr (HAS_BIAS) {
            if (e_bias) {
                int col = (tile_of(ti) % tiles_n) * BN + wc * WN + lane;
                col = col < N ? col : N - 1;
                p_glds1_dword(e_bias, (unsigned)col * 4u, stg_lds);
            }
        }
    };
#ifdef SSL4GIE_DEBUG_KNOBS
    // dbg 9 / 10 (timing only; feasibility of a row-interleaved schedule): the wr = 1 waves keep their LDS-DMA and
    // barrier duties but neither read fragments nor issue MFMAs (9), or run a GELU-sized VALU chunk (4 elements
    // per lane) in every MMA segment instead (10) — what does the other row's K-loop cost beside that?
    if ((dbg == 9 || dbg == 10) && wr == 1) {
        float fake[4] = {0.01f * lane, 0.02f * lane - 0.4f, 0.3f, -0.01f * lane};
        auto chunk = [&]() {
            if (dbg == 10) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float cdf, xpdf;
                    gelu_parts_fast(fake[q], cdf, xpdf);
                    fake[q] = fake[q] * cdf * 0.5f + (cdf + xpdf) * 0.25f - 0.3f;
                }
            }
        };
        for (int T = 0; T < total_kt; ++T) {
            issue(I3{}); __builtin_amdgcn_s_barrier(); chunk(); __builtin_amdgcn_s_barrier();
            issue(I0{}); __builtin_amdgcn_s_barrier(); chunk(); __builtin_amdgcn_s_barrier();
            issue(I1{}); __builtin_amdgcn_s_barrier(); chunk(); __builtin_amdgcn_s_barrier();
            issue(I2{}); stream_wait(T + 2 < total_kt); __builtin_amdgcn_s_barrier(); chunk();
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("" ::"v"(fake[0] + fake[1] + fake[2] + fake[3]));
        return;
    }
#endif
    int c_kt = 0, c_ti = 0;
    for (int T = 0; T < total_kt; ++T) {
        const int cb = T & 1;
        if (c_kt == 0) {
            stamp(c_ti, 0);
            if (dbg != 7) issue_bias(c_ti);
        }
        if constexpr (PH2) {
            // Two phases per K-tile (32 MFMAs per MMA segment, 4 barriers per K-tile instead of 8): the interval
            // is set by the MMA segment plus a fixed ~100 cycles of barrier / wait / issue overhead, so halving
            // the number of intervals per K-tile removes half of that overhead.  Register-neutral: Pa holds
            // A_h0 + B_h0 + B_h1 (quadrants (0,0), (0,1)), Pb replaces A_h0 by A_h1 (quadrants (1,1), (1,0)).
            // Hazards (interval I_n between barriers n-1 and n; the wr = 0 group runs LOAD(p) in I_2p, MMA(p) in
            // I_2p+1 with p = 2 T + {0: Pa, 1: Pb}; the wr = 1 group one interval later):
            //   WAR  every LOAD segment ends with s_waitcnt lgkmcnt(0) BEFORE its barrier, so a group's fragment
            //        reads are complete when the next interval starts.  Pb issues B_h0, A_h0, B_h1 of K-tile
            //        T + 2 over the slots read in Pa (wr = 0: two intervals earlier; wr = 1: one interval earlier,
            //        complete at that interval's barrier); Pa issues A_h1 of K-tile T + 1 over the slot read in
            //        Pb of K-tile T - 1 (same distances).
            //   RAW  one counted wait per K-tile at the end of LOAD(Pb): all but the three half-tiles just issued
            //        (K-tile T + 2) have landed, i.e. the whole of K-tile T + 1; it is read from LOAD(Pa(T + 1)),
            //        two barriers later for the waiting group and at least one barrier later for the other one.
            // ---------------- Pa
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) b0[ni][ks] = ldB(cb, 0, ni, ks);
#pragma unroll
            for (int ni = 0; ni < NI1; ++ni)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) b1[ni][ks] = ldB(cb, 1, ni, ks);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) a[mi][ks] = ldA(cb, 0, mi, ks);
            __builtin_amdgcn_sched_barrier(0);
            issue(I3{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            mma(I0{});
            mma(I1{});
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- Pb
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) a[mi][ks] = ldA(cb, 1, mi, ks);
            __builtin_amdgcn_sched_barrier(0);
            issue(I0{});
            issue(I1{});
            issue(I2{});
            stream_wait(T + 2 < total_kt);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            mma(I2{});
            mma(I3{});
            __builtin_amdgcn_sched_barrier(0);
        } else {
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
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // B_h0 reads retired (WAR, see header)
            __builtin_amdgcn_s_barrier();
            mma(I0{});
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- P1
    #pragma unroll
            for (int ni = 0; ni < NI1; ++ni)
    #pragma unroll
                for (int ks = 0; ks < 2; ++ks) b1[ni][ks] = ldB(cb, 1, ni, ks);
            __builtin_amdgcn_sched_barrier(0);
            issue(I0{});
            __builtin_amdgcn_s_barrier();
            mma(I1{});
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
            mma(I2{});
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- P3
            issue(I2{});
            // K-tile T+1 must have landed before the next phase reads it; the three half-tiles issued
            // in P1..P3 (of K-tile T+2) may stay in flight
            stream_wait(T + 2 < total_kt);
            __builtin_amdgcn_s_barrier();
            mma(I3{});
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c_ti > 0 && c_kt == 0) stamp(c_ti - 1, 3);
        if (c_ti > 0 && c_kt == 1) stamp(c_ti - 1, 4);
        if (++c_kt == nk) {
            // Output tile finished.  The wr=0 group ends its MMA interval first and runs its
            // epilogue in the next one, where the wr=1 group (one barrier behind) runs its own: the
            // two epilogues overlap instead of costing two MFMA-idle intervals per tile.
            c_kt = 0;
            const int tile = tile_of(c_ti);
            ++c_ti;
            const int m0 = (tile / tiles_n) * P_BM, n0 = (tile % tiles_n) * BN;
            if (wr == 0) __builtin_amdgcn_s_barrier();
            if constexpr (ROLE != 0 && HAS_BIAS) {
                if (!loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (its own older stores, and the bias row)
            }
            stamp(c_ti - 1, 1);
            char* stg = smem + 2 * P_BUF + (TAB ? P_TAB_BYTES + wave * 256 : wave * P_STG_WAVE);
            const bool full = m0 + P_BM <= M && n0 + BN <= N;
#ifdef SSL4GIE_DEBUG_KNOBS
            // one instantiation per FULL; the ablation modes only change its arguments
            const float* x_res = e_residual;
            const bf16_t* x_aux = e_aux;
            bf16_t* x_out2 = e_out2;
            TC* x_C = C;
            long long x_ldr = e_ldr;
            int x_rbase = m0 + wr * 128, x_M = M, reps = 1;
            bool x_full = full, skip = false;
            if (dbg == 1 || dbg == 9 || dbg == 10 || ((dbg == 5 || dbg == 6) && wr == 1)) skip = true;
            if (dbg == 2) { x_full = false; x_M = 0; }
            if (dbg == 3) {
                const size_t row = (size_t)(pos % (M > 0 ? M : 1));
                if (x_res) x_res += row * e_ldr;
                if (x_aux) x_aux += row * ldc;
                if (x_out2) x_out2 += row * ldc;
                x_C += row * ldc;
                x_ldr = 0; x_rbase = 0; x_full = true;
            }
            if (dbg == 6 && full) reps = 2;  // the partner's rows too (its values are not these: timing only)
            if (skip) {
                asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[7][NJ - 1][3]));
            } else if (x_full) {
                for (int rep = 0; rep < reps; ++rep)
                    p_epilogue<TC, MODE, true, STATS, true, NJ>(acc, stg, e_alpha, e_bias, x_res, x_ldr, x_aux, x_out2,
                                                      e_accumulate, x_C, dbg == 3 ? 0 : ldc, x_rbase + rep * 128,
                                                      n0 + wc * WN, x_M, N, lane, e_colstats,
                                                      (stamping && c_ti - 1 < NT256_STAMP_TILES)
                                                          ? &g_nt256_stamps[blockIdx.x >> 4][c_ti - 1][5] : nullptr,
                                                      dbg_prio & 59, !(dbg_prio & 4));
            } else {
                p_epilogue<TC, MODE, false, STATS, true, NJ>(acc, stg, e_alpha, e_bias, x_res, x_ldr, x_aux, x_out2,
                                                   e_accumulate, x_C, ldc, x_rbase, n0 + wc * WN, x_M, N, lane,
                                                   e_colstats, nullptr, dbg_prio & 59, !(dbg_prio & 4));
            }
#else
            if (full)
                p_epilogue<TC, MODE, true, STATS, true, NJ, NTS, TAB>(acc, stg, e_alpha, e_bias, e_residual, e_ldr, e_aux,
                                                  e_out2, e_accumulate, C, ldc, m0 + wr * 128,
                                                  n0 + wc * WN, M, N, lane, e_colstats, nullptr, P_EPI_PRIO_MODE,
                                                  P_EPI_XPOSE_SWAP, smem + 2 * P_BUF);
            else
                p_epilogue<TC, MODE, false, STATS, true, NJ, NTS, TAB>(acc, stg, e_alpha, e_bias, e_residual, e_ldr, e_aux,
                                                   e_out2, e_accumulate, C, ldc, m0 + wr * 128,
                                                   n0 + wc * WN, M, N, lane, e_colstats, nullptr, P_EPI_PRIO_MODE,
                                                   P_EPI_XPOSE_SWAP, smem + 2 * P_BUF);
#endif
            stamp(c_ti - 1, 2);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
            if (wr == 1) __builtin_amdgcn_s_barrier();
        } else {
            __builtin_amdgcn_s_barrier();
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // every wave executes the same number of barriers
}

// =====================================================================================
// host side
// =====================================================================================
static int nt256_mode() {  // SSL4GIE_NT256: "0" never, "1" whenever possible, unset = heuristic
    static int v = -2;
    if (v == -2) {
        const char* s = getenv("SSL4GIE_NT256");
        v = !s ? -1 : (s[0] == '0' ? 0 : 1);
    }
    return v;
}

bool ssl4gie_internal_nt256_ok(const ssl4gie_gemm_desc* d) {
    const int mode = nt256_mode();
    if (mode == 0 && !d->conv) return false;
    // 32-bit per-lane byte offsets; whole K-tiles; 16-byte row segments on the output
    if ((!d->conv && (long long)d->M * d->sAm * 2 >= (1LL << 32)) ||
        (long long)d->N * d->sBn * 2 >= (1LL << 32))
        return false;
    if (d->K % P_BK != 0 || d->N % 8 != 0 || d->ldc % 8 != 0) return false;
    const int ep = d->epilogue;
    if (d->colstats && !(d->dtype_c == SSL4GIE_BF16 && ep == SSL4GIE_EPI_NONE && !d->accumulate &&
                         d->N % 8 == 0))
        return false;
    if (!d->C && !d->colstats) return false;  // C == NULL: the statistics-only product
    if (ep == SSL4GIE_EPI_AFFINE_AUX_RELU)    // exists in this kernel only
        return !d->conv && d->dtype_c == SSL4GIE_BF16 && d->scale && d->bias && !d->accumulate && d->C;
    if (d->conv)  // gathered A: bf16 outputs with bias / plain only, whole K-tiles inside a tap
        return ssl4gie_internal_conv_geom_ok(d->conv) && d->conv->C % P_BK == 0 &&
               d->K == 9 * d->conv->C && (long long)d->M == conv_rows(d->conv) &&
               d->dtype_c == SSL4GIE_BF16 && !d->accumulate &&
               (ep == SSL4GIE_EPI_BIAS || ep == SSL4GIE_EPI_NONE ||
                (ep == SSL4GIE_EPI_RELU_MASK_AUX && d->aux && !d->conv->relu));
    if (d->dtype_c == SSL4GIE_BF16) {
        if (ep == SSL4GIE_EPI_BIAS_RESIDUAL || d->accumulate) return false;
    } else {
        if (ep == SSL4GIE_EPI_BIAS_GELU || ep == SSL4GIE_EPI_DGELU ||
            ep == SSL4GIE_EPI_BIAS_GELU_GRAD || ep == SSL4GIE_EPI_MUL_AUX || ep == SSL4GIE_EPI_RELU_MASK_AUX ||
            ep == SSL4GIE_EPI_ADD_AUX)
            return false;
    }
    if (ep == SSL4GIE_EPI_ADD_AUX) return d->aux != nullptr;  // exists in this kernel only
    if (mode == 1 || d->colstats) return true;  // the statistics only exist in this kernel
    // heuristic: enough 256x256 tiles to fill most of the chip
    const long long tiles = (long long)((d->M + P_BM - 1) / P_BM) * ((d->N + P_BN - 1) / P_BN);
    return tiles >= 128;
}

// Columns per tile, per shape: 256, or 192 (NJ = 3) when that means less work per CU — the tiles of a launch
// are dealt to `cus` persistent workgroups, so the launch takes ceil(tiles / cus) tile times, and a 192-wide
// tile costs about 0.78 of a 256-wide one (3/4 of the MFMAs and of the epilogue, shorter phases pipeline a
// little worse; tools/gemm_bench.py).  N = 768 at M = 12800 (ViT-B proj / fc2 and their data gradients):
// 150 tiles = one round at 62 % of the chip  ->  200 tiles = one round of 0.78.
// Round 5: a 256 x 128 tile (NJ = 2) for the narrow products of ResNet50 (N = 64 / 128: layer1 / layer2 conv1 and the
// data gradients into them): one column tile either way, but half (N = 128) or two thirds (N = 64 on the 192-wide
// tile) of the MFMAs of an HBM-streaming product were spent on columns that do not exist; exists for the epilogue
// kinds those products use (plain / bias, + statistics, + aux add, affine + aux + ReLU).
static bool nt256_nj2_mode(const ssl4gie_gemm_desc* d) {
#ifdef SSL4GIE_DEBUG_KNOBS
    return false;
#else
    static int on = -1;  // SSL4GIE_NT256_NJ2=0: never (A/B timing; same results)
    if (on < 0) { const char* s = getenv("SSL4GIE_NT256_NJ2"); on = (s && s[0] == '0') ? 0 : 1; }
    if (!on || d->conv || d->dtype_c != SSL4GIE_BF16) return false;
    const int ep = d->epilogue;
    return ep == SSL4GIE_EPI_NONE || ep == SSL4GIE_EPI_BIAS || ep == SSL4GIE_EPI_ADD_AUX || ep == SSL4GIE_EPI_AFFINE_AUX_RELU;
#endif
}
static int nt256_pick_nj(const ssl4gie_gemm_desc* d, int cus) {
    if (d->conv) return 4;
#ifdef SSL4GIE_DEBUG_KNOBS
    static int forced = -1;
    if (forced < 0) { const char* s = getenv("SSL4GIE_NT256_NJ"); forced = s ? atoi(s) : 0; }
    if (forced == 3 || forced == 4) return forced;
    static int stats4 = -1;  // SSL4GIE_NT256_STATS_NJ4=1: products with column statistics on 256-wide tiles only (A/B)
    if (stats4 < 0) { const char* s = getenv("SSL4GIE_NT256_STATS_NJ4"); stats4 = s ? atoi(s) : 0; }
    if (stats4 && d->colstats) return 4;
#endif
    const long long tm = (d->M + P_BM - 1) / P_BM;
    const long long r256 = (tm * ((d->N + 255) / 256) + cus - 1) / cus;
    const long long r192 = (tm * ((d->N + 191) / 192) + cus - 1) / cus;
    int nj = (double)r192 * 0.78 < (double)r256 ? 3 : 4;
    if (d->N <= 128 && nt256_nj2_mode(d)) nj = 2;  // one column tile whatever the width: the narrowest that holds N
    return nj;
}

int ssl4gie_internal_nt256_launch(const ssl4gie_gemm_desc* d, hipStream_t st) {
    const int cus = ssl4gie_internal_compute_cus();
    const int nj = nt256_pick_nj(d, cus);
    const int tm = (d->M + P_BM - 1) / P_BM, tn = (d->N + 64 * nj - 1) / (64 * nj);
    const int ntiles = tm * tn;
    // the launch takes ceil(ntiles / cus) rounds of tiles whatever happens: with that many rounds, the FEWEST
    // persistent workgroups that still finish in them (600 tiles: 3 rounds on 200 CUs instead of 240) leave the
    // other CUs to whatever runs beside the launch — the weight-gradient stream in the backward pass
    static int balance = -1;
    if (balance < 0) { const char* e = getenv("SSL4GIE_NT_BALANCE"); balance = e ? atoi(e) : 1; }
    int wgs = ntiles < cus ? ntiles : cus;
    if (balance && ntiles > cus) {
        const int rounds = (ntiles + cus - 1) / cus;
        wgs = (ntiles + rounds - 1) / rounds;
    }
    dim3 grid(wgs), block(512);
    // non-temporal output stores for the long products (SSL4GIE_NT_STREAM_M = least M, default 16384; 0 always,
    // -1 never): see gemm256.h est
    static long long stream_m = -2;
    if (stream_m == -2) { const char* s = getenv("SSL4GIE_NT_STREAM_M"); stream_m = s ? atoll(s) : 16384; }
    EpiArgs e{d->alpha, d->epilogue, d->bias, d->residual, d->ldr, d->aux, d->out2, d->accumulate,
              d->colstats, (stream_m >= 0 && d->M >= stream_m) ? 1 : 0};
    if (d->epilogue == SSL4GIE_EPI_AFFINE_AUX_RELU) {  // two borrowed slots (gemm256.h p_epilogue)
        e.residual = d->scale;
        e.accumulate = d->relu;
    }
    int skip_epi = 0;
    [[maybe_unused]] int role = NT256_DEFAULT_ROLE;
    [[maybe_unused]] bool ph2 = NT256_DEFAULT_PH2;
#ifdef SSL4GIE_DEBUG_KNOBS
    // debug library only (make DEBUG_KNOBS=1 -> libssl4gie_hip_dbg.so, loaded with SSL4GIE_DEBUG_LIB=1):
    // SSL4GIE_NT256_NOEPI=1..7 ablations (the kernel's `dbg`; all but 4 leave garbage outputs),
    // SSL4GIE_NT256_ROLE=0/1 the LDS-DMA ownership
    {
        static int k_epi = -1, k_role = -1, k_ph2 = 0;
        if (k_epi < 0) {
            const char* s = getenv("SSL4GIE_NT256_NOEPI");
            k_epi = s ? (atoi(s) & 15) : 0;
            s = getenv("SSL4GIE_NT256_ROLE");
            k_role = s ? (s[0] - '0') : NT256_DEFAULT_ROLE;
            if (k_epi || s) fprintf(stderr, "ssl4gie: DEBUG KNOBS active: NT256_NOEPI=%d NT256_ROLE=%d\n", k_epi, k_role);
            s = getenv("SSL4GIE_NT256_PH2");
            k_ph2 = s ? atoi(s) : (NT256_DEFAULT_PH2 ? 1 : 0);
            s = getenv("SSL4GIE_NT256_PRIO");
            k_epi |= (s ? atoi(s) : P_EPI_PRIO_MODE) << 4;
        }
        skip_epi = k_epi;
        role = k_role;
        ph2 = k_ph2 != 0;
    }
#endif
    static int gelu_tab = -1;
    if (gelu_tab < 0) { const char* s = getenv("SSL4GIE_GELU_TABLE"); gelu_tab = (s && s[0] == '0') ? 0 : 1; }
    ConvK ck{};
    if (d->conv) {
        const int rc = ssl4gie_internal_conv_k(d->conv, &ck);
        if (rc) return rc;
    }
    ProfScope prof(PROF_GEMM_NT, 2.0 * d->M * d->N * d->K, st);
#ifdef SSL4GIE_DEBUG_KNOBS
#define P_LAUNCH(TC_, MODE_)                                                   \
    do {                                                                        \
        if (nj == 3 && ph2) P_LAUNCH_R(TC_, MODE_, 0, false, 0, 3, true);       \
        else if (nj == 3) P_LAUNCH_R(TC_, MODE_, 0, false, 0, 3, false);        \
        else if (role == 1) P_LAUNCH_R(TC_, MODE_, 0, false, 1, 4, false);      \
        else if (ph2) P_LAUNCH_R(TC_, MODE_, 0, false, 0, 4, true);             \
        else P_LAUNCH_R(TC_, MODE_, 0, false, 0, 4, false);                     \
    } while (0)
#else
#define P_LAUNCH(TC_, MODE_)                                                              \
    do {                                                                                   \
        if (nj == 3) P_LAUNCH_R(TC_, MODE_, 0, false, 0, 3, NT256_DEFAULT_PH2);            \
        else P_LAUNCH_R(TC_, MODE_, 0, false, NT256_DEFAULT_ROLE, 4, NT256_DEFAULT_PH2);   \
    } while (0)
#endif
#ifdef SSL4GIE_DEBUG_KNOBS
#define P_LAUNCH2(TC_, MODE_) P_LAUNCH(TC_, MODE_)
#else
#define P_LAUNCH2(TC_, MODE_) /* the epilogue kinds that also exist on the 256 x 128 tile (nt256_nj2_mode) */ \
    do {                                                                                   \
        if (nj == 2) P_LAUNCH_R(TC_, MODE_, 0, false, 0, 2, NT256_DEFAULT_PH2);            \
        else P_LAUNCH(TC_, MODE_);                                                         \
    } while (0)
#endif
#ifdef SSL4GIE_DEBUG_KNOBS
#define P_LAUNCH_G(MODE_) P_LAUNCH(bf16_t, MODE_)
#else  /* the GELU pair on the 256 x 256 tile reads the table (SSL4GIE_GELU_TABLE=0: the polynomial form, A/B timing) */
#define P_LAUNCH_G(MODE_)                                                                                \
    do {                                                                                                  \
        if (nj == 4 && gelu_tab && e.nt_store) P_LAUNCH_KT(bf16_t, MODE_, 0, false, 0, 4, NT256_DEFAULT_PH2, true, true);  \
        else if (nj == 4 && gelu_tab) P_LAUNCH_KT(bf16_t, MODE_, 0, false, 0, 4, NT256_DEFAULT_PH2, false, true);          \
        else P_LAUNCH(bf16_t, MODE_);                                                                     \
    } while (0)
#endif
#define P_LAUNCH_C(TC_, MODE_, CONV_) P_LAUNCH_S(TC_, MODE_, CONV_, false)
#ifdef SSL4GIE_DEBUG_KNOBS
#define P_LAUNCH_S(TC_, MODE_, CONV_, STATS_)                            \
    do {                                                                  \
        if (ph2) P_LAUNCH_R(TC_, MODE_, CONV_, STATS_, 0, 4, true);       \
        else P_LAUNCH_R(TC_, MODE_, CONV_, STATS_, 0, 4, false);          \
    } while (0)
#else
#define P_LAUNCH_S(TC_, MODE_, CONV_, STATS_) P_LAUNCH_R(TC_, MODE_, CONV_, STATS_, 0, 4, NT256_DEFAULT_PH2)
#endif
#ifdef SSL4GIE_DEBUG_KNOBS
#define P_LAUNCH_R(TC_, MODE_, CONV_, STATS_, ROLE_, NJ_, PH2_) P_LAUNCH_K(TC_, MODE_, CONV_, STATS_, ROLE_, NJ_, PH2_, false)
#else  /* the streaming-store twin exists for the plain products only (no patch-matrix operand, no statistics) */
#define P_LAUNCH_R(TC_, MODE_, CONV_, STATS_, ROLE_, NJ_, PH2_)                                        \
    do {                                                                                           \
        if (CONV_ == 0 && !(STATS_) && e.nt_store) P_LAUNCH_K(TC_, MODE_, 0, false, ROLE_, NJ_, PH2_, true); \
        else P_LAUNCH_K(TC_, MODE_, CONV_, STATS_, ROLE_, NJ_, PH2_, false);                       \
    } while (0)
#endif
#define P_LAUNCH_K(TC_, MODE_, CONV_, STATS_, ROLE_, NJ_, PH2_, NTS_) P_LAUNCH_KT(TC_, MODE_, CONV_, STATS_, ROLE_, NJ_, PH2_, NTS_, false)
#define P_LAUNCH_KT(TC_, MODE_, CONV_, STATS_, ROLE_, NJ_, PH2_, NTS_, TAB_)                           \
    do {                                                                                           \
        auto kfn = gemm_bf16_nt256_kernel<TC_, MODE_, CONV_, STATS_, ROLE_, NJ_, PH2_, NTS_, TAB_>; \
        static bool attr_set = false; /* idempotent; a benign race only repeats the call */        \
        if (!attr_set) {                                                                           \
            HIP_RET(hipFuncSetAttribute((const void*)kfn,                                          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BYTES)); \
            attr_set = true;                                                                       \
        }                                                                                          \
        hipLaunchKernelGGL(kfn, grid, block, P_LDS_BYTES, st, (const bf16_t*)d->A, d->sAm,         \
                           (const bf16_t*)d->B, d->sBn, (TC_*)d->C, d->ldc, d->M, d->N, d->K, tn,  \
                           ntiles, e, skip_epi, ck);                                                \
    } while (0)
    if (d->colstats) {  // bf16, plain epilogue (checked by nt256_ok)
        const int cv = d->conv ? (d->conv->relu ? 2 : 1) : 0;
        if (cv == 0 && nj == 3) P_LAUNCH_R(bf16_t, SSL4GIE_EPI_NONE, 0, true, 0, 3, NT256_DEFAULT_PH2);
#ifndef SSL4GIE_DEBUG_KNOBS
        else if (cv == 0 && nj == 2) P_LAUNCH_R(bf16_t, SSL4GIE_EPI_NONE, 0, true, 0, 2, NT256_DEFAULT_PH2);
#endif
        else if (cv == 0) P_LAUNCH_S(bf16_t, SSL4GIE_EPI_NONE, 0, true);
        else if (cv == 1) P_LAUNCH_S(bf16_t, SSL4GIE_EPI_NONE, 1, true);
        else P_LAUNCH_S(bf16_t, SSL4GIE_EPI_NONE, 2, true);
    } else if (d->conv) {
        const bool relu = d->conv->relu != 0, bias = d->epilogue == SSL4GIE_EPI_BIAS;
        if (d->epilogue == SSL4GIE_EPI_RELU_MASK_AUX) P_LAUNCH_C(bf16_t, SSL4GIE_EPI_RELU_MASK_AUX, 1);
        else if (bias && relu) P_LAUNCH_C(bf16_t, SSL4GIE_EPI_BIAS, 2);
        else if (bias) P_LAUNCH_C(bf16_t, SSL4GIE_EPI_BIAS, 1);
        else if (relu) P_LAUNCH_C(bf16_t, SSL4GIE_EPI_NONE, 2);
        else P_LAUNCH_C(bf16_t, SSL4GIE_EPI_NONE, 1);
    } else if (d->dtype_c == SSL4GIE_BF16) {
        switch (d->epilogue) {
            case SSL4GIE_EPI_BIAS: P_LAUNCH2(bf16_t, SSL4GIE_EPI_BIAS); break;
            case SSL4GIE_EPI_BIAS_GELU: P_LAUNCH_G(SSL4GIE_EPI_BIAS_GELU); break;
            case SSL4GIE_EPI_DGELU: P_LAUNCH(bf16_t, SSL4GIE_EPI_DGELU); break;
            case SSL4GIE_EPI_BIAS_GELU_GRAD: P_LAUNCH_G(SSL4GIE_EPI_BIAS_GELU_GRAD); break;
            case SSL4GIE_EPI_MUL_AUX: P_LAUNCH(bf16_t, SSL4GIE_EPI_MUL_AUX); break;
            case SSL4GIE_EPI_ADD_AUX: P_LAUNCH2(bf16_t, SSL4GIE_EPI_ADD_AUX); break;
            case SSL4GIE_EPI_AFFINE_AUX_RELU: P_LAUNCH2(bf16_t, SSL4GIE_EPI_AFFINE_AUX_RELU); break;
            case SSL4GIE_EPI_NONE: P_LAUNCH2(bf16_t, SSL4GIE_EPI_NONE); break;
            default: return ARG_ERR;
        }
    } else {
        switch (d->epilogue) {
            case SSL4GIE_EPI_BIAS: P_LAUNCH(float, SSL4GIE_EPI_BIAS); break;
            case SSL4GIE_EPI_BIAS_RESIDUAL: P_LAUNCH(float, SSL4GIE_EPI_BIAS_RESIDUAL); break;
            case SSL4GIE_EPI_NONE: P_LAUNCH(float, SSL4GIE_EPI_NONE); break;
            default: return ARG_ERR;
        }
    }
#undef P_LAUNCH
#undef P_LAUNCH2
#undef P_LAUNCH_C
#undef P_LAUNCH_S
#undef P_LAUNCH_R
#undef P_LAUNCH_G
#undef P_LAUNCH_KT
    LAUNCH_CHECK();
    return 0;
}
