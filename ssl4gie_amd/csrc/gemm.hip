// GEMM kernels for gfx950 with fused epilogues.
//
//  * gemm_bf16_nt_kernel : C[M,N] = A[M,K] * B[N,K]^T  (nn.Linear forward, and the data-gradient
//                          with a pre-transposed weight copy).  128x128x64 tiles, 4 waves (2x2),
//                          each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 accumulators, operands
//                          staged by LDS-DMA (global_load_lds_dwordx4) into a double-buffered,
//                          XOR-swizzled LDS image, one barrier per K-tile.
//  * gemm_bf16_tn_kernel : C[M,N] = At[K,M]^T * Bt[K,N]  (weight gradients dW = dY^T X; the
//                          contraction runs over tokens).  Same tile; fragments come out of the
//                          k-major LDS image with ds_read_b64_tr_b16 (hardware transpose), split-K
//                          across workgroups into fp32 slabs + a deterministic slab reduction.
//  * gemm_generic_kernel : any strides / batches / dtypes on v_mfma_f32_16x16x4_f32 (exact fp32
//                          FMA chains) — the parity path and the fallback.
//
// Operand roles are swapped in the MFMA (a <- weight rows, b <- activation rows) so that every
// lane ends up with 4 *consecutive output columns* of one output row: epilogues (bias, GELU,
// residual add, GELU') and stores are 8/16-byte vector operations.
//
// Replaces the cuBLAS calls behind nn.Linear in timm Block — reference call
// sites Models/mae/models_mae.py:39-41,47,53-55,59 and Models/models.py:171-173.
#include "gemm_internal.h"

#include <mutex>
#include "prof.h"
#include <stdlib.h>

struct GemmArgs {
    int M, N, K, batch2;
    const void* A;
    long long sAm, sAk, sAb1, sAb2;
    const void* B;
    long long sBk, sBn, sBb1, sBb2;
    void* C;
    long long ldc, sCb1, sCb2;
    EpiArgs e;
};

// acc = 4 consecutive columns n..n+3 of row m
template <typename TC, bool VEC, bool FAST = false>
DEVI void epi_store4(const EpiArgs& e, TC* __restrict__ C, long long ldc, int m, int n, int M,
                     int N, f32x4 acc) {
    if (m >= M || n >= N) return;
    const size_t off = (size_t)m * ldc + n;
    if (VEC) {
        f32x4 v = acc * e.alpha;
        switch (e.mode) {
            case SSL4GIE_EPI_BIAS:
                v += ld4(e.bias + n);
                break;
            case SSL4GIE_EPI_BIAS_GELU: {
                if (e.bias) v += ld4(e.bias + n);
                st4(C + off, v);
                f32x4 g;
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = FAST ? gelu_fast(v[j]) : gelu_f(v[j]);
                st4((TC*)e.out2 + off, g);
                return;
            }
            case SSL4GIE_EPI_BIAS_RESIDUAL:
                if (e.bias) v += ld4(e.bias + n);
                v += ld4(e.residual + (size_t)m * e.ldr + n);
                break;
            case SSL4GIE_EPI_DGELU: {
                const f32x4 u = ld4((const TC*)e.aux + off);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= FAST ? dgelu_fast(u[j]) : dgelu_f(u[j]);
                break;
            }
            case SSL4GIE_EPI_BIAS_GELU_GRAD: {
                if (e.bias) v += ld4(e.bias + n);
                f32x4 g, d;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    g[j] = FAST ? gelu_fast(v[j]) : gelu_f(v[j]);
                    d[j] = FAST ? dgelu_fast(v[j]) : dgelu_f(v[j]);
                }
                st4(C + off, d);
                st4((TC*)e.out2 + off, g);
                return;
            }
            case SSL4GIE_EPI_MUL_AUX:
                v *= ld4((const TC*)e.aux + off);
                break;
            default:
                if (e.accumulate) v += ld4(C + off);
                break;
        }
        st4(C + off, v);
    } else {
        const int nn = (N - n) < 4 ? (N - n) : 4;
        for (int j = 0; j < nn; ++j) {
            float v = acc[j] * e.alpha;
            switch (e.mode) {
                case SSL4GIE_EPI_BIAS:
                    v += e.bias[n + j];
                    break;
                case SSL4GIE_EPI_BIAS_GELU:
                    if (e.bias) v += e.bias[n + j];
                    Elem<TC>::st(C + off + j, v);
                    Elem<TC>::st((TC*)e.out2 + off + j, gelu_f(v));
                    continue;
                case SSL4GIE_EPI_BIAS_RESIDUAL:
                    if (e.bias) v += e.bias[n + j];
                    v += e.residual[(size_t)m * e.ldr + n + j];
                    break;
                case SSL4GIE_EPI_DGELU:
                    v *= dgelu_f(Elem<TC>::ld((const TC*)e.aux + off + j));
                    break;
                case SSL4GIE_EPI_BIAS_GELU_GRAD:
                    if (e.bias) v += e.bias[n + j];
                    Elem<TC>::st(C + off + j, dgelu_f(v));
                    Elem<TC>::st((TC*)e.out2 + off + j, gelu_f(v));
                    continue;
                case SSL4GIE_EPI_MUL_AUX:
                    v *= Elem<TC>::ld((const TC*)e.aux + off + j);
                    break;
                default:
                    if (e.accumulate) v += Elem<TC>::ld(C + off + j);
                    break;
            }
            Elem<TC>::st(C + off + j, v);
        }
    }
}

// vector epilogue with the mode resolved at compile time (bf16 fast paths; fast GELU forms)
template <typename TC, int MODE>
DEVI void epi_vec(const EpiArgs& e, TC* __restrict__ C, long long ldc, int m, int n, int M, int N,
                  f32x4 acc) {
    if (m >= M || n >= N) return;
    const size_t off = (size_t)m * ldc + n;
    f32x4 v = acc * e.alpha;
    if (MODE == SSL4GIE_EPI_BIAS) {
        v += ld4(e.bias + n);
    } else if (MODE == SSL4GIE_EPI_BIAS_GELU) {
        if (e.bias) v += ld4(e.bias + n);
        st4(C + off, v);
        f32x4 g;
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = gelu_fast(v[j]);
        st4((TC*)e.out2 + off, g);
        return;
    } else if (MODE == SSL4GIE_EPI_BIAS_RESIDUAL) {
        if (e.bias) v += ld4(e.bias + n);
        v += ld4(e.residual + (size_t)m * e.ldr + n);
    } else if (MODE == SSL4GIE_EPI_DGELU) {
        const f32x4 u = ld4((const TC*)e.aux + off);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= dgelu_fast(u[j]);
    } else if (MODE == SSL4GIE_EPI_BIAS_GELU_GRAD) {
        if (e.bias) v += ld4(e.bias + n);
        f32x4 g, d;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float cdf, xpdf;
            gelu_parts_fast(v[j], cdf, xpdf);
            g[j] = v[j] * cdf;
            d[j] = cdf + xpdf;
        }
        st4(C + off, d);
        st4((TC*)e.out2 + off, g);
        return;
    } else if (MODE == SSL4GIE_EPI_MUL_AUX) {
        v *= ld4((const TC*)e.aux + off);
    } else {
        if (e.accumulate) v += ld4(C + off);
    }
    st4(C + off, v);
}
// a wave's 64x64 sub-tile (4x4 accumulators): rows m_base + 16i + (l&15), cols n_base + 16j + 4(l>>4)
template <typename TC, int MODE>
DEVI void epi_wave_tile(const EpiArgs& e, TC* __restrict__ C, long long ldc, int m_base, int n_base,
                        int M, int N, f32x4 (&acc)[4][4], int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            epi_vec<TC, MODE>(e, C, ldc, m_base + i * 16 + (lane & 15), n_base + j * 16 + 4 * (lane >> 4),
                              M, N, acc[i][j]);
}
template <typename TC>
DEVI void epi_wave_tile_dispatch(const EpiArgs& e, TC* __restrict__ C, long long ldc, int m_base,
                                 int n_base, int M, int N, f32x4 (&acc)[4][4], int lane) {
    switch (e.mode) {
        case SSL4GIE_EPI_BIAS: epi_wave_tile<TC, SSL4GIE_EPI_BIAS>(e, C, ldc, m_base, n_base, M, N, acc, lane); break;
        case SSL4GIE_EPI_BIAS_GELU: epi_wave_tile<TC, SSL4GIE_EPI_BIAS_GELU>(e, C, ldc, m_base, n_base, M, N, acc, lane); break;
        case SSL4GIE_EPI_BIAS_RESIDUAL: epi_wave_tile<TC, SSL4GIE_EPI_BIAS_RESIDUAL>(e, C, ldc, m_base, n_base, M, N, acc, lane); break;
        case SSL4GIE_EPI_DGELU: epi_wave_tile<TC, SSL4GIE_EPI_DGELU>(e, C, ldc, m_base, n_base, M, N, acc, lane); break;
        case SSL4GIE_EPI_BIAS_GELU_GRAD: epi_wave_tile<TC, SSL4GIE_EPI_BIAS_GELU_GRAD>(e, C, ldc, m_base, n_base, M, N, acc, lane); break;
        case SSL4GIE_EPI_MUL_AUX: epi_wave_tile<TC, SSL4GIE_EPI_MUL_AUX>(e, C, ldc, m_base, n_base, M, N, acc, lane); break;
        default: epi_wave_tile<TC, SSL4GIE_EPI_NONE>(e, C, ldc, m_base, n_base, M, N, acc, lane); break;
    }
}

// =====================================================================================
// generic strided / batched GEMM on f32 MFMA (16x16x4): 64x64x16 tiles, 4 waves (2x2)
// =====================================================================================
#define GT_M 64
#define GT_N 64
#define GT_K 16
#define GT_LD 80  // LDS row stride (floats): rows l>>4 land 16 banks apart -> conflict-free reads

template <typename TAB, typename TC>
__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmArgs g) {
    __shared__ float As[GT_K][GT_LD];
    __shared__ float Bs[GT_K][GT_LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.z, b1 = b / g.batch2, b2 = b % g.batch2;
    const TAB* A = (const TAB*)g.A + b1 * g.sAb1 + b2 * g.sAb2;
    const TAB* B = (const TAB*)g.B + b1 * g.sBb1 + b2 * g.sBb2;
    TC* C = (TC*)g.C + b1 * g.sCb1 + b2 * g.sCb2;
    const int m0 = blockIdx.y * GT_M, n0 = blockIdx.x * GT_N;
    const bool a_k = (g.sAk == 1), b_k = (g.sBk == 1);
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;

    for (int kt = 0; kt < g.K; kt += GT_K) {
        // ---- stage A (64 m x 16 k) and B (16 k x 64 n) as [k][x] in LDS
        {
            int mm, kk, dm, dk;
            if (a_k) { mm = t >> 2; kk = (t & 3) * 4; dm = 0; dk = 1; }
            else     { kk = t >> 4; mm = (t & 15) * 4; dm = 1; dk = 0; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = mm + i * dm, k = kk + i * dk;
                const int gm = m0 + m, gk = kt + k;
                float v = 0.f;
                if (gm < g.M && gk < g.K) v = Elem<TAB>::ld(A + gm * g.sAm + gk * g.sAk);
                As[k][m] = v;
            }
            int nn, dn;
            if (b_k) { nn = t >> 2; kk = (t & 3) * 4; dn = 0; dk = 1; }
            else     { kk = t >> 4; nn = (t & 15) * 4; dn = 1; dk = 0; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = nn + i * dn, k = kk + i * dk;
                const int gn = n0 + n, gk = kt + k;
                float v = 0.f;
                if (gn < g.N && gk < g.K) v = Elem<TAB>::ld(B + gk * g.sBk + gn * g.sBn);
                Bs[k][n] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < GT_K; ks += 4) {
            const int k = ks + (lane >> 4);
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                av[i] = As[k][wm + i * 16 + (lane & 15)];
                bv[i] = Bs[k][wn + i * 16 + (lane & 15)];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j], av[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + wm + i * 16 + (lane & 15);
            const int n = n0 + wn + j * 16 + 4 * (lane >> 4);
            epi_store4<TC, false>(g.e, C, g.ldc, m, n, g.M, g.N, acc[i][j]);
        }
}

// =====================================================================================
// bf16 fast paths
// =====================================================================================
#define BT_M 128
#define BT_N 128
#define BT_K 64
#define BT_TILE_BYTES (128 * 64 * 2)            // one operand tile (16 KiB)
#define BT_STAGE_BYTES (2 * BT_TILE_BYTES)      // A + B
#define BT_LDS_BYTES (2 * BT_STAGE_BYTES)       // double buffered: 64 KiB -> 2 workgroups / CU
#define NT_MAX_WGS 512                          // persistent NT grid: 2 workgroups x 256 CUs

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

DEVI void glds16(const void* gsrc, char* lds_wave_base) {
    // LDS destination = wave-uniform base + lane*16 (hardware rule), source address per lane
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)gsrc, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

// Same LDS-DMA issued from inline asm: invisible to hipcc's waitcnt pass, which otherwise puts a
// conservative `s_waitcnt vmcnt(0)` in front of the ds_read_b64_tr_b16 stream (the prefetch would
// then serialise with the MFMAs).  The caller owns the vmcnt accounting.  M0 is saved/restored
// inside the statement (cdna_hip_programming.md §5.7).
DEVI void glds16_asm(const void* gsrc, unsigned lds_wave_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_wave_addr)
        : "memory");
}
DEVI unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// ---- NT --------------------------------------------------------------------------------
// LDS image of an operand tile: 128 rows x 128 B; the 16-B chunk at position p of row r holds
// global chunk p ^ ((r>>1)&7)  (swizzle applied on the glds SOURCE address and again on the
// ds_read_b128 address; makes the 16 rows a lane group touches hit 16 distinct 16-B bank slots).
DEVI int nt_swz(int r) { return (r >> 1) & 7; }

// ---- LDS-staged epilogue for bf16 outputs -------------------------------------------------------
// Accumulator fragments give a lane 4 consecutive columns (8 B in bf16) of 16 different rows: a
// direct store writes 32-byte row fragments, which caps the store stream at ~1.5 TB/s (measured on
// the 4D-wide MLP outputs).  Instead the finished 128x128 bf16 tile is staged in the K-tile buffer
// that has just been consumed (exactly 32 KiB) and written out as whole 256-byte rows, 16 B/lane.
// LDS image: row r at r*256, 16-B chunk c stored at position c ^ (r & 15)  (ds_write_b64 2-way,
// ds_read_b128 conflict-free).
template <int XF>  // 0 as is, 1 gelu, 2 gelu'
DEVI void tile_lds_write(char* buf, int row_base, int col_base, const f32x4 (&v)[4][4], int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = row_base + 16 * i + (lane & 15);
            const int col = col_base + 16 * j + 4 * (lane >> 4);
            const int c = col >> 3;
            f32x4 x = v[i][j];
            if (XF != 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) x[q] = XF == 1 ? gelu_fast(x[q]) : dgelu_fast(x[q]);
            }
            u32x2 pk;
            pk[0] = pack_bf2(x[0], x[1]);
            pk[1] = pack_bf2(x[2], x[3]);
            *(u32x2*)(buf + row * 256 + ((c ^ (row & 15)) << 4) + ((col & 4) << 1)) = pk;
            if (XF != 0) __builtin_amdgcn_sched_barrier(0);
        }
}
DEVI void tile_lds_store(const char* buf, bf16_t* __restrict__ C, long long ldc, int m0, int n0, int M,
                         int N, int wave, int lane) {
#pragma unroll 2
    for (int it = 0; it < 8; ++it) {
        const int row = wave * 32 + it * 4 + (lane >> 4), c = lane & 15;
        const u32x4 v = *(const u32x4*)(buf + row * 256 + ((c ^ (row & 15)) << 4));
        const int gm = m0 + row, gn = n0 + c * 8;
        if (gm < M && gn < N) __builtin_nontemporal_store(v, (u32x4*)(C + (size_t)gm * ldc + gn));
    }
}
// whole-tile epilogue (all 4 waves; contains workgroup barriers -> call uniformly)
// stage the (already transformed) 128x128 tile and store whole rows; `gelu2` adds the second,
// GELU-activated output.  Contains workgroup barriers -> call uniformly.
template <bool GELU2, int XF1 = 0>  // XF1: transform of the first output (2 = gelu' for GELU_GRAD)
DEVI void tile_out_lds(char* buf, bf16_t* __restrict__ C, bf16_t* __restrict__ C2, long long ldc, int m0,
                       int n0, int wm, int wn, int M, int N, f32x4 (&acc)[4][4], int wave, int lane) {
    __builtin_amdgcn_s_barrier();  // every wave has finished reading this K-tile buffer
    tile_lds_write<XF1>(buf, wm, wn, acc, lane);
    __syncthreads();
    tile_lds_store(buf, C, ldc, m0, n0, M, N, wave, lane);
    if (GELU2) {
        __syncthreads();
        tile_lds_write<1>(buf, wm, wn, acc, lane);
        __syncthreads();
        tile_lds_store(buf, C2, ldc, m0, n0, M, N, wave, lane);
    }
}

// Persistent: gridDim.x = min(#tiles, 2 per CU) workgroups walk tiles pos, pos+G, ... and the
// LDS-DMA stream never drains at a tile boundary — the first K-tile of the next output tile is
// already in flight while the epilogue (bias / GELU / residual, vector stores) of the current one
// runs.  With K = 512..768 a tile is only 8-12 K-steps long, so fill/drain would otherwise be a
// large fraction of its life.
template <typename TC, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_kernel(
    const bf16_t* __restrict__ A, long long lda, const bf16_t* __restrict__ B, long long ldb,
    TC* __restrict__ C, long long ldc, int M, int N, int K, int tiles_n, int ntiles, EpiArgs e) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int G = gridDim.x;
    const int pos = xcd_remap(blockIdx.x, G);  // consecutive positions share an XCD (L2 reuse)
    const int my_tiles = (ntiles - pos + G - 1) / G;
    const int nk = K / BT_K;

    // per-lane global source pointers of the 4 A and 4 B LDS-DMA pieces this wave issues per K-tile
    const bf16_t* asrc[4];
    const bf16_t* bsrc[4];
    auto point_at = [&](int ti) {
        const int tile = pos + ti * G;
        const int sm0 = (tile / tiles_n) * BT_M, sn0 = (tile % tiles_n) * BT_N;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (wave * 4 + i) * 8 + (lane >> 3);  // tile row
            const int c = (lane & 7) ^ nt_swz(r);            // global chunk stored at position lane&7
            int ga = sm0 + r; ga = ga < M ? ga : M - 1;
            int gb = sn0 + r; gb = gb < N ? gb : N - 1;
            asrc[i] = A + (size_t)ga * lda + c * 8;
            bsrc[i] = B + (size_t)gb * ldb + c * 8;
        }
    };
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * BT_STAGE_BYTES + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(asrc[i] + (size_t)kt * BT_K, base + i * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            glds16(bsrc[i] + (size_t)kt * BT_K, base + BT_TILE_BYTES + i * 1024);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    // per-lane LDS read offsets: row (l&15) of each 16-row fragment, logical chunk (l>>4)+4*ks
    int a_off[4], b_off[4], a_sw[4], b_sw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = wm + i * 16 + (lane & 15), rb = wn + i * 16 + (lane & 15);
        a_off[i] = ra * 128; a_sw[i] = nt_swz(ra);
        b_off[i] = BT_TILE_BYTES + rb * 128; b_sw[i] = nt_swz(rb);
    }
    auto compute = [&](int buf) {
        const char* base = smem + buf * BT_STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int cc = ks * 4 + (lane >> 4);
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *(const bf16x8*)(base + a_off[i] + ((cc ^ a_sw[i]) << 4));
                bfr[i] = *(const bf16x8*)(base + b_off[i] + ((cc ^ b_sw[i]) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    constexpr bool PRE_RES = false;  // measured: prefetching the fp32 residual tile does not pay
    constexpr bool PRE_AUX = (MODE == SSL4GIE_EPI_DGELU || MODE == SSL4GIE_EPI_MUL_AUX) && (sizeof(TC) == 2);
    f32x4 pre_res[PRE_RES ? 4 : 1][4];
    u32x2 pre_aux[PRE_AUX ? 4 : 1][4];
    (void)pre_res; (void)pre_aux;

    point_at(0);
    stage(0, 0);
    int st_ti = 0, st_kt = 1;  // next K-tile to stage
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int ti = 0; ti < my_tiles; ++ti) {
        const int tile = pos + ti * G;
        const int m0 = (tile / tiles_n) * BT_M, n0 = (tile % tiles_n) * BT_N;
        for (int kt = 0; kt < nk; ++kt) {
            if (st_kt == nk) { st_kt = 0; ++st_ti; if (st_ti < my_tiles) point_at(st_ti); }
            if (st_ti < my_tiles) { stage(cur ^ 1, st_kt); ++st_kt; }
            if (kt == nk - 1) {
                // epilogue operands (fp32 residual tile / bf16 pre-activation tile) are requested
                // BEFORE the last K-step's MFMAs so their HBM latency hides under them
                if constexpr (PRE_RES) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int m = m0 + wm + i * 16 + (lane & 15);
                            const int n = n0 + wn + j * 16 + 4 * (lane >> 4);
                            pre_res[i][j] = (m < M && n < N) ? ld4(e.residual + (size_t)m * e.ldr + n)
                                                             : f32x4{0, 0, 0, 0};
                        }
                }
                if constexpr (PRE_AUX) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int m = m0 + wm + i * 16 + (lane & 15);
                            const int n = n0 + wn + j * 16 + 4 * (lane >> 4);
                            pre_aux[i][j] = (m < M && n < N)
                                ? *(const u32x2*)((const bf16_t*)e.aux + (size_t)m * ldc + n) : u32x2{0, 0};
                        }
                }
            }
            compute(cur);
            if (kt == nk - 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int n = n0 + wn + j * 16 + 4 * (lane >> 4);
                        acc[i][j] *= e.alpha;
                        if constexpr (MODE == SSL4GIE_EPI_BIAS || MODE == SSL4GIE_EPI_BIAS_GELU ||
                                      MODE == SSL4GIE_EPI_BIAS_RESIDUAL ||
                                      MODE == SSL4GIE_EPI_BIAS_GELU_GRAD) {
                            if (e.bias && n < N) acc[i][j] += ld4(e.bias + n);
                        }
                        if constexpr (PRE_RES) acc[i][j] += pre_res[i][j];
                        if constexpr (PRE_AUX) {
                            const u32x2 r = pre_aux[i][j];
                            f32x4 u = {__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u),
                                       __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u)};
                            if constexpr (MODE == SSL4GIE_EPI_DGELU) {
#pragma unroll
                                for (int q = 0; q < 4; ++q) u[q] = dgelu_fast(u[q]);
                            }
                            acc[i][j] *= u;
                        }
                    }
                if constexpr (sizeof(TC) == 2 && MODE != SSL4GIE_EPI_BIAS_RESIDUAL) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    tile_out_lds<MODE == SSL4GIE_EPI_BIAS_GELU || MODE == SSL4GIE_EPI_BIAS_GELU_GRAD,
                                 MODE == SSL4GIE_EPI_BIAS_GELU_GRAD ? 2 : 0>(
                        smem + cur * BT_STAGE_BYTES, (bf16_t*)C, (bf16_t*)e.out2, ldc, m0, n0, wm, wn, M,
                        N, acc, wave, lane);
                } else if constexpr (MODE == SSL4GIE_EPI_BIAS_RESIDUAL) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int m = m0 + wm + i * 16 + (lane & 15);
                            const int n = n0 + wn + j * 16 + 4 * (lane >> 4);
                            if (m < M && n < N) {
                                acc[i][j] += ld4(e.residual + (size_t)m * e.ldr + n);
                                st4(C + (size_t)m * ldc + n, acc[i][j]);
                            }
                        }
                } else {
                    EpiArgs e2 = e;  // alpha / bias already applied above
                    e2.alpha = 1.f;
                    e2.bias = nullptr;
                    // fp32 outputs: bias already added; DGELU / MUL_AUX re-read aux in the tile epilogue
                    constexpr int REST = (MODE == SSL4GIE_EPI_BIAS) ? SSL4GIE_EPI_NONE : MODE;
                    epi_wave_tile<TC, REST>(e2, C, ldc, m0 + wm, n0 + wn, M, N, acc, lane);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            cur ^= 1;
        }
    }
}

// ---- TN --------------------------------------------------------------------------------
// LDS image of an operand tile: 64 k-rows x 256 B (128 x-values); chunk position p of row k holds
// global chunk p ^ tn_swz(k).  ds_read_b64_tr_b16 then delivers, per 16-lane group, a 4(k) x 16(x)
// block column-major: two reads give a lane its 8 k-values of one x — the MFMA operand.
DEVI int tn_swz(int k) { return ((k & 3) | ((k >> 1) & 4)) << 1; }

DEVI bf16x8 tr_frag(const char* tile, int krow0 /* ks*32 + 8*(l>>4) */, int x0, int lane) {
    const int q = (lane & 15) >> 2, p = lane & 3;
    const int c = (x0 >> 3) + (p >> 1);  // logical 16-B chunk
    const int r0 = krow0 + q, r1 = krow0 + 4 + q;
    const int o0 = r0 * 256 + ((c ^ tn_swz(r0)) << 4) + ((p & 1) << 3);
    const int o1 = r1 * 256 + ((c ^ tn_swz(r1)) << 4) + ((p & 1) << 3);
    typedef __attribute__((address_space(3))) s16x4* lp_t;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(tile + o0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(tile + o1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// grid.x = tiles_m*tiles_n*splits.  splits==1: epilogue straight to C; else fp32 slabs.
__global__ __launch_bounds__(256, 2) void gemm_bf16_tn_kernel(
    const bf16_t* __restrict__ At, long long ldat, const bf16_t* __restrict__ Bt, long long ldbt,
    float* __restrict__ C, long long ldc, float* __restrict__ slabs, int M, int N, int K,
    int tiles_n, int ntiles, int splits, EpiArgs e) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = bid / ntiles, tile = bid % ntiles;
    const int m0 = (tile / tiles_n) * BT_M, n0 = (tile % tiles_n) * BT_N;
    const int nkt = (K + BT_K - 1) / BT_K;
    const int kt0 = (int)((long long)nkt * split / splits);
    const int kt1 = (int)((long long)nkt * (split + 1) / splits);

    // this lane's piece of each 4-row LDS-DMA instruction: row (w*4+i)*4 + (l>>4), position l&15
    int krow[4];
    int acol[4], bcol[4];
    const int mch = (M >> 3) - 1, nch = (N >> 3) - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = (wave * 4 + i) * 4 + (lane >> 4);
        krow[i] = k;
        const int c = (lane & 15) ^ tn_swz(k);
        int ca = (m0 >> 3) + c; ca = ca < mch ? ca : mch;
        int cb = (n0 >> 3) + c; cb = cb < nch ? cb : nch;
        acol[i] = ca * 8;
        bcol[i] = cb * 8;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * 4096);
    auto stage = [&](int buf, int kt) {
        const unsigned base = lds0 + buf * BT_STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int gk = kt * BT_K + krow[i]; gk = gk < K ? gk : K - 1;
            glds16_asm(At + (size_t)gk * ldat + acol[i], base + i * 1024);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int gk = kt * BT_K + krow[i]; gk = gk < K ? gk : K - 1;
            glds16_asm(Bt + (size_t)gk * ldbt + bcol[i], base + BT_TILE_BYTES + i * 1024);
        }
    };
    // zero the k-rows of a tail tile that lie beyond K (sources were clamped)
    auto zero_tail = [&](int buf, int kt) {
        const int kv = K - kt * BT_K;  // valid rows (< 64)
        if (kv < BT_K) {
            char* base = smem + buf * BT_STAGE_BYTES;
            for (int o = kv * 256 + t * 16; o < BT_K * 256; o += 256 * 16) {
                *(u32x4*)(base + o) = u32x4{0, 0, 0, 0};
                *(u32x4*)(base + BT_TILE_BYTES + o) = u32x4{0, 0, 0, 0};
            }
            __syncthreads();
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    auto compute = [&](int buf) {
        const char* ta = smem + buf * BT_STAGE_BYTES;
        const char* tb = ta + BT_TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int kr = ks * 32 + 8 * (lane >> 4);
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = tr_frag(ta, kr, wm + i * 16, lane);
                bfr[i] = tr_frag(tb, kr, wn + i * 16, lane);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    if (kt0 < kt1) {
        stage(0, kt0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int cur = 0;
        for (int kt = kt0; kt < kt1 - 1; ++kt) {
            stage(cur ^ 1, kt + 1);
            compute(cur);  // tiles before the last are always full
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            cur ^= 1;
        }
        zero_tail(cur, kt1 - 1);
        compute(cur);
    }

    float* out = (splits == 1) ? C : slabs + (size_t)split * M * N;
    const long long ldo = (splits == 1) ? ldc : N;
    EpiArgs es = e;
    if (splits != 1) { es.mode = SSL4GIE_EPI_NONE; es.accumulate = 0; es.alpha = 1.f; }
    epi_wave_tile<float, SSL4GIE_EPI_NONE>(es, out, ldo, m0 + wm, n0 + wn, M, N, acc, lane);
}

// C[m, n] (+)= alpha * sum_s slabs[s][m][n]   (N % 4 == 0); the blocks past the C range reduce the
// column-sum partials cs_part[s][m] into cs_out[m] (bias gradient of the fused TN kernel)
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ C,
                                   long long ldc, int M, int N, int splits, float alpha,
                                   int accumulate, const float* __restrict__ cs_part,
                                   float* __restrict__ cs_out, unsigned c_blocks) {
    if (blockIdx.x >= c_blocks) {
        const int m = (blockIdx.x - c_blocks) * blockDim.x + threadIdx.x;
        if (m < M) {
            float s = 0.f;
            for (int k = 0; k < splits; ++k) s += cs_part[(size_t)k * M + m];
            cs_out[m] = accumulate ? cs_out[m] + s : s;
        }
        return;
    }
    const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total4 = (size_t)M * N / 4;
    if (i4 >= total4) return;
    const size_t idx = i4 * 4;
    const int m = (int)(idx / N), n = (int)(idx % N);
    f32x4 s = ld4(slabs + idx);
    int k = 1;
    for (; k + 3 < splits; k += 4) {  // four slabs' loads in flight, added in slab order
        const f32x4 a = ld4(slabs + (size_t)k * M * N + idx), b = ld4(slabs + (size_t)(k + 1) * M * N + idx);
        const f32x4 c2 = ld4(slabs + (size_t)(k + 2) * M * N + idx), d2 = ld4(slabs + (size_t)(k + 3) * M * N + idx);
        s += a; s += b; s += c2; s += d2;
    }
    for (; k < splits; ++k) s += ld4(slabs + (size_t)k * M * N + idx);
    s *= alpha;
    float* c = C + (size_t)m * ldc + n;
    if (accumulate) s += ld4(c);
    st4(c, s);
}

// The same reduction for MANY slabs of a small output (the long-contraction weight gradients of ResNet's 1x1
// convolutions: up to 256 splits of a 64 KiB tile): 64 float4 columns x 4 slab groups per block — group g sums
// slabs g, g + 4, g + 8, ... in that order, then the four group sums are added in group order (fixed order:
// deterministic), so four times as many loads are in flight as with one thread per column.
// G slab groups per block (4, or 16 from 128 slabs on: a dW [256, 64] tile over 802 816 pixels has 256 slabs and only
// 4096 float4 columns — with 4 groups 64 blocks each summed 64 slabs per thread, eight loads at a time: 30 us of
// latency for 17 MB; with 16 groups a thread sums 16, round 5)
template <int G>
__global__ __launch_bounds__(64 * G) void slab_reduce_wide_kernel(const float* __restrict__ slabs,
                                                                  float* __restrict__ C, long long ldc, int M, int N,
                                                                  int splits, float alpha, int accumulate) {
    __shared__ f32x4 part[G - 1][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t i4 = (size_t)blockIdx.x * 64 + c;
    const size_t total4 = (size_t)M * N / 4;
    const bool ok = i4 < total4;
    const size_t idx = i4 * 4, slab = (size_t)M * N;
    f32x4 s = {0, 0, 0, 0};
    if (ok)
    {
        int k = g;
        for (; k + 7 * G < splits; k += 8 * G) {  // eight of this group's slabs in flight, added in slab order
            f32x4 a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = ld4(slabs + (size_t)(k + G * u) * slab + idx);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += a[u];
        }
        for (; k < splits; k += G) s += ld4(slabs + (size_t)k * slab + idx);
    }
    if (g > 0) part[g - 1][c] = s;
    __syncthreads();
    if (g != 0 || !ok) return;
#pragma unroll
    for (int w = 0; w < G - 1; ++w) s += part[w][c];
    s *= alpha;
    const int m = (int)(idx / N), n = (int)(idx % N);
    float* o = C + (size_t)m * ldc + n;
    if (accumulate) s += ld4(o);
    st4(o, s);
}
// slab reduction of one product: the wide form from 32 slabs on when no column-sum partials ride along
static int launch_slab_reduce(const float* slabs, float* C, long long ldc, int M, int N, int splits, float alpha,
                              int accumulate, const float* cs_part, float* cs_out, bool fused_cs, hipStream_t st) {
    const size_t total4 = (size_t)M * N / 4;
    if (splits >= 32 && !fused_cs) {
        if (splits >= 128)
            hipLaunchKernelGGL(slab_reduce_wide_kernel<16>, dim3((unsigned)((total4 + 63) / 64)), dim3(1024), 0, st,
                               slabs, C, ldc, M, N, splits, alpha, accumulate);
        else
            hipLaunchKernelGGL(slab_reduce_wide_kernel<4>, dim3((unsigned)((total4 + 63) / 64)), dim3(256), 0, st, slabs,
                               C, ldc, M, N, splits, alpha, accumulate);
        LAUNCH_CHECK();
        return 0;
    }
    const unsigned c_blocks = (unsigned)((total4 + 255) / 256);
    const unsigned b_blocks = fused_cs ? (unsigned)((M + 255) / 256) : 0;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(c_blocks + b_blocks), dim3(256), 0, st, slabs, C, ldc, M, N, splits,
                       alpha, accumulate, cs_part, cs_out, c_blocks);
    LAUNCH_CHECK();
    return 0;
}

// =====================================================================================
// dispatch
// =====================================================================================
static bool is16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

static bool nt_ok(const ssl4gie_gemm_desc* d) {
    return d->dtype_ab == SSL4GIE_BF16 && d->batch1 * d->batch2 == 1 && d->sAk == 1 &&
           d->sBk == 1 && d->K % BT_K == 0 && d->K >= BT_K && d->sAm % 8 == 0 &&
           d->sBn % 8 == 0 && d->N % 4 == 0 && d->ldc % 4 == 0 && is16(d->A) && is16(d->B) &&
           is16(d->C) && (!d->residual || (d->ldr % 4 == 0 && is16(d->residual))) &&
           (!d->bias || is16(d->bias)) && (!d->aux || is16(d->aux)) && (!d->scale || is16(d->scale)) &&
           (!d->out2 || is16(d->out2)) && d->M > 0 && d->N > 0 &&
           // bf16 outputs leave through the LDS-staged 16-byte row stores
           (d->dtype_c == SSL4GIE_F32 || d->epilogue == SSL4GIE_EPI_BIAS_RESIDUAL ||
            (d->N % 8 == 0 && d->ldc % 8 == 0 && !d->accumulate));
}
static bool tn_ok(const ssl4gie_gemm_desc* d) {
    return d->dtype_ab == SSL4GIE_BF16 && d->dtype_c == SSL4GIE_F32 &&
           d->batch1 * d->batch2 == 1 && d->sAm == 1 && d->sBn == 1 && d->M % 8 == 0 &&
           d->N % 8 == 0 && d->sAk % 8 == 0 && d->sBk % 8 == 0 && d->ldc % 4 == 0 &&
           d->epilogue == SSL4GIE_EPI_NONE && is16(d->A) && is16(d->B) && is16(d->C) &&
           d->K > 0 && d->M >= 8 && d->N >= 8;
}
static int tn_splits(const ssl4gie_gemm_desc* d) {
    const int tiles = ((d->M + BT_M - 1) / BT_M) * ((d->N + BT_N - 1) / BT_N);
    const int nkt = (d->K + BT_K - 1) / BT_K;
    int s = (1024 + tiles - 1) / tiles;  // aim at ~4 workgroups per CU worth of work items
    if (s > nkt / 4) s = nkt / 4;        // at least 4 K-tiles per split
    if (s < 1) s = 1;
    // up to 256 splits: the 1x1-convolution weight gradients of ResNet's wide maps (dW [64, 256] over 802 816
    // pixels) have ONE or two output tiles and a contraction 25 000 K-tiles long — at the former cap of 64 they
    // ran on a quarter of the chip (225 us for 0.5 GB: 2.3 TB/s); their slabs are 64 KiB each
    if (s > 256) s = 256;
    return s;
}

// ---- CUs available to the GEMM grids
// -1: SSL4GIE_COMPUTE_CUS or 240 — measured (profiles/r01w_ab_grid_sizing.log): with the
// weight-gradient side stream two GEMM streams share the chip, and grids sized for 240 CUs pack
// better than grids sized for all 256 (MAE step +1.4 %)
static int g_compute_cus = -1;
int ssl4gie_internal_compute_cus() {
    if (g_compute_cus < 0) {
        const char* e = getenv("SSL4GIE_COMPUTE_CUS");
        const int v = e ? atoi(e) : 240;
        g_compute_cus = v < 8 ? 8 : (v > 256 ? 256 : v);
    }
    return g_compute_cus;
}
extern "C" int ssl4gie_set_compute_cus(int n) {
    REQUIRE(n >= 8 && n <= 256);
    g_compute_cus = n;
    return 0;
}

// ---- implicit 3x3 patch-matrix operand (ssl4gie_gemm_desc::conv)
bool ssl4gie_internal_conv_geom_ok(const ssl4gie_conv3x3_geom* g) {
    if (!g || g->B < 1 || g->H < 1 || g->W < 1 || g->C < 8 || g->C % 8 != 0) return false;
    if (g->stride != 1 && g->stride != 2) return false;
    const long long Ho = (g->H - 1) / g->stride + 1, Wo = (g->W - 1) / g->stride + 1;
    if (Wo < 2 || Ho * Wo < 2 || g->H > 32767 || g->W > 32767) return false;
    // signed 32-bit byte offsets into the map (incl. the row above it), pixel index < 2^31
    return ((long long)g->B * g->H + 1) * g->W * g->C * 2 < (1LL << 31) &&
           (long long)g->B * Ho * Wo < (1LL << 31);
}
// 256 zero bytes per device, allocated on first use (never freed: lives as long as the library)
static const void* conv_zero_page(int* rc) {
    static void* page[64] = {nullptr};
    static std::mutex mu;
    int dev = 0;
    *rc = (int)hipGetDevice(&dev);
    if (*rc) return nullptr;
    if (dev < 0 || dev >= 64) { *rc = ARG_ERR; return nullptr; }
    std::lock_guard<std::mutex> lk(mu);
    if (!page[dev]) {
        void* p = nullptr;
        *rc = (int)hipMalloc(&p, 256);
        if (*rc) return nullptr;
        *rc = (int)hipMemset(p, 0, 256);
        if (!*rc) *rc = (int)hipStreamSynchronize(nullptr);  // visible to every stream from here on
        if (*rc) return nullptr;
        page[dev] = p;
    }
    return page[dev];
}
int ssl4gie_internal_conv_k(const ssl4gie_conv3x3_geom* g, ConvK* k) {
    if (!ssl4gie_internal_conv_geom_ok(g)) return ARG_ERR;
    const int Ho = (g->H - 1) / g->stride + 1, Wo = (g->W - 1) / g->stride + 1;
    k->H = g->H; k->W = g->W; k->C = g->C; k->Wo = Wo; k->HoWo = Ho * Wo; k->stride = g->stride;
    conv_magic((unsigned)Wo, &k->mg_wo, &k->sh_wo);
    conv_magic((unsigned)(Ho * Wo), &k->mg_hw, &k->sh_hw);
    int rc = 0;
    k->zero = conv_zero_page(&rc);
    return rc;
}

static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
// workspace layout of the TN paths: [slabs (splits > 1)] [column-sum scratch (colsum_a)]
struct TnPlan {
    bool big;
    int splits;
    size_t slab_bytes, cs_off, cs_bytes, total;
};
static TnPlan tn_plan(const ssl4gie_gemm_desc* d) {
    TnPlan p;
    p.big = ssl4gie_internal_tn256_ok(d);
    p.splits = p.big ? ssl4gie_internal_tn256_splits(d) : tn_splits(d);
    p.slab_bytes = p.splits > 1 ? (size_t)p.splits * d->M * d->N * sizeof(float) : 0;
    p.cs_off = al256(p.slab_bytes);
    p.cs_bytes = 0;
    if (d->colsum_a)
        p.cs_bytes = p.big ? (p.splits > 1 ? (size_t)p.splits * d->M * sizeof(float) : 0)
                           : ssl4gie_colsum_workspace_bytes(d->K, d->M);
    p.total = p.cs_bytes ? p.cs_off + p.cs_bytes : p.slab_bytes;
    return p;
}

extern "C" size_t ssl4gie_gemm_workspace_bytes(const ssl4gie_gemm_desc* d) {
    if (!d) return 0;
    if (d->dtype_ab == SSL4GIE_BF16 && !nt_ok(d) && tn_ok(d)) return tn_plan(d).total;
    if (d->colsum_a) return ssl4gie_colsum_workspace_bytes(d->K, d->M);
    return 0;
}

extern "C" int ssl4gie_gemm(const ssl4gie_gemm_desc* d, void* workspace, size_t workspace_bytes,
                            void* stream) {
    REQUIRE(d && d->A && d->B && d->M >= 0 && d->N >= 0 && d->K >= 0);
    // C == NULL: the statistics-only product of ssl4gie_gemm_desc::colstats (256x256 NT kernel only)
    REQUIRE(d->C || (d->colstats && d->dtype_ab == SSL4GIE_BF16 && nt_ok(d) && ssl4gie_internal_nt256_ok(d)));
    REQUIRE(d->batch1 >= 1 && d->batch2 >= 1);
    REQUIRE(d->dtype_ab == SSL4GIE_F32 || d->dtype_ab == SSL4GIE_BF16);
    REQUIRE(d->dtype_c == SSL4GIE_F32 || d->dtype_c == SSL4GIE_BF16);
    REQUIRE(d->epilogue >= SSL4GIE_EPI_NONE && d->epilogue <= SSL4GIE_EPI_AFFINE_AUX_RELU);
    REQUIRE(d->epilogue != SSL4GIE_EPI_AFFINE_AUX_RELU ||  // 256x256 NT kernel only
            (d->scale && d->bias && d->batch1 * d->batch2 == 1 && d->dtype_ab == SSL4GIE_BF16 && nt_ok(d) &&
             ssl4gie_internal_nt256_ok(d)));
    REQUIRE(d->epilogue != SSL4GIE_EPI_ADD_AUX ||
            (d->aux && !d->conv && nt_ok(d) && ssl4gie_internal_nt256_ok(d)));  // 256x256 NT kernel only
    REQUIRE(d->epilogue != SSL4GIE_EPI_RELU_MASK_AUX || (d->conv && d->aux));  // implicit conv only
    REQUIRE(d->epilogue != SSL4GIE_EPI_BIAS || d->bias);
    REQUIRE((d->epilogue != SSL4GIE_EPI_BIAS_GELU && d->epilogue != SSL4GIE_EPI_BIAS_GELU_GRAD) || d->out2);
    REQUIRE(d->epilogue != SSL4GIE_EPI_BIAS_RESIDUAL || d->residual);
    REQUIRE((d->epilogue != SSL4GIE_EPI_DGELU && d->epilogue != SSL4GIE_EPI_MUL_AUX) || d->aux);
    REQUIRE(d->batch1 * d->batch2 == 1 || d->epilogue == SSL4GIE_EPI_NONE);
    REQUIRE(!d->accumulate || d->epilogue == SSL4GIE_EPI_NONE);
    REQUIRE(!d->colsum_a || (d->sAm == 1 && d->batch1 * d->batch2 == 1 && d->sAk >= d->M));
    if (d->M == 0 || d->N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    EpiArgs e{d->alpha, d->epilogue, d->bias, d->residual, d->ldr, d->aux, d->out2, d->accumulate};

    REQUIRE(!d->conv || (d->dtype_ab == SSL4GIE_BF16 && d->batch1 * d->batch2 == 1));
    REQUIRE(!d->colstats || (d->dtype_ab == SSL4GIE_BF16 && d->batch1 * d->batch2 == 1));
    if (nt_ok(d)) {
        if (ssl4gie_internal_nt256_ok(d)) return ssl4gie_internal_nt256_launch(d, st);
        REQUIRE(!d->conv && !d->colstats);  // these only exist in the 256x256 kernels
        const int tm = (d->M + BT_M - 1) / BT_M, tn = (d->N + BT_N - 1) / BT_N;
        const int ntiles = tm * tn;
        const int max_wgs = 2 * ssl4gie_internal_compute_cus();  // 2 workgroups per CU
        dim3 grid(ntiles < max_wgs ? ntiles : max_wgs), block(256);
        ProfScope prof(PROF_GEMM_NT, 2.0 * d->M * d->N * d->K, st);
#define NT_LAUNCH(TC_, MODE_)                                                                  \
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<TC_, MODE_>), grid, block, BT_LDS_BYTES, st,       \
                       (const bf16_t*)d->A, d->sAm, (const bf16_t*)d->B, d->sBn, (TC_*)d->C,   \
                       d->ldc, d->M, d->N, d->K, tn, ntiles, e)
#define NT_MODES(TC_)                                                                    \
    switch (d->epilogue) {                                                               \
        case SSL4GIE_EPI_BIAS: NT_LAUNCH(TC_, SSL4GIE_EPI_BIAS); break;                  \
        case SSL4GIE_EPI_BIAS_GELU: NT_LAUNCH(TC_, SSL4GIE_EPI_BIAS_GELU); break;        \
        case SSL4GIE_EPI_BIAS_RESIDUAL: NT_LAUNCH(TC_, SSL4GIE_EPI_BIAS_RESIDUAL); break; \
        case SSL4GIE_EPI_DGELU: NT_LAUNCH(TC_, SSL4GIE_EPI_DGELU); break;                \
        case SSL4GIE_EPI_BIAS_GELU_GRAD: NT_LAUNCH(TC_, SSL4GIE_EPI_BIAS_GELU_GRAD); break; \
        case SSL4GIE_EPI_MUL_AUX: NT_LAUNCH(TC_, SSL4GIE_EPI_MUL_AUX); break;            \
        default: NT_LAUNCH(TC_, SSL4GIE_EPI_NONE); break;                                \
    }
        if (d->dtype_c == SSL4GIE_BF16) { NT_MODES(bf16_t) } else { NT_MODES(float) }
#undef NT_MODES
#undef NT_LAUNCH
        LAUNCH_CHECK();
        return 0;
    }
    if (tn_ok(d)) {
        const int tm = (d->M + BT_M - 1) / BT_M, tn = (d->N + BT_N - 1) / BT_N;
        const TnPlan p = tn_plan(d);
        const int splits = p.splits;
        REQUIRE(!d->conv || p.big);
        REQUIRE(!d->colstats);
        REQUIRE(p.total == 0 || (workspace && workspace_bytes >= p.total));
        float* slabs = (float*)workspace;
        float* cs_ws = p.cs_bytes ? (float*)((char*)workspace + p.cs_off) : nullptr;
        {
            ProfScope prof(PROF_GEMM_TN, 2.0 * d->M * d->N * d->K, st);
            if (p.big) {
                const int rc = ssl4gie_internal_tn256_launch(d, slabs, cs_ws, st);
                if (rc) return rc;
            } else {
                dim3 grid(tm * tn * splits), block(256);
                hipLaunchKernelGGL(gemm_bf16_tn_kernel, grid, block, BT_LDS_BYTES, st,
                                   (const bf16_t*)d->A, d->sAk, (const bf16_t*)d->B, d->sBk,
                                   (float*)d->C, d->ldc, slabs, d->M, d->N, d->K, tn, tm * tn,
                                   splits, e);
                LAUNCH_CHECK();
            }
        }
        const bool fused_cs = p.big && d->colsum_a;
        if (splits > 1) {
            const int rc = launch_slab_reduce((const float*)slabs, (float*)d->C, d->ldc, d->M, d->N, splits, d->alpha,
                                              d->accumulate, (const float*)cs_ws, d->colsum_a, fused_cs, st);
            if (rc) return rc;
        }
        if (d->colsum_a && !fused_cs)
            return ssl4gie_colsum(d->A, d->dtype_ab, d->colsum_a, d->accumulate, cs_ws, d->K, d->M,
                                  d->sAk, stream);
        return 0;
    }
    // generic
    REQUIRE(!d->conv && !d->colstats);
    GemmArgs g;
    g.M = d->M; g.N = d->N; g.K = d->K; g.batch2 = d->batch2;
    g.A = d->A; g.sAm = d->sAm; g.sAk = d->sAk; g.sAb1 = d->sAb1; g.sAb2 = d->sAb2;
    g.B = d->B; g.sBk = d->sBk; g.sBn = d->sBn; g.sBb1 = d->sBb1; g.sBb2 = d->sBb2;
    g.C = d->C; g.ldc = d->ldc; g.sCb1 = d->sCb1; g.sCb2 = d->sCb2;
    g.e = e;
    dim3 grid((d->N + GT_N - 1) / GT_N, (d->M + GT_M - 1) / GT_M, d->batch1 * d->batch2);
    dim3 block(256);
    REQUIRE(grid.y <= 65535 && grid.z <= 65535);
    ProfScope prof(PROF_GEMM_GENERIC, 2.0 * d->M * d->N * d->K * d->batch1 * d->batch2, st);
    if (d->dtype_ab == SSL4GIE_F32 && d->dtype_c == SSL4GIE_F32)
        hipLaunchKernelGGL((gemm_generic_kernel<float, float>), grid, block, 0, st, g);
    else if (d->dtype_ab == SSL4GIE_BF16 && d->dtype_c == SSL4GIE_BF16)
        hipLaunchKernelGGL((gemm_generic_kernel<bf16_t, bf16_t>), grid, block, 0, st, g);
    else if (d->dtype_ab == SSL4GIE_BF16 && d->dtype_c == SSL4GIE_F32)
        hipLaunchKernelGGL((gemm_generic_kernel<bf16_t, float>), grid, block, 0, st, g);
    else
        hipLaunchKernelGGL((gemm_generic_kernel<float, bf16_t>), grid, block, 0, st, g);
    LAUNCH_CHECK();
    if (d->colsum_a) {
        REQUIRE(workspace && workspace_bytes >= ssl4gie_colsum_workspace_bytes(d->K, d->M));
        return ssl4gie_colsum(d->A, d->dtype_ab, d->colsum_a, d->accumulate, (float*)workspace,
                              d->K, d->M, d->sAk, stream);
    }
    return 0;
}


// ---------------------------------------------------------------------------------------------
// Two weight-gradient (TN) products with the same contraction length in ONE launch.  A split-K TN
// product alone spreads over ~one workgroup per CU, each writing a full 256 x 256 fp32 slab (64 MiB
// per product whatever its shape); two products sharing the grid need half the splits each, hence
// half the slab traffic and reduction work.  Falls back to two ssl4gie_gemm calls when the pair does
// not qualify (shape rules of the 256x256 TN kernel, equal K, no implicit-conv operand).
static bool tn_pair_ok(const ssl4gie_gemm_desc* a, const ssl4gie_gemm_desc* b) {
    static int enabled = -1;  // SSL4GIE_TN_PAIR=0: always two launches (A/B measurements)
    if (enabled < 0) {
        const char* e = getenv("SSL4GIE_TN_PAIR");
        enabled = (e && e[0] == '0') ? 0 : 1;
    }
    return enabled && a && b && !nt_ok(a) && !nt_ok(b) && tn_ok(a) && tn_ok(b) && !a->conv && !b->conv &&
           ssl4gie_internal_tn256_ok(a) && ssl4gie_internal_tn256_ok(b) && a->K == b->K &&
           a->accumulate == b->accumulate && a->alpha == 1.f && b->alpha == 1.f &&
           a->N % 4 == 0 && b->N % 4 == 0;
}
struct PairPlan {
    int splits;
    size_t slab_a, slab_b, cs_a, cs_b, off_b, off_csa, off_csb, total;
};
static PairPlan pair_plan(const ssl4gie_gemm_desc* a, const ssl4gie_gemm_desc* b) {
    PairPlan p;
    p.splits = ssl4gie_internal_tn256_pair_splits(a, b);
    const bool sp = p.splits > 1;
    p.slab_a = sp ? al256((size_t)p.splits * a->M * a->N * sizeof(float)) : 0;
    p.slab_b = sp ? al256((size_t)p.splits * b->M * b->N * sizeof(float)) : 0;
    p.cs_a = (sp && a->colsum_a) ? al256((size_t)p.splits * a->M * sizeof(float)) : 0;
    p.cs_b = (sp && b->colsum_a) ? al256((size_t)p.splits * b->M * sizeof(float)) : 0;
    p.off_b = p.slab_a;
    p.off_csa = p.off_b + p.slab_b;
    p.off_csb = p.off_csa + p.cs_a;
    p.total = p.off_csb + p.cs_b;
    return p;
}
extern "C" size_t ssl4gie_gemm_tn_pair_workspace_bytes(const ssl4gie_gemm_desc* a,
                                                       const ssl4gie_gemm_desc* b) {
    if (!a || !b) return 0;
    if (tn_pair_ok(a, b)) return pair_plan(a, b).total;
    const size_t wa = ssl4gie_gemm_workspace_bytes(a), wb = ssl4gie_gemm_workspace_bytes(b);
    return wa > wb ? wa : wb;
}
extern "C" int ssl4gie_gemm_tn_pair(const ssl4gie_gemm_desc* a, const ssl4gie_gemm_desc* b,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    REQUIRE(a && b);
    if (!tn_pair_ok(a, b)) {
        const int rc = ssl4gie_gemm(a, workspace, workspace_bytes, stream);
        if (rc) return rc;
        return ssl4gie_gemm(b, workspace, workspace_bytes, stream);
    }
    REQUIRE(a->A && a->B && a->C && b->A && b->B && b->C);
    REQUIRE(!a->colsum_a || a->sAk >= a->M);
    REQUIRE(!b->colsum_a || b->sAk >= b->M);
    hipStream_t st = (hipStream_t)stream;
    const PairPlan p = pair_plan(a, b);
    REQUIRE(p.total == 0 || (workspace && workspace_bytes >= p.total));
    char* ws = (char*)workspace;
    float* slabs_a = (float*)ws;
    float* slabs_b = (float*)(ws + p.off_b);
    float* cs_a = p.cs_a ? (float*)(ws + p.off_csa) : nullptr;
    float* cs_b = p.cs_b ? (float*)(ws + p.off_csb) : nullptr;
    {
        ProfScope prof(PROF_GEMM_TN, 2.0 * a->K * ((double)a->M * a->N + (double)b->M * b->N), st);
        const int rc = ssl4gie_internal_tn256_launch_pair(a, b, p.splits, slabs_a, cs_a, slabs_b, cs_b, st);
        if (rc) return rc;
    }
    if (p.splits > 1) {
        const ssl4gie_gemm_desc* ds[2] = {a, b};
        float* sl[2] = {slabs_a, slabs_b};
        float* cs[2] = {cs_a, cs_b};
        for (int i = 0; i < 2; ++i) {
            const ssl4gie_gemm_desc* d = ds[i];
            const size_t total4 = (size_t)d->M * d->N / 4;
            const unsigned c_blocks = (unsigned)((total4 + 255) / 256);
            const unsigned b_blocks = d->colsum_a ? (unsigned)((d->M + 255) / 256) : 0;
            hipLaunchKernelGGL(slab_reduce_kernel, dim3(c_blocks + b_blocks), dim3(256), 0, st,
                               (const float*)sl[i], (float*)d->C, d->ldc, d->M, d->N, p.splits, d->alpha,
                               d->accumulate, (const float*)cs[i], d->colsum_a, c_blocks);
            LAUNCH_CHECK();
        }
    }
    return 0;
}

// ---- grouped weight-gradient products (ssl4gie_gemm_tn_group)
static bool tn_group_ok(const ssl4gie_gemm_desc* ds, int n) {
    if (!ds || n < 1 || n > TN_GROUP_MAX) return false;
    for (int i = 0; i < n; ++i) {
        const ssl4gie_gemm_desc* d = &ds[i];
        if (nt_ok(d) || !tn_ok(d) || d->conv || !ssl4gie_internal_tn256_ok(d) || d->K != ds[0].K ||
            d->N % 4 != 0 || d->dtype_c != SSL4GIE_F32)
            return false;
    }
    return true;
}
struct GroupPlan {
    int splits;
    size_t slab_off[TN_GROUP_MAX], cs_off[TN_GROUP_MAX], total;
};
static GroupPlan group_plan(const ssl4gie_gemm_desc* ds, int n) {
    GroupPlan p;
    p.splits = ssl4gie_internal_tn256_group_splits(ds, n);
    size_t o = 0;
    for (int i = 0; i < n; ++i) {
        p.slab_off[i] = o;
        if (p.splits > 1) o += al256((size_t)p.splits * ds[i].M * ds[i].N * sizeof(float));
    }
    for (int i = 0; i < n; ++i) {
        p.cs_off[i] = o;
        if (p.splits > 1 && ds[i].colsum_a) o += al256((size_t)p.splits * ds[i].M * sizeof(float));
    }
    p.total = o;
    return p;
}
extern "C" size_t ssl4gie_gemm_tn_group_workspace_bytes(const ssl4gie_gemm_desc* descs, int n) {
    if (!descs || n < 1) return 0;
    if (tn_group_ok(descs, n)) return group_plan(descs, n).total;
    size_t m = 0;
    for (int i = 0; i < n; ++i) {
        const size_t w = ssl4gie_gemm_workspace_bytes(&descs[i]);
        if (w > m) m = w;
    }
    return m;
}
extern "C" int ssl4gie_gemm_tn_group(const ssl4gie_gemm_desc* descs, int n, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    REQUIRE(descs && n >= 1);
    if (!tn_group_ok(descs, n)) {  // products that do not qualify run one by one
        for (int i = 0; i < n; ++i) {
            const int rc = ssl4gie_gemm(&descs[i], workspace, workspace_bytes, stream);
            if (rc) return rc;
        }
        return 0;
    }
    for (int i = 0; i < n; ++i) {
        REQUIRE(descs[i].A && descs[i].B && descs[i].C);
        REQUIRE(!descs[i].colsum_a || descs[i].sAk >= descs[i].M);
    }
    hipStream_t st = (hipStream_t)stream;
    const GroupPlan p = group_plan(descs, n);
    REQUIRE(p.total == 0 || (workspace && workspace_bytes >= p.total));
    char* ws = (char*)workspace;
    float* slabs[TN_GROUP_MAX];
    float* cs[TN_GROUP_MAX];
    double flops = 0;
    for (int i = 0; i < n; ++i) {
        slabs[i] = p.splits > 1 ? (float*)(ws + p.slab_off[i]) : nullptr;
        cs[i] = (p.splits > 1 && descs[i].colsum_a) ? (float*)(ws + p.cs_off[i]) : nullptr;
        flops += 2.0 * descs[i].K * (double)descs[i].M * descs[i].N;
    }
    {
        ProfScope prof(PROF_GEMM_TN, flops, st);
        const int rc = ssl4gie_internal_tn256_launch_group(descs, n, p.splits, slabs, cs, st);
        if (rc) return rc;
    }
    if (p.splits > 1) {
        for (int i = 0; i < n; ++i) {
            const ssl4gie_gemm_desc* d = &descs[i];
            const size_t total4 = (size_t)d->M * d->N / 4;
            const unsigned c_blocks = (unsigned)((total4 + 255) / 256);
            const unsigned b_blocks = d->colsum_a ? (unsigned)((d->M + 255) / 256) : 0;
            hipLaunchKernelGGL(slab_reduce_kernel, dim3(c_blocks + b_blocks), dim3(256), 0, st,
                               (const float*)slabs[i], (float*)d->C, d->ldc, d->M, d->N, p.splits, d->alpha,
                               d->accumulate, (const float*)cs[i], d->colsum_a, c_blocks);
            LAUNCH_CHECK();
        }
    }
    return 0;
}
