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
#include "common.h"
#include "ssl4gie_hip.h"
#include "prof.h"

struct EpiArgs {
    float alpha;
    int mode;
    const float* bias;
    const float* residual;
    long long ldr;
    const void* aux;
    void* out2;
    int accumulate;
};

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
template <typename TC, bool VEC>
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
                for (int j = 0; j < 4; ++j) g[j] = gelu_f(v[j]);
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
                for (int j = 0; j < 4; ++j) v[j] *= dgelu_f(u[j]);
                break;
            }
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
                default:
                    if (e.accumulate) v += Elem<TC>::ld(C + off + j);
                    break;
            }
            Elem<TC>::st(C + off + j, v);
        }
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

template <typename TC>
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_kernel(
    const bf16_t* __restrict__ A, long long lda, const bf16_t* __restrict__ B, long long ldb,
    TC* __restrict__ C, long long ldc, int M, int N, int K, int tiles_n, EpiArgs e) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nwg = gridDim.x;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int m0 = (tile / tiles_n) * BT_M, n0 = (tile % tiles_n) * BT_N;

    // per-lane global source pointers for the 4 A and 4 B LDS-DMA pieces this wave issues
    const bf16_t* asrc[4];
    const bf16_t* bsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);  // tile row
        const int c = (lane & 7) ^ nt_swz(r);            // global chunk stored at position lane&7
        int ga = m0 + r; ga = ga < M ? ga : M - 1;
        int gb = n0 + r; gb = gb < N ? gb : N - 1;
        asrc[i] = A + (size_t)ga * lda + c * 8;
        bsrc[i] = B + (size_t)gb * ldb + c * 8;
    }
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

    const int nk = K / BT_K;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int kt = 0; kt < nk - 1; ++kt) {
        stage(cur ^ 1, kt + 1);  // LDS-DMA of the next tile flies under this tile's MFMAs
        compute(cur);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    }
    compute(cur);

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm + i * 16 + (lane & 15);
            const int n = n0 + wn + j * 16 + 4 * (lane >> 4);
            epi_store4<TC, true>(e, C, ldc, m, n, M, N, acc[i][j]);
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
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm + i * 16 + (lane & 15);
            const int n = n0 + wn + j * 16 + 4 * (lane >> 4);
            epi_store4<float, true>(es, out, ldo, m, n, M, N, acc[i][j]);
        }
}

// C[m, n] (+)= alpha * sum_s slabs[s][m][n]   (N % 4 == 0)
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ C,
                                   long long ldc, int M, int N, int splits, float alpha,
                                   int accumulate) {
    const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total4 = (size_t)M * N / 4;
    if (i4 >= total4) return;
    const size_t idx = i4 * 4;
    const int m = (int)(idx / N), n = (int)(idx % N);
    f32x4 s = ld4(slabs + idx);
    for (int k = 1; k < splits; ++k) s += ld4(slabs + (size_t)k * M * N + idx);
    s *= alpha;
    float* c = C + (size_t)m * ldc + n;
    if (accumulate) s += ld4(c);
    st4(c, s);
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
           (!d->bias || is16(d->bias)) && (!d->aux || is16(d->aux)) &&
           (!d->out2 || is16(d->out2)) && d->M > 0 && d->N > 0;
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
    if (s > 64) s = 64;
    return s;
}

extern "C" size_t ssl4gie_gemm_workspace_bytes(const ssl4gie_gemm_desc* d) {
    if (!d) return 0;
    if (d->dtype_ab == SSL4GIE_BF16 && !nt_ok(d) && tn_ok(d)) {
        const int s = tn_splits(d);
        return s > 1 ? (size_t)s * d->M * d->N * sizeof(float) : 0;
    }
    return 0;
}

extern "C" int ssl4gie_gemm(const ssl4gie_gemm_desc* d, void* workspace, size_t workspace_bytes,
                            void* stream) {
    REQUIRE(d && d->A && d->B && d->C && d->M >= 0 && d->N >= 0 && d->K >= 0);
    REQUIRE(d->batch1 >= 1 && d->batch2 >= 1);
    REQUIRE(d->dtype_ab == SSL4GIE_F32 || d->dtype_ab == SSL4GIE_BF16);
    REQUIRE(d->dtype_c == SSL4GIE_F32 || d->dtype_c == SSL4GIE_BF16);
    REQUIRE(d->epilogue >= SSL4GIE_EPI_NONE && d->epilogue <= SSL4GIE_EPI_DGELU);
    REQUIRE(d->epilogue != SSL4GIE_EPI_BIAS || d->bias);
    REQUIRE(d->epilogue != SSL4GIE_EPI_BIAS_GELU || d->out2);
    REQUIRE(d->epilogue != SSL4GIE_EPI_BIAS_RESIDUAL || d->residual);
    REQUIRE(d->epilogue != SSL4GIE_EPI_DGELU || d->aux);
    REQUIRE(d->batch1 * d->batch2 == 1 || d->epilogue == SSL4GIE_EPI_NONE);
    REQUIRE(!d->accumulate || d->epilogue == SSL4GIE_EPI_NONE);
    if (d->M == 0 || d->N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    EpiArgs e{d->alpha, d->epilogue, d->bias, d->residual, d->ldr, d->aux, d->out2, d->accumulate};

    if (nt_ok(d)) {
        const int tm = (d->M + BT_M - 1) / BT_M, tn = (d->N + BT_N - 1) / BT_N;
        dim3 grid(tm * tn), block(256);
        ProfScope prof(PROF_GEMM_NT, 2.0 * d->M * d->N * d->K, st);
        if (d->dtype_c == SSL4GIE_BF16)
            hipLaunchKernelGGL(gemm_bf16_nt_kernel<bf16_t>, grid, block, BT_LDS_BYTES, st,
                               (const bf16_t*)d->A, d->sAm, (const bf16_t*)d->B, d->sBn,
                               (bf16_t*)d->C, d->ldc, d->M, d->N, d->K, tn, e);
        else
            hipLaunchKernelGGL(gemm_bf16_nt_kernel<float>, grid, block, BT_LDS_BYTES, st,
                               (const bf16_t*)d->A, d->sAm, (const bf16_t*)d->B, d->sBn,
                               (float*)d->C, d->ldc, d->M, d->N, d->K, tn, e);
        LAUNCH_CHECK();
        return 0;
    }
    if (tn_ok(d)) {
        const int tm = (d->M + BT_M - 1) / BT_M, tn = (d->N + BT_N - 1) / BT_N;
        const int splits = tn_splits(d);
        if (splits > 1)
            REQUIRE(workspace && workspace_bytes >= (size_t)splits * d->M * d->N * sizeof(float));
        dim3 grid(tm * tn * splits), block(256);
        ProfScope prof(PROF_GEMM_TN, 2.0 * d->M * d->N * d->K, st);
        hipLaunchKernelGGL(gemm_bf16_tn_kernel, grid, block, BT_LDS_BYTES, st,
                           (const bf16_t*)d->A, d->sAk, (const bf16_t*)d->B, d->sBk,
                           (float*)d->C, d->ldc, (float*)workspace, d->M, d->N, d->K, tn,
                           tm * tn, splits, e);
        LAUNCH_CHECK();
        if (splits > 1) {
            const size_t total4 = (size_t)d->M * d->N / 4;
            hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)),
                               dim3(256), 0, st, (const float*)workspace, (float*)d->C, d->ldc,
                               d->M, d->N, splits, d->alpha, d->accumulate);
            LAUNCH_CHECK();
        }
        return 0;
    }
    // generic
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
    return 0;
}
