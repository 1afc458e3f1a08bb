// Direct 3x3 convolutions (stride 1, pad 1) for the NARROW ends of the DPT output head on gfx950.
//
// Replaces nn.Conv2d(128, 32, 3, 1, 1) of output_conv.2 (Models/DPT_decoder.py:473-478) and its two
// gradients.  As implicit GEMMs these have N = 32 (forward), K-per-tap = 32 (data gradient) or a
// 32 x 1152 output (weight gradient): in the 256x256-tile GEMM kernels 7/8 of every tile is padding
// (forward 133 TFLOP/s, weight gradient 113 TFLOP/s, and the data gradient needs a materialised
// patch matrix — tools/dpt_head_bench.py).  Here a workgroup owns a tile of 256 output pixels instead
// (8 x 32 of one image; 16 x 16 for maps up to 16 wide; 8 x 8 of FOUR images for 7 x 7 maps — the
// geometry that wastes the fewest pixels is chosen per map):
//
//  * conv3x3_direct_kernel (forward; the data gradient is the same kernel on dy with the flipped,
//    transposed weight): the tile's input halo (10 x 34 pixels x CK channels) and the weight slice
//    [32 NCB couts][9 taps][CK] are staged ONCE into LDS, and the nine taps are nine shifted reads
//    of the same halo — each input element is written to LDS once and read nine times, where the
//    gathered GEMM re-stages it per tap.  v_mfma_f32_32x32x16_bf16 (a <- 32 couts, b <- 32 pixels
//    of one tile row): twice the flops per LDS byte of the 16x16x32 form, which is what a 32-wide
//    output needs.  A lane ends up with 4 x 4 consecutive couts of one pixel.  XOR swizzles on the
//    16-byte chunk index make both ds_read_b128 streams conflict-free (see DcLay).  Two or three
//    workgroups share a CU (58 KiB LDS each), so one stages while the others compute.
//  * conv3x3_wgrad_direct_kernel: dW[co][tap][ci] = sum_pixels dy[p][co] x[p + tap][ci].  The
//    contraction runs over pixels, so both MFMA operands are read TRANSPOSED out of pixel-major LDS
//    images with ds_read_b64_tr_b16 (k = 16 consecutive pixels of a tile row); the nine taps are
//    nine shifted reads again.  Persistent workgroups (2 per CU) each keep their share of the
//    32 x 9 x 64 accumulator block in registers across all their tiles and write ONE fp32 partial;
//    a small kernel sums the partials in a fixed order (no atomics).
#include "common.h"
#include "internal.h"
#include "gemm_internal.h"
#include "ssl4gie_hip.h"
#include "prof.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define DC_THREADS 256
#define DC_PIX 256  // output pixels per tile: 4 waves x 2 MFMA operands of 32 pixels

// Tile geometry: TB images x TH rows x TW columns = 256 pixels, in that (row-major) order; the halo
// is one pixel wider on every side, per image.
template <int TB_, int TH_, int TW_> struct TileG {
    static constexpr int TB = TB_, TH = TH_, TW = TW_, HH = TH_ + 2, HW = TW_ + 2;
    static constexpr int NH = TB_ * HH * HW;  // halo pixels
    static_assert(TB_ * TH_ * TW_ == DC_PIX && (TW_ == 32 || TW_ == 16 || TW_ == 8), "tile = 256 pixels");
    static DEVI int img(int p) { return p / (TH * TW); }
    static DEVI int row(int p) { return (p / TW) % TH; }
    static DEVI int col(int p) { return p % TW; }
    static DEVI int lin(int im, int hy, int hx) { return (im * HH + hy) * HW + hx; }  // halo pixel index
};
struct TilePos {  // which tile a workgroup owns
    int tx0, ty0, b0;
};
DEVI TilePos tile_pos(int t, int tiles_x, int tiles_y, int tb, int th, int tw) {
    TilePos q;
    q.tx0 = (t % tiles_x) * tw;
    t /= tiles_x;
    q.ty0 = (t % tiles_y) * th;
    q.b0 = (t / tiles_y) * tb;
    return q;
}

DEVI u32x4 relu_bf16x8(u32x4 v) {  // clears every 16-bit half whose sign bit is set
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned s = (v[i] >> 15) & 0x00010001u;
        o[i] = v[i] & ~((s << 16) - s);
    }
    return o;
}

// LDS images of the forward kernel (32 channels = four 16-byte chunks = 64 B per pixel).
// X: pixel-major halo; chunk c of a halo pixel sits at position c ^ f, with f chosen per geometry so
// that every ds_read_b128 of the tap loop and every staging write takes its ideal 4 LDS cycles under
// the hardware's 16-lane grouping (tools/lds_bank_sim.py, confirmed with SQ_LDS_BANK_CONFLICT = 0):
// TW = 32: f = (hx >> 2) & 3; TW = 16: f = (hx >> 1) & 3; TW = 8: f = ((hx >> 2) & 1) | (hy & 1) << 1.
// W: one row of 9 x 32 values per cout, chunk cc at cc ^ ((row >> 2) & 3): rows are 36 slots
// (4 mod 16) apart.
#define DC_CK 32
#define DC_CPP 4
#define DC_PS 64
#define DC_WROW (9 * DC_CK * 2)
template <typename G> DEVI int dc_x_off(int lin, int hy, int hx, int c) {
    const int f = G::TW == 8 ? (((hx >> 2) & 1) | ((hy & 1) << 1)) : G::TW == 16 ? ((hx >> 1) & 3) : ((hx >> 2) & 3);
    return lin * DC_PS + ((c ^ f) << 4);
}
DEVI int dc_w_off(int row, int cc) { return row * DC_WROW + ((cc ^ ((row >> 2) & 3)) << 4); }

// ---- epilogue of the forward kernels (contains workgroup barriers: call uniformly) -------------
template <int NCB, typename G>
DEVI void dc_epilogue(f32x16 (&acc)[2][NCB], char* smem, const TilePos tp, const int co0, const int tile_id,
                      const float* __restrict__ bias, const bf16_t* __restrict__ relu_mask,
                      bf16_t* __restrict__ y, float* __restrict__ colstats, const int Bn, const int H,
                      const int W, const int Cout) {
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int p_im[2], p_r[2], p_c[2];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
        const int p = (2 * wave + pb) * 32 + l31;
        p_im[pb] = G::img(p); p_r[pb] = G::row(p); p_c[pb] = G::col(p);
    }
    // ---- epilogue.  Accumulator j of a lane = cout 8 (j >> 2) + 4 half + (j & 3) of pixel l31: 8-byte
    // pieces 64-128 B apart.  The finished bf16 tile goes through LDS ([256 pixels][32 NCB couts],
    // 8-byte slots XOR-ed with (p ^ p >> 2): conflict-free writes and reads) and leaves as whole 16-byte
    // chunks of consecutive pixels; the ReLU mask is read the same way.  Pixels outside the map are
    // staged as zeros, so the BatchNorm statistics below need no mask.
    constexpr int NC = 32 * NCB, ROWB = NC * 2, SLOTS = ROWB / 8, SM = SLOTS - 1;
    __syncthreads();  // every wave is done with the operand images
    char* Os = smem;
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
        const int p = (2 * wave + pb) * 32 + l31;
        const bool valid = tp.b0 + p_im[pb] < Bn && tp.ty0 + p_r[pb] < H && tp.tx0 + p_c[pb] < W;
#pragma unroll
        for (int n = 0; n < NCB; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int slot = n * 8 + 2 * q + half, co = co0 + slot * 4;
                f32x4 v = {acc[pb][n][4 * q], acc[pb][n][4 * q + 1], acc[pb][n][4 * q + 2],
                           acc[pb][n][4 * q + 3]};
                if (bias && co < Cout) v += ld4(bias + co);
                if (!valid || co >= Cout) v = f32x4{0, 0, 0, 0};
                st4((bf16_t*)(Os + p * ROWB + ((slot ^ ((p ^ (p >> 2)) & SM)) << 3)), v);
            }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SLOTS / 2; ++i) {
        const int idx = tid + i * DC_THREADS, p = idx / (SLOTS / 2), j = idx % (SLOTS / 2);
        const int sw = (p ^ (p >> 2)) & SM;
        u32x4 v = *(const u32x4*)(Os + p * ROWB + ((j ^ (sw >> 1)) << 4));
        if (sw & 1) v = u32x4{v[2], v[3], v[0], v[1]};  // the two slots of the chunk sit swapped
        const int gb = tp.b0 + G::img(p), gy = tp.ty0 + G::row(p), gx = tp.tx0 + G::col(p), co = co0 + 8 * j;
        if (gb < Bn && gy < H && gx < W && co < Cout) {
            const size_t o = (((size_t)gb * H + gy) * W + gx) * Cout + co;
            if (relu_mask) {  // keep where mask > 0: sign bit clear and not zero
                const u32x4 m = *(const u32x4*)(relu_mask + o);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned lo = ((m[k] & 0x8000u) == 0 && (m[k] & 0x7fffu) != 0) ? 0xffffu : 0u;
                    const unsigned hi = ((m[k] & 0x80000000u) == 0 && (m[k] & 0x7fff0000u) != 0) ? 0xffff0000u : 0u;
                    v[k] &= lo | hi;
                }
            }
            *(u32x4*)(y + o) = v;
        }
    }
    if (colstats) {
        // per-tile column sums and sums of squares of the STORED values (ssl4gie_gemm_desc.colstats
        // semantics with one partial per tile): thread = (cout c, one of 256 / NC pixel groups)
        constexpr int NG = DC_THREADS / NC, PPG = DC_PIX / NG;
        float* red = (float*)(smem + DC_PIX * ROWB);  // [NG][2][NC]
        const int c = tid % NC, g = tid / NC;
        float sum = 0.f, sq = 0.f;
        for (int p = g * PPG; p < (g + 1) * PPG; ++p) {
            const float v = bf2f(*(const bf16_t*)(Os + p * ROWB + (((c >> 2) ^ ((p ^ (p >> 2)) & SM)) << 3) + ((c & 3) << 1)));
            sum += v;
            sq += v * v;
        }
        red[(g * 2) * NC + c] = sum;
        red[(g * 2 + 1) * NC + c] = sq;
        __syncthreads();
        if (tid < NC && co0 + tid < Cout) {
            sum = 0.f, sq = 0.f;
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                sum += red[(k * 2) * NC + tid];
                sq += red[(k * 2 + 1) * NC + tid];
            }
            float* o = colstats + (size_t)tile_id * 2 * Cout + co0 + tid;
            o[0] = sum;
            o[Cout] = sq;
        }
    }
}

// y[b, oy, ox, co] = sum_{tap, ci} act(x)[b, oy + dy - 1, ox + dx - 1, ci] w2[co, tap * Cin + ci]
// (+ bias[co]) (masked by relu_mask > 0); colstats [tiles][2][Cout]: per-tile sums / sums of squares
// of y.  grid = tiles * ngroups, ngroups = ceil(Cout / (32 NCB)).
// AFF: act(x) = act(x in_coef[0][ci] + in_coef[1][ci]) rounded to bf16 (act = ReLU if relu_in), applied to the
// in-image halo pixels between their global load and their LDS write — the training-mode BatchNorm (+ ReLU) in
// front of the convolution (torchvision Bottleneck bn1 -> relu -> conv2) without a pass of its own: the
// normalised map is never written.  Padding stays zero (it pads the NORMALISED map).
template <int NCB, typename G, bool AFF = false>
__global__ __launch_bounds__(DC_THREADS, (NCB == 1 && !AFF) ? 3 : 2) void conv3x3_direct_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ w2, const float* __restrict__ bias,
    const bf16_t* __restrict__ relu_mask, bf16_t* __restrict__ y, float* __restrict__ colstats, int Bn,
    int H, int W, int Cin, int Cout, int relu_in, int tiles_x, int tiles_y, int ngroups,
    const float* __restrict__ in_coef) {
    constexpr int CPP = DC_CPP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Xs = smem;
    char* Ws = smem + G::NH * DC_PS;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the cout groups of a tile are neighbours in an XCD's share of the grid (common.h xcd_remap), so
    // that the halo they all read is fetched into that XCD's L2 once
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_id = lid / ngroups;
    const TilePos tp = tile_pos(tile_id, tiles_x, tiles_y, G::TB, G::TH, G::TW);
    const int co0 = (lid % ngroups) * (32 * NCB);

    f32x16 acc[2][NCB];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int n = 0; n < NCB; ++n)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[pb][n][j] = 0.f;

    // this lane's two pixels (one per 32-pixel operand of the wave) in halo coordinates
    int p_im[2], p_r[2], p_c[2];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
        const int p = (2 * wave + pb) * 32 + l31;
        p_im[pb] = G::img(p); p_r[pb] = G::row(p); p_c[pb] = G::col(p);
    }

    constexpr int NX = G::NH * CPP, IX = (NX + DC_THREADS - 1) / DC_THREADS;
    constexpr int NW = NCB * 32 * 9 * CPP, IW = (NW + DC_THREADS - 1) / DC_THREADS;
    // AFF: the coefficient table [2][Cin] sits in LDS behind the operand images for the whole kernel (a global
    // fetch per channel slice between the barrier and the halo writes would expose an L2 round trip per slice)
    [[maybe_unused]] float* Cs = (float*)(smem + G::NH * DC_PS + NCB * 32 * DC_WROW);
    if constexpr (AFF) {
        for (int i = tid; i < 2 * Cin; i += DC_THREADS) Cs[i] = in_coef[i];
        // (visible after the first __syncthreads below: every thread passes one before its first halo write)
    }
    for (int cin0 = 0; cin0 < Cin; cin0 += DC_CK) {
        // ---- stage the halo and the weight slice: all global loads first, then the LDS writes
        u32x4 xv[IX], wv[IW];
        [[maybe_unused]] unsigned inimg = 0;  // AFF: which of this thread's halo chunks lie inside the image
#pragma unroll
        for (int i = 0; i < IX; ++i) {
            const int idx = tid + i * DC_THREADS, pi = idx / CPP, c = idx % CPP;
            const int im = pi / (G::HH * G::HW), hy = (pi / G::HW) % G::HH, hx = pi % G::HW;
            const int gy = tp.ty0 + hy - 1, gx = tp.tx0 + hx - 1, gb = tp.b0 + im;
            xv[i] = u32x4{0, 0, 0, 0};
            if (idx < NX && gb < Bn && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                xv[i] = *(const u32x4*)(x + (((size_t)gb * H + gy) * W + gx) * Cin + cin0 + c * 8);
                if constexpr (AFF) inimg |= 1u << i;
            }
        }
#pragma unroll
        for (int i = 0; i < IW; ++i) {
            const int idx = tid + i * DC_THREADS, row = idx / (9 * CPP), cc = idx % (9 * CPP);
            const int tap = cc / CPP, c = cc % CPP;
            wv[i] = u32x4{0, 0, 0, 0};
            if (idx < NW && co0 + row < Cout)
                wv[i] = *(const u32x4*)(w2 + (size_t)(co0 + row) * 9 * Cin + tap * Cin + cin0 + c * 8);
        }
        if (cin0 || AFF) __syncthreads();  // the previous channel slice has been consumed (AFF: the table is in place)
        if constexpr (AFF) {
            // this thread's chunk index is the same for all its halo chunks (DC_THREADS % CPP == 0): one
            // coefficient set per channel slice, read here so that it is live only across the x writes
            static_assert(DC_THREADS % CPP == 0, "one channel chunk per thread");
            const float* ca = Cs + cin0 + (tid % CPP) * 8;
            const f32x4 a8[2] = {*(const f32x4*)ca, *(const f32x4*)(ca + 4)};
            const f32x4 b8[2] = {*(const f32x4*)(ca + Cin), *(const f32x4*)(ca + Cin + 4)};
#pragma unroll
            for (int i = 0; i < IX; ++i) {
                const int idx = tid + i * DC_THREADS, pi = idx / CPP, c = idx % CPP;
                if (idx < NX)
                    *(u32x4*)(Xs + dc_x_off<G>(pi, (pi / G::HW) % G::HH, pi % G::HW, c)) =
                        ((inimg >> i) & 1) ? bn_affine_act8(xv[i], a8, b8, relu_in) : xv[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < IX; ++i) {
                const int idx = tid + i * DC_THREADS, pi = idx / CPP, c = idx % CPP;
                if (idx < NX)
                    *(u32x4*)(Xs + dc_x_off<G>(pi, (pi / G::HW) % G::HH, pi % G::HW, c)) =
                        relu_in ? relu_bf16x8(xv[i]) : xv[i];
            }
        }
#pragma unroll
        for (int i = 0; i < IW; ++i) {
            const int idx = tid + i * DC_THREADS;
            if (idx < NW) *(u32x4*)(Ws + dc_w_off(idx / (9 * CPP), idx % (9 * CPP))) = wv[i];
        }
        __syncthreads();
        // ---- nine shifted reads of the same halo
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int ks = 0; ks < DC_CK / 16; ++ks) {
                const int c = ks * 2 + half;
                bf16x8 a[NCB];
#pragma unroll
                for (int n = 0; n < NCB; ++n)
                    a[n] = *(const bf16x8*)(Ws + dc_w_off(n * 32 + l31, tap * CPP + c));
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    const int hy = p_r[pb] + dy, hx = p_c[pb] + dx;
                    const bf16x8 xf = *(const bf16x8*)(Xs + dc_x_off<G>(G::lin(p_im[pb], hy, hx), hy, hx, c));
#pragma unroll
                    for (int n = 0; n < NCB; ++n) acc[pb][n] = MFMA32(a[n], xf, acc[pb][n]);
                }
            }
        }
    }
    dc_epilogue<NCB, G>(acc, smem, tp, co0, tile_id, bias, relu_mask, y, colstats, Bn, H, W, Cout);
}

// ------------------------------------------------------------------ weight gradient
// One workgroup = (64 input channels, 32 couts, a strided share of the tiles).
// Wave w: input-channel block cb = w >> 1 (32 channels), taps 0-4 (w & 1 == 0) or 5-8.
// LDS: X halo [halo pixels][64 ch] (the 64-byte halves of a pixel swapped by (hx >> 1) & 1, which
// keeps the four consecutive pixels of a transposed read on different banks for every tap shift)
// and the dy tile [256 pixels][32 couts].  k-step = a run of 16 consecutive tile pixels: half a row
// (TW = 32), a row (16) or two rows (8).
#define WG_DY_BYTES (DC_PIX * 64)
DEVI int wg_x_off(int lin, int hx, int ch) {  // byte offset of channel ch (0..63) of a halo pixel
    return lin * 128 + ((((ch >> 5) ^ (hx >> 1)) & 1) << 6) + ((ch & 31) << 1);
}
// transposed 32 x 16 MFMA operand: lane (n = lane & 31, half) gets, for column n of a 32-column block,
// the 8 k-rows (pixels) 8 half .. 8 half + 7 of a run.  Each 16-lane group transposes a 4 (k) x 16
// (columns) block per ds_read_b64_tr_b16: lane i supplies the address of row i >> 2, columns
// 4 (i & 3) .. + 3, and receives column i.  `p` = this lane's address in the first block (row
// 8 half + (i >> 2)), `rs` = bytes per pixel: the second block is 4 pixels further on (same map row).
DEVI bf16x8 tr_operand32(const char* p, const int rs) {
    typedef __attribute__((address_space(3))) s16x4* lp_t;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(p + 4 * rs));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// one staged tile: 16 runs x the wave's taps.  Fully unrolled with compile-time taps: every LDS
// address is a per-lane base (dbase; xbase[dx], which carries the lane's pixel inside a run and the
// swizzle of its column) plus an immediate (the run's first pixel, shifted by the tap).
// BIAS (one four-tap wave of the slice-0 workgroups): the spare fifth accumulator takes the bias
// gradient — dy^T against a ones operand puts sum_pixels dy[p][co] into every column of row co.
template <typename G, int TAP0, int NTAP, bool BIAS>
DEVI void wgrad_tile(f32x16 (&acc)[5], const char* dbase, const char* const (&xbase)[3]) {
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;
#pragma unroll
    for (int u = 0; u < DC_PIX / 16; ++u) {
        const int p0 = u * 16, im = G::img(p0), r = G::row(p0), c0 = G::col(p0);
        const bf16x8 a = tr_operand32(dbase + p0 * 64, 64);
        if (BIAS) acc[4] = MFMA32(a, ones, acc[4]);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            const int tap = TAP0 + t, dyy = tap / 3, dxx = tap % 3;
            const bf16x8 xf = tr_operand32(xbase[dxx] + G::lin(im, r + dyy, c0) * 128, 128);
            acc[t] = MFMA32(a, xf, acc[t]);
        }
    }
}

// partial[wg][co 32][tap 9][ci 64] fp32, then partial_b[wg][co 32] (written by the workgroups of
// slice 0 only); wg = the workgroup's logical id (xcd_remap of blockIdx.x): combo = wg % (nslice * ngroups) = group * nslice + slice picks the
// 64 input channels and the 32 couts, wg / ncombo the share of the tiles
template <typename G, bool AFF = false>
__global__ __launch_bounds__(DC_THREADS, 2) void conv3x3_wgrad_direct_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, float* __restrict__ partial,
    float* __restrict__ partial_b,
    int Bn, int H, int W, int Cin, int Cout, int relu_in, int tiles_x, int tiles_y, int ntiles,
    int nslice, int ncombo, const float* __restrict__ in_coef) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Xs = smem;
    char* Ds = smem + G::NH * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the (slice, group) combinations that walk the same tiles are neighbours in an XCD's share of
    // the grid: the x halo slice is shared by the cout groups, the dy tile by the channel slices
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int combo = wg % ncombo, slice = combo % nslice, cin0 = slice * 64;
    const int co0 = (combo / nslice) * 32;
    const int wgs_per_combo = gridDim.x / ncombo, me = wg / ncombo;
    const int cb = wave >> 1, tap0 = (wave & 1) * 5, ntap = (wave & 1) ? 4 : 5;  // wgrad_tile<tap0, ntap>

    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

    // per-lane LDS bases of the transposed reads: pixel k0 = 8 half + (i >> 2) of a 16-pixel run
    // (TW = 8: map row k0 >> 3, column k0 & 7), columns 4 (i & 3) .. of the lane group's 16-column
    // half.  wg_x_off's swizzle depends on the pixel column only through (kc + dx) >> 1: runs start
    // at column 0 or 16.
    const int li = lane & 15, k0 = 8 * (lane >> 5) + (li >> 2), colb = (((lane >> 4) & 1) * 16 + 4 * (li & 3)) * 2;
    const int kr = G::TW == 8 ? (k0 >> 3) : 0, kc = G::TW == 8 ? (k0 & 7) : k0;
    const char* dbase = Ds + k0 * 64 + colb;
    const char* const xbase[3] = {Xs + wg_x_off(G::lin(0, kr, kc), kc, cb * 32) + colb,
                                  Xs + wg_x_off(G::lin(0, kr, kc + 1), kc + 1, cb * 32) + colb,
                                  Xs + wg_x_off(G::lin(0, kr, kc + 2), kc + 2, cb * 32) + colb};
    constexpr int NX = G::NH * 8, IX = (NX + DC_THREADS - 1) / DC_THREADS;  // 16-byte chunks
    constexpr int ND = DC_PIX * 4, ID = ND / DC_THREADS;
    [[maybe_unused]] float* Cs = (float*)(smem + G::NH * 128 + WG_DY_BYTES);  // AFF: this slice's [2][64] coefficients
    if constexpr (AFF) {
        if (tid < 128) Cs[tid] = in_coef[(tid >> 6) * Cin + cin0 + (tid & 63)];
    }
    for (int tile = me; tile < ntiles; tile += wgs_per_combo) {
        const TilePos tp = tile_pos(tile, tiles_x, tiles_y, G::TB, G::TH, G::TW);
        u32x4 xv[IX], dv[ID];
        [[maybe_unused]] unsigned inimg = 0;
#pragma unroll
        for (int i = 0; i < IX; ++i) {
            const int idx = tid + i * DC_THREADS, pi = idx >> 3, c = idx & 7;
            const int im = pi / (G::HH * G::HW), hy = (pi / G::HW) % G::HH, hx = pi % G::HW;
            const int gy = tp.ty0 + hy - 1, gx = tp.tx0 + hx - 1, gb = tp.b0 + im;
            xv[i] = u32x4{0, 0, 0, 0};
            if (idx < NX && gb < Bn && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                xv[i] = *(const u32x4*)(x + (((size_t)gb * H + gy) * W + gx) * Cin + cin0 + c * 8);
                if constexpr (AFF) inimg |= 1u << i;
            }
        }
#pragma unroll
        for (int i = 0; i < ID; ++i) {  // dy tile in tile-pixel order, 4 chunks of 8 couts
            const int idx = tid + i * DC_THREADS, p = idx >> 2, c = idx & 3;
            const int gb = tp.b0 + G::img(p), gy = tp.ty0 + G::row(p), gx = tp.tx0 + G::col(p);
            dv[i] = u32x4{0, 0, 0, 0};
            if (gb < Bn && gy < H && gx < W)
                dv[i] = *(const u32x4*)(dy + (((size_t)gb * H + gy) * W + gx) * Cout + co0 + c * 8);
        }
        __syncthreads();  // the previous tile has been consumed
        if constexpr (AFF) {  // as conv3x3_direct_kernel<.., AFF>: the operand is the normalised map, rebuilt on the way in
            const float* ca = Cs + (tid & 7) * 8;
            const f32x4 a8[2] = {*(const f32x4*)ca, *(const f32x4*)(ca + 4)};
            const f32x4 b8[2] = {*(const f32x4*)(ca + 64), *(const f32x4*)(ca + 68)};
#pragma unroll
            for (int i = 0; i < IX; ++i) {
                const int idx = tid + i * DC_THREADS, pi = idx >> 3, c = idx & 7;
                if (idx < NX)
                    *(u32x4*)(Xs + wg_x_off(pi, pi % G::HW, c * 8)) =
                        ((inimg >> i) & 1) ? bn_affine_act8(xv[i], a8, b8, relu_in) : xv[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < IX; ++i) {
                const int idx = tid + i * DC_THREADS, pi = idx >> 3, c = idx & 7;
                if (idx < NX)
                    *(u32x4*)(Xs + wg_x_off(pi, pi % G::HW, c * 8)) = relu_in ? relu_bf16x8(xv[i]) : xv[i];
            }
        }
#pragma unroll
        for (int i = 0; i < ID; ++i) {
            const int idx = tid + i * DC_THREADS;
            *(u32x4*)(Ds + (idx >> 2) * 64 + ((idx & 3) << 4)) = dv[i];
        }
        __syncthreads();
        if (wave == 1 && slice == 0) wgrad_tile<G, 5, 4, true>(acc, dbase, xbase);
        else if (wave & 1) wgrad_tile<G, 5, 4, false>(acc, dbase, xbase);
        else wgrad_tile<G, 0, 5, false>(acc, dbase, xbase);
    }
    // acc[t][j]: row (cout) = 8 (j >> 2) + 4 half + (j & 3), column (ci) = cb * 32 + (lane & 31)
    float* out = partial + (size_t)wg * (32 * 9 * 64);
    const int half = lane >> 5, ci = cb * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < 5; ++t)
        if (t < ntap)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int co = 8 * (j >> 2) + 4 * half + (j & 3);
                out[(co * 9 + tap0 + t) * 64 + ci] = acc[t][j];
            }
    if (wave == 1 && slice == 0 && (lane & 31) == 0)
#pragma unroll
        for (int j = 0; j < 16; ++j) partial_b[wg * 32 + 8 * (j >> 2) + 4 * half + (j & 3)] = acc[4][j];
}

// dW2[co, tap * Cin + ci] (+)= sum over the workgroups of (co's group, ci's slice), in workgroup
// order; the last Cout threads do the same for dbias[co] over the slice-0 workgroups of co's group
__global__ void conv3x3_wgrad_reduce_kernel(const float* __restrict__ partial,
                                            const float* __restrict__ partial_b, float* __restrict__ dw2,
                                            float* __restrict__ dbias, int Cin, int Cout, int nslice,
                                            int ncombo, int wgs_per_combo, int accumulate) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over Cout * 9 * Cin (+ Cout)
    const int nw = Cout * 9 * Cin;
    if (idx >= nw) {
        const int co = idx - nw;
        if (co < Cout && dbias) {
            float s = 0.f;
            const float* src = partial_b + (size_t)((co >> 5) * nslice) * 32 + (co & 31);
            int g = 0;
            for (; g + 7 < wgs_per_combo; g += 8) {  // loads in flight, adds in workgroup order (as below)
                float a[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = src[(size_t)(g + u) * ncombo * 32];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += a[u];
            }
            for (; g < wgs_per_combo; ++g) s += src[(size_t)g * ncombo * 32];
            dbias[co] = accumulate ? dbias[co] + s : s;
        }
        return;
    }
    const int ci = idx % Cin, tap = (idx / Cin) % 9, co = idx / (9 * Cin);
    const int combo = (co >> 5) * nslice + (ci >> 6), cl = ci & 63;
    float s = 0.f;
    const float* src = partial + (size_t)combo * (32 * 9 * 64) + ((co & 31) * 9 + tap) * 64 + cl;
    const size_t gs = (size_t)ncombo * (32 * 9 * 64);
    int g = 0;
    for (; g + 7 < wgs_per_combo; g += 8) {  // eight workgroups' partials in flight, added in workgroup order
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = src[(size_t)(g + u) * gs];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += a[u];
    }
    for (; g < wgs_per_combo; ++g) s += src[(size_t)g * gs];
    float* o = dw2 + (size_t)co * 9 * Cin + tap * Cin + ci;
    *o = accumulate ? *o + s : s;
}

// ------------------------------------------------------------------ 7x7 stride-2 stem
// torchvision ResNet.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False) (reference Models/models.py:63-69)
// without a patch matrix.  The fp32 NCHW image is packed ONCE per step into bf16 [B, Hp, Wp, 4]
// (three channels + a zero, three pixels / rows of zero padding in front, Hp = 2 Ho + 6,
// Wp = 2 Wo + 6): an output pixel's patch is then 7 image rows x 8 pixels x 4 channels, every MFMA
// k-chunk (2 pixels) a 16-byte read at a 16-byte-aligned address (the window starts at the even
// column 2 ox), and 32 consecutive output pixels read 512 contiguous bytes: no swizzle needed.
// Weights [64][8][8][4] bf16 (row 7, pixel 7, channel 3 zero).  Same tiles, MFMA shape and epilogue
// (incl. the BatchNorm partial statistics) as conv3x3_direct_kernel<2, G>.
#define ST_WROW 512
template <typename G> struct StemG {
    static constexpr int RH = 2 * G::TH + 6, RW = 2 * G::TW + 6, XS_BYTES = RH * RW * 8;
    static_assert(G::TB == 1, "one image per tile");
};
__global__ void stem7x7_pack_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int H, int W,
                                    int Hp, int Wp, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % Wp), yy = (int)((idx / Wp) % Hp);
    const long long b = idx / ((long long)Wp * Hp);
    const int iy = yy - 3, ix = x - 3;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
        const float* p = img + ((size_t)b * 3 * H + iy) * W + ix;
        c0 = p[0]; c1 = p[(size_t)H * W]; c2 = p[(size_t)2 * H * W];
    }
    u32x2 v;
    v[0] = pack_bf2(c0, c1);
    v[1] = pack_bf2(c2, 0.f);
    *(u32x2*)(out + idx * 4) = v;
}
template <typename G>
DEVI void stem_stage_x(char* Xs, const bf16_t* __restrict__ P, const TilePos tp, int Hp, int Wp, int tid) {
    using S = StemG<G>;
    constexpr int NC = S::RH * S::RW / 2, IT = (NC + DC_THREADS - 1) / DC_THREADS;  // 16-byte chunks = 2 pixels
    u32x4 v[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int idx = tid + i * DC_THREADS, hy = idx / (S::RW / 2), hx = 2 * (idx % (S::RW / 2));
        const int gy = 2 * tp.ty0 + hy, gx = 2 * tp.tx0 + hx;
        v[i] = u32x4{0, 0, 0, 0};
        if (idx < NC && gy < Hp && gx < Wp) v[i] = *(const u32x4*)(P + (((size_t)tp.b0 * Hp + gy) * Wp + gx) * 4);
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int idx = tid + i * DC_THREADS;
        if (idx < NC) *(u32x4*)(Xs + (size_t)idx * 16) = v[i];
    }
}
template <typename G>
__global__ __launch_bounds__(DC_THREADS, 2) void stem7x7_direct_kernel(
    const bf16_t* __restrict__ P, const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
    float* __restrict__ colstats, int Bn, int Ho, int Wo, int Hp, int Wp, int tiles_x, int tiles_y) {
    using S = StemG<G>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Xs = smem;
    char* Ws = smem + S::XS_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile_id = xcd_remap(blockIdx.x, gridDim.x);
    const TilePos tp = tile_pos(tile_id, tiles_x, tiles_y, 1, G::TH, G::TW);
    f32x16 acc[2][2];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[pb][n][j] = 0.f;
    {
        u32x4 wv[8];  // 64 rows x 32 chunks
#pragma unroll
        for (int i = 0; i < 8; ++i) wv[i] = *(const u32x4*)(w + (size_t)(tid + i * DC_THREADS) * 8);
        stem_stage_x<G>(Xs, P, tp, Hp, Wp, tid);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + i * DC_THREADS, row = idx >> 5, cc = idx & 31;
            *(u32x4*)(Ws + row * ST_WROW + ((cc ^ (row & 15)) << 4)) = wv[i];
        }
    }
    __syncthreads();
    int xo[2];  // this lane's two output pixels: byte offset of their window's first pixel
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
        const int p = (2 * wave + pb) * 32 + l31;
        xo[pb] = (2 * G::row(p) * S::RW + 2 * G::col(p) + 2 * half) * 8;
    }
#pragma unroll
    for (int dy = 0; dy < 7; ++dy)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int cc = dy * 4 + ks * 2 + half;
            bf16x8 a[2];
#pragma unroll
            for (int n = 0; n < 2; ++n)
                a[n] = *(const bf16x8*)(Ws + (n * 32 + l31) * ST_WROW + ((cc ^ (l31 & 15)) << 4));
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const bf16x8 xf = *(const bf16x8*)(Xs + xo[pb] + (dy * S::RW + 4 * ks) * 8);
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[pb][n] = MFMA32(a[n], xf, acc[pb][n]);
            }
        }
    dc_epilogue<2, G>(acc, smem, tp, 0, tile_id, nullptr, nullptr, y, colstats, Bn, Ho, Wo, 64);
}
// dW[co][dy][px * 4 + c] = sum over output pixels of dy_out[p][co] P[2 oy + dy][2 ox + px][c]; wave w:
// cout block w & 1, image rows dy = (w >> 1) + 2 t.  partial[wg][64][7][32] fp32.
template <typename G>
__global__ __launch_bounds__(DC_THREADS, 2) void stem7x7_wgrad_kernel(
    const bf16_t* __restrict__ dyo, const bf16_t* __restrict__ P, float* __restrict__ partial, int Bn,
    int Ho, int Wo, int Hp, int Wp, int tiles_x, int tiles_y, int ntiles) {
    using S = StemG<G>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Xs = smem;
    char* Ds = smem + S::XS_BYTES;  // two images [256 pixels][32 couts]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int cbk = wave & 1, dy0 = wave >> 1, nrow = dy0 ? 3 : 4;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    const int li = lane & 15, k0 = 8 * (lane >> 5) + (li >> 2), colb = (((lane >> 4) & 1) * 16 + 4 * (li & 3)) * 2;
    const char* dbase = Ds + cbk * WG_DY_BYTES + k0 * 64 + colb;
    const char* xbase = Xs + 2 * k0 * 8 + colb;
    constexpr int ND = DC_PIX * 8, ID = ND / DC_THREADS;
    for (int tile = wg; tile < ntiles; tile += gridDim.x) {
        const TilePos tp = tile_pos(tile, tiles_x, tiles_y, 1, G::TH, G::TW);
        u32x4 dv[ID];
#pragma unroll
        for (int i = 0; i < ID; ++i) {
            const int idx = tid + i * DC_THREADS, p = idx >> 3, c = idx & 7;
            const int gy = tp.ty0 + G::row(p), gx = tp.tx0 + G::col(p);
            dv[i] = u32x4{0, 0, 0, 0};
            if (gy < Ho && gx < Wo) dv[i] = *(const u32x4*)(dyo + (((size_t)tp.b0 * Ho + gy) * Wo + gx) * 64 + c * 8);
        }
        __syncthreads();  // the previous tile has been consumed
        stem_stage_x<G>(Xs, P, tp, Hp, Wp, tid);
#pragma unroll
        for (int i = 0; i < ID; ++i) {
            const int idx = tid + i * DC_THREADS, p = idx >> 3, c = idx & 7;
            *(u32x4*)(Ds + (c >> 2) * WG_DY_BYTES + p * 64 + ((c & 3) << 4)) = dv[i];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < DC_PIX / 16; ++u) {
            const int p0 = u * 16, r = G::row(p0), c0 = G::col(p0);
            const bf16x8 a = tr_operand32(dbase + p0 * 64, 64);
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < nrow) {
                    const int dyy = dy0 + 2 * t;  // wave-uniform
                    const bf16x8 xf = tr_operand32(xbase + ((2 * r + dyy) * S::RW + 2 * c0) * 8, 16);
                    acc[t] = MFMA32(a, xf, acc[t]);
                }
        }
    }
    float* out = partial + (size_t)wg * (64 * 7 * 32);
    const int half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (t < nrow)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int co = cbk * 32 + 8 * (j >> 2) + 4 * half + (j & 3);
                out[(co * 7 + dy0 + 2 * t) * 32 + col] = acc[t][j];
            }
}

// ------------------------------------------------------------------ C ABI
typedef TileG<1, 8, 32> G32;
typedef TileG<1, 16, 16> G16;
typedef TileG<4, 8, 8> G8;
struct Geom {
    int id, tb, th, tw, tiles_x, tiles_y, tiles;
};
// the geometry with the fewest tiles (every tile costs the same): 8 x 32 unless the map is narrow
static Geom pick_geom(int B, int H, int W) {
    static const int G_[3][3] = {{1, 8, 32}, {1, 16, 16}, {4, 8, 8}};
    Geom best = {};
    for (int i = 0; i < 3; ++i) {
        Geom g = {i, G_[i][0], G_[i][1], G_[i][2], 0, 0, 0};
        g.tiles_x = (W + g.tw - 1) / g.tw;
        g.tiles_y = (H + g.th - 1) / g.th;
        g.tiles = g.tiles_x * g.tiles_y * ((B + g.tb - 1) / g.tb);
        if (i == 0 || g.tiles < best.tiles) best = g;
    }
    return best;
}

// 32 staged channels per pass and up to 64 couts per workgroup: 58 KiB of LDS, so two to three
// workgroups overlap their staging on a CU.  (64-channel passes with 32 couts — 80 KiB — measured
// slower at every geometry tried: 128 -> 32 @224 746 vs 666 us, 256 -> 128 @112 1532 vs 1227 us.)
extern "C" int ssl4gie_conv3x3_direct_ok(int B, int H, int W, int Cin, int Cout) {
    return B > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 8 == 0 && Cin > 0 && Cin % DC_CK == 0 &&
           (long long)B * H * W * (Cin > Cout ? Cin : Cout) * 2 < (1LL << 40);
}

extern "C" int ssl4gie_conv3x3_direct_tiles(int B, int H, int W) {
    return (B > 0 && H > 0 && W > 0) ? pick_geom(B, H, W).tiles : 0;
}

template <int NCB, typename G, bool AFF = false>
static int launch_direct(const Geom& g, const void* x, const void* w2, const float* bias,
                         const void* relu_mask, void* y, float* colstats, int B, int H, int W, int Cin,
                         int Cout, int relu_in, hipStream_t st, const float* in_coef = nullptr) {
    auto k = conv3x3_direct_kernel<NCB, G, AFF>;
    // AFF: + the coefficient table [2][Cin] (Cin <= 2048 by the REQUIRE of the entry point: 16 KiB)
    constexpr int lds_max = G::NH * DC_PS + NCB * 32 * DC_WROW + (AFF ? 2 * 2048 * (int)sizeof(float) : 0);
    int lds = G::NH * DC_PS + NCB * 32 * DC_WROW + (AFF ? 2 * Cin * (int)sizeof(float) : 0);
    const int lds_out = DC_PIX * NCB * 64 + 2 * DC_THREADS * (int)sizeof(float);
    if (lds < lds_out) lds = lds_out;
    static bool attr = false;  // one per instantiation
    if (!attr) {
        HIP_RET(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    lds_max > lds_out ? lds_max : lds_out));
        attr = true;
    }
    const int ngroups = (Cout + 32 * NCB - 1) / (32 * NCB);
    REQUIRE((long long)g.tiles * ngroups < (1LL << 31));
    hipLaunchKernelGGL(k, dim3((unsigned)(g.tiles * ngroups)), dim3(DC_THREADS), lds, st, (const bf16_t*)x,
                       (const bf16_t*)w2, bias, (const bf16_t*)relu_mask, (bf16_t*)y, colstats, B, H, W, Cin,
                       Cout, relu_in, g.tiles_x, g.tiles_y, ngroups, in_coef);
    LAUNCH_CHECK();
    return 0;
}

// the same convolution over act(x in_coef[0][ci] + in_coef[1][ci]) (in_coef [2][Cin] as ssl4gie_bn_coef_partials
// writes it; act = ReLU if relu_in; zero padding of the NORMALISED map): Bottleneck bn1 -> relu -> conv2 with no
// BatchNorm pass of its own
extern "C" int ssl4gie_conv3x3_direct_fwd_affine(const void* x, const float* in_coef, const void* w2,
                                                 const float* bias, void* y, float* colstats, int B, int H,
                                                 int W, int Cin, int Cout, int relu_in, void* stream) {
    REQUIRE(x && in_coef && w2 && y && ssl4gie_conv3x3_direct_ok(B, H, W, Cin, Cout) && Cin <= 2048);
    hipStream_t st = (hipStream_t)stream;
    const Geom g = pick_geom(B, H, W);
    ProfScope prof(PROF_GEMM_NT, 2.0 * B * H * (double)W * Cout * 9 * Cin, st);
#define DC_ARGS g, x, w2, bias, nullptr, y, colstats, B, H, W, Cin, Cout, relu_in, st, in_coef
    if (Cout > 32) {
        if (g.id == 0) return launch_direct<2, G32, true>(DC_ARGS);
        if (g.id == 1) return launch_direct<2, G16, true>(DC_ARGS);
        return launch_direct<2, G8, true>(DC_ARGS);
    }
    if (g.id == 0) return launch_direct<1, G32, true>(DC_ARGS);
    if (g.id == 1) return launch_direct<1, G16, true>(DC_ARGS);
    return launch_direct<1, G8, true>(DC_ARGS);
#undef DC_ARGS
}

extern "C" int ssl4gie_conv3x3_direct_fwd(const void* x, const void* w2, const float* bias,
                                          const void* relu_mask, void* y, float* colstats, int B, int H,
                                          int W, int Cin, int Cout, int relu_in, void* stream) {
    REQUIRE(x && w2 && y && ssl4gie_conv3x3_direct_ok(B, H, W, Cin, Cout));
    REQUIRE(!(colstats && relu_mask));
    hipStream_t st = (hipStream_t)stream;
    const Geom g = pick_geom(B, H, W);
    ProfScope prof(PROF_GEMM_NT, 2.0 * B * H * (double)W * Cout * 9 * Cin, st);
#define DC_ARGS g, x, w2, bias, relu_mask, y, colstats, B, H, W, Cin, Cout, relu_in, st
    if (Cout > 32) {
        if (g.id == 0) return launch_direct<2, G32>(DC_ARGS);
        if (g.id == 1) return launch_direct<2, G16>(DC_ARGS);
        return launch_direct<2, G8>(DC_ARGS);
    }
    if (g.id == 0) return launch_direct<1, G32>(DC_ARGS);
    if (g.id == 1) return launch_direct<1, G16>(DC_ARGS);
    return launch_direct<1, G8>(DC_ARGS);
#undef DC_ARGS
}

static int wgrad_grid(const Geom& g, int Cin, int Cout, int* nslice, int* ncombo, int* per_combo) {
    *nslice = Cin / 64;
    *ncombo = *nslice * (Cout / 32);
    int per = (2 * ssl4gie_internal_compute_cus()) / *ncombo;  // two workgroups per CU in all
    if (per < 1) per = 1;
    if (per > g.tiles) per = g.tiles;
    *per_combo = per;
    return *ncombo * per;
}

extern "C" int ssl4gie_conv3x3_direct_wgrad_ok(int B, int H, int W, int Cin, int Cout) {
    return B > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 32 == 0 && Cout <= 512 && Cin > 0 &&
           Cin % 64 == 0 && Cin <= 512;
}

extern "C" size_t ssl4gie_conv3x3_direct_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (!ssl4gie_conv3x3_direct_wgrad_ok(B, H, W, Cin, Cout)) return 0;
    int ns, nc, per;
    return (size_t)wgrad_grid(pick_geom(B, H, W), Cin, Cout, &ns, &nc, &per) * (32 * 9 * 64 + 32) * sizeof(float);
}

template <typename G, bool AFF = false>
static int launch_wgrad(const Geom& g, int grid, const void* dy, const void* x, float* part, float* part_b,
                        int B, int H, int W, int Cin, int Cout, int relu_in, int ns, int nc,
                        hipStream_t st, const float* in_coef = nullptr) {
    auto k = conv3x3_wgrad_direct_kernel<G, AFF>;
    const int lds = G::NH * 128 + WG_DY_BYTES + (AFF ? 128 * (int)sizeof(float) : 0);
    static bool attr = false;
    if (!attr) {
        HIP_RET(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr = true;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(DC_THREADS), lds, st, (const bf16_t*)dy, (const bf16_t*)x, part,
                       part_b, B, H, W, Cin, Cout, relu_in, g.tiles_x, g.tiles_y, g.tiles, ns, nc, in_coef);
    LAUNCH_CHECK();
    return 0;
}

static int wgrad_impl(const void* dy, const void* x, const float* in_coef, float* dw2, float* dbias,
                      void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout,
                      int relu_in, int accumulate, void* stream);
extern "C" int ssl4gie_conv3x3_direct_wgrad(const void* dy, const void* x, float* dw2, float* dbias,
                                            void* workspace, size_t workspace_bytes, int B, int H,
                                            int W, int Cin, int Cout, int relu_in, int accumulate,
                                            void* stream) {
    return wgrad_impl(dy, x, nullptr, dw2, dbias, workspace, workspace_bytes, B, H, W, Cin, Cout, relu_in,
                      accumulate, stream);
}
// weight gradient against the operand of ssl4gie_conv3x3_direct_fwd_affine: act(x in_coef[0] + in_coef[1])
extern "C" int ssl4gie_conv3x3_direct_wgrad_affine(const void* dy, const void* x, const float* in_coef,
                                                   float* dw2, float* dbias, void* workspace,
                                                   size_t workspace_bytes, int B, int H, int W, int Cin,
                                                   int Cout, int relu_in, int accumulate, void* stream) {
    REQUIRE(in_coef);
    return wgrad_impl(dy, x, in_coef, dw2, dbias, workspace, workspace_bytes, B, H, W, Cin, Cout, relu_in,
                      accumulate, stream);
}
static int wgrad_impl(const void* dy, const void* x, const float* in_coef, float* dw2, float* dbias,
                      void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout,
                      int relu_in, int accumulate, void* stream) {
    REQUIRE(dy && x && dw2 && workspace && ssl4gie_conv3x3_direct_wgrad_ok(B, H, W, Cin, Cout));
    REQUIRE(workspace_bytes >= ssl4gie_conv3x3_direct_wgrad_workspace_bytes(B, H, W, Cin, Cout));
    hipStream_t st = (hipStream_t)stream;
    const Geom g = pick_geom(B, H, W);
    int ns, nc, per;
    const int grid = wgrad_grid(g, Cin, Cout, &ns, &nc, &per);
    float* part_b = (float*)workspace + (size_t)grid * 32 * 9 * 64;
    {
        ProfScope prof(PROF_GEMM_TN, 2.0 * B * H * (double)W * Cout * 9 * Cin, st);
        int rc;
#define WG_ARGS g, grid, dy, x, (float*)workspace, part_b, B, H, W, Cin, Cout, relu_in, ns, nc, st, in_coef
        if (in_coef) {
            if (g.id == 0) rc = launch_wgrad<G32, true>(WG_ARGS);
            else if (g.id == 1) rc = launch_wgrad<G16, true>(WG_ARGS);
            else rc = launch_wgrad<G8, true>(WG_ARGS);
        } else if (g.id == 0) rc = launch_wgrad<G32>(WG_ARGS);
        else if (g.id == 1) rc = launch_wgrad<G16>(WG_ARGS);
        else rc = launch_wgrad<G8>(WG_ARGS);
#undef WG_ARGS
        if (rc) return rc;
    }
    const int n = Cout * 9 * Cin + Cout;
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st,
                       (const float*)workspace, (const float*)part_b, dw2, dbias, Cin, Cout, ns, nc, per,
                       accumulate);
    LAUNCH_CHECK();
    return 0;
}

// ---- stem
static Geom stem_geom(int B, int Ho, int Wo) {
    Geom g = pick_geom(B, Ho, Wo);
    if (g.id == 2) {  // the four-image geometry does not exist for the stem: 16 x 16 instead
        g = Geom{1, 1, 16, 16, (Wo + 15) / 16, (Ho + 15) / 16, 0};
        g.tiles = g.tiles_x * g.tiles_y * B;
    }
    return g;
}
static bool stem_ok(int B, int H, int W) {
    return B > 0 && H >= 7 && W >= 7 && (long long)B * H * W < (1LL << 31);
}
extern "C" size_t ssl4gie_stem7x7_packed_bytes(int B, int H, int W) {
    if (!stem_ok(B, H, W)) return 0;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    return (size_t)B * (2 * Ho + 6) * (2 * Wo + 6) * 4 * sizeof(bf16_t);
}
extern "C" int ssl4gie_stem7x7_pack(const float* img, void* packed, int B, int H, int W, void* stream) {
    REQUIRE(img && packed && stem_ok(B, H, W));
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, Hp = 2 * Ho + 6, Wp = 2 * Wo + 6;
    const long long total = (long long)B * Hp * Wp;
    hipLaunchKernelGGL(stem7x7_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, img, (bf16_t*)packed, H, W, Hp, Wp, total);
    LAUNCH_CHECK();
    return 0;
}
extern "C" int ssl4gie_stem7x7_tiles(int B, int H, int W) {
    if (!stem_ok(B, H, W)) return 0;
    return stem_geom(B, (H - 1) / 2 + 1, (W - 1) / 2 + 1).tiles;
}
template <typename G>
static int launch_stem_fwd(const Geom& g, const void* packed, const void* w, void* y, float* colstats, int B,
                           int Ho, int Wo, hipStream_t st) {
    auto k = stem7x7_direct_kernel<G>;
    int lds = StemG<G>::XS_BYTES + 64 * ST_WROW;
    const int lds_out = DC_PIX * 128 + 2 * DC_THREADS * (int)sizeof(float);
    if (lds < lds_out) lds = lds_out;
    hipLaunchKernelGGL(k, dim3((unsigned)g.tiles), dim3(DC_THREADS), lds, st, (const bf16_t*)packed,
                       (const bf16_t*)w, (bf16_t*)y, colstats, B, Ho, Wo, 2 * Ho + 6, 2 * Wo + 6, g.tiles_x,
                       g.tiles_y);
    LAUNCH_CHECK();
    return 0;
}
extern "C" int ssl4gie_stem7x7_fwd(const void* packed, const void* w2s, void* y, float* colstats, int B,
                                   int H, int W, void* stream) {
    REQUIRE(packed && w2s && y && stem_ok(B, H, W));
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const Geom g = stem_geom(B, Ho, Wo);
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_GEMM_NT, 2.0 * B * Ho * (double)Wo * 64 * 147, st);
    if (g.id == 0) return launch_stem_fwd<G32>(g, packed, w2s, y, colstats, B, Ho, Wo, st);
    return launch_stem_fwd<G16>(g, packed, w2s, y, colstats, B, Ho, Wo, st);
}
static int stem_wgrad_grid(const Geom& g) {
    int n = 2 * ssl4gie_internal_compute_cus();
    return n < g.tiles ? n : g.tiles;
}
extern "C" size_t ssl4gie_stem7x7_wgrad_workspace_bytes(int B, int H, int W) {
    if (!stem_ok(B, H, W)) return 0;
    return (size_t)stem_wgrad_grid(stem_geom(B, (H - 1) / 2 + 1, (W - 1) / 2 + 1)) * 64 * 7 * 32 * sizeof(float);
}
template <typename G>
static int launch_stem_wgrad(const Geom& g, int grid, const void* dy, const void* packed, float* part, int B,
                             int Ho, int Wo, hipStream_t st) {
    auto k = stem7x7_wgrad_kernel<G>;
    const int lds = StemG<G>::XS_BYTES + 2 * WG_DY_BYTES;
    hipLaunchKernelGGL(k, dim3(grid), dim3(DC_THREADS), lds, st, (const bf16_t*)dy, (const bf16_t*)packed, part,
                       B, Ho, Wo, 2 * Ho + 6, 2 * Wo + 6, g.tiles_x, g.tiles_y, g.tiles);
    LAUNCH_CHECK();
    return 0;
}
extern "C" int ssl4gie_stem7x7_wgrad(const void* dy, const void* packed, float* dw2s, void* workspace,
                                     size_t workspace_bytes, int B, int H, int W, int accumulate,
                                     void* stream) {
    REQUIRE(dy && packed && dw2s && workspace && stem_ok(B, H, W));
    REQUIRE(workspace_bytes >= ssl4gie_stem7x7_wgrad_workspace_bytes(B, H, W));
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const Geom g = stem_geom(B, Ho, Wo);
    const int grid = stem_wgrad_grid(g);
    hipStream_t st = (hipStream_t)stream;
    {
        ProfScope prof(PROF_GEMM_TN, 2.0 * B * Ho * (double)Wo * 64 * 147, st);
        int rc = g.id == 0 ? launch_stem_wgrad<G32>(g, grid, dy, packed, (float*)workspace, B, Ho, Wo, st)
                           : launch_stem_wgrad<G16>(g, grid, dy, packed, (float*)workspace, B, Ho, Wo, st);
        if (rc) return rc;
    }
    return ssl4gie_internal_reduce_partials((const float*)workspace, dw2s, grid, 64 * 7 * 32, (size_t)64 * 7 * 32,
                                            accumulate, st);
}
