// Native executor for one pre-LN transformer block (forward and backward): issues the whole
// kernel sequence on the caller's stream from C++ — no Python per-op dispatch, nothing allocated,
// nothing synchronised, so a stack of blocks can be replayed from a hipGraph.
//
// Reference: timm Block (un-vendored) instantiated at models_mae.py:39-41,53-55 and
// looped at models_mae.py:166-167,186-187 / Models/models.py:451-454.
//
// Forward (7 launches):  LN1 -> QKV GEMM(+bias) -> fused attention -> proj GEMM(+bias+residual)
//                        -> LN2 -> fc1 GEMM(+bias; saves gelu'(u) and gelu(u)) -> fc2 GEMM(+bias+res).
// Backward: data-gradient GEMMs are NT products against pre-transposed weight copies (the saved GELU'
// multiplied in by the fc2 one), weight-gradient GEMMs are split-K TN products over the token axis with the
// bias gradients (column sums of dY) fused in, and both LayerNorm backward kernels add the residual
// gradient and emit the operand-type copy the next GEMM consumes.
#include "common.h"
#include "ssl4gie_hip.h"

#include <string.h>
#include <stdlib.h>
#include <mutex>
#include "prof.h"
#include "internal.h"

extern "C" int ssl4gie_abi_version(void) { return 7; }

namespace { extern int g_wgrad_stream; }
// 1: block weight gradients on the library's side stream (default), 0: everything on the caller's
extern "C" int ssl4gie_set_wgrad_stream(int on) {
    g_wgrad_stream = on ? 1 : 0;
    return 0;
}

// ---- launch profiler (bench only; see prof.h)
ProfState g_prof = {false, 0, 0, nullptr, nullptr, nullptr};

extern "C" int ssl4gie_prof_begin(int max_launches) {
    REQUIRE(max_launches > 0 && !g_prof.ev);
    g_prof.ev = (hipEvent_t*)calloc((size_t)2 * max_launches, sizeof(hipEvent_t));
    g_prof.kind = (int*)calloc(max_launches, sizeof(int));
    g_prof.flops = (double*)calloc(max_launches, sizeof(double));
    REQUIRE(g_prof.ev && g_prof.kind && g_prof.flops);
    for (int i = 0; i < 2 * max_launches; ++i) HIP_RET(hipEventCreate(&g_prof.ev[i]));
    g_prof.cap = max_launches;
    g_prof.n = 0;
    g_prof.on = true;
    return 0;
}
// Synchronises the device, sums elapsed ms / flops / launch counts per kernel kind
// (arrays of SSL4GIE_PROF_KINDS entries), resets the record list; profiling stays on.
extern "C" int ssl4gie_prof_collect(double* ms, double* flops, long long* launches) {
    REQUIRE(g_prof.ev && ms && flops && launches);
    HIP_RET(hipDeviceSynchronize());
    for (int k = 0; k < PROF_KINDS; ++k) { ms[k] = 0; flops[k] = 0; launches[k] = 0; }
    for (int i = 0; i < g_prof.n; ++i) {
        float t = 0.f;
        HIP_RET(hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
        ms[g_prof.kind[i]] += t;
        flops[g_prof.kind[i]] += g_prof.flops[i];
        launches[g_prof.kind[i]] += 1;
    }
    g_prof.n = 0;
    return 0;
}
extern "C" int ssl4gie_prof_end(void) {
    g_prof.on = false;
    if (g_prof.ev) {
        for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
        free(g_prof.ev); free(g_prof.kind); free(g_prof.flops);
    }
    g_prof = ProfState{false, 0, 0, nullptr, nullptr, nullptr};
    return 0;
}

namespace {

// ---- weight-gradient side stream
// The four dW products of a block do not feed the data-gradient chain, and the chain's NT GEMMs
// often leave CUs idle (150 or 394 output tiles on 256 CUs): the dW products run on a second,
// non-blocking stream so the dispatcher can fill those CUs.  Ordering is by events only (no host
// synchronisation): each dW waits for the event recorded after its dY producer, and the caller's
// stream waits for the last dW before ssl4gie_block_bwd's work is considered complete, so from
// outside the call nothing changes.  One stream + event set per device, created on first use and
// kept for the life of the library.  SSL4GIE_WGRAD_STREAM=0 (or ssl4gie_set_wgrad_stream(0))
// folds everything back onto the caller's stream — used when per-kernel durations are measured.
struct SideStream {
    hipStream_t s = nullptr;
    // SSL4GIE_WGRAD_STREAMS=2 (round 5): the (proj, qkv) pair of a block on a second weight-gradient stream, so that
    // the two pair launches of a block — and those of neighbouring blocks — can be in flight together with fewer,
    // longer workgroups each; ev2: 0..3 completion under the caller's slot, 4 join, 5 fork
    hipStream_t s2 = nullptr;
    hipEvent_t ev2[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // 0..4: fork / join points of ssl4gie_block_bwd; 5: fork of ssl4gie_wgrad_group;
    // 6..9: completion of the group launched with slot 0..3 (ssl4gie_wgrad_wait)
    hipEvent_t ev[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool slot_used[4] = {false, false, false, false};
};
int g_wgrad_stream = -1;  // -1: read the environment on first use
SideStream* side_stream() {
    if (g_wgrad_stream < 0) {
        const char* e = getenv("SSL4GIE_WGRAD_STREAM");
        g_wgrad_stream = (e && e[0] == '0') ? 0 : 1;
    }
    if (!g_wgrad_stream) return nullptr;
    static SideStream per_dev[64];
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    SideStream* ss = &per_dev[dev];
    if (!ss->s) {
        hipStream_t st = nullptr;
        // SSL4GIE_WGRAD_PRIO=high|low: scheduling priority of the weight-gradient stream against the caller's
        // (default: the same).  Measured on the MAE step (profiles/r04bs): see profiles/HISTORY.md section 5.
        const char* pe = getenv("SSL4GIE_WGRAD_PRIO");
        int least = 0, greatest = 0;
        if (pe && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) {
            const int prio = (pe[0] == 'h' || pe[0] == 'H') ? greatest : least;
            if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio) != hipSuccess) return nullptr;
        } else if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
            return nullptr;
        }
        for (int i = 0; i < 10; ++i) {
            if (hipEventCreateWithFlags(&ss->ev[i], hipEventDisableTiming) != hipSuccess) {
                (void)hipStreamDestroy(st);
                return nullptr;
            }
        }
        // (measured null, profiles/HISTORY.md: the second stream exists in the debug library only — ADVICE r5)
        int two = 0;
#ifdef SSL4GIE_DEBUG_KNOBS
        { const char* e = getenv("SSL4GIE_WGRAD_STREAMS"); two = (e && e[0] == '2') ? 1 : 0; }
#endif
        if (two) {
            hipStream_t st2 = nullptr;
            if (hipStreamCreateWithFlags(&st2, hipStreamNonBlocking) == hipSuccess) {
                bool ok = true;
                for (int i = 0; i < 6 && ok; ++i) ok = hipEventCreateWithFlags(&ss->ev2[i], hipEventDisableTiming) == hipSuccess;
                if (ok) ss->s2 = st2; else (void)hipStreamDestroy(st2);
            }
        }
        ss->s = st;
    }
    return ss;
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
size_t esize(int dt) { return dt == SSL4GIE_BF16 ? 2 : 4; }

struct BwdLayout {
    size_t du, dh, dxmid, dxmid_lp, dattn, dqkv, ln_ws, ln_ws1, gemm_ws, gemm_ws2, attn_ws, total;
    size_t gemm_ws_bytes;
};

ssl4gie_gemm_desc lin_desc(int M, int N, int K, int dt) {
    ssl4gie_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.M = M; d.N = N; d.K = K; d.batch1 = 1; d.batch2 = 1;
    d.dtype_ab = dt; d.dtype_c = dt; d.alpha = 1.f; d.epilogue = SSL4GIE_EPI_NONE;
    return d;
}
// dW[N_out, K_in] = dY[T, N_out]^T X[T, K_in]
// (db[N_out] = column sums of dY rides on the same product: ssl4gie_gemm_desc.colsum_a)
ssl4gie_gemm_desc wgrad_desc(int n_out, int k_in, int T, const void* dY, const void* X, float* dW,
                             float* db, int dt, int accumulate) {
    ssl4gie_gemm_desc d = lin_desc(n_out, k_in, T, dt);
    d.A = dY; d.sAm = 1; d.sAk = n_out;
    d.B = X; d.sBk = k_in; d.sBn = 1;
    d.C = dW; d.ldc = k_in; d.dtype_c = SSL4GIE_F32; d.accumulate = accumulate;
    d.colsum_a = db;
    return d;
}

BwdLayout bwd_layout(const ssl4gie_block_dims* d) {
    const size_t T = (size_t)d->B * d->N, D = d->D, F = d->F, es = esize(d->dtype);
    BwdLayout L;
    size_t o = 0;
    L.du = o; o += align_up(T * F * es);
    L.dh = o; o += align_up(T * D * es);
    L.dxmid = o; o += align_up(T * D * 4);
    L.dxmid_lp = o; o += (d->dtype == SSL4GIE_F32) ? 0 : align_up(T * D * es);
    L.dattn = o; o += align_up(T * D * es);
    L.dqkv = o; o += align_up(T * 3 * D * es);
    L.ln_ws = o; o += align_up(ssl4gie_layernorm_bwd_workspace_bytes((int)T, (int)D));
    L.ln_ws1 = o; o += align_up(ssl4gie_layernorm_bwd_workspace_bytes((int)T, (int)D));  // LN1's own partials: LN2's are still being summed on the side stream
    size_t g = 0;
    // the weight gradients run as two pairs: (fc2, fc1) and (proj, qkv)
    const int dims[4][2] = {{(int)D, (int)F}, {(int)F, (int)D}, {(int)D, (int)D}, {(int)(3 * D), (int)D}};
    for (int i = 0; i < 4; i += 2) {
        // pointers only matter for alignment checks: use 16-B aligned dummies
        ssl4gie_gemm_desc w0 = wgrad_desc(dims[i][0], dims[i][1], (int)T, (const void*)256,
                                          (const void*)256, (float*)256, (float*)256, d->dtype, 0);
        ssl4gie_gemm_desc w1 = wgrad_desc(dims[i + 1][0], dims[i + 1][1], (int)T, (const void*)256,
                                          (const void*)256, (float*)256, (float*)256, d->dtype, 0);
        const size_t b = ssl4gie_gemm_tn_pair_workspace_bytes(&w0, &w1);
        if (b > g) g = b;
    }
    L.gemm_ws_bytes = g;
    L.gemm_ws = o; o += align_up(g);
#ifdef SSL4GIE_DEBUG_KNOBS
    L.gemm_ws2 = o; o += align_up(g);  // the (proj, qkv) pair's own slabs: it may run beside the (fc2, fc1) pair (SSL4GIE_WGRAD_STREAMS=2)
#else
    L.gemm_ws2 = L.gemm_ws;  // one weight-gradient stream: the two pairs run in order and share the slabs
#endif
    L.attn_ws = o; o += align_up(ssl4gie_attn_workspace_bytes(d->dtype, d->B, d->N, d->H, d->D / d->H));
    L.total = o;
    return L;
}

bool dims_ok(const ssl4gie_block_dims* d) {
    return d && d->B > 0 && d->N > 0 && d->D > 0 && d->H > 0 && d->F > 0 && d->D % d->H == 0 &&
           d->D % 4 == 0 && d->F % 4 == 0 &&
           (d->dtype == SSL4GIE_F32 || d->dtype == SSL4GIE_BF16);
}

// y[T, n_out] = x[T, k_in] W[n_out, k_in]^T (+ epilogue)
int linear_fwd(const void* x, const void* W, int T, int n_out, int k_in, int dt,
               ssl4gie_gemm_desc ep /* C, dtype_c, epilogue fields set */, void* stream) {
    ep.M = T; ep.N = n_out; ep.K = k_in; ep.batch1 = ep.batch2 = 1;
    ep.A = x; ep.sAm = k_in; ep.sAk = 1;
    ep.B = W; ep.sBk = 1; ep.sBn = k_in;
    ep.dtype_ab = dt; ep.alpha = 1.f;
    return ssl4gie_gemm(&ep, nullptr, 0, stream);
}
// dx[T, k_in] = dy[T, n_out] W[n_out, k_in]; uses the transposed copy Wt[k_in, n_out] if given
int linear_bwd_data(const void* dy, const void* W, const void* Wt, int T, int n_out, int k_in,
                    int dt, ssl4gie_gemm_desc ep, void* stream) {
    ep.M = T; ep.N = k_in; ep.K = n_out; ep.batch1 = ep.batch2 = 1;
    ep.A = dy; ep.sAm = n_out; ep.sAk = 1;
    if (Wt) { ep.B = Wt; ep.sBk = 1; ep.sBn = n_out; }
    else    { ep.B = W;  ep.sBk = k_in; ep.sBn = 1; }
    ep.dtype_ab = dt; ep.alpha = 1.f;
    return ssl4gie_gemm(&ep, nullptr, 0, stream);
}

#define RC(expr)                 \
    do {                         \
        int rc__ = (expr);       \
        if (rc__) return rc__;   \
    } while (0)

}  // namespace

extern "C" size_t ssl4gie_block_workspace_bytes(const ssl4gie_block_dims* d) {
    if (!dims_ok(d)) return 0;
    return bwd_layout(d).total;  // forward only needs the attention workspace (subset)
}

extern "C" int ssl4gie_block_fwd(const ssl4gie_block_dims* d, const ssl4gie_block_weights* w,
                                 const ssl4gie_block_act* a, const float* x_in, float* x_out,
                                 void* workspace, void* stream) {
    REQUIRE(dims_ok(d) && w && a && x_in && x_out);
    const int T = d->B * d->N, D = d->D, F = d->F, dt = d->dtype;
    const BwdLayout L = bwd_layout(d);
    REQUIRE(workspace || L.total == 0);
    char* ws = (char*)workspace;

    RC(ssl4gie_layernorm_fwd(x_in, w->ln1_g, w->ln1_b, a->h1, dt, a->mean1, a->rstd1, T, D,
                             d->eps, stream));
    ssl4gie_gemm_desc e;
    memset(&e, 0, sizeof(e));
    e.C = a->qkv; e.ldc = 3 * D; e.dtype_c = dt; e.epilogue = SSL4GIE_EPI_BIAS; e.bias = w->bqkv;
    RC(linear_fwd(a->h1, w->wqkv, T, 3 * D, D, dt, e, stream));
    RC(ssl4gie_attn_fwd(a->qkv, a->attn, a->lse, dt, d->B, d->N, d->H, D / d->H, ws + L.attn_ws,
                        stream));
    memset(&e, 0, sizeof(e));
    e.C = a->xmid; e.ldc = D; e.dtype_c = SSL4GIE_F32; e.epilogue = SSL4GIE_EPI_BIAS_RESIDUAL;
    e.bias = w->bproj; e.residual = x_in; e.ldr = D;
    RC(linear_fwd(a->attn, w->wproj, T, D, D, dt, e, stream));
    RC(ssl4gie_layernorm_fwd(a->xmid, w->ln2_g, w->ln2_b, a->h2, dt, a->mean2, a->rstd2, T, D,
                             d->eps, stream));
    memset(&e, 0, sizeof(e));
    e.C = a->u; e.ldc = F; e.dtype_c = dt; e.epilogue = SSL4GIE_EPI_BIAS_GELU_GRAD; e.bias = w->bfc1;
    e.out2 = a->g;
    RC(linear_fwd(a->h2, w->wfc1, T, F, D, dt, e, stream));
    memset(&e, 0, sizeof(e));
    e.C = x_out; e.ldc = D; e.dtype_c = SSL4GIE_F32; e.epilogue = SSL4GIE_EPI_BIAS_RESIDUAL;
    e.bias = w->bfc2; e.residual = a->xmid; e.ldr = D;
    RC(linear_fwd(a->g, w->wfc2, T, D, F, dt, e, stream));
    return 0;
}

extern "C" int ssl4gie_block_bwd(const ssl4gie_block_dims* d, const ssl4gie_block_weights* w,
                                 const ssl4gie_block_act* a, const ssl4gie_block_grads* g,
                                 const float* x_in, const float* dx_out, const void* dx_out_lp,
                                 float* dx_in, void* dx_in_lp, int accumulate, void* workspace,
                                 void* stream) {
    REQUIRE(dims_ok(d) && w && a && g && x_in && dx_out && dx_in && workspace);
    const int T = d->B * d->N, D = d->D, F = d->F, dt = d->dtype;
    REQUIRE(dt == SSL4GIE_F32 || dx_out_lp);
    // `accumulate` carries flags: bit 0 accumulate, bit 1 SSL4GIE_BWD_DEFER_WGRAD (the caller launches
    // the four dW products later: ssl4gie_block_wgrad_descs + ssl4gie_wgrad_group)
    const bool defer = (accumulate & SSL4GIE_BWD_DEFER_WGRAD) != 0;
    const bool no_join = (accumulate & SSL4GIE_BWD_NO_JOIN) != 0;
    const int slot = (accumulate >> 4) & 3;
    accumulate &= SSL4GIE_BWD_ACCUMULATE;
    const BwdLayout L = bwd_layout(d);
    char* ws = (char*)workspace;
    void* du = ws + L.du;
    void* dh = ws + L.dh;
    float* dxmid = (float*)(ws + L.dxmid);
    void* dxmid_lp = (dt == SSL4GIE_F32) ? (void*)dxmid : (void*)(ws + L.dxmid_lp);
    void* dattn = ws + L.dattn;
    void* dqkv = ws + L.dqkv;
    float* ln_ws = (float*)(ws + L.ln_ws);
    float* ln_ws1 = (float*)(ws + L.ln_ws1);
    void* gws = ws + L.gemm_ws;
    void* gws2 = ws + L.gemm_ws2;
    const void* dy = (dt == SSL4GIE_F32) ? (const void*)dx_out : dx_out_lp;
    ssl4gie_gemm_desc e, wg, wg2;
    // weight gradients go to the side stream when there is one (see SideStream)
    hipStream_t main_st = (hipStream_t)stream;
    SideStream* ss = defer ? nullptr : side_stream();
    void* wst = ss ? (void*)ss->s : stream;
    int evi = 0;
    auto fork = [&]() -> int {  // side stream: wait for everything enqueued on the caller's so far
        if (!ss) return 0;
        HIP_RET(hipEventRecord(ss->ev[evi], main_st));
        HIP_RET(hipStreamWaitEvent(ss->s, ss->ev[evi], 0));
        ++evi;
        return 0;
    };

    // ---- fc2
    memset(&e, 0, sizeof(e));
    e.C = du; e.ldc = F; e.dtype_c = dt; e.epilogue = SSL4GIE_EPI_MUL_AUX; e.aux = a->u;
    RC(linear_bwd_data(dy, w->wfc2, w->wfc2_t, T, D, F, dt, e, stream));
    // ---- dW_fc2 and dW_fc1 as one paired launch (both inputs exist once du does)
    if (!defer) {
        RC(fork());
        wg = wgrad_desc(D, F, T, dy, a->g, g->wfc2, g->bfc2, dt, accumulate);
        wg2 = wgrad_desc(F, D, T, du, a->h2, g->wfc1, g->bfc1, dt, accumulate);
        RC(ssl4gie_gemm_tn_pair(&wg, &wg2, gws, L.gemm_ws_bytes, wst));
    }
    // ---- fc1
    memset(&e, 0, sizeof(e));
    e.C = dh; e.ldc = D; e.dtype_c = dt; e.epilogue = SSL4GIE_EPI_NONE;
    RC(linear_bwd_data(du, w->wfc1, w->wfc1_t, T, F, D, dt, e, stream));
    // ---- LN2 (adds the residual gradient dx_out)
    // with a weight-gradient stream the second stage of both LayerNorm backward passes (the sum of the per-block
    // dgamma / dbeta partials: a 6-us launch the data-gradient chain would wait for, 40 per MAE step) runs there
    static int ln_side_on = -1;  // SSL4GIE_LN_SIDE=0: the reductions stay on the caller's stream (A/B timing; same results)
    if (ln_side_on < 0) { const char* e = getenv("SSL4GIE_LN_SIDE"); ln_side_on = (e && e[0] == '0') ? 0 : 1; }
    const bool ln_side = ss != nullptr && ln_side_on;
    RC(ssl4gie_layernorm_bwd(dh, dt, a->xmid, w->ln2_g, a->mean2, a->rstd2, dx_out, dxmid,
                             dt == SSL4GIE_F32 ? nullptr : dxmid_lp, dt, ln_side ? nullptr : g->ln2_g,
                             ln_side ? nullptr : g->ln2_b, accumulate, ln_ws, T, D, stream));
    // ---- proj
    memset(&e, 0, sizeof(e));
    e.C = dattn; e.ldc = D; e.dtype_c = dt; e.epilogue = SSL4GIE_EPI_NONE;
    RC(linear_bwd_data(dxmid_lp, w->wproj, w->wproj_t, T, D, D, dt, e, stream));
    // ---- attention
    RC(ssl4gie_attn_bwd(a->qkv, a->attn, dattn, a->lse, dqkv, dt, d->B, d->N, d->H, D / d->H,
                        ws + L.attn_ws, stream));
    // ---- dW_proj and dW_qkv as one paired launch (on the second weight-gradient stream when there is one)
    bool used_s2 = false;
    if (!defer) {
        void* wstb = wst;
        if (ss && ss->s2) {
            HIP_RET(hipEventRecord(ss->ev2[5], main_st));
            HIP_RET(hipStreamWaitEvent(ss->s2, ss->ev2[5], 0));
            wstb = (void*)ss->s2;
            used_s2 = true;
        } else {
            RC(fork());
        }
        if (ln_side)  // (the stream it is enqueued on has just waited for the caller's: LN2's partials are there)
            RC(ssl4gie_internal_ln_reduce(ln_ws, g->ln2_g, g->ln2_b, T, D, accumulate, (hipStream_t)wstb));
        wg = wgrad_desc(D, D, T, dxmid_lp, a->attn, g->wproj, g->bproj, dt, accumulate);
        wg2 = wgrad_desc(3 * D, D, T, dqkv, a->h1, g->wqkv, g->bqkv, dt, accumulate);
        RC(ssl4gie_gemm_tn_pair(&wg, &wg2, gws2, L.gemm_ws_bytes, wstb));
    }
    // ---- qkv
    memset(&e, 0, sizeof(e));
    e.C = dh; e.ldc = D; e.dtype_c = dt; e.epilogue = SSL4GIE_EPI_NONE;
    RC(linear_bwd_data(dqkv, w->wqkv, w->wqkv_t, T, 3 * D, D, dt, e, stream));
    // ---- LN1 (adds the residual gradient dxmid)
    RC(ssl4gie_layernorm_bwd(dh, dt, x_in, w->ln1_g, a->mean1, a->rstd1, dxmid, dx_in,
                             dt == SSL4GIE_F32 ? nullptr : dx_in_lp, dt, ln_side ? nullptr : g->ln1_g,
                             ln_side ? nullptr : g->ln1_b, accumulate, ln_side ? ln_ws1 : ln_ws, T, D, stream));
    if (ln_side) {
        RC(fork());
        RC(ssl4gie_internal_ln_reduce(ln_ws1, g->ln1_g, g->ln1_b, T, D, accumulate, ss->s));
    }
    if (ss && no_join) {  // completion under the caller's slot: ssl4gie_wgrad_wait(slot, stream) before reuse
        HIP_RET(hipEventRecord(ss->ev[6 + slot], ss->s));
        if (ss->s2) HIP_RET(hipEventRecord(ss->ev2[slot], ss->s2));  // (recorded even when unused: the wait is unconditional)
        ss->slot_used[slot] = true;
    } else if (ss) {  // join: the caller's stream continues after the last weight gradient
        HIP_RET(hipEventRecord(ss->ev[4], ss->s));
        HIP_RET(hipStreamWaitEvent(main_st, ss->ev[4], 0));
        if (used_s2) {
            HIP_RET(hipEventRecord(ss->ev2[4], ss->s2));
            HIP_RET(hipStreamWaitEvent(main_st, ss->ev2[4], 0));
        }
    }
    return 0;
}

// ---- deferred weight gradients (SSL4GIE_BWD_DEFER_WGRAD)
extern "C" int ssl4gie_block_wgrad_descs(const ssl4gie_block_dims* d, const ssl4gie_block_act* a,
                                         const ssl4gie_block_grads* g, const void* dy, void* workspace,
                                         int accumulate, ssl4gie_gemm_desc* out4) {
    REQUIRE(dims_ok(d) && a && g && dy && workspace && out4);
    const int T = d->B * d->N, D = d->D, F = d->F, dt = d->dtype;
    const BwdLayout L = bwd_layout(d);
    char* ws = (char*)workspace;
    const void* du = ws + L.du;
    const void* dxmid_lp = (dt == SSL4GIE_F32) ? (const void*)(ws + L.dxmid) : (const void*)(ws + L.dxmid_lp);
    const void* dqkv = ws + L.dqkv;
    accumulate &= SSL4GIE_BWD_ACCUMULATE;
    out4[0] = wgrad_desc(D, F, T, dy, a->g, g->wfc2, g->bfc2, dt, accumulate);
    out4[1] = wgrad_desc(F, D, T, du, a->h2, g->wfc1, g->bfc1, dt, accumulate);
    out4[2] = wgrad_desc(D, D, T, dxmid_lp, a->attn, g->wproj, g->bproj, dt, accumulate);
    out4[3] = wgrad_desc(3 * D, D, T, dqkv, a->h1, g->wqkv, g->bqkv, dt, accumulate);
    return 0;
}

extern "C" int ssl4gie_wgrad_group(const ssl4gie_gemm_desc* descs, int n, void* workspace,
                                   size_t workspace_bytes, int slot, void* stream) {
    REQUIRE(descs && n >= 1 && slot >= 0 && slot < 4);
    hipStream_t main_st = (hipStream_t)stream;
    SideStream* ss = side_stream();
    if (!ss) return ssl4gie_gemm_tn_group(descs, n, workspace, workspace_bytes, stream);
    HIP_RET(hipEventRecord(ss->ev[5], main_st));        // everything the products read is enqueued
    HIP_RET(hipStreamWaitEvent(ss->s, ss->ev[5], 0));
    RC(ssl4gie_gemm_tn_group(descs, n, workspace, workspace_bytes, (void*)ss->s));
    HIP_RET(hipEventRecord(ss->ev[6 + slot], ss->s));
    ss->slot_used[slot] = true;
    return 0;
}

extern "C" int ssl4gie_wgrad_wait(int slot, void* stream) {
    REQUIRE(slot >= 0 && slot < 4);
    SideStream* ss = side_stream();
    if (!ss || !ss->slot_used[slot]) return 0;
    HIP_RET(hipStreamWaitEvent((hipStream_t)stream, ss->ev[6 + slot], 0));
    if (ss->s2) HIP_RET(hipStreamWaitEvent((hipStream_t)stream, ss->ev2[slot], 0));
    return 0;
}
