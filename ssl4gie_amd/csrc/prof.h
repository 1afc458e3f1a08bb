// Optional launch timing for bench.py's roofline line (HIP events on the launch stream around
// every launch of the heavy kernels).  Off by default; process-global and NOT thread-safe —
// the one deliberate exception to the "no mutable globals" rule of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

// kinds 0-4 carry FLOPs in `flops`; the HBM-bound kinds 5 (BatchNorm entry points) and 6 (LayerNorm) carry their
// ALGORITHMIC BYTES there (every tensor the op must read once / write once) — bench.py prices them against 8 TB/s
enum { PROF_GEMM_NT = 0, PROF_GEMM_TN = 1, PROF_ATTN_FWD = 2, PROF_ATTN_BWD = 3,
       PROF_GEMM_GENERIC = 4, PROF_BN = 5, PROF_LN = 6, PROF_KINDS = 7 };

struct ProfState {
    bool on;
    int cap, n;
    hipEvent_t* ev;  // 2 per record
    int* kind;
    double* flops;
};
extern ProfState g_prof;

struct ProfScope {
    int idx;
    hipStream_t st;
    ProfScope(int kind, double flops, hipStream_t s) : idx(-1), st(s) {
        if (g_prof.on && g_prof.n < g_prof.cap) {
            idx = g_prof.n++;
            g_prof.kind[idx] = kind;
            g_prof.flops[idx] = flops;
            (void)hipEventRecord(g_prof.ev[2 * idx], st);
        }
    }
    ~ProfScope() {
        if (idx >= 0) (void)hipEventRecord(g_prof.ev[2 * idx + 1], st);
    }
};
