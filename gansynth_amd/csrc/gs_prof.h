// Per-launch HIP-event profiling hooks shared by the kernel files (gs_prof_enable / gs_prof_records / gs_prof_collect).
// A ProfScope around a launch records an event pair on the launch stream plus the launch's algorithmic flops / bytes and an
// 8-int descriptor; nothing is recorded (and no event is created) while profiling is off.
#pragma once
#include "gs_common.h"

namespace gs {

// --------------------------------------------------------------------------- profiling hooks
struct ProfState {
    bool on = false;
    int burst = 1;                         // gs_prof_enable(n > 1): an idempotent launch runs n times back to back inside its event pair
    static constexpr int MAXEV = 8192;
    hipEvent_t ev[MAXEV][2];
    int created = 0;
    int used = 0;
    double flops = 0.0;
    double lflops[MAXEV], lbytes[MAXEV];   // per launch: algorithmic flops and bytes (operands read once + result written once)
    int lreps[MAXEV];                      // per launch: launches between the two events (the recorded time is divided by it)
    int ldesc[MAXEV][8];                   // per launch: {kind, N, Hb, Wb, IC, OC, masked, fused norm}; kind = conv mode, +10 for weight gradients
};
extern ProfState g_prof;   // (core.cpp)

struct ProfScope {
    hipStream_t s;
    int idx = -1;
    // `reps`: how many times the caller launches the (idempotent) kernel inside the scope -- see prof_reps()
    ProfScope(hipStream_t st, double flops, double bytes, int kind, int N, int Hb, int Wb, int IC, int OC, int masked, int norm, int reps = 1) : s(st) {
        if (!g_prof.on || g_prof.used >= ProfState::MAXEV) return;
        idx = g_prof.used++;
        g_prof.lreps[idx] = reps > 0 ? reps : 1;
        g_prof.lflops[idx] = flops;
        g_prof.lbytes[idx] = bytes;
        const int d[8] = {kind, N, Hb, Wb, IC, OC, masked, norm};
        for (int i = 0; i < 8; ++i) g_prof.ldesc[idx][i] = d[i];
        if (kind >= 10) flops = 0.0;   // (the family totals of gs_prof_collect count the implicit-GEMM launches only)
        if (idx >= g_prof.created) {
            (void)hipEventCreate(&g_prof.ev[idx][0]);
            (void)hipEventCreate(&g_prof.ev[idx][1]);
            g_prof.created = idx + 1;
        }
        g_prof.flops += flops;
        (void)hipEventRecord(g_prof.ev[idx][0], s);
    }
    ~ProfScope() {
        if (idx >= 0) (void)hipEventRecord(g_prof.ev[idx][1], s);
    }
};

// An event pair around ONE eager launch also times the host's launch latency (the stream runs dry between launches of a
// few microseconds each: +4-5 us per launch measured against back-to-back launches of the same kernel, scripts/probe/igemm_trace).
// In burst mode a launch whose result does not depend on how often it runs (the implicit-GEMM convs: pure functions of their
// inputs) is issued prof_reps() times inside its scope and the elapsed time divided: the steady-state launch-to-launch time,
// one launch boundary included -- what the launch costs inside a replayed graph.
inline int prof_reps() { return g_prof.on && g_prof.burst > 1 ? g_prof.burst : 1; }

}  // namespace gs
