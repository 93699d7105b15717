// Per-launch HIP-event profiling hooks shared by the kernel files (gs_prof_enable / gs_prof_records / gs_prof_collect).
// A ProfScope around a launch records an event pair on the launch stream plus the launch's algorithmic flops / bytes and an
// 8-int descriptor; nothing is recorded (and no event is created) while profiling is off.
#pragma once
#include "gs_common.h"

namespace gs {

// --------------------------------------------------------------------------- profiling hooks
struct ProfState {
    bool on = false;
    static constexpr int MAXEV = 8192;
    hipEvent_t ev[MAXEV][2];
    int created = 0;
    int used = 0;
    double flops = 0.0;
    double lflops[MAXEV], lbytes[MAXEV];   // per launch: algorithmic flops and bytes (operands read once + result written once)
    int ldesc[MAXEV][8];                   // per launch: {kind, N, Hb, Wb, IC, OC, masked, fused norm}; kind = conv mode, +10 for weight gradients
};
extern ProfState g_prof;   // (core.cpp)

struct ProfScope {
    hipStream_t s;
    int idx = -1;
    ProfScope(hipStream_t st, double flops, double bytes, int kind, int N, int Hb, int Wb, int IC, int OC, int masked, int norm) : s(st) {
        if (!g_prof.on || g_prof.used >= ProfState::MAXEV) return;
        idx = g_prof.used++;
        g_prof.lflops[idx] = flops;
        g_prof.lbytes[idx] = bytes;
        const int d[8] = {kind, N, Hb, Wb, IC, OC, masked, norm};
        for (int i = 0; i < 8; ++i) g_prof.ldesc[idx][i] = d[i];
        if (kind >= 10) flops = 0.0;   // (the family totals of gs_prof_collect count the implicit-GEMM launches only)
        if (idx >= g_prof.created) {
            hipEventCreate(&g_prof.ev[idx][0]);
            hipEventCreate(&g_prof.ev[idx][1]);
            g_prof.created = idx + 1;
        }
        g_prof.flops += flops;
        hipEventRecord(g_prof.ev[idx][0], s);
    }
    ~ProfScope() {
        if (idx >= 0) hipEventRecord(g_prof.ev[idx][1], s);
    }
};

}  // namespace gs
