// In what order does hipGraphLaunch submit the nodes of a graph with two parallel branches?  (round 6: the side branch of a captured run
// starts ~390 us after its graph although it depends on the root only -- after exactly the ~86 main-stream nodes captured before it.)
// Two independent chains of spin kernels, captured in different ISSUE orders; every kernel stamps its start (s_memrealtime, 100 MHz).
//   hipcc --offload-arch=gfx950 -O2 graph_order.hip -o graph_order && ./graph_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin(unsigned long long* stamps, int idx, int ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) stamps[idx] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) {}
}

int main() {
    const int NA = 80, NB = 80;
    unsigned long long* stamps;
    CK(hipMalloc(&stamps, (NA + NB + 8) * sizeof(unsigned long long)));
    hipStream_t s1, s2, levels[64];
    CK(hipStreamCreate(&s1));
    CK(hipStreamCreate(&s2));
    const char* names[] = {"A chain first, then B chain", "B chain first, then A chain", "A, B alternating", "A x8, B x8 alternating",
                           "A first; A long kernels (25 us), B short (4 us)", "alternating; A long (25 us), B short (4 us), 1 A : 5 B"};
    for (int variant = 0; variant < 6; ++variant) {
        const int ta = (variant >= 4) ? 2500 : 400, tb = 400;   // ticks of 10 ns
        hipGraph_t g;
        hipGraphExec_t ge;
        hipEvent_t fork, join;
        CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
        for (int i = 0; i < 64; ++i) CK(hipStreamCreate(&levels[i]));   // (level the runtime's hardware-queue pool, see gs_streams_create)
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeRelaxed));
        CK(hipEventRecord(fork, s1));
        CK(hipStreamWaitEvent(s2, fork, 0));
        int ia = 0, ib = 0;
        auto A = [&]() { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, stamps, ia, ta); ++ia; };
        auto B = [&]() { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s2, stamps, NA + ib, tb); ++ib; };
        const int nb_total = (variant >= 4) ? NB : NB, na_total = (variant >= 4) ? NA / 5 : NA;
        if (variant == 0 || variant == 4) { while (ia < na_total) A(); while (ib < nb_total) B(); }
        else if (variant == 1) { while (ib < nb_total) B(); while (ia < na_total) A(); }
        else if (variant == 2) { while (ia < na_total || ib < nb_total) { if (ia < na_total) A(); if (ib < nb_total) B(); } }
        else if (variant == 3) { while (ia < na_total || ib < nb_total) { for (int k = 0; k < 8 && ia < na_total; ++k) A(); for (int k = 0; k < 8 && ib < nb_total; ++k) B(); } }
        else { while (ia < na_total || ib < nb_total) { if (ia < na_total) A(); for (int k = 0; k < 5 && ib < nb_total; ++k) B(); } }
        CK(hipEventRecord(join, s2));
        CK(hipStreamWaitEvent(s1, join, 0));
        CK(hipStreamEndCapture(s1, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 64; ++i) CK(hipStreamDestroy(levels[i]));
        std::vector<unsigned long long> h(NA + NB);
        double host_us = 0, span = 0, b_first = 0, a_first_to_last = 0, b_first_to_last = 0;
        const int reps = 5;
        for (int r = 0; r < reps + 2; ++r) {
            CK(hipMemsetAsync(stamps, 0, (NA + NB) * sizeof(unsigned long long), s1));
            CK(hipStreamSynchronize(s1));
            auto t0 = std::chrono::steady_clock::now();
            CK(hipGraphLaunch(ge, s1));
            auto t1 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(s1));
            CK(hipMemcpy(h.data(), stamps, (NA + NB) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            if (r < 2) continue;
            unsigned long long a0 = h[0], al = h[ia - 1], b0 = h[NA], bl = h[NA + ib - 1];
            unsigned long long first = a0 < b0 ? a0 : b0, last = al > bl ? al : bl;
            host_us += std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
            span += (last - first) / 100.0 / reps;
            b_first += ((double)b0 - (double)a0) / 100.0 / reps;
            a_first_to_last += (al - a0) / 100.0 / reps;
            b_first_to_last += (bl - b0) / 100.0 / reps;
        }
        printf("%-72s | A %3d x %4.1f us, B %3d x %4.1f us | hipGraphLaunch host %7.1f us | first B - first A %8.1f us | A chain %7.1f us  B chain %7.1f us | both %7.1f us\n",
               names[variant], ia, ta / 100.0, ib, tb / 100.0, host_us, b_first, a_first_to_last, b_first_to_last, span);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
