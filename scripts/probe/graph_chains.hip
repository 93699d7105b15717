// How many branches of a captured graph really run side by side?  N independent chains of spin kernels forked from the root of one capture
// (chain 0 on the capturing stream), every kernel stamps its start: per chain, when it started and ended relative to the graph's first kernel.
// (round 6: part A of the generator run ran entirely BEFORE or entirely AFTER the discriminator's backward, never beside it.)
//   hipcc --offload-arch=gfx950 -O2 graph_chains.hip -o graph_chains && ./graph_chains
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin(unsigned long long* stamps, int idx, int ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) stamps[idx] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) {}
}

int main(int argc, char** argv) {
    const int LEN = 40, TICKS = 1000;   // 40 kernels of 10 us per chain
    const int level = argc > 1 ? atoi(argv[1]) : 64;
    unsigned long long* stamps;
    CK(hipMalloc(&stamps, 8 * LEN * sizeof(unsigned long long)));
    hipStream_t s[8], levels[128];
    for (int i = 0; i < 8; ++i) CK(hipStreamCreate(&s[i]));
    for (int nch = 2; nch <= 6; ++nch) {
        hipGraph_t g;
        hipGraphExec_t ge;
        hipEvent_t fork, join[8];
        CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        for (int i = 0; i < 8; ++i) CK(hipEventCreateWithFlags(&join[i], hipEventDisableTiming));
        for (int i = 0; i < level; ++i) CK(hipStreamCreate(&levels[i]));
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeRelaxed));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[0], stamps, 0, 100);   // a common root node
        CK(hipEventRecord(fork, s[0]));
        for (int c = 1; c < nch; ++c) CK(hipStreamWaitEvent(s[c], fork, 0));
        for (int c = 0; c < nch; ++c)
            for (int k = (c == 0 ? 1 : 0); k < LEN; ++k) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[c], stamps, c * LEN + k, TICKS);
        for (int c = 1; c < nch; ++c) { CK(hipEventRecord(join[c], s[c])); CK(hipStreamWaitEvent(s[0], join[c], 0)); }
        CK(hipStreamEndCapture(s[0], &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < level; ++i) CK(hipStreamDestroy(levels[i]));
        std::vector<unsigned long long> h(8 * LEN);
        for (int r = 0; r < 3; ++r) {
            CK(hipGraphLaunch(ge, s[0]));
            CK(hipStreamSynchronize(s[0]));
        }
        CK(hipMemcpy(h.data(), stamps, 8 * LEN * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        printf("%d chains x %d kernels of %d us (levelled with %d streams):", nch, LEN, TICKS / 100, level);
        for (int c = 0; c < nch; ++c)
            printf("  [%d] %6.0f .. %6.0f us", c, (h[c * LEN + (c == 0 ? 1 : 0)] - h[0]) / 100.0, (h[c * LEN + LEN - 1] - h[0]) / 100.0 + TICKS / 100.0);
        printf("\n");
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
