// A one-thread kernel that writes the 100 MHz constant clock into slot `idx`: phase boundaries of a replayed hipGraph WITHOUT a profiler in the
// way (rocprofv3 / roctracer intercept every dispatch and make the host the bottleneck: the replay they show is not the one that runs).
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC stamp.hip -o libstamp.so      (scripts/phase_timeline.py builds and loads it)
#include <hip/hip_runtime.h>
__global__ void stamp_kernel(unsigned long long* slots, int idx) { slots[idx] = __builtin_amdgcn_s_memrealtime(); }
extern "C" int stamp(unsigned long long* slots, int idx, void* stream) {
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slots, idx);
    return (int)hipGetLastError();
}
