// Shader clock and MFMA issue rate under load: every wave runs K x 16 independent v_mfma_f32_32x32x16_bf16 and stamps
// s_memtime (shader-clock counter) and s_memrealtime (constant 100 MHz) around them.
// usage: mfma_clock <blocks> <threads per block> <K>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(unsigned long long* out, float* sink, int K, float seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 0.001f + i); b[i] = (__bf16)(seed * 0.5f + i * 0.25f + threadIdx.x * 0.002f); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < K; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = r1 - r0;
    }
}

int main(int argc, char** argv) {
    const int blocks = atoi(argv[1]), threads = atoi(argv[2]), K = atoi(argv[3]);
    const int waves = blocks * threads / 64;
    unsigned long long* d; float* sink;
    hipMalloc(&d, waves * 16); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, sink, K, 1.0f);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(2 * waves);
    hipMemcpy(h.data(), d, waves * 16, hipMemcpyDeviceToHost);
    double st = 0, sr = 0;
    for (int w = 0; w < waves; ++w) { st += h[2 * w]; sr += h[2 * w + 1]; }
    st /= waves; sr /= waves;
    const double n = (double)K * 16;
    printf("blocks %d x %d threads, %d MFMAs per wave: event %.1f us | memtime %.0f ticks (%.2f per MFMA), realtime %.0f ticks = %.2f us -> memtime clock %.0f MHz | %.2f ns per MFMA per wave, %.1f TFLOP/s\n",
           blocks, threads, (int)n, ms * 1e3, st, st / n, sr, sr / 100.0, st / sr * 100.0, sr * 10.0 / n, 32768.0 * n * waves / (ms * 1e-3) / 1e12);
    return 0;
}
