// LDS-DMA throughput per CU by source access pattern (L2-resident or HBM-streamed source).
// Every wave keeps W pieces (1 KiB wave-instructions) in flight; patterns differ in how the 64 lanes' 16-byte fetches
// fall on 128-byte cache lines.
// usage: dma_rate <pattern 0..4> <blocks> <waves per block> <pieces per wave> <window> <footprint MiB>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, i32x4 rs, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int W> __device__ __forceinline__ void wait_w() {
    if constexpr (W == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (W == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (W == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (W == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
}
template <int W>
__global__ void k(const unsigned char* x, unsigned long long* out, unsigned span, int pattern, int pieces) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;
    const unsigned long long base = (unsigned long long)x;
    i32x4 rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
    rs[1] = __builtin_amdgcn_readfirstlane((int)((base >> 32) & 0xffff));
    rs[2] = __builtin_amdgcn_readfirstlane((int)span);
    rs[3] = 0x00020000;
    unsigned voff, step;
    // bytes a piece advances (step) so that the whole run covers distinct data in every pattern
    if (pattern == 0) { voff = lane * 16; step = 1024; }                                   // contiguous 1 KiB
    else if (pattern == 1) { voff = (lane >> 2) * 256 + (lane & 3) * 16; step = 16 * 256; } // 16 rows x 64 B, row stride 256 (half lines, other halves never read)
    else if (pattern == 2) { voff = (lane >> 3) * 256 + (lane & 7) * 16; step = 8 * 256; }  // 8 rows x 128 B, row stride 256 (whole lines)
    else if (pattern == 3) { voff = (lane >> 2) * 128 + (lane & 3) * 16; step = 16 * 128; } // 16 rows x 64 B, row stride 128 (half lines; the other half by the next pass)
    else { voff = (lane >> 2) * 64 + (lane & 3) * 16; step = 1024; }                        // contiguous again, written as rows
    const unsigned wave_id = blockIdx.x * nw + wv;
    const unsigned lbase = (unsigned)(uintptr_t)lds + wv * (W * 1024);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned soff = (unsigned)(((unsigned long long)wave_id * pieces * step) % (span - 65536));
    for (int i = 0; i < pieces; ++i) {
        dma16(lbase + (i % W) * 1024, voff, rs, soff);
        soff += step;
        if (soff >= span - 65536) soff -= span - 65536;
        wait_w<W>();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[wave_id] = t1 - t0;
}
int main(int argc, char** argv) {
    const int pattern = atoi(argv[1]), blocks = atoi(argv[2]), nw = atoi(argv[3]), pieces = atoi(argv[4]), W = atoi(argv[5]);
    const size_t mib = atoi(argv[6]);
    unsigned char* x; unsigned long long* out;
    hipMalloc(&x, mib << 20); hipMemset(x, 1, mib << 20); hipMalloc(&out, blocks * nw * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        const unsigned span = (unsigned)(mib << 20);
        const size_t lds = (size_t)nw * W * 1024;
#define L(WW) hipLaunchKernelGGL(k<WW>, dim3(blocks), dim3(nw * 64), lds, 0, x, out, span, pattern, pieces)
        if (W == 1) L(1); else if (W == 2) L(2); else if (W == 4) L(4); else if (W == 8) L(8); else L(16);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(blocks * nw);
    hipMemcpy(h.data(), out, blocks * nw * 8, hipMemcpyDeviceToHost);
    double t = 0; for (auto v : h) t += v; t /= h.size();
    const double bytes_cu = (double)nw * pieces * 1024 * (blocks > 256 ? blocks / 256.0 : 1.0);
    printf("pattern %d, %d blocks x %d waves, %d pieces/wave, window %d, footprint %zu MiB: %.1f us, %.0f ticks/wave -> %.1f B/tick/CU, %.2f TB/s total\n",
           pattern, blocks, nw, pieces, W, mib, ms * 1e3, t, bytes_cu / t, (double)blocks * nw * pieces * 1024 / (ms * 1e-3) / 1e12);
    return 0;
}
