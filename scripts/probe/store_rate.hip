// Global-store rate on gfx950 by lane -> address pattern (what an MFMA epilogue's stores look like vs fully contiguous ones).
// Every wave streams its own slice of a large buffer, 16 bytes per lane per instruction; only the mapping of lanes to addresses differs.
// build: hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip ; run: ./store_rate [MB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// PAT 0: lane L -> 16 L (1 KiB contiguous per instruction)
// PAT 1: 32-channel bf16 tile, lane (px = L & 31, hi = L >> 5): instr k in {0,1} -> px * 64 + k * 32 + hi * 16     (2 KiB per pair)
// PAT 2: 64-channel tile of a 64-channel tensor: instr k in 0..3 -> px * 128 + k * 32 + hi * 16                  (4 KiB per four)
// PAT 3: 64-channel tile of a 256-channel tensor: instr k in 0..3 -> px * 512 + k * 32 + hi * 16                 (rows 512 B apart)
// PAT 4: as 3 after a transpose: instr k in 0..3 -> (8 k + L / 8) * 512 + (L & 7) * 16                           (8 whole 128-B rows per instruction)
// PAT 5: as 1 after a transpose: instr k in {0,1} -> k * 1024 + 16 L
template <int PAT>
__global__ __launch_bounds__(256) void k(uint4* out, long bytes_per_wave, int nt) {
    const int lane = threadIdx.x & 63, px = lane & 31, hi = lane >> 5;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const u4 v = {1u, 2u, 3u, (unsigned)lane};
    if (PAT == 3 || PAT == 4) {
        // the tensor is [pixels][256 ch] bf16; the wave owns channel tile (wave & 3) of pixel groups of 32
        char* base = (char*)out + (wave >> 2) * (bytes_per_wave * 4) + (wave & 3) * 128;
        for (long g = 0; g < bytes_per_wave / 4096; ++g) {   // 32 pixels x 128 B per group
            char* b = base + g * 32 * 512;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                char* a = PAT == 3 ? b + px * 512 + kk * 32 + hi * 16 : b + (8 * kk + lane / 8) * 512 + (lane & 7) * 16;
                if (nt) __builtin_nontemporal_store(v, (u4*)a); else *(u4*)a = v;
            }
        }
        return;
    }
    char* base = (char*)out + wave * bytes_per_wave;
    constexpr int CH = PAT == 0 ? 1024 : (PAT == 2 ? 4096 : 2048);
    for (long g = 0; g < bytes_per_wave / CH; ++g) {
        char* b = base + g * CH;
        if (PAT == 0) { if (nt) __builtin_nontemporal_store(v, (u4*)(b + lane * 16)); else *(u4*)(b + lane * 16) = v; }
        if (PAT == 1) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) { char* a = b + px * 64 + kk * 32 + hi * 16; if (nt) __builtin_nontemporal_store(v, (u4*)a); else *(u4*)a = v; }
        }
        if (PAT == 2) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { char* a = b + px * 128 + kk * 32 + hi * 16; if (nt) __builtin_nontemporal_store(v, (u4*)a); else *(u4*)a = v; }
        }
        if (PAT == 5) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) { char* a = b + kk * 1024 + lane * 16; if (nt) __builtin_nontemporal_store(v, (u4*)a); else *(u4*)a = v; }
        }
    }
}

template <int PAT>
static void run(uint4* buf, long total, int blocks, int nt) {
    const long waves = (long)blocks * 4;
    long bpw = total / waves / 4096 * 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 0, 0, buf, bpw, nt);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("pattern %d nt %d blocks %5d: %.1f MB in %.1f us = %.0f GB/s\n", PAT, nt, blocks, bpw * waves / 1e6, best * 1e3, bpw * waves / (best * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    const long mb = argc > 1 ? atol(argv[1]) : 64;
    uint4* buf;
    hipMalloc(&buf, (mb + 8) << 20);
    for (int blocks : {256, 512, 2048})
        for (int nt : {0, 1}) {
            run<0>(buf, mb << 20, blocks, nt); run<1>(buf, mb << 20, blocks, nt); run<5>(buf, mb << 20, blocks, nt);
            run<2>(buf, mb << 20, blocks, nt); run<3>(buf, mb << 20, blocks, nt); run<4>(buf, mb << 20, blocks, nt);
        }
    // small: one 32 KB tile per block, every block at once (the epilogue burst of a mid layer)
    for (int nt : {0, 1}) { run<3>(buf, 8 << 20, 256, nt); run<4>(buf, 8 << 20, 256, nt); run<0>(buf, 8 << 20, 256, nt); }
    return 0;
}
