// Probe: buffer_load_dwordx4 ... lds  (LDS-DMA) semantics on gfx950: destination order, out-of-range lanes, soffset in the range check.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ inline void dma16(unsigned lds_addr, unsigned voff, i32x4 rs, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

__global__ void k(const unsigned* x, unsigned* y, unsigned nbytes, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned* l32 = reinterpret_cast<unsigned*>(lds);
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) l32[i] = 0xFFFFFFFFu;
    __syncthreads();
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const unsigned long long base = (unsigned long long)x;
    i32x4 rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
    rs[1] = __builtin_amdgcn_readfirstlane((int)((base >> 32) & 0xffff));
    rs[2] = __builtin_amdgcn_readfirstlane((int)nbytes);
    rs[3] = 0x00020000;
    const unsigned lbase = (unsigned)(uintptr_t)lds + wv * 1024;
    unsigned voff, soff = 0;
    if (mode == 0) voff = (63 - lane) * 16;                       // reversed gather
    else if (mode == 1) voff = (lane & 1) ? 0x80000000u : lane * 16;  // odd lanes out of range
    else if (mode == 2) { voff = lane * 16; soff = nbytes - 512; }  // soffset pushes half the lanes past the end
    else voff = (lane & 1) ? (unsigned)(-16 - lane * 16) : lane * 16;  // negative offsets
    dma16(lbase, voff, rs, soff);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) y[i] = l32[i];
}

int main() {
    unsigned *x, *y;
    const unsigned n = 4096;  // dwords
    hipMalloc(&x, n * 4 + 4096);
    hipMalloc(&y, 1024);
    unsigned h[n + 1024];
    for (unsigned i = 0; i < n + 1024; ++i) h[i] = i;
    hipMemcpy(x, h, sizeof(h), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, x, y, n * 4, mode);
        unsigned o[256];
        hipMemcpy(o, y, 1024, hipMemcpyDeviceToHost);
        printf("mode %d:", mode);
        for (int l = 0; l < 64; l += (mode == 0 ? 21 : 1)) printf(" [%d]=%x", l, o[l * 4]);
        printf("\n");
    }
    return 0;
}
