// Kernels and constants shared by conv_igemm.hip and conv_api.hip (each TU gets its own copy).
#pragma once
#include "gs_common.h"

namespace gs {

enum { MODE_S1 = 0, MODE_S2 = 1, MODE_T2 = 2 };

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// ------------------------------------------------------------------------------ weight prep
// Re-lays the fp32 HWIO master weight into the kernel operand Wp[tap][OCk][ICk] (ICk contiguous,
// storage type T, no scaling -- alpha is applied to the fp32 accumulators).
//   variant 0 (fwd)        : Wp[t][co][ci]       = w[t][ci][co]      (OCk = co, ICk = ci)
//   variant 1 (bwd-data S1): Wp[taps-1-t][ci][co] = w[t][ci][co]     (OCk = ci, ICk = co; taps flipped)
//   variant 2 (bwd-data T2): Wp[t][ci][co]       = w[t][ci][co]      (OCk = ci, ICk = co)
template <typename T>
__global__ void weight_prep_kernel(const float* __restrict__ w, T* __restrict__ wp, int taps, int ci, int co, int variant) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)taps * ci * co;
    if (idx >= total) return;
    if (variant == 0) {
        int c_i = idx % ci;
        int c_o = (idx / ci) % co;
        int t = idx / ((long)ci * co);
        DT<T>::st(wp + idx, w[((long)t * ci + c_i) * co + c_o]);
    } else {
        int t = idx / ((long)ci * co);
        long rem = idx % ((long)ci * co);
        int tt = variant == 1 ? taps - 1 - t : t;
        DT<T>::st(wp + (long)tt * ci * co + rem, w[idx]);
    }
}

// gw[e] = alpha * sum_s part[s][e]; `transpose` swaps the last two dims on output
// (used by conv2d_transpose's weight gradient, whose stored variable is [k][k][Cin_T][Cout_T]).
// Block = 64 consecutive elements x 4 slice lanes over the slab [blockIdx.y*per, +per) of slices; fixed
// summation order -> deterministic.  With `raw` the slab sum is written un-scaled to out[blockIdx.y][e]
// (first level of a two-level reduction when there are many slices).
static __global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, int nslices,
                                                                  int taps, int ic, int oc, float alpha, int transpose, int per, int raw, int accumulate) {
    __shared__ float red[256];
    const long total = (long)taps * ic * oc;
    const long e = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    const int sl = threadIdx.x >> 6;
    const int k0 = blockIdx.y * per;
    int k1 = k0 + per;
    if (k1 > nslices) k1 = nslices;
    float s0 = 0.f, s1 = 0.f;
    if (e < total) {
        int k = k0 + sl;
        for (; k + 4 < k1; k += 8) {
            s0 += part[(long)k * total + e];
            s1 += part[(long)(k + 4) * total + e];
        }
        if (k < k1) s0 += part[(long)k * total + e];
    }
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (sl == 0 && e < total) {
        float s = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
        if (raw) {
            gw[(long)blockIdx.y * total + e] = s;
            return;
        }
        s *= alpha;
        long dst = e;
        if (transpose) {
            const int o = e % oc;
            const int i = (e / oc) % ic;
            const int t = e / ((long)ic * oc);
            dst = ((long)t * oc + o) * ic + i;
        }
        gw[dst] = accumulate ? gw[dst] + s : s;
    }
}

constexpr int WG_SLAB = 64;  // slices per first-level block
// extra fp32 elements the partial buffer needs behind its nslices*total partials
static inline size_t wgrad_reduce_extra(long nslices, long total) { return nslices > WG_SLAB ? (size_t)((nslices + WG_SLAB - 1) / WG_SLAB) * total : 0; }
static inline void wgrad_reduce_launch(float* part, float* gw, int nslices, int taps, int ic, int oc, float alpha, int transpose, int accumulate, hipStream_t st) {
    const long total = (long)taps * ic * oc;
    const unsigned gx = (unsigned)((total + 63) / 64);
    if (nslices <= WG_SLAB) {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gx, 1), dim3(256), 0, st, part, gw, nslices, taps, ic, oc, alpha, transpose, nslices, 0, accumulate);
    } else {
        const int nsplit = (nslices + WG_SLAB - 1) / WG_SLAB;
        float* part2 = part + (long)nslices * total;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gx, nsplit), dim3(256), 0, st, part, part2, nslices, taps, ic, oc, 1.f, 0, WG_SLAB, 1, 0);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gx, 1), dim3(256), 0, st, part2, gw, nsplit, taps, ic, oc, alpha, transpose, nsplit, 0, accumulate);
    }
}

}  // namespace gs
