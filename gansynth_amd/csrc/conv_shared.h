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
// Block = (256 / L) consecutive elements x L slice lanes; one launch whatever the slice count (L = 4 for a few slices,
// 16 for the hundreds of slices of the thin top-of-pyramid layers); fixed summation order -> deterministic.
template <int L>
static __global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, float* __restrict__ gb, int nslices,
                                                                  int taps, int ic, int oc, float alpha, int transpose, int accumulate) {
    constexpr int EPB = 256 / L;
    __shared__ float red[256];
    const long total = (long)taps * ic * oc;
    const long pstride = total + (gb ? oc : 0);   // a slice = the taps (+ one row of bias sums when gb is given)
    const long e = (long)blockIdx.x * EPB + (threadIdx.x % EPB);
    const int sl = threadIdx.x / EPB;
    float s0 = 0.f, s1 = 0.f;
    if (e < pstride) {
        int k = sl;
        for (; k + L < nslices; k += 2 * L) {
            s0 += part[(long)k * pstride + e];
            s1 += part[(long)(k + L) * pstride + e];
        }
        if (k < nslices) s0 += part[(long)k * pstride + e];
    }
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (sl == 0 && e < pstride) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < L; ++j) s += red[threadIdx.x + j * EPB];
        if (e >= total) {   // bias gradient: no equalized-LR scale
            float* o = gb + (e - total);
            *o = accumulate ? *o + s : s;
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

static inline size_t wgrad_reduce_extra(long nslices, long total) { (void)nslices; (void)total; return 0; }
static inline void wgrad_reduce_launch(float* part, float* gw, float* gb, int nslices, int taps, int ic, int oc, float alpha, int transpose, int accumulate, hipStream_t st) {
    const long n = (long)taps * ic * oc + (gb ? oc : 0);
    if (nslices <= 32) {
        hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, part, gw, gb, nslices, taps, ic, oc, alpha, transpose, accumulate);
    } else {
        hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, part, gw, gb, nslices, taps, ic, oc, alpha, transpose, accumulate);
    }
}

}  // namespace gs
