// Kernels and constants shared by conv_igemm.hip and conv_api.hip (each TU gets its own copy).
#pragma once
#include "gs_common.h"

namespace gs {

enum { MODE_S1 = 0, MODE_S2 = 1, MODE_T2 = 2 };
#define GS_WGRAD_MAX_SRC 4   // (x, gy) pairs one weight-gradient launch contracts (gs_conv2d_bwd_weight_bias_multi)

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Several (x, gy) pairs of ONE layer -- the real and the fake discriminator pass, the second-order contribution -- are contracted
// by one launch: the images of all sources form one list (source s owns images n_end[s-1] .. n_end[s] - 1), so the layer costs one set of block
// partials and one slice reduction instead of one per pair.  bias_mask: which sources contribute to the bias gradient.
struct WgradSrcs {
    const void* x[GS_WGRAD_MAX_SRC];
    const void* gy[GS_WGRAD_MAX_SRC];
    int n_end[GS_WGRAD_MAX_SRC];   // cumulative image counts (unused entries = the total)
    unsigned bias_mask;
};
__device__ __forceinline__ int wgrad_source(const WgradSrcs& s, int nn, int& n_local) {
    const int src = (nn >= s.n_end[0]) + (nn >= s.n_end[1]) + (nn >= s.n_end[2]);
    n_local = nn - (src ? s.n_end[src - 1] : 0);
    return src;
}

// ---- grouped weight gradients (gs_conv_wgrad_jobs): the 64 x 64 channel-tile kernel run ONCE over the layers of a backward pass that
// share its instantiation (conv mode, tile width).  The work of the group is one list of units -- (layer, channel tile, pixel tile), in that
// order -- cut into equal contiguous ranges, one per block (a "stream-K" schedule).  A block keeps its accumulators across the pixel
// tiles of one (layer, channel tile) RUN and writes a partial only where its range leaves the run: partials = blocks + runs per
// launch instead of blocks per LAYER, and the summation order of a run (ascending block index) is a function of the shapes alone.
#define GS_SK_MAX_JOBS 16
#define GS_SK_PSTRIDE (9 * 64 * 64 + 64)   // floats of one partial: 9 taps of a 64 x 64 tile + 64 bias sums
struct SkJob {
    WgradSrcs srcs;
    float* gw;                      // [9][IC][OC] (or [9][OC][IC] when transpose), fp32
    float* gb;                      // optional [OC]
    float alpha;
    int transpose, accumulate;
    int Hi, Wi, IC, OC, Hb, Wb;     // kernel-role geometry (input side / gradient side)
    int ICld;                       // input-channel rows of the stored variable (IC, or more when gw is a channel slice of a wider one)
    int tiles_x, tiles_y, ntiles;   // pixel tiles over the images of all sources
    int n_ict, nct;                 // IC / 64, channel tiles (IC / 64) * (OC / 64)
    int unit_base, run_base;        // first unit / first run of the layer inside the group
};
struct SkGroup {
    int njobs, total_units, total_runs, nblocks;
    SkJob job[GS_SK_MAX_JOBS];
};
// block that owns unit u when block b owns [b * total / nb, (b + 1) * total / nb)
__host__ __device__ inline int sk_block_of(long u, long nb, long total) { return (int)(((u + 1) * nb - 1) / total); }

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
    // a thread sums 4 consecutive elements (one 16-byte load per slice) over its share of the slices; EPB element quads per block
    constexpr int EPB = 256 / L;
    __shared__ float4 red[256];
    const long total = (long)taps * ic * oc;
    const long pstride = total + (gb ? oc : 0);   // a slice = the taps (+ one row of bias sums when gb is given); multiple of 4
    const long e = ((long)blockIdx.x * EPB + (threadIdx.x % EPB)) * 4;
    const int sl = threadIdx.x / EPB;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (e < pstride) {
        int k = sl;
        // eight slices per trip, all eight loads in flight together (the thin colour-block gradients fold 1024 slices of 64 floats in ONE
        // block: with two loads per trip that was 32 dependent round trips, 12 us)
        for (; k + 7 * L < nslices; k += 8 * L) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(part + (long)(k + j * L) * pstride + e);
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                s0.x += v[j].x; s0.y += v[j].y; s0.z += v[j].z; s0.w += v[j].w;
                s1.x += v[j + 1].x; s1.y += v[j + 1].y; s1.z += v[j + 1].z; s1.w += v[j + 1].w;
            }
        }
        for (; k + L < nslices; k += 2 * L) {
            const float4 a = *reinterpret_cast<const float4*>(part + (long)k * pstride + e);
            const float4 b = *reinterpret_cast<const float4*>(part + (long)(k + L) * pstride + e);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        }
        if (k < nslices) {
            const float4 a = *reinterpret_cast<const float4*>(part + (long)k * pstride + e);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
    }
    red[threadIdx.x] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    if (sl == 0 && e < pstride) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const float4 v = red[threadIdx.x + j * EPB];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        const float s[4] = {t.x, t.y, t.z, t.w};
        if (e >= total) {   // bias gradient: no equalized-LR scale
            float4* o = reinterpret_cast<float4*>(gb + (e - total));
            const float4 old = accumulate ? *o : make_float4(0.f, 0.f, 0.f, 0.f);
            *o = make_float4(old.x + s[0], old.y + s[1], old.z + s[2], old.w + s[3]);
            return;
        }
        if (!transpose) {
            float4* o = reinterpret_cast<float4*>(gw + e);
            const float4 old = accumulate ? *o : make_float4(0.f, 0.f, 0.f, 0.f);
            *o = make_float4(old.x + s[0] * alpha, old.y + s[1] * alpha, old.z + s[2] * alpha, old.w + s[3] * alpha);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const long ee = e + c;
                const int o = ee % oc;
                const int i = (ee / oc) % ic;
                const int tp = ee / ((long)ic * oc);
                const long dst = ((long)tp * oc + o) * ic + i;
                gw[dst] = accumulate ? gw[dst] + s[c] * alpha : s[c] * alpha;
            }
        }
    }
}

// element counts that are not a multiple of 4 (odd channel counts of the direct kernels): one element per thread
static __global__ __launch_bounds__(256) void wgrad_reduce_scalar_kernel(const float* __restrict__ part, float* __restrict__ gw, int nslices, int taps, int ic,
                                                                         int oc, float alpha, int transpose, int accumulate) {
    const long total = (long)taps * ic * oc;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    float s = 0.f;
    for (int k = 0; k < nslices; ++k) s += part[(long)k * total + e];
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

// ---- many reductions per launch (gs_wgrad_reduce_batch): blockIdx.y = entry, blockIdx.x = element chunk of that entry
#define GS_REDUCE_BATCH 16
struct ReduceBatch {
    GsWgradReduce e[GS_REDUCE_BATCH];
};
// L = 4: block = 64 consecutive element quads (1 KiB per slice row: whole DRAM bursts) x 4 slice lanes; L = 16: 16 quads x 16 slice lanes
// (the thin top-level layers leave 256 slices of 18 K floats each: with 4 slice lanes that is 72 blocks per entry walking 64 slices per
// thread, four loads in flight -- 32-37 us for 19 MB; 16 lanes put four times the loads in flight on four times the blocks).  A thread keeps
// four slice rows in flight.
// The lane count is a function of the ENTRY (its slice count), never of what else shares the launch: the association of an entry's sum must
// not depend on how the caller batches the folds (a bucketed flush batches them differently, and two schedules of one iteration must agree).
static __global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const ReduceBatch b) {
    const GsWgradReduce& d = b.e[blockIdx.y];
    const int nslices = d.nslices, oc = d.oc, ic = d.ic;
    const int L = nslices > 32 ? 16 : 4, EPB = 256 / L;
    __shared__ float4 red[256];
    const long total = (long)d.taps * ic * oc;
    const long pstride = total + (d.gb ? oc : 0);
    const long e = ((long)blockIdx.x * EPB + (threadIdx.x % EPB)) * 4;
    if ((long)blockIdx.x * EPB * 4 >= pstride) return;   // (whole block past the end of this entry)
    const int sl = threadIdx.x / EPB;
    const float* __restrict__ part = d.partials;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    if (e < pstride) {
        int k = sl;
        for (; k + 3 * L < nslices; k += 4 * L) {
            const float4 a0 = *reinterpret_cast<const float4*>(part + (long)k * pstride + e);
            const float4 a1 = *reinterpret_cast<const float4*>(part + (long)(k + L) * pstride + e);
            const float4 a2 = *reinterpret_cast<const float4*>(part + (long)(k + 2 * L) * pstride + e);
            const float4 a3 = *reinterpret_cast<const float4*>(part + (long)(k + 3 * L) * pstride + e);
            s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
            s1.x += a1.x; s1.y += a1.y; s1.z += a1.z; s1.w += a1.w;
            s2.x += a2.x; s2.y += a2.y; s2.z += a2.z; s2.w += a2.w;
            s3.x += a3.x; s3.y += a3.y; s3.z += a3.z; s3.w += a3.w;
        }
        for (; k < nslices; k += L) {
            const float4 a = *reinterpret_cast<const float4*>(part + (long)k * pstride + e);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
    }
    s0.x += s2.x; s0.y += s2.y; s0.z += s2.z; s0.w += s2.w;
    s1.x += s3.x; s1.y += s3.y; s1.z += s3.z; s1.w += s3.w;
    red[threadIdx.x] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    if (sl != 0 || e >= pstride) return;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < L; ++j) {
        const float4 v = red[threadIdx.x + j * EPB];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    const float s[4] = {t.x, t.y, t.z, t.w};
    const int accumulate = d.accumulate;
    if (e >= total) {   // bias gradient: no equalized-LR scale
        float4* o = reinterpret_cast<float4*>(d.gb + (e - total));
        const float4 old = accumulate ? *o : make_float4(0.f, 0.f, 0.f, 0.f);
        *o = make_float4(old.x + s[0], old.y + s[1], old.z + s[2], old.w + s[3]);
        return;
    }
    const float alpha = d.alpha;
    float* __restrict__ gw = d.gw;
    if (!d.transpose) {
        long dst = e;
        if (d.ic_ld > ic) {   // channel slice of a wider variable: tap t starts at t * ic_ld * oc (a quad never straddles taps: ic * oc % 4 == 0)
            const long per_tap = (long)ic * oc;
            const long tp = e / per_tap;
            dst = tp * d.ic_ld * oc + (e - tp * per_tap);
        }
        float4* o = reinterpret_cast<float4*>(gw + dst);
        const float4 old = accumulate ? *o : make_float4(0.f, 0.f, 0.f, 0.f);
        *o = make_float4(old.x + s[0] * alpha, old.y + s[1] * alpha, old.z + s[2] * alpha, old.w + s[3] * alpha);
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long ee = e + c;
            const int o = ee % oc;
            const int i = (ee / oc) % ic;
            const int tp = ee / ((long)ic * oc);
            const long dst = ((long)tp * oc + o) * ic + i;
            gw[dst] = accumulate ? gw[dst] + s[c] * alpha : s[c] * alpha;
        }
    }
}

static inline size_t wgrad_reduce_extra(long nslices, long total) { (void)nslices; (void)total; return 0; }
static inline void wgrad_reduce_launch(float* part, float* gw, float* gb, int nslices, int taps, int ic, int oc, float alpha, int transpose, int accumulate, hipStream_t st,
                                       GsWgradReduce* defer = nullptr) {
    const bool vec = !((((long)taps * ic * oc) & 3) != 0 || (gb && (oc & 3) != 0));
    if (defer && vec) {   // phase 2 is left to gs_wgrad_reduce_batch
        defer->partials = part; defer->gw = gw; defer->gb = gb;
        defer->nslices = nslices; defer->taps = taps; defer->ic = ic; defer->oc = oc;
        defer->alpha = alpha; defer->transpose = transpose; defer->accumulate = accumulate; defer->ic_ld = 0;
        return;
    }
    if (!vec) {
        const long total = (long)taps * ic * oc;   // (gb never comes with such shapes: the fused bias path needs oc % 32 == 0)
        hipLaunchKernelGGL(wgrad_reduce_scalar_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part, gw, nslices, taps, ic, oc, alpha, transpose, accumulate);
        return;
    }
    const long n4 = ((long)taps * ic * oc + (gb ? oc : 0)) / 4;   // element quads
    if (nslices <= 32) {
        hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, st, part, gw, gb, nslices, taps, ic, oc, alpha, transpose, accumulate);
    } else {
        hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, part, gw, gb, nslices, taps, ic, oc, alpha, transpose, accumulate);
    }
}

}  // namespace gs
