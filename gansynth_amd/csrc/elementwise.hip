// HBM-bound elementwise / row-reduction kernels of the PGGAN hot path (channels-last).
//   bias + leaky_relu / tanh          ops.py:244-246, networks.py:55,66,80,91,106,184,194,216,227,241
//   pixel_normalization (+1st/2nd order gradients)          ops.py:330-333
//   upscale2d / downscale2d                                  ops.py:283-305
//   lerp                                                     networks.py:10-11
//   R1 penalty reductions                                    models.py:47-49
//   TF-form Adam                                             models.py:67-89
// All are one-read/one-write streaming kernels with 16-byte (f32) / 8-byte (bf16) accesses and
// wave64 shuffle reductions; none goes near the MFMA.
#include "gs_common.h"

namespace gs {

static inline int ew_grid(long nvec) {
    long g = (nvec + 255) / 256;
    static const long cap = getenv("GS_EW_GRID_CAP") ? atol(getenv("GS_EW_GRID_CAP")) : 8192;   // (measurement knob)
    if (g > cap) g = cap;  // grid-stride the rest (256 CUs x 8 blocks x 4)
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------ bias + activation
template <typename T, int ACT, bool BIAS>
__global__ void bias_act_kernel(const T* __restrict__ x, const float* __restrict__ bias, T* __restrict__ y, long nvec, int cvec) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        float v[4];
        ld4(x + i * 4, v);
        if (BIAS) {
            const float4 b = *reinterpret_cast<const float4*>(bias + (i % cvec) * 4);
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (ACT == GS_ACT_LRELU) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
            if (ACT == GS_ACT_TANH) v[e] = tanhf(v[e]);
        }
        st4(y + i * 4, v);
    }
}
// scalar fallback (c not a multiple of 4: the 2-channel images)
template <typename T>
__global__ void bias_act_scalar_kernel(const T* __restrict__ x, const float* __restrict__ bias, T* __restrict__ y, long n, int c, int act) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = DT<T>::ld(x + i);
        if (bias) v += bias[i % c];
        if (act == GS_ACT_LRELU) v = v > 0.f ? v : 0.2f * v;
        if (act == GS_ACT_TANH) v = tanhf(v);
        DT<T>::st(y + i, v);
    }
}

// gx = g * act'(y) through the activation output y
template <typename T, int ACT>
__global__ void act_bwd_kernel(const T* __restrict__ g, const T* __restrict__ y, T* __restrict__ gx, long n) {
    const long nvec = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        float gv[4], yv[4];
        ld4(g + i * 4, gv);
        ld4(y + i * 4, yv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (ACT == GS_ACT_LRELU) gv[e] = yv[e] > 0.f ? gv[e] : 0.2f * gv[e];
            if (ACT == GS_ACT_TANH) gv[e] = gv[e] * (1.f - yv[e] * yv[e]);
        }
        st4(gx + i * 4, gv);
    }
    // tail
    const long t = nvec * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        float gv = DT<T>::ld(g + t), yv = DT<T>::ld(y + t);
        if (ACT == GS_ACT_LRELU) gv = yv > 0.f ? gv : 0.2f * gv;
        if (ACT == GS_ACT_TANH) gv = gv * (1.f - yv * yv);
        DT<T>::st(gx + t, gv);
    }
}

template <typename T>
__global__ void tanh_bwd_bwd_kernel(const T* __restrict__ gg, const T* __restrict__ g, const T* __restrict__ y, T* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        DT<T>::st(out + i, -2.f * DT<T>::ld(y + i) * DT<T>::ld(g + i) * DT<T>::ld(gg + i));
}

// out = ca*a + cb*b
template <typename T>
__global__ void axpby_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, long n, float ca, float cb) {
    const long nvec = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        float av[4], bv[4];
        ld4(a + i * 4, av);
        ld4(b + i * 4, bv);
#pragma unroll
        for (int e = 0; e < 4; ++e) av[e] = ca * av[e] + cb * bv[e];
        st4(out + i * 4, av);
    }
    const long t = nvec * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) DT<T>::st(out + t, ca * DT<T>::ld(a + t) + cb * DT<T>::ld(b + t));
}

// ------------------------------------------------------------------------ channel sum
// out[c] = sum_p g[p][c].  Pass 1: block b sums rows b, b+G, ... into part[b][c]; pass 2 sums parts.
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(const T* __restrict__ g, float* __restrict__ part, long p, int c) {
    // thread handles channel quad q = tid % (c/4) (or single channels when c%4 != 0), row lane r = tid / quads
    __shared__ float red[256 * 4];
    const int tid = threadIdx.x;
    if ((c == 1 || c == 2 || c == 4) && ((p * c) & 7) == 0) {
        // colour-width tensors (the images: 2 channels): stream the flat array 8 elements per lane, element k belongs to
        // channel k % c
        const long n8 = (p * c) >> 3;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (long i = (long)blockIdx.x * 256 + tid; i < n8; i += (long)gridDim.x * 256) {
            float v[8];
            ld4(g + i * 8, *reinterpret_cast<float(*)[4]>(v));
            ld4(g + i * 8 + 4, *reinterpret_cast<float(*)[4]>(v + 4));
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += v[k];
        }
        for (int e = 0; e < c; ++e) {
            float mine = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) mine += (k & (c - 1)) == e ? a[k] : 0.f;   // (c is a power of two here)
            const float tot = block_sum<256>(mine, red);
            if (tid == 0) part[(long)blockIdx.x * c + e] = tot;
        }
    } else if ((c & 3) == 0 && c <= 1024) {
        const int quads = c >> 2;
        const int rl = 256 / quads > 0 ? 256 / quads : 1;  // row lanes per block
        const int q = tid % quads, r = tid / quads;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < rl && quads <= 256) {
            // four rows per trip: the loads of a trip are independent and in flight together (one load per trip leaves the
            // kernel waiting a full memory latency per 8 bytes)
            const long step = (long)gridDim.x * rl;
            long row = (long)blockIdx.x * rl + r;
            for (; row + 3 * step < p; row += 4 * step) {
                float v0[4], v1[4], v2[4], v3[4];
                ld4(g + row * c + q * 4, v0);
                ld4(g + (row + step) * c + q * 4, v1);
                ld4(g + (row + 2 * step) * c + q * 4, v2);
                ld4(g + (row + 3 * step) * c + q * 4, v3);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] += (v0[e] + v1[e]) + (v2[e] + v3[e]);
            }
            for (; row < p; row += step) {
                float v[4];
                ld4(g + row * c + q * 4, v);
                a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) red[tid * 4 + e] = a[e];
        __syncthreads();
        if (tid < quads) {
            float s[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < rl; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] += red[(k * quads + tid) * 4 + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) part[(long)blockIdx.x * c + tid * 4 + e] = s[e];
        }
    } else {
        // generic: thread per (channel, row lane)
        const int cl = c < 256 ? c : 256;
        const int rl = 256 / cl;
        const int ch = tid % cl, r = tid / cl;
        for (int c0 = 0; c0 < c; c0 += cl) {
            float a = 0.f;
            if (r < rl && c0 + ch < c)
                for (long row = (long)blockIdx.x * rl + r; row < p; row += (long)gridDim.x * rl) a += DT<T>::ld(g + row * c + c0 + ch);
            __syncthreads();
            red[tid] = a;
            __syncthreads();
            if (tid < cl && c0 + tid < c) {
                float s = 0.f;
                for (int k = 0; k < rl; ++k) s += red[k * cl + tid];
                part[(long)blockIdx.x * c + c0 + tid] = s;
            }
        }
    }
}
// fused backward of (bias + activation): gx = g * act'(y) AND part[block][c] = sum over the block's rows of gx
// (one pass over g and y instead of two; c % 4 == 0, c <= 1024)
template <typename T, int ACT>
__global__ __launch_bounds__(256) void act_bwd_sum_kernel(const T* __restrict__ g, const T* __restrict__ y, T* __restrict__ gx,
                                                          float* __restrict__ part, long p, int c) {
    __shared__ float red[256 * 4];
    const int tid = threadIdx.x;
    const int quads = c >> 2;
    const int rl = 256 / quads > 0 ? 256 / quads : 1;
    const int q = tid % quads, r = tid / quads;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < rl) {
        const long step = (long)gridDim.x * rl;
        long row = (long)blockIdx.x * rl + r;
        for (; row + step < p; row += 2 * step) {   // two rows per trip: four independent loads in flight
            float g0[4], y0[4], g1[4], y1[4];
            ld4(g + row * c + q * 4, g0);
            ld4(y + row * c + q * 4, y0);
            ld4(g + (row + step) * c + q * 4, g1);
            ld4(y + (row + step) * c + q * 4, y1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (ACT == GS_ACT_LRELU) { g0[e] = y0[e] > 0.f ? g0[e] : 0.2f * g0[e]; g1[e] = y1[e] > 0.f ? g1[e] : 0.2f * g1[e]; }
                if (ACT == GS_ACT_TANH) { g0[e] = g0[e] * (1.f - y0[e] * y0[e]); g1[e] = g1[e] * (1.f - y1[e] * y1[e]); }
                a[e] += g0[e] + g1[e];
            }
            st4(gx + row * c + q * 4, g0);
            st4(gx + (row + step) * c + q * 4, g1);
        }
        for (; row < p; row += step) {
            float gv[4], yv[4];
            ld4(g + row * c + q * 4, gv);
            ld4(y + row * c + q * 4, yv);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (ACT == GS_ACT_LRELU) gv[e] = yv[e] > 0.f ? gv[e] : 0.2f * gv[e];
                if (ACT == GS_ACT_TANH) gv[e] = gv[e] * (1.f - yv[e] * yv[e]);
                a[e] += gv[e];
            }
            st4(gx + row * c + q * 4, gv);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[tid * 4 + e] = a[e];
    __syncthreads();
    if (tid < quads) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < rl; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += red[(k * quads + tid) * 4 + e];
#pragma unroll
        for (int e = 0; e < 4; ++e) part[(long)blockIdx.x * c + tid * 4 + e] = s[e];
    }
}

// out[s][ch] = sum over the s-th slab of `per` parts of part[k][ch]: block = 32 channels x 8 part lanes.
// Called once (nsplit = 1) or twice (first level writes nsplit rows, second level sums them): fixed order.
static __global__ __launch_bounds__(256) void channel_sum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, int c, int per, int accumulate) {
    __shared__ float red[256];
    const int ch = blockIdx.x * 32 + (threadIdx.x & 31);
    const int pl = threadIdx.x >> 5;
    const int k0 = blockIdx.y * per;
    int k1 = k0 + per;
    if (k1 > nparts) k1 = nparts;
    float s = 0.f;
    if (ch < c)
        for (int k = k0 + pl; k < k1; k += 8) s += part[(long)k * c + ch];
    red[threadIdx.x] = s;
    __syncthreads();
    if (pl == 0 && ch < c) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k * 32 + threadIdx.x];
        float* o = out + (long)blockIdx.y * c + ch;
        *o = accumulate ? *o + t : t;
    }
}
constexpr int CS_SLAB = 64;  // parts per first-level block
static int channel_sum_finalize(float* part, float* out, int nparts, int c, int accumulate, hipStream_t st) {
    if (nparts <= CS_SLAB) {
        hipLaunchKernelGGL(channel_sum_final_kernel, dim3(cdiv(c, 32), 1), dim3(256), 0, st, part, out, nparts, c, nparts, accumulate);
    } else {
        const int nsplit = cdiv(nparts, CS_SLAB);
        float* part2 = part + (long)nparts * c;
        hipLaunchKernelGGL(channel_sum_final_kernel, dim3(cdiv(c, 32), nsplit), dim3(256), 0, st, part, part2, nparts, c, CS_SLAB, 0);
        hipLaunchKernelGGL(channel_sum_final_kernel, dim3(cdiv(c, 32), 1), dim3(256), 0, st, part2, out, nsplit, c, nsplit, accumulate);
    }
    GS_CHECK_LAUNCH();
    return 0;
}
// ---- deferred folds.  A backward pass has ~25 bias gradients whose producers (act_bwd_sum, the pixel-norm backward, channel_sum)
// leave one partial row per block; folding each right behind its producer costs one or two 3-us launches plus the gap between
// kernels every time.  With GS_SUM_PARTIALS in `accumulate` a producer only writes its partial rows (ws = the caller's own
// buffer of gs_bias_partial_rows x c floats, kept until the fold) and gs_channel_fold_batch folds ALL of them in one launch
// (two when a producer left more than CS_SLAB rows), in the order and association channel_sum_finalize uses: same bits.
#define GS_FOLD_HEADS 32
#define GS_FOLD_SRCS 48
struct FoldSrc {
    const float* part;
    int nparts, per;
};
struct FoldHead {   // one target: its sources are folded one after the other by the same block (call order: (out + t1) + t2 like the immediate folds)
    float* out;
    int c, accumulate, blk0, chblocks, src0, nsrc;
};
struct FoldBatch {
    FoldHead h[GS_FOLD_HEADS];
    FoldSrc s[GS_FOLD_SRCS];
    int n;
};
static __global__ __launch_bounds__(256) void channel_fold_batch_kernel(const FoldBatch b) {
    __shared__ float red[256];
    int k = 0;
    while (k + 1 < b.n && (int)blockIdx.x >= b.h[k + 1].blk0) ++k;
    float* const out = b.h[k].out;
    const int c = b.h[k].c, src0 = b.h[k].src0, nsrc = b.h[k].nsrc;
    const int local = (int)blockIdx.x - b.h[k].blk0, chb = local % b.h[k].chblocks, slab = local / b.h[k].chblocks;
    const int ch = chb * 32 + (threadIdx.x & 31);
    const int pl = threadIdx.x >> 5;
    float acc = 0.f;
    if (pl == 0 && ch < c && b.h[k].accumulate) acc = out[(long)slab * c + ch];
    for (int i = 0; i < nsrc; ++i) {
        const float* const part = b.s[src0 + i].part;
        const int nparts = b.s[src0 + i].nparts, per = b.s[src0 + i].per;
        const int k0 = slab * per;
        int k1 = k0 + per;
        if (k1 > nparts) k1 = nparts;
        float s = 0.f;
        if (ch < c)
            for (int r = k0 + pl; r < k1; r += 8) s += part[(long)r * c + ch];
        if (i) __syncthreads();
        red[threadIdx.x] = s;
        __syncthreads();
        if (pl == 0 && ch < c) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += red[q * 32 + threadIdx.x];
            acc = (i || b.h[k].accumulate) ? acc + t : t;
        }
    }
    if (pl == 0 && ch < c) out[(long)slab * c + ch] = acc;
}
struct FoldTarget {
    float* out;
    int c, accumulate;
    std::vector<FoldSrc> src;
};
static int fold_launch(const std::vector<FoldTarget>& targets, hipStream_t st) {
    size_t i = 0;
    while (i < targets.size()) {
        FoldBatch b;
        b.n = 0;
        int blocks = 0, nsrc = 0;
        while (i < targets.size() && b.n < GS_FOLD_HEADS && nsrc + (int)targets[i].src.size() <= GS_FOLD_SRCS) {
            const FoldTarget& t = targets[i];
            FoldHead& h = b.h[b.n++];
            h.out = t.out; h.c = t.c; h.accumulate = t.accumulate; h.blk0 = blocks; h.chblocks = cdiv(t.c, 32); h.src0 = nsrc; h.nsrc = (int)t.src.size();
            for (const FoldSrc& q : t.src) b.s[nsrc++] = q;
            blocks += h.chblocks * cdiv(t.src[0].nparts, t.src[0].per);   // (several slabs: single-source targets only)
            ++i;
        }
        if (b.n == 0) return fail(GS_ERR_UNSUPPORTED, "channel_fold_batch: %zu folds into one target (at most %d)", targets[i].src.size(), GS_FOLD_SRCS);
        hipLaunchKernelGGL(channel_fold_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, st, b);
    }
    GS_CHECK_LAUNCH();
    return 0;
}
extern "C" size_t gs_channel_fold_batch_workspace_bytes(const GsFoldJob* jobs, int njobs) {
    size_t rows = 0;
    for (int i = 0; jobs && i < njobs; ++i)
        if (jobs[i].nparts > CS_SLAB) rows += (size_t)cdiv(jobs[i].nparts, CS_SLAB) * jobs[i].c;
    return rows * sizeof(float);
}
extern "C" int gs_channel_fold_batch(const GsFoldJob* jobs, int njobs, void* ws, size_t ws_bytes, void* stream) {
    GS_CHECK_ARG(jobs && njobs > 0, "channel_fold_batch: no jobs");
    if (ws_bytes < gs_channel_fold_batch_workspace_bytes(jobs, njobs)) return fail(GS_ERR_WORKSPACE, "channel_fold_batch: workspace too small");
    std::vector<FoldTarget> first, second;
    float* part2 = (float*)ws;
    for (int i = 0; i < njobs; ++i) {
        const GsFoldJob& j = jobs[i];
        GS_CHECK_ARG(j.part && j.out && j.nparts > 0 && j.c > 0, "channel_fold_batch: bad job %d", i);
        FoldSrc src{j.part, j.nparts, j.nparts};
        if (j.nparts > CS_SLAB) {   // slab sums first (their own rows of ws), then the fold of those
            const int nsplit = cdiv(j.nparts, CS_SLAB);
            first.push_back(FoldTarget{part2, j.c, 0, {FoldSrc{j.part, j.nparts, CS_SLAB}}});
            src = FoldSrc{part2, nsplit, nsplit};
            part2 += (size_t)nsplit * j.c;
        }
        size_t t = 0;
        while (t < second.size() && second[t].out != j.out) ++t;   // jobs into one target: folded in call order by the same blocks
        if (t == second.size()) second.push_back(FoldTarget{j.out, j.c, j.accumulate ? 1 : 0, {}});
        GS_CHECK_ARG(second[t].c == j.c, "channel_fold_batch: job %d adds %d channels into a target of %d", i, j.c, second[t].c);
        GS_CHECK_ARG(second[t].src.empty() || j.accumulate, "channel_fold_batch: job %d overwrites a target an earlier job wrote", i);
        second[t].src.push_back(src);
    }
    hipStream_t st = as_stream(stream);
    if (!first.empty())
        if (int e = fold_launch(first, st)) return e;
    return fold_launch(second, st);
}
// rows of partials a producer leaves with GS_SUM_PARTIALS (0: it writes the sums itself, nothing to fold)
extern "C" int gs_bias_partial_rows(int producer, int64_t p, int c, int dtype);

// few rows, many channels (dense-layer biases: [batch][8192]): a thread per channel, no partials
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_rows_kernel(const T* __restrict__ g, float* __restrict__ out, int p, int c, int accumulate) {
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= c) return;
    float s = 0.f;
    for (int r = 0; r < p; ++r) s += DT<T>::ld(g + (long)r * c + ch);
    out[ch] = accumulate ? out[ch] + s : s;
}

static int channel_sum_parts(long p, int c) {
    long rows_per_block = 256 / (c >= 4 ? ((c & 3) == 0 ? c / 4 : (c < 256 ? c : 256)) : c);
    if (rows_per_block < 1) rows_per_block = 1;
    long nb = (p + rows_per_block * 8 - 1) / (rows_per_block * 8);
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    return (int)nb;
}

// ------------------------------------------------------------------------- pixel norm
// Row p of C channels is owned by L = min(64, C/E) lanes, E = 16 bytes of channels per lane per pass (4 fp32 / 8 bf16; E = 4
// for narrower rows), P passes per row (1 up to 256 fp32 / 512 bf16 channels), U independent row groups per loop trip so that
// a wave keeps 2U..3U 1-KiB loads in flight -- the kernel is a pure HBM stream.
// MODE 0: y = x*r ; MODE 1: gx = r*(g - y*mean(y*g)) ; MODE 2: second-order term (see header).
template <typename T, int E> __device__ inline void pn_ld(const T* p, float* o) {
    if constexpr (E == Wide<T>::N) ld_wide<T>(p, o);
    else ld4(p, *reinterpret_cast<float(*)[4]>(o));
}
template <typename T, int E> __device__ inline void pn_st(T* p, const float* o) {
    if constexpr (E == Wide<T>::N) st_wide<T>(p, o);
    else st4(p, *reinterpret_cast<const float(*)[4]>(o));
}
// BS (MODE 1 only): also the per-channel sums of the written gradient as one partial row per block, bsum[block][c] -- the bias gradient
// of the conv block that produced x (z = act(conv + bias)), folded afterwards by channel_sum_finalize: no second pass over the gradient.
template <typename T, int MODE, int E, int P, int U, bool BS = false>
__global__ __launch_bounds__(256) void pixel_norm_kernel(const T* __restrict__ a0, const T* __restrict__ a1, const T* __restrict__ a2,
                                                         T* __restrict__ out, long p, int c, float eps, int act, int pre,
                                                         const T* __restrict__ addend, T* __restrict__ out2, float* __restrict__ bsum = nullptr) {
    // x is itself the output of an activation in the generator blocks (conv -> act -> norm), and the passes around this
    // kernel fold into it:
    //   act    (MODE 1): the result is multiplied by act'(.) through x  -> gradient w.r.t. the PRE-activation
    //   addend (MODE 1): added to the norm's gradient before that       -> a second gradient into x (second-order terms)
    //   pre    (MODE 1, 2): the incoming g / gg is first multiplied by act'(.) through x (transpose of the `act` form)
    //   out2   (MODE 2): also write pixel_norm_bwd(gg', x) -- the two gradients of a differentiated norm-backward in one pass
    // a0 = x (MODE 0) | g (MODE 1) | gg (MODE 2);  a1 = x (MODE 1) | g (MODE 2);  a2 = x (MODE 2)
    const int vecs = c / E;
    const int L = vecs < 64 ? vecs : 64;         // lanes per row (power of two)
    const int rows_per_wave = 64 / L;
    const int lane = threadIdx.x & 63;
    const int sub = lane % L;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const float invc = 1.f / (float)c;
    const T* const xbase = MODE == 0 ? a0 : (MODE == 1 ? a1 : a2);
    float bs[BS ? P : 1][BS ? E : 1];   // this lane's channels, summed over its rows
    if (BS) {
#pragma unroll
        for (int k = 0; k < P; ++k)
#pragma unroll
            for (int e = 0; e < E; ++e) bs[k][e] = 0.f;
    }
    auto dact = [&](int kind, float xv) __attribute__((always_inline)) {   // act'(.) through the activation output
        return kind == GS_ACT_LRELU ? (xv > 0.f ? 1.f : 0.2f) : (kind == GS_ACT_TANH ? 1.f - xv * xv : 1.f);
    };
    for (long row0 = wave * rows_per_wave; row0 < p; row0 += nwaves * rows_per_wave * U) {
        long off[U];
        bool ok[U];
        float xv[U][P][E], av[U][P][E], bv[U][P][E];   // x ; g (MODE 1) / gg (MODE 2) ; g (MODE 2)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long row = row0 + u * nwaves * rows_per_wave + lane / L;
            ok[u] = row < p;
            off[u] = (ok[u] ? row : 0) * c + sub * E;     // clamped: the loads stay unconditional
#pragma unroll
            for (int k = 0; k < P; ++k) {
                pn_ld<T, E>(xbase + off[u] + k * L * E, xv[u][k]);
                if (MODE >= 1) pn_ld<T, E>(a0 + off[u] + k * L * E, av[u][k]);
                if (MODE == 2) pn_ld<T, E>(a1 + off[u] + k * L * E, bv[u][k]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < P; ++k)
#pragma unroll
                for (int e = 0; e < E; ++e) s += xv[u][k][e] * xv[u][k][e];
            s = group_sum(s, L);
            const float r = rsqrtf(s * invc + eps);
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    float o[E];
#pragma unroll
                    for (int e = 0; e < E; ++e) o[e] = xv[u][k][e] * r;
                    if (ok[u]) pn_st<T, E>(out + off[u] + k * L * E, o);
                }
            } else if (MODE == 1) {
                float q = 0.f;
#pragma unroll
                for (int k = 0; k < P; ++k)
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        av[u][k][e] *= dact(pre, xv[u][k][e]);
                        q += xv[u][k][e] * r * av[u][k][e];
                    }
                q = group_sum(q, L);
                q *= invc;
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    float o[E], ad[E];
#pragma unroll
                    for (int e = 0; e < E; ++e) ad[e] = 0.f;
                    if (addend) pn_ld<T, E>(addend + off[u] + k * L * E, ad);
#pragma unroll
                    for (int e = 0; e < E; ++e) o[e] = (r * (av[u][k][e] - xv[u][k][e] * r * q) + ad[e]) * dact(act, xv[u][k][e]);
                    if (ok[u]) pn_st<T, E>(out + off[u] + k * L * E, o);
                    if (BS && ok[u]) {
#pragma unroll
                        for (int e = 0; e < E; ++e) bs[k][e] += o[e];
                    }
                }
            } else {
                float sa = 0.f, sp = 0.f, sq = 0.f;
#pragma unroll
                for (int k = 0; k < P; ++k)
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        av[u][k][e] *= dact(pre, xv[u][k][e]);
                        const float yv = xv[u][k][e] * r;
                        sa += av[u][k][e] * bv[u][k][e];
                        sp += yv * av[u][k][e];
                        sq += yv * bv[u][k][e];
                    }
                sa = group_sum(sa, L);
                sp = group_sum(sp, L);
                sq = group_sum(sq, L);
                const float k0 = r * r * invc;
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    float o[E];
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const float yv = xv[u][k][e] * r;
                        o[e] = k0 * (-sa * yv - sq * av[u][k][e] - sp * bv[u][k][e] + 3.f * sp * sq * yv * invc);
                    }
                    if (ok[u]) pn_st<T, E>(out + off[u] + k * L * E, o);
                    if (out2) {   // r * (gg' - y * mean(y * gg')), sp = sum(y * gg')
                        float o2[E];
#pragma unroll
                        for (int e = 0; e < E; ++e) o2[e] = r * (av[u][k][e] - xv[u][k][e] * r * (sp * invc));
                        if (ok[u]) pn_st<T, E>(out2 + off[u] + k * L * E, o2);
                    }
                }
            }
        }
    }
    if constexpr (BS) {
        __shared__ float bred[4][P * 64 * E];   // [wave][channel]
#pragma unroll
        for (int k = 0; k < P; ++k)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                bs[k][e] = residue_sum(bs[k][e], L);   // the lanes that own the same channels (other rows)
            }
        const int wv = threadIdx.x >> 6;
        if (lane < L) {
#pragma unroll
            for (int k = 0; k < P; ++k)
#pragma unroll
                for (int e = 0; e < E; ++e) bred[wv][k * L * E + sub * E + e] = bs[k][e];
        }
        __syncthreads();
        for (int ch = threadIdx.x; ch < c; ch += 256) bsum[(long)blockIdx.x * c + ch] = bred[0][ch] + bred[1][ch] + bred[2][ch] + bred[3][ch];
    }
}

// ----------------------------------------------------------------- upscale / block sum
template <typename T>
__global__ void upscale_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c, int fy, int fx, float scale) {
    const long total = (long)n * h * fy * w * fx * c;
    const int wo = w * fx, ho = h * fy;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = i % c;
        long r = i / c;
        const int ox = r % wo;
        r /= wo;
        const int oy = r % ho;
        const int b = r / ho;
        const float v = DT<T>::ld(x + (((long)b * h + oy / fy) * w + ox / fx) * c + ch);
        DT<T>::st(y + i, scale == 1.f ? v : v * scale);
    }
}
// one wave per output element group: y[b][oy][ox][ch] = scale * sum_{fy,fx} x
template <typename T>
__global__ __launch_bounds__(256) void blocksum_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c, int fy, int fx, float scale) {
    const int ho = h / fy, wo = w / fx;
    const long total = (long)n * ho * wo * c;
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const int blk = fy * fx;
    for (long o = wave; o < total; o += nwaves) {
        const int ch = o % c;
        long r = o / c;
        const int ox = r % wo;
        r /= wo;
        const int oy = r % ho;
        const int b = r / ho;
        float s = 0.f;
        for (int k = lane; k < blk; k += 64) {
            const int dy = k / fx, dx = k % fx;
            s += DT<T>::ld(x + (((long)b * h + oy * fy + dy) * w + ox * fx + dx) * c + ch);
        }
        s = wave_sum(s);
        if (lane == 0) DT<T>::st(y + o, s * scale);
    }
}
// small blocks: one thread per output element
template <typename T>
__global__ void blocksum_small_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c, int fy, int fx, float scale) {
    const int ho = h / fy, wo = w / fx;
    const long total = (long)n * ho * wo * c;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const int ch = o % c;
        long r = o / c;
        const int ox = r % wo;
        r /= wo;
        const int oy = r % ho;
        const int b = r / ho;
        float s = 0.f;
        for (int dy = 0; dy < fy; ++dy)
            for (int dx = 0; dx < fx; ++dx) s += DT<T>::ld(x + (((long)b * h + oy * fy + dy) * w + ox * fx + dx) * c + ch);
        DT<T>::st(y + o, s * scale);
    }
}

// --------------------------------------------------------------------- row reductions
// out[r] = sum_j x[r][j]^2.  A row is split over SSQ_CHUNKS blocks (8 rows of 262144 elements would otherwise keep 8 CUs busy);
// pass 1 writes one partial per (row, chunk), pass 2 (one wave per row) sums them in a fixed order: deterministic.
constexpr int SSQ_CHUNKS = 64;
template <typename T>
__global__ __launch_bounds__(256) void sumsq_rows_kernel(const T* __restrict__ x, float* __restrict__ part, long cols) {
    __shared__ float red[4];
    const int r = blockIdx.y, ck = blockIdx.x;
    const T* xr = x + (long)r * cols;
    const long nvec = cols >> 2;
    const long per = (nvec + SSQ_CHUNKS - 1) / SSQ_CHUNKS;
    const long i0 = ck * per, i1 = i0 + per < nvec ? i0 + per : nvec;
    float s = 0.f;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        float v[4];
        ld4(xr + i * 4, v);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (ck == SSQ_CHUNKS - 1)
        for (long i = nvec * 4 + threadIdx.x; i < cols; i += 256) { const float v = DT<T>::ld(xr + i); s += v * v; }
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) part[(long)r * SSQ_CHUNKS + ck] = s;
}
static __global__ __launch_bounds__(64) void sumsq_rows_final_kernel(const float* __restrict__ part, float* __restrict__ out) {
    const float s = wave_sum(part[(long)blockIdx.x * SSQ_CHUNKS + threadIdx.x]);   // SSQ_CHUNKS == 64 lanes
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
template <typename T>
__global__ void row_scale_kernel(const T* __restrict__ x, const float* __restrict__ s, T* __restrict__ out, int rows, long cols, float alpha) {
    const long nvec = ((long)rows * cols) >> 2;  // cols % 4 == 0 checked by the caller
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const float f = alpha * s[(i * 4) / cols];
        float v[4];
        ld4(x + i * 4, v);
        v[0] *= f; v[1] *= f; v[2] *= f; v[3] *= f;
        st4(out + i * 4, v);
    }
}

// -------------------------------------------------------------------------------- Adam
// ZERO: the gradient is cleared behind the update (the trainer's next run accumulates into it from zero: no separate fill pass)
// DEV: the bias-corrected step size is read from device memory (`lr_dev[0]`; a NEGATIVE value means "no step": the launch leaves every
// buffer untouched) -- the by-value scalar would be frozen into a captured hipGraph, and with the optimizer step inside the iteration's
// graph there is no eager launch left between two runs.
template <bool ZERO, bool DEV>
static __global__ void adam_tf_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                      long n, float lr_t, const float* __restrict__ lr_dev, float b1, float b2, float eps, float gs) {
    if (DEV) {
        lr_t = __builtin_nontemporal_load(lr_dev);
        if (lr_t < 0.f) return;
    }
    const long nvec = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv0 = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* pp = &pv.x; float* mm = &mv.x; float* vp = &vv.x; const float* gg = &gv0.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gg[e] * gs;
            mm[e] = b1 * mm[e] + (1.f - b1) * gr;
            vp[e] = b2 * vp[e] + (1.f - b2) * gr * gr;
            pp[e] -= lr_t * mm[e] / (sqrtf(vp[e]) + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (ZERO) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const long t = nvec * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        const float gr = g[t] * gs;
        m[t] = b1 * m[t] + (1.f - b1) * gr;
        v[t] = b2 * v[t] + (1.f - b2) * gr * gr;
        p[t] -= lr_t * m[t] / (sqrtf(v[t]) + eps);
        if (ZERO) g[t] = 0.f;
    }
}

}  // namespace gs

using namespace gs;

extern "C" int gs_bias_act_fwd(const void* x, const float* bias, void* y, int64_t p, int c, int act, int dtype, void* stream) {
    GS_CHECK_ARG(p > 0 && c > 0 && act >= 0 && act <= 2, "bias_act: bad args");
    hipStream_t st = as_stream(stream);
    const long n = (long)p * c;
    if ((c & 3) == 0) {
        const long nvec = n >> 2;
        dim3 grid(ew_grid(nvec));
#define GS_BA(ACT, B) hipLaunchKernelGGL((bias_act_kernel<T, ACT, B>), grid, dim3(256), 0, st, (const T*)x, bias, (T*)y, nvec, c >> 2)
        GS_DISPATCH_DTYPE(dtype, {
            if (bias) { if (act == 0) GS_BA(0, true); else if (act == 1) GS_BA(1, true); else GS_BA(2, true); }
            else { if (act == 0) GS_BA(0, false); else if (act == 1) GS_BA(1, false); else GS_BA(2, false); }
        });
#undef GS_BA
    } else {
        GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bias_act_scalar_kernel<T>), dim3(ew_grid(n)), dim3(256), 0, st, (const T*)x, bias, (T*)y, n, c, act));
    }
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_act_bwd(const void* g, const void* y, void* gx, int64_t numel, int act, int dtype, void* stream) {
    GS_CHECK_ARG(numel > 0 && (act == 1 || act == 2), "act_bwd: bad args");
    hipStream_t st = as_stream(stream);
    dim3 grid(ew_grid((numel >> 2) + 4));
    GS_DISPATCH_DTYPE(dtype, {
        if (act == 1) hipLaunchKernelGGL((act_bwd_kernel<T, 1>), grid, dim3(256), 0, st, (const T*)g, (const T*)y, (T*)gx, (long)numel);
        else hipLaunchKernelGGL((act_bwd_kernel<T, 2>), grid, dim3(256), 0, st, (const T*)g, (const T*)y, (T*)gx, (long)numel);
    });
    GS_CHECK_LAUNCH();
    return 0;
}

// The generator's first block (networks.py:41-56): dense -> reshape to [n, c, h, w] -> leaky_relu -> pixel_normalization.  The dense layer's
// units are channel-major (unit u = ch * hw + p) while every activation of this library is channels-last ([n][p][ch]): the change of order
// rides in the bias / activation pass instead of a copy of its own.
//   units_to_nhwc:  z[n][p][ch] = act(y[n][u] + bias[u])            (mask == NULL: the forward)
//                   z[n][p][ch] = y[n][u] * act'(mask[n][p][ch])    (mask given:  the backward of nhwc_to_units, second-order pass)
//   nhwc_to_units:  gu[n][u] = g[n][p][ch] * act'(z[n][p][ch])      (the backward of the forward, in the units' order for the dense kernels)
template <typename T>
__global__ void units_to_nhwc_kernel(const T* __restrict__ y, const float* __restrict__ bias, const T* __restrict__ mask, T* __restrict__ z, long total, int c, int hw, int act) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // channels-last index: coalesced stores
    if (i >= total) return;
    const int ch = (int)(i % c);
    const long r = i / c;
    const int p = (int)(r % hw);
    const long n = r / hw;
    const long u = (long)ch * hw + p;
    float v = DT<T>::ld(y + n * c * hw + u) + (bias ? bias[u] : 0.f);
    if (mask) {
        const float m = DT<T>::ld(mask + i);
        v *= act == GS_ACT_LRELU ? (m > 0.f ? 1.f : 0.2f) : (act == GS_ACT_TANH ? 1.f - m * m : 1.f);
    } else {
        v = act == GS_ACT_LRELU ? fmaxf(v, 0.2f * v) : (act == GS_ACT_TANH ? tanhf(v) : v);
    }
    DT<T>::st(z + i, v);
}
template <typename T>
__global__ void nhwc_to_units_kernel(const T* __restrict__ g, const T* __restrict__ zm, T* __restrict__ gu, long total, int c, int hw, int act) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // units index: coalesced stores
    if (i >= total) return;
    const long n = i / ((long)c * hw);
    const int u = (int)(i - n * c * hw);
    const int ch = u / hw, p = u - ch * hw;
    const long src = (n * hw + p) * c + ch;
    const float m = DT<T>::ld(zm + src);
    const float f = act == GS_ACT_LRELU ? (m > 0.f ? 1.f : 0.2f) : (act == GS_ACT_TANH ? 1.f - m * m : 1.f);
    DT<T>::st(gu + i, DT<T>::ld(g + src) * f);
}
extern "C" int gs_units_bias_act_to_nhwc(const void* y, const float* bias, const void* mask, void* z, int n, int c, int hw, int act, int dtype, void* stream) {
    GS_CHECK_ARG(n > 0 && c > 0 && hw > 0 && y && z && act >= 0 && act <= 2, "units_bias_act_to_nhwc: bad args");
    const long total = (long)n * c * hw;
    GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((units_to_nhwc_kernel<T>), dim3((unsigned)cdiv(total, 256)), dim3(256), 0, as_stream(stream), (const T*)y, bias,
                                                (const T*)mask, (T*)z, total, c, hw, act));
    GS_CHECK_LAUNCH();
    return 0;
}
extern "C" int gs_nhwc_act_bwd_to_units(const void* g, const void* z, void* gu, int n, int c, int hw, int act, int dtype, void* stream) {
    GS_CHECK_ARG(n > 0 && c > 0 && hw > 0 && g && z && gu && act >= 0 && act <= 2, "nhwc_act_bwd_to_units: bad args");
    const long total = (long)n * c * hw;
    GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((nhwc_to_units_kernel<T>), dim3((unsigned)cdiv(total, 256)), dim3(256), 0, as_stream(stream), (const T*)g, (const T*)z,
                                                (T*)gu, total, c, hw, act));
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_tanh_bwd_bwd(const void* gg, const void* g, const void* y, void* out, int64_t numel, int dtype, void* stream) {
    GS_CHECK_ARG(numel > 0, "tanh_bwd_bwd: bad args");
    GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((tanh_bwd_bwd_kernel<T>), dim3(ew_grid(numel)), dim3(256), 0, as_stream(stream),
                                                (const T*)gg, (const T*)g, (const T*)y, (T*)out, (long)numel));
    GS_CHECK_LAUNCH();
    return 0;
}

// out = coef[ia] * a + coef[ib] * b with the coefficients read from device memory: the fade-in weight of the progressive schedule
// changes every step, and a by-value scalar would be frozen into a captured hipGraph
template <typename T>
__global__ void axpby_dev_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, long n, const float* __restrict__ coef, int ia, int ib) {
    const float ca = coef[ia], cb = coef[ib];
    const long nvec = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        float av[4], bv[4];
        ld4(a + i * 4, av);
        ld4(b + i * 4, bv);
#pragma unroll
        for (int e = 0; e < 4; ++e) av[e] = ca * av[e] + cb * bv[e];
        st4(out + i * 4, av);
    }
    const long t = nvec * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) DT<T>::st(out + t, ca * DT<T>::ld(a + t) + cb * DT<T>::ld(b + t));
}

extern "C" int gs_axpby_dev(const void* a, const void* b, void* out, int64_t numel, const float* coef, int ia, int ib, int dtype, void* stream) {
    GS_CHECK_ARG(numel > 0 && coef && ia >= 0 && ib >= 0, "axpby_dev: bad args");
    GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((axpby_dev_kernel<T>), dim3(ew_grid((numel >> 2) + 4)), dim3(256), 0, as_stream(stream),
                                                (const T*)a, (const T*)b, (T*)out, (long)numel, coef, ia, ib));
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_axpby(const void* a, const void* b, void* out, int64_t numel, float ca, float cb, int dtype, void* stream) {
    GS_CHECK_ARG(numel > 0, "axpby: bad args");
    GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((axpby_kernel<T>), dim3(ew_grid((numel >> 2) + 4)), dim3(256), 0, as_stream(stream),
                                                (const T*)a, (const T*)b, (T*)out, (long)numel, ca, cb));
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t gs_channel_sum_workspace_bytes(int64_t p, int c) {
    const int np = channel_sum_parts(p, c);
    return (size_t)(np + cdiv(np, CS_SLAB)) * c * sizeof(float);
}

extern "C" int gs_channel_sum(const void* g, float* out, int64_t p, int c, int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream) {
    GS_CHECK_ARG(p > 0 && c > 0, "channel_sum: bad args");
    const bool partials_only = (accumulate & GS_SUM_PARTIALS) != 0;
    accumulate &= 1;
    if (p <= 64 && c >= 256) {
        GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((channel_sum_rows_kernel<T>), dim3(cdiv(c, 256)), dim3(256), 0, as_stream(stream), (const T*)g, out, (int)p, c, accumulate));
        GS_CHECK_LAUNCH();
        return 0;
    }
    const int nparts = channel_sum_parts(p, c);
    if (ws_bytes < (partials_only ? (size_t)nparts * c * sizeof(float) : gs_channel_sum_workspace_bytes(p, c))) return fail(GS_ERR_WORKSPACE, "channel_sum: workspace too small");
    hipStream_t st = as_stream(stream);
    float* part = (float*)ws;
    GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((channel_sum_kernel<T>), dim3(nparts), dim3(256), 0, st, (const T*)g, part, (long)p, c));
    GS_CHECK_LAUNCH();
    if (partials_only) return 0;
    return channel_sum_finalize(part, out, nparts, c, accumulate, st);
}

extern "C" int gs_act_bwd(const void* g, const void* y, void* gx, int64_t numel, int act, int dtype, void* stream);
extern "C" size_t gs_channel_sum_workspace_bytes(int64_t p, int c);

extern "C" int gs_act_bwd_bias(const void* g, const void* y, void* gx, float* gb, int64_t p, int c, int act, int accumulate, int dtype,
                               void* ws, size_t ws_bytes, void* stream) {
    GS_CHECK_ARG(p > 0 && c > 0 && (act == 1 || act == 2), "act_bwd_bias: bad args");
    if ((c & 3) != 0 || c > 1024 || 256 % (c >> 2) != 0) {  // generic shapes: two passes
        if (int e = gs_act_bwd(g, y, gx, p * c, act, dtype, stream)) return e;
        return gs_channel_sum(gx, gb, p, c, accumulate, dtype, ws, ws_bytes, stream);
    }
    const bool partials_only = (accumulate & GS_SUM_PARTIALS) != 0;
    accumulate &= 1;
    const int nparts = channel_sum_parts(p, c);
    if (ws_bytes < (partials_only ? (size_t)nparts * c * sizeof(float) : gs_channel_sum_workspace_bytes(p, c))) return fail(GS_ERR_WORKSPACE, "act_bwd_bias: workspace too small");
    hipStream_t st = as_stream(stream);
    float* part = (float*)ws;
    GS_DISPATCH_DTYPE(dtype, {
        if (act == 1) hipLaunchKernelGGL((act_bwd_sum_kernel<T, 1>), dim3(nparts), dim3(256), 0, st, (const T*)g, (const T*)y, (T*)gx, part, (long)p, c);
        else hipLaunchKernelGGL((act_bwd_sum_kernel<T, 2>), dim3(nparts), dim3(256), 0, st, (const T*)g, (const T*)y, (T*)gx, part, (long)p, c);
    });
    GS_CHECK_LAUNCH();
    if (partials_only) return 0;
    return channel_sum_finalize(part, gb, nparts, c, accumulate, st);
}

static int pixel_norm_grid(long p, int c, int wn, bool bias_sums = false) {
    const int E = c % wn == 0 ? wn : 4;
    const int vecs = c / E, L = vecs < 64 ? vecs : 64, P = vecs / L;
    const int U = P == 1 ? 2 : 1;
    const long rows_per_block = 4L * (64 / L) * U;
    int g = ew_grid((p + rows_per_block - 1) / rows_per_block * 256);
    if (bias_sums) {   // every block ends with a cross-lane + LDS fold of its channel sums and one partial row: as many trips per block as
        static const int cap = getenv("GS_PN_BIAS_GRID_CAP") ? atoi(getenv("GS_PN_BIAS_GRID_CAP")) : 1024;   // the plain pass has blocks
        if (g > cap) g = cap;   // (top level, 32 channels: 61.5 us at 8192 blocks, 50.5 at 1024 -- 42.3 without the sums; scripts/bench_ew.py)
    }
    return g;
}
template <typename T, int MODE, bool BS = false>
static void pixel_norm_launch_t(const void* a0, const void* a1, const void* a2, void* out, long p, int c, float eps, int act, int pre, const void* addend,
                                void* out2, hipStream_t st, float* bsum = nullptr) {
    constexpr int WN = Wide<T>::N;
    const int E = c % WN == 0 ? WN : 4;
    const int vecs = c / E, L = vecs < 64 ? vecs : 64, P = vecs / L;   // P in {1, 2, 4}
    dim3 grid(pixel_norm_grid(p, c, WN, BS));
#define GS_PN(EE, PP, UU)                                                                                                          \
    hipLaunchKernelGGL((pixel_norm_kernel<T, MODE, EE, PP, UU, BS>), grid, dim3(256), 0, st, (const T*)a0, (const T*)a1, (const T*)a2, (T*)out, p, c, eps, \
                       act, pre, (const T*)addend, (T*)out2, bsum)
    if (E == WN) {
        if (P == 1) GS_PN(WN, 1, 2); else if (P == 2) GS_PN(WN, 2, 1); else GS_PN(WN, 4, 1);
    } else {
        if (P == 1) GS_PN(4, 1, 2); else if (P == 2) GS_PN(4, 2, 1); else GS_PN(4, 4, 1);
    }
#undef GS_PN
}
static int pixel_norm_launch(int mode, const void* a0, const void* a1, const void* a2, void* out, int64_t p, int c, float eps, int dtype, void* stream, int act = 0,
                             int pre = 0, const void* addend = nullptr, void* out2 = nullptr) {
    GS_CHECK_ARG(p > 0 && c >= 4 && c <= 1024 && (c & (c - 1)) == 0, "pixel_norm: c=%d must be a power of two in [4,1024]", c);
    hipStream_t st = as_stream(stream);
    GS_DISPATCH_DTYPE(dtype, {
        if (mode == 0) pixel_norm_launch_t<T, 0>(a0, a1, a2, out, (long)p, c, eps, act, pre, addend, out2, st);
        else if (mode == 1) pixel_norm_launch_t<T, 1>(a0, a1, a2, out, (long)p, c, eps, act, pre, addend, out2, st);
        else pixel_norm_launch_t<T, 2>(a0, a1, a2, out, (long)p, c, eps, act, pre, addend, out2, st);
    });
    GS_CHECK_LAUNCH();
    return 0;
}
extern "C" int gs_pixel_norm_fwd(const void* x, void* y, int64_t p, int c, float eps, int dtype, void* stream) {
    return pixel_norm_launch(0, x, nullptr, nullptr, y, p, c, eps, dtype, stream);
}
extern "C" int gs_pixel_norm_bwd(const void* g, const void* x, void* gx, int64_t p, int c, float eps, int dtype, void* stream) {
    return pixel_norm_launch(1, g, x, nullptr, gx, p, c, eps, dtype, stream);
}
static bool pn_act_ok(int a) { return a == GS_ACT_NONE || a == GS_ACT_LRELU || a == GS_ACT_TANH; }
extern "C" int gs_pixel_norm_bwd_fused(const void* g, const void* x, const void* addend, void* gx, int64_t p, int c, float eps, int pre_act, int post_act,
                                       int dtype, void* stream) {
    GS_CHECK_ARG(pn_act_ok(pre_act) && pn_act_ok(post_act), "pixel_norm_bwd_fused: bad activation %d / %d", pre_act, post_act);
    return pixel_norm_launch(1, g, x, nullptr, gx, p, c, eps, dtype, stream, post_act, pre_act, addend);
}
// ... and the per-channel sums of gx on the side (gb (+)= sum over pixels: the bias gradient of the conv block that produced x)
extern "C" size_t gs_pixel_norm_bwd_bias_workspace_bytes(int64_t p, int c, int dtype) {
    const int nb = pixel_norm_grid((long)p, c, dtype == GS_F32 ? 4 : 8, true);
    return (size_t)(nb + cdiv(nb, CS_SLAB)) * c * sizeof(float);
}
extern "C" int gs_pixel_norm_bwd_fused_bias(const void* g, const void* x, const void* addend, void* gx, float* gb, int64_t p, int c, float eps, int pre_act,
                                            int post_act, int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream) {
    GS_CHECK_ARG(pn_act_ok(pre_act) && pn_act_ok(post_act), "pixel_norm_bwd_fused_bias: bad activation %d / %d", pre_act, post_act);
    GS_CHECK_ARG(p > 0 && c >= 4 && c <= 1024 && (c & (c - 1)) == 0 && gb && g && x && gx, "pixel_norm_bwd_fused_bias: bad args (c=%d)", c);
    const bool partials_only = (accumulate & GS_SUM_PARTIALS) != 0;
    accumulate &= 1;
    const int nb = pixel_norm_grid((long)p, c, dtype == GS_F32 ? 4 : 8, true);
    if (ws_bytes < (partials_only ? (size_t)nb * c * sizeof(float) : gs_pixel_norm_bwd_bias_workspace_bytes(p, c, dtype)))
        return fail(GS_ERR_WORKSPACE, "pixel_norm_bwd_fused_bias: workspace too small");
    hipStream_t st = as_stream(stream);
    float* part = (float*)ws;
    GS_DISPATCH_DTYPE(dtype, (pixel_norm_launch_t<T, 1, true>(g, x, nullptr, gx, (long)p, c, eps, post_act, pre_act, addend, nullptr, st, part)));
    GS_CHECK_LAUNCH();
    if (partials_only) return 0;
    return channel_sum_finalize(part, gb, nb, c, accumulate, st);
}
extern "C" int gs_bias_partial_rows(int producer, int64_t p, int c, int dtype) {
    if (p <= 0 || c <= 0) return 0;
    const bool act_fast = !((c & 3) != 0 || c > 1024 || 256 % (c >> 2) != 0);   // gs_act_bwd_bias's one-pass shapes
    switch (producer) {
        case GS_BIAS_FROM_CHANNEL_SUM: return (p <= 64 && c >= 256) ? 0 : channel_sum_parts(p, c);
        case GS_BIAS_FROM_ACT_BWD: return (!act_fast && p <= 64 && c >= 256) ? 0 : channel_sum_parts(p, c);
        case GS_BIAS_FROM_PIXEL_NORM_BWD: return (c >= 4 && c <= 1024 && (c & (c - 1)) == 0) ? pixel_norm_grid((long)p, c, dtype == GS_F32 ? 4 : 8, true) : 0;
        default: return 0;
    }
}
extern "C" int gs_pixel_norm_bwd_bwd_fused(const void* gg, const void* g, const void* x, void* out, void* out_g, int64_t p, int c, float eps, int pre_act,
                                           int dtype, void* stream) {
    GS_CHECK_ARG(pn_act_ok(pre_act), "pixel_norm_bwd_bwd_fused: bad activation %d", pre_act);
    return pixel_norm_launch(2, gg, g, x, out, p, c, eps, dtype, stream, 0, pre_act, nullptr, out_g);
}
extern "C" int gs_pixel_norm_bwd_bwd(const void* gg, const void* g, const void* x, void* out, int64_t p, int c, float eps, int dtype, void* stream) {
    return pixel_norm_launch(2, gg, g, x, out, p, c, eps, dtype, stream);
}

extern "C" int gs_upscale2d(const void* x, void* y, int n, int h, int w, int c, int fy, int fx, float scale, int dtype, void* stream) {
    GS_CHECK_ARG(n > 0 && h > 0 && w > 0 && c > 0 && fy > 0 && fx > 0, "upscale2d: bad args");
    const long total = (long)n * h * fy * w * fx * c;
    GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((upscale_kernel<T>), dim3(ew_grid(total)), dim3(256), 0, as_stream(stream),
                                                (const T*)x, (T*)y, n, h, w, c, fy, fx, scale));
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_blocksum2d(const void* x, void* y, int n, int h, int w, int c, int fy, int fx, float scale, int dtype, void* stream) {
    GS_CHECK_ARG(n > 0 && c > 0 && fy > 0 && fx > 0 && h % fy == 0 && w % fx == 0, "blocksum2d: bad args");
    const long total = (long)n * (h / fy) * (w / fx) * c;
    hipStream_t st = as_stream(stream);
    if (fy * fx >= 32) {
        GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((blocksum_kernel<T>), dim3(ew_grid(total * 64)), dim3(256), 0, st, (const T*)x, (T*)y, n, h, w, c, fy, fx, scale));
    } else {
        GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((blocksum_small_kernel<T>), dim3(ew_grid(total)), dim3(256), 0, st, (const T*)x, (T*)y, n, h, w, c, fy, fx, scale));
    }
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t gs_sumsq_rows_workspace_bytes(int rows) { return (size_t)(rows > 0 ? rows : 0) * SSQ_CHUNKS * sizeof(float); }

extern "C" int gs_sumsq_rows(const void* x, float* out, int rows, int64_t cols, int dtype, void* ws, size_t ws_bytes, void* stream) {
    GS_CHECK_ARG(rows > 0 && cols > 0, "sumsq_rows: bad args");
    if (ws_bytes < gs_sumsq_rows_workspace_bytes(rows)) return fail(GS_ERR_WORKSPACE, "sumsq_rows: workspace too small");
    float* part = (float*)ws;
    GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((sumsq_rows_kernel<T>), dim3(SSQ_CHUNKS, rows), dim3(256), 0, as_stream(stream), (const T*)x, part, (long)cols));
    GS_CHECK_LAUNCH();
    hipLaunchKernelGGL(sumsq_rows_final_kernel, dim3(rows), dim3(64), 0, as_stream(stream), part, out);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_row_scale(const void* x, const float* s, float alpha, void* out, int rows, int64_t cols, int dtype, void* stream) {
    GS_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0, "row_scale: cols must be a multiple of 4");
    const long nvec = ((long)rows * cols) >> 2;
    GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((row_scale_kernel<T>), dim3(ew_grid(nvec)), dim3(256), 0, as_stream(stream), (const T*)x, s, (T*)out, rows, (long)cols, alpha));
    GS_CHECK_LAUNCH();
    return 0;
}

// sign bits of a bf16 activation in the layout the conv epilogues use (include/gansynth_hip.h): one 32-bit word per (pixel, 32-channel tile),
// bit 8 (2 h + q) + k = channel 16 q + 8 h + k > 0.  A thread per word: 64 bytes in, 4 bytes out.
static __global__ void pack_act_bits_kernel(const unsigned short* __restrict__ z, unsigned* __restrict__ bits, long nwords) {
    const long w = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    const uint4* src = reinterpret_cast<const uint4*>(z + w * 32);
    unsigned word = 0u;
#pragma unroll
    for (int piece = 0; piece < 4; ++piece) {   // piece = 16-byte group of 8 channels: channels 8 piece ..: q = piece >> 1, h = piece & 1
        const uint4 v = src[piece];
        const unsigned b8 = ((int)(v.x << 16) > 0 ? 1u : 0u) | ((int)v.x >= 0x10000 ? 2u : 0u) | ((int)(v.y << 16) > 0 ? 4u : 0u) | ((int)v.y >= 0x10000 ? 8u : 0u) |
                            ((int)(v.z << 16) > 0 ? 16u : 0u) | ((int)v.z >= 0x10000 ? 32u : 0u) | ((int)(v.w << 16) > 0 ? 64u : 0u) | ((int)v.w >= 0x10000 ? 128u : 0u);
        word |= b8 << (8 * (2 * (piece & 1) + (piece >> 1)));
    }
    bits[w] = word;
}

extern "C" int gs_pack_act_bits(void* z, int64_t p, int c, int dtype, void* stream) {
    GS_CHECK_ARG(z && p > 0 && c > 0 && c % 32 == 0 && dtype == GS_BF16, "pack_act_bits: bf16 activations with a multiple of 32 channels only");
    const long nwords = (long)p * c / 32;
    unsigned* bits = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(z) + (long)p * c);
    hipLaunchKernelGGL(pack_act_bits_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, as_stream(stream), reinterpret_cast<const unsigned short*>(z), bits, nwords);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_adam_tf_step(float* p, const float* g, float* m, float* v, int64_t numel, float lr_t, float beta1, float beta2,
                               float eps, float grad_scale, void* stream) {
    GS_CHECK_ARG(numel > 0, "adam: bad args");
    hipLaunchKernelGGL((adam_tf_kernel<false, false>), dim3(ew_grid((numel >> 2) + 4)), dim3(256), 0, as_stream(stream), p, const_cast<float*>(g), m, v,
                       (long)numel, lr_t, (const float*)nullptr, beta1, beta2, eps, grad_scale);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_adam_tf_step_zero_grad(float* p, float* g, float* m, float* v, int64_t numel, float lr_t, float beta1, float beta2,
                                         float eps, float grad_scale, void* stream) {
    GS_CHECK_ARG(numel > 0, "adam: bad args");
    hipLaunchKernelGGL((adam_tf_kernel<true, false>), dim3(ew_grid((numel >> 2) + 4)), dim3(256), 0, as_stream(stream), p, g, m, v, (long)numel, lr_t,
                       (const float*)nullptr, beta1, beta2, eps, grad_scale);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_adam_tf_step_dev(float* p, float* g, float* m, float* v, int64_t numel, const float* lr_t_dev, float beta1, float beta2,
                                   float eps, float grad_scale, int zero_grad, void* stream) {
    GS_CHECK_ARG(numel > 0 && lr_t_dev != nullptr, "adam: bad args");
    const dim3 grid(ew_grid((numel >> 2) + 4));
    if (zero_grad)
        hipLaunchKernelGGL((adam_tf_kernel<true, true>), grid, dim3(256), 0, as_stream(stream), p, g, m, v, (long)numel, 0.f, lr_t_dev, beta1, beta2, eps,
                           grad_scale);
    else
        hipLaunchKernelGGL((adam_tf_kernel<false, true>), grid, dim3(256), 0, as_stream(stream), p, g, m, v, (long)numel, 0.f, lr_t_dev, beta1, beta2, eps,
                           grad_scale);
    GS_CHECK_LAUNCH();
    return 0;
}
