// C-ABI entry points of the conv family + the generic direct (VALU) kernels used for the
// channel counts the MFMA implicit GEMM does not take: the 2-channel colour 1x1 convs
// (networks.py:98-105, 233-240) and the 1-channel minibatch-stddev plane (networks.py:174-176).
#include "conv_shared.h"
#include <utility>
#include <vector>

namespace gs {


// from conv_igemm.hip
bool igemm_supported(int ic, int oc, int dtype);
bool igemm_normbwd_fused(int mode, int N, int Hb, int Wb, int IC, int OC, int dtype, int form = 1);
extern "C" int gs_pixel_norm_bwd_bwd_fused(const void* gg, const void* g, const void* x, void* out, void* out_g, int64_t p, int c, float eps, int pre_act,
                                           int dtype, void* stream);
bool wgrad_mfma_supported(int ic, int oc, int dtype);
size_t igemm_prep_bytes(int ic, int oc, int dtype);
int run_igemm(int mode, int variant, const void* x, const float* w_hwio, void* y, int N, int Hi, int Wi, int ICk,
              int OCk, int w_ci, int w_co, int Hb, int Wb, float alpha, const float* bias, int act, int dtype, int w_prepared,
              void* ws, size_t ws_bytes, hipStream_t st, const void* mask = nullptr, int mask_act = 0, void* y2 = nullptr, float pn_eps = 0.f,
              const void* addend = nullptr, int normbwd = 0);
extern "C" int gs_pixel_norm_bwd_fused(const void* g, const void* x, const void* addend, void* gx, int64_t p, int c, float eps, int pre_act, int post_act, int dtype,
                                       void* stream);
size_t wgrad_mfma_bytes(int mode, int dtype, int N, int Hb, int Wb, int IC, int OC);
bool wgrad_mfma_has_bias(int dtype);
int run_wgrad_mfma(int mode, const WgradSrcs& srcs, int nsrc, float* gw, float* gb, int N, int Hi, int Wi, int IC, int OC, int Hb,
                   int Wb, float alpha, int transpose, int accumulate, int dtype, void* ws, size_t ws_bytes, hipStream_t st,
                   GsWgradReduce* defer = nullptr);
bool wgrad_sk_supported(int mode, int dtype, int IC, int OC);
int wgrad_sk_tile_width(int Wb);
void wgrad_sk_job_geometry(int mode, int tw, int N, SkJob& q);
void wgrad_sk_plan(int mode, SkGroup& g);
size_t wgrad_sk_bytes(const SkGroup& g);
int run_wgrad_sk(int mode, int tw, const SkGroup& g, void* ws, size_t ws_bytes, hipStream_t st);

// ------------------------------------------------------------------------- direct gather conv
// y[n][oy][ox][oc0..oc0+OCV) = alpha * sum_{tap,ic} x[n][iy][ix][ic] * wp[tap][oc][ic]   (wp fp32)
template <typename T, int OCV, int VEC>
__global__ void conv_direct_kernel(const T* __restrict__ x, const float* __restrict__ wp, T* __restrict__ y, int mode,
                                   int ks, int N, int Hi, int Wi, int IC, int OC, int Ho, int Wo, float alpha) {
    const int ngrp = OC / OCV;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)N * Ho * Wo * ngrp;
    if (idx >= total) return;
    const int g = idx % ngrp;
    long pix = idx / ngrp;
    const int ox = pix % Wo;
    pix /= Wo;
    const int oy = pix % Ho;
    const int n = pix / Ho;
    const int oc = g * OCV;
    float acc[OCV];
#pragma unroll
    for (int v = 0; v < OCV; ++v) acc[v] = 0.f;
    for (int ky = 0; ky < ks; ++ky) {
        int iy;
        if (mode == MODE_S1) iy = oy + ky - (ks >> 1);
        else if (mode == MODE_S2) iy = 2 * oy + ky;
        else { const int d = oy - ky; if (d < 0 || (d & 1)) continue; iy = d >> 1; }
        if (iy < 0 || iy >= Hi) continue;
        for (int kx = 0; kx < ks; ++kx) {
            int ix;
            if (mode == MODE_S1) ix = ox + kx - (ks >> 1);
            else if (mode == MODE_S2) ix = 2 * ox + kx;
            else { const int d = ox - kx; if (d < 0 || (d & 1)) continue; ix = d >> 1; }
            if (ix < 0 || ix >= Wi) continue;
            const T* xp = x + (((long)n * Hi + iy) * Wi + ix) * IC;
            const float* wr = wp + ((long)(ky * ks + kx) * OC + oc) * IC;
            if (VEC == 4) {
                for (int ic = 0; ic < IC; ic += 4) {
                    float xv[4];
                    ld4(xp + ic, xv);
#pragma unroll
                    for (int v = 0; v < OCV; ++v) {
                        const float4 wv = *reinterpret_cast<const float4*>(wr + (long)v * IC + ic);
                        acc[v] += xv[0] * wv.x + xv[1] * wv.y + xv[2] * wv.z + xv[3] * wv.w;
                    }
                }
            } else {
                for (int ic = 0; ic < IC; ++ic) {
                    const float xv = DT<T>::ld(xp + ic);
#pragma unroll
                    for (int v = 0; v < OCV; ++v) acc[v] += xv * wr[(long)v * IC + ic];
                }
            }
        }
    }
    T* yp = y + (((long)n * Ho + oy) * Wo + ox) * OC + oc;
    if (OCV == 4) {
        float o[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) o[v] = acc[v < OCV ? v : 0] * alpha;
        st4(yp, o);
    } else {
#pragma unroll
        for (int v = 0; v < OCV; ++v) DT<T>::st(yp + v, acc[v] * alpha);
    }
}

template <typename T>
static int launch_direct(const T* x, const float* wp, T* y, int mode, int ks, int N, int Hi, int Wi, int IC, int OC,
                         int Ho, int Wo, float alpha, hipStream_t st) {
    const int ocv = OC % 4 == 0 ? 4 : (OC % 2 == 0 ? 2 : 1);
    const int vec = IC % 4 == 0 ? 4 : 1;
    const long total = (long)N * Ho * Wo * (OC / ocv);
    dim3 grid(cdiv(total, 256)), block(256);
#define GS_DL(OCVV, VECV)                                                                                              \
    hipLaunchKernelGGL((conv_direct_kernel<T, OCVV, VECV>), grid, block, 0, st, x, wp, y, mode, ks, N, Hi, Wi, IC, OC, Ho, \
                       Wo, alpha)
    if (ocv == 4 && vec == 4) GS_DL(4, 4);
    else if (ocv == 4) GS_DL(4, 1);
    else if (ocv == 2 && vec == 4) GS_DL(2, 4);
    else if (ocv == 2) GS_DL(2, 1);
    else if (vec == 4) GS_DL(1, 4);
    else GS_DL(1, 1);
#undef GS_DL
    GS_CHECK_LAUNCH();
    return 0;
}


// ------------------------------------------------------------------ thin 1x1 convolutions
// The colour blocks (ops.py:237-243 with a [1,1,C,2] / [1,1,2,C] kernel) are pure streaming: 64 bytes in and 4 out per
// pixel, or the reverse.  One lane per 16 bytes of the wide side keeps every access coalesced; the bias / activation
// epilogue of the block is applied in the same pass.
// The activation of a streaming kernel is a TEMPLATE parameter: as a run-time argument the three-way choice (tanhf's own branches
// included) was compiled into the pixel loop once per output VALUE -- ~6 scalar branches per value in thin_expand_kernel's hot loop
// (hipcc -S), 23 us for the 67 MB of the top-level colour block where the write rate allows ~14.
template <int ACT> __device__ inline float thin_act(float v) {
    if constexpr (ACT == GS_ACT_LRELU) return fmaxf(v, 0.2f * v);
    else if constexpr (ACT == GS_ACT_TANH) return fast_tanh(v);
    else return v;
}
// few -> many channels: y[p][oc] = act(alpha * sum_ic x[p][ic] wp[oc][ic] + bias[oc]),  IC <= 4, OC % Wide::N == 0, 256 % (OC / N) == 0.
// A thread keeps its Wide::N output channels (weights + bias in registers) and strides over pixels.
template <typename T, int IC, int ACT, bool MASKED>
// `mask` (MASKED, y's shape): y *= mask_act'(.) through that activation output -- the second-order pass of the R1 penalty runs the colour
// block forward on a cotangent and the next node's first step is that multiplication (gs_conv2d_fwd_mask)
__global__ __launch_bounds__(256) void thin_expand_kernel(const T* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                                                          T* __restrict__ y, long P, int OC, float alpha,
                                                          const T* __restrict__ mask = nullptr, int mask_act = 0, unsigned* __restrict__ bits = nullptr) {
    // `bits` (bf16 leaky relu, OC % 32 == 0): the sign words of the result behind it (include/gansynth_hip.h, GS_ACT_WRITE_BITS) -- a lane's 8 channels
    // are one byte of the pixel's word, the four lanes of a 32-channel tile are a DPP quad
    constexpr int WN = Wide<T>::N;
    const int groups = OC / WN;
    const int oc0 = (threadIdx.x % groups) * WN;
    float wr[WN][IC], br[WN];
    {   // this lane's WN x IC weights and WN biases: contiguous floats, fetched as 16-byte vectors (a scalar load each made the prologue of
        // a block cost as much as its pixels: 22.8 us at 4096 blocks, 51 us at 16384, scripts/bench_thin.py)
        typedef float f4_t __attribute__((ext_vector_type(4)));
        float flat[WN * IC];
        if constexpr ((WN * IC) % 4 == 0) {
#pragma unroll
            for (int q = 0; q < WN * IC / 4; ++q) {
                const f4_t t = *reinterpret_cast<const f4_t*>(wp + (long)oc0 * IC + 4 * q);
                flat[4 * q] = t.x; flat[4 * q + 1] = t.y; flat[4 * q + 2] = t.z; flat[4 * q + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < WN * IC; ++q) flat[q] = wp[(long)oc0 * IC + q];
        }
#pragma unroll
        for (int v = 0; v < WN; ++v)
#pragma unroll
            for (int i = 0; i < IC; ++i) wr[v][i] = flat[v * IC + i] * alpha;
#pragma unroll
        for (int q = 0; q < WN / 4; ++q) {
            f4_t t = {0.f, 0.f, 0.f, 0.f};
            if (bias) t = *reinterpret_cast<const f4_t*>(bias + oc0 + 4 * q);
            br[4 * q] = t.x; br[4 * q + 1] = t.y; br[4 * q + 2] = t.z; br[4 * q + 3] = t.w;
        }
    }
    const long ppb = 256 / groups;  // pixels per block pass
    const long stride = (long)gridDim.x * ppb;
    long pix = (long)blockIdx.x * ppb + threadIdx.x / groups;
    auto one = [&](long px, const float* xv) __attribute__((always_inline)) {
        float o[WN];
        // two output channels per instruction (v_pk_fma_f32 / v_pk_mul_f32): the kernel's time IS its VALU count -- 50 instructions per 16
        // bytes stored with scalar arithmetic (hipcc split every multiply-add into a packed multiply and an add), ~30 like this
        typedef float f2_t __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int v = 0; v < WN; v += 2) {
            f2_t a = {br[v], br[v + 1]};
#pragma unroll
            for (int i = 0; i < IC; ++i) {
                const f2_t xx = {xv[i], xv[i]}, ww = {wr[v][i], wr[v + 1][i]};
                a = __builtin_elementwise_fma(xx, ww, a);
            }
            if constexpr (ACT == GS_ACT_LRELU) {
                const f2_t t = a * 0.2f;
                o[v] = fmaxf(a.x, t.x);
                o[v + 1] = fmaxf(a.y, t.y);
            } else {
                o[v] = thin_act<ACT>(a.x);
                o[v + 1] = thin_act<ACT>(a.y);
            }
        }
        if constexpr (MASKED) {
            float mv[WN];
            ld_wide<T>(mask + px * OC + oc0, mv);
#pragma unroll
            for (int v = 0; v < WN; ++v) o[v] *= mask_act == GS_ACT_LRELU ? (mv[v] > 0.f ? 1.f : 0.2f) : (mask_act == GS_ACT_TANH ? 1.f - mv[v] * mv[v] : 1.f);
        }
        if constexpr (ACT == GS_ACT_LRELU && !MASKED && sizeof(T) == 2) {
            if (bits) {
                uint4 v;
                v.x = pack_bf16x2(o[0], o[1]); v.y = pack_bf16x2(o[2], o[3]); v.z = pack_bf16x2(o[4], o[5]); v.w = pack_bf16x2(o[6], o[7]);
                *reinterpret_cast<uint4*>(y + px * OC + oc0) = v;
                const unsigned b8 = ((int)(v.x << 16) > 0 ? 1u : 0u) | ((int)v.x >= 0x10000 ? 2u : 0u) | ((int)(v.y << 16) > 0 ? 4u : 0u) | ((int)v.y >= 0x10000 ? 8u : 0u) |
                                    ((int)(v.z << 16) > 0 ? 16u : 0u) | ((int)v.z >= 0x10000 ? 32u : 0u) | ((int)(v.w << 16) > 0 ? 64u : 0u) | ((int)v.w >= 0x10000 ? 128u : 0u);
                const int piece = (oc0 >> 3) & 3;   // 16-byte piece of the tile: channels 8 piece .. -> byte 2 (piece & 1) + (piece >> 1) of the word
                unsigned word = b8 << (8 * (2 * (piece & 1) + (piece >> 1)));
                word |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)word, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
                word |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)word, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
                if (piece == 0) bits[(px * OC + oc0) >> 5] = word;
                return;
            }
        }
        st_wide<T>(y + px * OC + oc0, o);
    };
    auto ldpix = [&](long px, float* xv) __attribute__((always_inline)) {
        if constexpr (IC == 2 && sizeof(T) == 2) {   // both colour channels of a pixel in one 4-byte load
            const unsigned u = *reinterpret_cast<const unsigned*>(x + px * 2);
            xv[0] = __uint_as_float(u << 16);
            xv[1] = __uint_as_float(u & 0xffff0000u);
        } else {
#pragma unroll
            for (int i = 0; i < IC; ++i) xv[i] = DT<T>::ld(x + px * IC + i);
        }
    };
    for (; pix + 3 * stride < P; pix += 4 * stride) {   // four pixels per trip (loads first)
        float xv[4][IC];
#pragma unroll
        for (int k = 0; k < 4; ++k) ldpix(pix + k * stride, xv[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) one(pix + k * stride, xv[k]);
    }
    for (; pix < P; pix += stride) {
        float xv[IC];
        ldpix(pix, xv);
        one(pix, xv);
    }
}

// The same map as a DATA GRADIENT continued through the pixel norm and activation of the block that produced the conv's input
// (gs_conv2d_bwd_data_pnbwd for the colour block, networks.py:98-104): g[p][oc] = alpha * sum_ic x[p][ic] wp[oc][ic] is the gradient
// w.r.t. pixel_norm(z); out = (r (g - z r^2 mean_c(z g)) + addend) * act'(z), r = rsqrt(mean_c z^2 + eps).  The OC / Wide::N lanes of
// a pixel are neighbours in the wave: two xor-shuffle folds give the pixel's sums.  One pass instead of thin_expand + a 4-tensor norm pass.
template <typename T, int IC>
__global__ __launch_bounds__(256) void thin_expand_pnbwd_kernel(const T* __restrict__ x, const float* __restrict__ wp, const T* __restrict__ z,
                                                                const T* __restrict__ addend, T* __restrict__ y, long P, int OC, float alpha, float eps, int act) {
    constexpr int WN = Wide<T>::N;
    const int groups = OC / WN;   // a power of two <= 64 dividing 256 (checked by the launcher)
    const int oc0 = (threadIdx.x % groups) * WN;
    float wr[WN][IC];
#pragma unroll
    for (int v = 0; v < WN; ++v)
#pragma unroll
        for (int i = 0; i < IC; ++i) wr[v][i] = wp[(long)(oc0 + v) * IC + i] * alpha;
    const long ppb = 256 / groups;
    const long stride = (long)gridDim.x * ppb;
    const long npass = (P + stride - 1) / stride;   // same trip count for every lane (shuffles)
    const float inv_c = 1.f / (float)OC;
    for (long k = 0; k < npass; ++k) {
        const long pix = (long)blockIdx.x * ppb + threadIdx.x / groups + k * stride;
        const long px = pix < P ? pix : P - 1;      // (clamped: loads and shuffles stay unconditional)
        float xv[IC], zv[WN], av[WN], g[WN];
        if constexpr (IC == 2 && sizeof(T) == 2) {
            const unsigned u = *reinterpret_cast<const unsigned*>(x + px * 2);
            xv[0] = __uint_as_float(u << 16);
            xv[1] = __uint_as_float(u & 0xffff0000u);
        } else {
#pragma unroll
            for (int i = 0; i < IC; ++i) xv[i] = DT<T>::ld(x + px * IC + i);
        }
        ld_wide<T>(z + px * OC + oc0, zv);
        if (addend) ld_wide<T>(addend + px * OC + oc0, av);
        float ssq = 0.f, szg = 0.f;
#pragma unroll
        for (int v = 0; v < WN; ++v) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < IC; ++i) a += xv[i] * wr[v][i];
            g[v] = a;
            ssq += zv[v] * zv[v];
            szg += zv[v] * a;
        }
        ssq = group_sum(ssq, groups);
        szg = group_sum(szg, groups);
        const float r = rsqrtf(ssq * inv_c + eps);
        const float m = szg * inv_c * r * r;
        float out[WN];
#pragma unroll
        for (int v = 0; v < WN; ++v) {
            float t = r * (g[v] - zv[v] * m);
            if (addend) t += av[v];
            out[v] = act == GS_ACT_LRELU ? (zv[v] > 0.f ? t : 0.2f * t) : t;
        }
        if (pix < P) st_wide<T>(y + pix * OC + oc0, out);
    }
}
static bool thin_expand_pnbwd_ok(int ks, int ICk, int OCk, int dtype) {
    const int wn = dtype == GS_F32 ? 4 : 8;
    if (ks != 1 || ICk < 1 || ICk > 4 || OCk % wn != 0) return false;
    const int groups = OCk / wn;
    return groups >= 1 && groups <= 64 && (groups & (groups - 1)) == 0;
}
// x: the thin gradient [P][ICk], wp: fp32 [OCk][ICk] (bwd-data layout of the 1x1 kernel), z / addend / y: [P][OCk]
static int run_thin_expand_pnbwd(const void* x, const float* wp, const void* z, const void* addend, void* y, long P, int ICk, int OCk, float alpha, float eps, int act,
                                 int dtype, hipStream_t st) {
    const int wn = dtype == GS_F32 ? 4 : 8;
    long nb = cdiv(P * (OCk / wn), 256);
    if (nb > 4096) nb = 4096;
    const unsigned grid = (unsigned)nb;
#define GS_TEP(TT, ICV) hipLaunchKernelGGL((thin_expand_pnbwd_kernel<TT, ICV>), dim3(grid), dim3(256), 0, st, (const TT*)x, wp, (const TT*)z, (const TT*)addend, (TT*)y, P, OCk, alpha, eps, act)
#define GS_TEP_ALL(TT) do { if (ICk == 1) GS_TEP(TT, 1); else if (ICk == 2) GS_TEP(TT, 2); else if (ICk == 3) GS_TEP(TT, 3); else GS_TEP(TT, 4); } while (0)
    GS_DISPATCH_DTYPE(dtype, GS_TEP_ALL(T));
#undef GS_TEP_ALL
#undef GS_TEP
    GS_CHECK_LAUNCH();
    return 0;
}

// many -> few channels: a pixel is read by L = IC / Wide::N lanes (a power of two <= 64), partial dots are folded with
// xor-shuffles, lane 0 of the group writes the OC <= 4 results.  Weights stay in registers across the pixel loop.
// LC: the lanes per pixel as a compile-time constant (4: the 32-channel bf16 colour block; 8: 64 bf16 / 32 fp32 channels; 0: any power of
// two, run-time) -- with a run-time lane count the folds are loops of ds_bpermute round trips through the LDS pipe, with a constant one
// they unroll into DPP moves.
template <typename T, int OC, int ACT, int LC>
__global__ __launch_bounds__(256) void thin_reduce_kernel(const T* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                                                          T* __restrict__ y, long P, int IC, float alpha) {
    constexpr int WN = Wide<T>::N;
    const int L = LC ? LC : IC / WN;
    const int l = threadIdx.x % L;
    float wr[OC][WN];
#pragma unroll
    for (int v = 0; v < OC; ++v)
#pragma unroll
        for (int i = 0; i < WN; ++i) wr[v][i] = wp[v * IC + l * WN + i] * alpha;
    const long ppb = 256 / L;
    const long npass = (P + (long)gridDim.x * ppb - 1) / ((long)gridDim.x * ppb);  // same trip count for every lane (shuffles)
    float bv[OC];
#pragma unroll
    for (int v = 0; v < OC; ++v) bv[v] = bias ? bias[v] : 0.f;
    auto fold = [&](float* a) __attribute__((always_inline)) {
        // VALU only (DPP inside a row, permlane swaps across rows; gs_common.h: a __shfl_xor is a ds_bpermute whatever its offset)
#pragma unroll
        for (int v = 0; v < OC; ++v) a[v] = group_sum(a[v], LC != 0 ? LC : L);
    };
    auto finish = [&](long pix, float* a) __attribute__((always_inline)) {
        fold(a);
        if (pix < P && l == 0) {
            if constexpr (OC == 2 && sizeof(T) == 2) {   // both colour channels of a pixel in one 4-byte store
                *reinterpret_cast<unsigned*>(y + pix * 2) = pack_bf16x2(thin_act<ACT>(a[0] + bv[0]), thin_act<ACT>(a[1] + bv[1]));
            } else {
#pragma unroll
                for (int v = 0; v < OC; ++v) DT<T>::st(y + pix * OC + v, thin_act<ACT>(a[v] + bv[v]));
            }
        }
    };
    constexpr int U = 4;   // pixels per trip: their loads go out together
    long k = 0;
    for (; k + U <= npass; k += U) {
        float xv[U][WN], a[U][OC];
        long pix[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pix[u] = ((k + u) * gridDim.x + blockIdx.x) * ppb + threadIdx.x / L;
            ld_wide<T>(x + (pix[u] < P ? pix[u] : 0) * IC + l * WN, xv[u]);   // (clamped: the loads stay unconditional)
        }
        if constexpr (LC == U) {
            // after the folds each of the pixel's LC lanes holds the sums of all U pixels of the trip: lane l finishes pixel l -- bias,
            // activation (a tanh on the generator's colour block) and the store run ONCE per lane instead of U times on a quarter of them
            float mine[OC];
            long mypix = pix[0];
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int v = 0; v < OC; ++v) {
                    a[u][v] = 0.f;
#pragma unroll
                    for (int i = 0; i < WN; ++i) a[u][v] += xv[u][i] * wr[v][i];
                }
                fold(a[u]);
#pragma unroll
                for (int v = 0; v < OC; ++v) mine[v] = (u == 0 || l == u) ? a[u][v] : mine[v];
                mypix = l == u ? pix[u] : mypix;
            }
            if (mypix < P) {
                if constexpr (OC == 2 && sizeof(T) == 2) {
                    *reinterpret_cast<unsigned*>(y + mypix * 2) = pack_bf16x2(thin_act<ACT>(mine[0] + bv[0]), thin_act<ACT>(mine[1] + bv[1]));
                } else {
#pragma unroll
                    for (int v = 0; v < OC; ++v) DT<T>::st(y + mypix * OC + v, thin_act<ACT>(mine[v] + bv[v]));
                }
            }
            continue;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int v = 0; v < OC; ++v) {
                a[u][v] = 0.f;
#pragma unroll
                for (int i = 0; i < WN; ++i) a[u][v] += xv[u][i] * wr[v][i];
            }
            finish(pix[u], a[u]);
        }
    }
    for (; k < npass; ++k) {
        const long pix = (k * gridDim.x + blockIdx.x) * ppb + threadIdx.x / L;
        float a[OC];
#pragma unroll
        for (int v = 0; v < OC; ++v) a[v] = 0.f;
        if (pix < P) {
            float xv[WN];
            ld_wide<T>(x + pix * IC + l * WN, xv);
#pragma unroll
            for (int v = 0; v < OC; ++v)
#pragma unroll
                for (int i = 0; i < WN; ++i) a[v] += xv[i] * wr[v][i];
        }
        finish(pix, a);
    }
}

// one output channel, many input channels, any tap geometry (the data gradient of the minibatch-stddev plane of the last
// discriminator block): a wave per output pixel, lanes over the input channels.
template <typename T>
__global__ __launch_bounds__(256) void thin_single_kernel(const T* __restrict__ x, const float* __restrict__ wp, T* __restrict__ y, int mode,
                                                          int ks, int N, int Hi, int Wi, int IC, int Ho, int Wo, float alpha) {
    const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pix >= (long)N * Ho * Wo) return;
    const int ox = pix % Wo;
    const int oy = (pix / Wo) % Ho;
    const int n = pix / ((long)Wo * Ho);
    float a = 0.f;
    for (int ky = 0; ky < ks; ++ky) {
        int iy;
        if (mode == MODE_S1) iy = oy + ky - (ks >> 1);
        else if (mode == MODE_S2) iy = 2 * oy + ky;
        else { const int d = oy - ky; if (d < 0 || (d & 1)) continue; iy = d >> 1; }
        if (iy < 0 || iy >= Hi) continue;
        for (int kx = 0; kx < ks; ++kx) {
            int ix;
            if (mode == MODE_S1) ix = ox + kx - (ks >> 1);
            else if (mode == MODE_S2) ix = 2 * ox + kx;
            else { const int d = ox - kx; if (d < 0 || (d & 1)) continue; ix = d >> 1; }
            if (ix < 0 || ix >= Wi) continue;
            const T* xp = x + (((long)n * Hi + iy) * Wi + ix) * IC;
            const float* wr = wp + (long)(ky * ks + kx) * IC;
            for (int ic = lane; ic < IC; ic += 64) a += DT<T>::ld(xp + ic) * wr[ic];
        }
    }
    a = wave_sum(a);
    if (lane == 0) DT<T>::st(y + pix, a * alpha);
}

// the same for 3x3 kernels with IC a multiple of 256: a lane owns 4 consecutive channels per 256-channel block and ALL its loads (9 taps x
// IC / 256 blocks, inputs and weights) are issued before the first multiply -- the generic kernel above walks 9 x IC / 64 dependent
// load pairs (11.6 us for the 256-channel stddev-plane gradient at 2 x 16: a pure latency chain)
template <typename T, int NB>
__global__ __launch_bounds__(256) void thin_single3_kernel(const T* __restrict__ x, const float* __restrict__ wp, T* __restrict__ y, int mode,
                                                           int N, int Hi, int Wi, int Ho, int Wo, float alpha) {
    constexpr int IC = 256 * NB;
    const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pix >= (long)N * Ho * Wo) return;
    const int ox = pix % Wo;
    const int oy = (pix / Wo) % Ho;
    const int n = pix / ((long)Wo * Ho);
    float xv[9][NB][4];
    float4 wv[9][NB];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t % 3;
        int iy, ix;
        bool ok = true;
        if (mode == MODE_S1) { iy = oy + ky - 1; ix = ox + kx - 1; }
        else if (mode == MODE_S2) { iy = 2 * oy + ky; ix = 2 * ox + kx; }
        else {
            const int dy = oy - ky, dx = ox - kx;
            ok = dy >= 0 && dx >= 0 && !(dy & 1) && !(dx & 1);
            iy = dy >> 1; ix = dx >> 1;
        }
        ok = ok && iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;
        const T* xp = x + (((long)n * Hi + (ok ? iy : 0)) * Wi + (ok ? ix : 0)) * IC + 4 * lane;   // (clamped: the loads stay unconditional)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            ld4(xp + 256 * b, xv[t][b]);
            wv[t][b] = ok ? *reinterpret_cast<const float4*>(wp + (long)t * IC + 256 * b + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float a = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b) a += xv[t][b][0] * wv[t][b].x + xv[t][b][1] * wv[t][b].y + xv[t][b][2] * wv[t][b].z + xv[t][b][3] * wv[t][b].w;
    a = wave_sum(a);
    if (lane == 0) DT<T>::st(y + pix, a * alpha);
}

// direct conv through the fp32 prepped weights living in ws; *fused is set when bias / act went into the same pass
static int run_direct(int mode, int ks, int variant, const void* x, const float* w_hwio, void* y, int N, int Hi, int Wi,
                      int ICk, int OCk, int w_ci, int w_co, int Ho, int Wo, float alpha, int dtype, int w_prepared, void* ws,
                      size_t ws_bytes, hipStream_t st, const float* bias = nullptr, int act = GS_ACT_NONE, bool* fused = nullptr,
                      const void* mask = nullptr, int mask_act = 0, bool* mask_fused = nullptr, unsigned* bits_out = nullptr, bool* bits_done = nullptr) {
    const long total = (long)ks * ks * w_ci * w_co;
    if (ws_bytes < (size_t)total * 4) return fail(GS_ERR_WORKSPACE, "conv direct: workspace %zu < %zu", ws_bytes, (size_t)total * 4);
    float* wp = reinterpret_cast<float*>(ws);
    if (!w_prepared) {
        hipLaunchKernelGGL((weight_prep_kernel<float>), dim3(cdiv(total, 256)), dim3(256), 0, st, w_hwio, wp, ks * ks, w_ci, w_co, variant);
        GS_CHECK_LAUNCH();
    }
    if (fused) *fused = false;
    const long P = (long)N * Ho * Wo;
    const int wn = dtype == GS_F32 ? 4 : 8;
    if (ks == 1 && mode == MODE_S1 && ICk <= 4 && OCk % wn == 0 && 256 % (OCk / wn) == 0) {   // colour -> features
        long nb = cdiv(P * (OCk / wn), 256);
        // (2048 blocks: 18.5 us for the 67 MB of the top level against 19.4 at 4096 and 23.0 at 8192 -- every block pays the weight prologue)
        static const long te_cap = getenv("GS_THIN_EXPAND_BLOCKS") ? atol(getenv("GS_THIN_EXPAND_BLOCKS")) : 2048;
        if (nb > te_cap) nb = te_cap;
        const unsigned grid = (unsigned)nb;
        unsigned* te_bits = (bits_out && bits_done && !mask && act == GS_ACT_LRELU && dtype == GS_BF16 && OCk % 32 == 0) ? bits_out : nullptr;
        if (te_bits) *bits_done = true;
#define GS_TE(TT, ICV, ACTV, MK) hipLaunchKernelGGL((thin_expand_kernel<TT, ICV, ACTV, MK>), dim3(grid), dim3(256), 0, st, (const TT*)x, wp, bias, (TT*)y, P, OCk, alpha, (const TT*)mask, mask_act, te_bits)
#define GS_TE_ACT(TT, ICV)                                                                                               \
    do {                                                                                                                 \
        if (mask) {   /* (masked: a forward on a cotangent -- the activation, if any, is applied by the slower generic form) */ \
            if (act == GS_ACT_NONE) GS_TE(TT, ICV, GS_ACT_NONE, true);                                                   \
            else if (act == GS_ACT_LRELU) GS_TE(TT, ICV, GS_ACT_LRELU, true);                                            \
            else GS_TE(TT, ICV, GS_ACT_TANH, true);                                                                      \
        } else if (act == GS_ACT_LRELU) GS_TE(TT, ICV, GS_ACT_LRELU, false);                                             \
        else if (act == GS_ACT_TANH) GS_TE(TT, ICV, GS_ACT_TANH, false);                                                 \
        else GS_TE(TT, ICV, GS_ACT_NONE, false);                                                                         \
    } while (0)
#define GS_TE_ALL(TT) do { if (ICk == 1) GS_TE_ACT(TT, 1); else if (ICk == 2) GS_TE_ACT(TT, 2); else if (ICk == 3) GS_TE_ACT(TT, 3); else GS_TE_ACT(TT, 4); } while (0)
        GS_DISPATCH_DTYPE(dtype, GS_TE_ALL(T));
#undef GS_TE_ALL
#undef GS_TE_ACT
#undef GS_TE
        GS_CHECK_LAUNCH();
        if (mask_fused) *mask_fused = mask != nullptr;
        if (fused) *fused = true;
        else if (bias || act != GS_ACT_NONE) return fail(GS_ERR_ARG, "conv direct: epilogue requested without a fused flag");
        return 0;
    }
    const int lanes = ICk / wn;
    if (ks == 1 && mode == MODE_S1 && OCk <= 4 && ICk % wn == 0 && lanes <= 64 && (lanes & (lanes - 1)) == 0) {  // features -> colour
        long nb = cdiv(P * lanes, 256);
        static const long tr_cap = getenv("GS_THIN_REDUCE_BLOCKS") ? atol(getenv("GS_THIN_REDUCE_BLOCKS")) : 4096;
        if (nb > tr_cap) nb = tr_cap;
        const unsigned grid = (unsigned)nb;
#define GS_TRL(TT, OCV, ACTV, LV) hipLaunchKernelGGL((thin_reduce_kernel<TT, OCV, ACTV, LV>), dim3(grid), dim3(256), 0, st, (const TT*)x, wp, bias, (TT*)y, P, ICk, alpha)
#define GS_TR(TT, OCV, ACTV) do { if (lanes == 4) GS_TRL(TT, OCV, ACTV, 4); else if (lanes == 8) GS_TRL(TT, OCV, ACTV, 8); else GS_TRL(TT, OCV, ACTV, 0); } while (0)
#define GS_TR_ACT(TT, OCV) do { if (act == GS_ACT_LRELU) GS_TR(TT, OCV, GS_ACT_LRELU); else if (act == GS_ACT_TANH) GS_TR(TT, OCV, GS_ACT_TANH); else GS_TR(TT, OCV, GS_ACT_NONE); } while (0)
#define GS_TR_ALL(TT) do { if (OCk == 1) GS_TR_ACT(TT, 1); else if (OCk == 2) GS_TR_ACT(TT, 2); else if (OCk == 3) GS_TR_ACT(TT, 3); else GS_TR_ACT(TT, 4); } while (0)
        GS_DISPATCH_DTYPE(dtype, GS_TR_ALL(T));
#undef GS_TR_ALL
#undef GS_TR_ACT
#undef GS_TR
#undef GS_TRL
        GS_CHECK_LAUNCH();
        if (fused) *fused = true;
        return 0;
    }
    if (OCk == 1 && ICk == 256 && ks == 3) {
        GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((thin_single3_kernel<T, 1>), dim3((unsigned)cdiv(P, 4)), dim3(256), 0, st, (const T*)x, wp, (T*)y, mode,
                                                    N, Hi, Wi, Ho, Wo, alpha));
        GS_CHECK_LAUNCH();
        return 0;
    }
    if (OCk == 1 && ICk >= 64) {
        GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((thin_single_kernel<T>), dim3((unsigned)cdiv(P, 4)), dim3(256), 0, st, (const T*)x, wp, (T*)y, mode,
                                                    ks, N, Hi, Wi, ICk, Ho, Wo, alpha));
        GS_CHECK_LAUNCH();
        return 0;
    }
    GS_DISPATCH_DTYPE(dtype, return launch_direct<T>(reinterpret_cast<const T*>(x), wp, reinterpret_cast<T*>(y), mode, ks,
                                                     N, Hi, Wi, ICk, OCk, Ho, Wo, alpha, st));
}

// ------------------------------------------------------------------ batched weight re-layout
// which operand a map's entry point builds: (variant of weight_prep_kernel, fp32 storage for the direct kernels / T for MFMA)
__device__ __host__ inline void prep_plan(int map, int ci, int co, int ksize, int dtype, int* variant, bool* store_f32) {
    const int bk = dtype == GS_F32 ? 16 : 32;
    bool mfma;
    if (map == GS_PREP_CONV_FWD || map == GS_PREP_CONVT_FWD) {
        *variant = 0;
        mfma = ksize == 3 && ci % bk == 0 && co % 32 == 0;
    } else if (map == GS_PREP_CONV_BWD_DATA) {   // stride 1: flipped taps; stride 2: transposed-conv walk (set by the caller)
        *variant = 1;
        mfma = ksize == 3 && co % bk == 0 && ci % 32 == 0;
    } else {
        *variant = 2;
        mfma = co % bk == 0 && ci % 32 == 0;
    }
    *store_f32 = !mfma || dtype == GS_F32;
}
// One 64 x 64 (ci x co) tile of one tap per trip, staged through LDS so that both sides move whole 256-byte rows: the forward operand
// is the TRANSPOSE of the stored [ci][co] slab (a thread-per-element gather read it with a stride of co floats: 31 us per launch for the
// ~20 MB of a network, now bandwidth-bound); the data-gradient operands are (tap-flipped) copies.
static __global__ __launch_bounds__(256) void weight_prep_batch_kernel(const GsPrepDesc* __restrict__ descs) {
    __shared__ float tile[64][65];
    const GsPrepDesc d = descs[blockIdx.y];
    int variant;
    bool f32;
    prep_plan(d.map, d.ci, d.co, d.ksize, d.dtype, &variant, &f32);
    if (d.map == GS_PREP_CONV_BWD_DATA && d.stride == 2) variant = 2;
    const int taps = d.ksize * d.ksize, ci = d.ci, co = d.co;
    const int tci = (ci + 63) >> 6, tco = (co + 63) >> 6;
    const int ntiles = taps * tci * tco;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 x 16 threads, 4 consecutive elements each
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int t = tl / (tci * tco), r = tl % (tci * tco);
        const int i0 = (r / tco) * 64, o0 = (r % tco) * 64;
        const float* src = d.w_hwio + (long)t * ci * co;
        if (variant == 0) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {   // rows i0 + ty + 16 k, columns o0 + 4 tx ..
                const int i = i0 + ty + 16 * k, o = o0 + 4 * tx;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (i < ci) {
                    if (o + 3 < co && (co & 3) == 0) { const float4 q = *reinterpret_cast<const float4*>(src + (long)i * co + o); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
                    else { for (int e = 0; e < 4; ++e) if (o + e < co) v[e] = src[(long)i * co + o + e]; }
                }
                for (int e = 0; e < 4; ++e) tile[ty + 16 * k][4 * tx + e] = v[e];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {   // output rows o0 + ty + 16 k, 4 consecutive input channels i0 + 4 tx ..
                const int o = o0 + ty + 16 * k, i = i0 + 4 * tx;
                if (o >= co) continue;
                float v[4];
                for (int e = 0; e < 4; ++e) v[e] = tile[4 * tx + e][ty + 16 * k];
                const long dst = ((long)t * co + o) * ci + i;
                if (i + 3 < ci && (ci & 3) == 0) {
                    if (f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.ws) + dst) = make_float4(v[0], v[1], v[2], v[3]);
                    else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(d.ws) + dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (i + e < ci) {
                            if (f32) reinterpret_cast<float*>(d.ws)[dst + e] = v[e];
                            else reinterpret_cast<bf16_t*>(d.ws)[dst + e] = f32_to_bf16(v[e]);
                        }
                }
            }
        } else {
            const int tt = variant == 1 ? taps - 1 - t : t;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + ty + 16 * k, o = o0 + 4 * tx;
                if (i >= ci) continue;
                const long so = (long)i * co + o, dst = (long)tt * ci * co + so;
                if (o + 3 < co && (co & 3) == 0) {
                    const float4 q = *reinterpret_cast<const float4*>(src + so);
                    if (f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.ws) + dst) = q;
                    else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(d.ws) + dst) = make_uint2(pack_bf16x2(q.x, q.y), pack_bf16x2(q.z, q.w));
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (o + e < co) {
                            if (f32) reinterpret_cast<float*>(d.ws)[dst + e] = src[so + e];
                            else reinterpret_cast<bf16_t*>(d.ws)[dst + e] = f32_to_bf16(src[so + e]);
                        }
                }
            }
        }
    }
}

extern "C" int gs_weight_prep_batch(const GsPrepDesc* descs, int n, void* stream) {
    GS_CHECK_ARG(descs != nullptr && n > 0 && n <= 65535, "weight_prep_batch: bad args");
    hipLaunchKernelGGL(weight_prep_batch_kernel, dim3(48, n), dim3(256), 0, as_stream(stream), descs);
    GS_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ direct weight gradient
// part[slice][t][ic][oc] = sum over the slice's output pixels of x[in(p,t)][ic] * gy[p][oc]
template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_direct_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                                float* __restrict__ part, int mode, int ks, int N, int Hi,
                                                                int Wi, int IC, int OC, int Hb, int Wb, long E, long npix,
                                                                long pps) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const long e = (long)blockIdx.x * 64 + (tid & 63);
    const int sub = tid >> 6;
    const int slice = blockIdx.y;
    float acc = 0.f;
    if (e < E) {
        const int oc = e % OC;
        const int ic = (e / OC) % IC;
        const int t = e / ((long)IC * OC);
        const int ky = t / ks, kx = t % ks;
        const long p0 = (long)slice * pps;
        long p1 = p0 + pps;
        if (p1 > npix) p1 = npix;
        // (branch-free body, clamped addresses: unrolled, the loads of four pixels are in flight together -- with a `continue` per pixel the
        //  loop was a chain of dependent load pairs: 10 us for the 256 pixels of the stddev-plane conv at 2 x 16)
#pragma unroll 4
        for (long p = p0 + sub; p < p1; p += 4) {
            const int px = (int)(p % Wb);
            const long q = p / Wb;
            const int py = (int)(q % Hb);
            const int n = (int)(q / Hb);
            int iy, ix;
            if (mode == MODE_S1) { iy = py + ky - (ks >> 1); ix = px + kx - (ks >> 1); }
            else { iy = 2 * py + ky; ix = 2 * px + kx; }
            const bool ok = iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;
            const float xv = DT<T>::ld(x + (((long)n * Hi + (ok ? iy : 0)) * Wi + (ok ? ix : 0)) * IC + ic);
            const float gv = DT<T>::ld(gy + p * OC + oc);
            acc += ok ? xv * gv : 0.f;
        }
    }
    red[tid] = acc;
    __syncthreads();
    if (sub == 0 && e < E) part[(long)slice * E + e] = red[tid] + red[tid + 64] + red[tid + 128] + red[tid + 192];
}

// ------------------------------------------------------------- thin 1x1 weight gradient
// The colour convs have 2 channels on one side: gw = sum_p wide[p][C] (x) thin[p][2].  One streaming pass:
// C/4 lanes per pixel hold 4 wide channels x 2 thin values = 8 accumulators, pixel lanes reduce through LDS.
// WIDE_IS_X: x is the wide tensor (gw[c][j], Cout = 2), else gy is (gw[j][c], Cin = 2).
template <typename T, bool WIDE_IS_X>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const T* __restrict__ wide, const T* __restrict__ thin, float* __restrict__ part,
                                                         int C, long npix, long pps) {
    __shared__ float red[256 * 8];
    const int quads = C >> 2;
    const int rpi = 256 / quads;  // pixel lanes per iteration
    const int q = threadIdx.x % quads, r = threadIdx.x / quads;
    const long p0 = (long)blockIdx.x * pps;
    long p1 = p0 + pps;
    if (p1 > npix) p1 = npix;
    float acc[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    if (r < rpi) {
        long p = p0 + r;
        for (; p + 3 * rpi < p1; p += 4 * rpi) {   // four pixels per trip: their loads are in flight together
            float w4[4][4], t[4][2];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ld4(wide + (p + k * rpi) * C + q * 4, w4[k]);
                t[k][0] = DT<T>::ld(thin + (p + k * rpi) * 2);
                t[k][1] = DT<T>::ld(thin + (p + k * rpi) * 2 + 1);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[e][0] += w4[k][e] * t[k][0]; acc[e][1] += w4[k][e] * t[k][1]; }
        }
        for (; p < p1; p += rpi) {
            float w4[4];
            ld4(wide + p * C + q * 4, w4);
            const float t0 = DT<T>::ld(thin + p * 2), t1 = DT<T>::ld(thin + p * 2 + 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[e][0] += w4[e] * t0; acc[e][1] += w4[e] * t1; }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[threadIdx.x * 8 + e * 2] = acc[e][0]; red[threadIdx.x * 8 + e * 2 + 1] = acc[e][1]; }
    __syncthreads();
    if (threadIdx.x < quads) {
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < rpi; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += red[(k * quads + threadIdx.x) * 8 + e];
        float* out = part + (long)blockIdx.x * C * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = threadIdx.x * 4 + e;
            if (WIDE_IS_X) { out[c * 2] = s[e * 2]; out[c * 2 + 1] = s[e * 2 + 1]; }
            else { out[c] = s[e * 2]; out[C + c] = s[e * 2 + 1]; }
        }
    }
}

static bool thin_wgrad_ok(int ks, int ci, int co) {
    if (ks != 1) return false;
    const int c = ci == 2 ? co : (co == 2 ? ci : 0);
    return c >= 4 && c <= 1024 && (c & 3) == 0 && 256 % (c >> 2) == 0 && !(ci == 2 && co == 2);
}
static void thin_wgrad_geometry(long npix, int c, long* nslices, long* pps) {
    const long rpi = 256 / (c >> 2);
    long ns = (npix + rpi * 32 - 1) / (rpi * 32);
    if (ns > 1024) ns = 1024;
    if (ns < 1) ns = 1;
    *pps = (npix + ns - 1) / ns;
    *nslices = (npix + *pps - 1) / *pps;
}

static void wgrad_direct_geometry(long npix, long* nslices, long* pps) {
    long ns = (npix + 63) / 64;   // a thread walks its slice serially (index arithmetic + two dependent loads per pixel): keep it short
    if (ns > 1024) ns = 1024;
    if (ns < 1) ns = 1;
    *pps = (npix + ns - 1) / ns;
    *nslices = (npix + *pps - 1) / *pps;
}

static size_t wgrad_direct_bytes(int ks, int N, int Hb, int Wb, int IC, int OC) {
    long ns, pps;
    wgrad_direct_geometry((long)N * Hb * Wb, &ns, &pps);
    if (thin_wgrad_ok(ks, IC, OC)) {
        long tns, tpps;
        thin_wgrad_geometry((long)N * Hb * Wb, IC == 2 ? OC : IC, &tns, &tpps);
        if (tns > ns) ns = tns;
    }
    return align256(((size_t)ns * ks * ks * IC * OC + wgrad_reduce_extra(ns, (long)ks * ks * IC * OC)) * 4);
}

static int run_wgrad_direct(int mode, int ks, const void* x, const void* gy, float* gw, int N, int Hi, int Wi, int IC,
                            int OC, int Hb, int Wb, float alpha, int transpose, int accumulate, int dtype, void* ws, size_t ws_bytes,
                            hipStream_t st, GsWgradReduce* defer = nullptr) {
    long ns, pps;
    const long npix = (long)N * Hb * Wb;
    const long E = (long)ks * ks * IC * OC;
    if (thin_wgrad_ok(ks, IC, OC) && mode == MODE_S1) {
        const int C = IC == 2 ? OC : IC;
        thin_wgrad_geometry(npix, C, &ns, &pps);
        if (ws_bytes < ((size_t)ns * E + wgrad_reduce_extra(ns, E)) * 4) return fail(GS_ERR_WORKSPACE, "conv wgrad thin: workspace too small (%zu)", ws_bytes);
        float* tpart = reinterpret_cast<float*>(ws);
        if (IC == 2) {
            GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((thin_wgrad_kernel<T, false>), dim3((unsigned)ns), dim3(256), 0, st,
                                                        reinterpret_cast<const T*>(gy), reinterpret_cast<const T*>(x), tpart, C, npix, pps));
        } else {
            GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((thin_wgrad_kernel<T, true>), dim3((unsigned)ns), dim3(256), 0, st,
                                                        reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(gy), tpart, C, npix, pps));
        }
        GS_CHECK_LAUNCH();
        wgrad_reduce_launch(tpart, gw, nullptr, (int)ns, 1, IC, OC, alpha, transpose, accumulate, st, defer);
        GS_CHECK_LAUNCH();
        return 0;
    }
    wgrad_direct_geometry(npix, &ns, &pps);
    if (ws_bytes < ((size_t)ns * E + wgrad_reduce_extra(ns, E)) * 4) return fail(GS_ERR_WORKSPACE, "conv wgrad direct: workspace too small (%zu)", ws_bytes);
    float* part = reinterpret_cast<float*>(ws);
    dim3 grid(cdiv(E, 64), (unsigned)ns);
    GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((conv_wgrad_direct_kernel<T>), grid, dim3(256), 0, st,
                                                reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(gy), part, mode,
                                                ks, N, Hi, Wi, IC, OC, Hb, Wb, E, npix, pps));
    GS_CHECK_LAUNCH();
    wgrad_reduce_launch(part, gw, nullptr, (int)ns, ks * ks, IC, OC, alpha, transpose, accumulate, st, defer);
    GS_CHECK_LAUNCH();
    return 0;
}

static int check_conv_args(int n, int h, int w, int ci, int co, int ksize, int stride, int dtype) {
    GS_CHECK_ARG(n > 0 && h > 0 && w > 0 && ci > 0 && co > 0, "conv2d: non-positive dim");
    GS_CHECK_ARG(ksize == 1 || ksize == 3, "conv2d: ksize %d not in {1,3}", ksize);
    GS_CHECK_ARG(stride == 1 || (stride == 2 && ksize == 3), "conv2d: stride %d unsupported with ksize %d", stride, ksize);
    GS_CHECK_ARG(stride == 1 || (h % 2 == 0 && w % 2 == 0), "conv2d: stride 2 needs even h,w (got %d,%d)", h, w);
    GS_CHECK_ARG(dtype == GS_F32 || dtype == GS_BF16, "conv2d: bad dtype %d", dtype);
    return 0;
}

}  // namespace gs

using namespace gs;

extern "C" size_t gs_conv2d_workspace_bytes(int which, int n, int h, int w, int ci, int co, int ksize, int stride, int dtype) {
    const int hb = h / stride, wb = w / stride;
    if (which == GS_CONV_BWD_WEIGHT) {
        const size_t cs = align256(gs_channel_sum_workspace_bytes((int64_t)n * hb * wb, co));  // bias-gradient fallback of gs_conv2d_bwd_weight_bias
        size_t wg;
        if (ksize == 3 && wgrad_mfma_supported(ci, co, dtype)) wg = wgrad_mfma_bytes(stride == 2 ? MODE_S2 : MODE_S1, dtype, n, hb, wb, ci, co);
        else wg = wgrad_direct_bytes(ksize, n, hb, wb, ci, co);
        return wg > cs ? wg : cs;
    }
    return align256((size_t)ksize * ksize * ci * co * 4);
}

// the bias/activation epilogue for the shapes that do not go through the MFMA kernel
extern "C" int gs_bias_act_fwd(const void* x, const float* bias, void* y, int64_t p, int c, int act, int dtype, void* stream);

// y2 (optional): y2 = pixel_norm(act(conv + bias)) as well -- fused into the conv epilogue where the tile owns all channels of a
// pixel, a separate pass otherwise; y (the activation itself) may then be NULL
static int mask_fuse_min_channels() {
    static const int v = [] { const char* e = getenv("GS_MASK_FUSE_MIN_CI"); return e ? atoi(e) : 32; }();   // (env: measurement knob)
    return v;
}

static int conv2d_fwd_impl(const void* x, const float* w_hwio, const float* bias, int act, void* y, int n, int h, int w, int ci, int co, int ksize,
                           int stride, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream, void* y2 = nullptr,
                           float pn_eps = 0.f) {
    if (int e = check_conv_args(n, h, w, ci, co, ksize, stride, dtype)) return e;
    const bool want_bits = (act & GS_ACT_WRITE_BITS) != 0;   // (the caller's y has room for the sign bits behind it: include/gansynth_hip.h)
    act &= ~GS_ACT_WRITE_BITS;
    GS_CHECK_ARG(act == GS_ACT_NONE || act == GS_ACT_LRELU || act == GS_ACT_TANH, "conv2d: bad activation %d", act);
    GS_CHECK_ARG(!want_bits || (act == GS_ACT_LRELU && dtype == GS_BF16 && co % 32 == 0 && y), "conv2d: sign bits go with a bf16 leaky-relu result of 32 k channels");
    GS_CHECK_ARG(y || y2, "conv2d: no output");
    hipStream_t st = as_stream(stream);
    const int hb = h / stride, wb = w / stride;
    const int mode = stride == 2 ? MODE_S2 : MODE_S1;
    if (ksize == 3 && igemm_supported(ci, co, dtype) && act != GS_ACT_TANH)
        return run_igemm(mode, 0, x, w_hwio, y, n, h, w, ci, co, ci, co, hb, wb, alpha, bias, act | (want_bits ? GS_ACT_WRITE_BITS : 0), dtype, w_prepared, ws,
                         ws_bytes, st, nullptr, 0, y2, pn_eps);
    void* z = y ? y : y2;
    bool fused = false;
    bool bits_done = false;
    unsigned* bits_out = (want_bits && !y2) ? reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(y) + (size_t)n * hb * wb * co) : nullptr;
    if (int e = run_direct(mode, ksize, 0, x, w_hwio, z, n, h, w, ci, co, ci, co, hb, wb, alpha, dtype, w_prepared, ws, ws_bytes, st, bias, act, &fused, nullptr, 0, nullptr,
                           bits_out, &bits_done))
        return e;
    if (!fused && (bias || act != GS_ACT_NONE))
        if (int e = gs_bias_act_fwd(z, bias, z, (int64_t)n * hb * wb, co, act, dtype, stream)) return e;
    if (y2)
        if (int e = gs_pixel_norm_fwd(z, y2, (int64_t)n * hb * wb, co, pn_eps, dtype, stream)) return e;
    if (want_bits && !bits_done) return gs_pack_act_bits(y, (int64_t)n * hb * wb, co, dtype, stream);   // (direct kernels other than the colour block's)
    return 0;
}

extern "C" int gs_conv2d_fwd(const void* x, const float* w_hwio, void* y, int n, int h, int w, int ci, int co, int ksize,
                             int stride, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream) {
    return conv2d_fwd_impl(x, w_hwio, nullptr, GS_ACT_NONE, y, n, h, w, ci, co, ksize, stride, alpha, dtype, w_prepared, ws, ws_bytes, stream);
}

extern "C" int gs_conv2d_fwd_bias_act(const void* x, const float* w_hwio, const float* bias, void* y, int n, int h, int w, int ci, int co,
                                      int ksize, int stride, float alpha, int act, int dtype, int w_prepared, void* ws, size_t ws_bytes,
                                      void* stream) {
    return conv2d_fwd_impl(x, w_hwio, bias, act, y, n, h, w, ci, co, ksize, stride, alpha, dtype, w_prepared, ws, ws_bytes, stream);
}

extern "C" int gs_act_bwd(const void* g, const void* y, void* gx, int64_t numel, int act, int dtype, void* stream);

// y = conv2d(x, w) * mask_act'(.) through `mask` (an activation OUTPUT of y's shape): the second-order pass of the R1 penalty runs
// the discriminator's convs forward on cotangents and multiplies each result by the derivative of the activation that follows
// the conv; in the epilogue on the MFMA path (see gs_conv2d_bwd_data_mask), in place after the conv otherwise
extern "C" int gs_conv2d_fwd_mask(const void* x, const float* w_hwio, const void* mask, int mask_act, void* y, int n, int h, int w, int ci, int co,
                                  int ksize, int stride, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream) {
    if (int e = check_conv_args(n, h, w, ci, co, ksize, stride, dtype)) return e;
    GS_CHECK_ARG(mask == nullptr || mask_act == GS_ACT_LRELU || mask_act == GS_ACT_TANH || mask_act == GS_ACT_LRELU_BITS, "conv2d_fwd_mask: bad activation %d", mask_act);
    hipStream_t st = as_stream(stream);
    const int hb = h / stride, wb = w / stride;
    const int mode = stride == 2 ? MODE_S2 : MODE_S1;
    const bool fused = mask != nullptr && co >= mask_fuse_min_channels() && ksize == 3 && igemm_supported(ci, co, dtype);
    int rc;
    const int bits_act = mask_act;   // (only the MFMA epilogue reads the sign bits; every other path reads the values they stand behind)
    if (mask_act == GS_ACT_LRELU_BITS) mask_act = GS_ACT_LRELU;
    if (ksize == 3 && igemm_supported(ci, co, dtype))
        rc = run_igemm(mode, 0, x, w_hwio, y, n, h, w, ci, co, ci, co, hb, wb, alpha, nullptr, GS_ACT_NONE, dtype, w_prepared, ws, ws_bytes, st, fused ? mask : nullptr, bits_act);
    else {   // (the colour block's streaming kernel applies the mask itself; the other direct kernels leave it to the pass below)
        bool epi = false, mdone = false;
        rc = run_direct(mode, ksize, 0, x, w_hwio, y, n, h, w, ci, co, ci, co, hb, wb, alpha, dtype, w_prepared, ws, ws_bytes, st, nullptr, GS_ACT_NONE, &epi, mask, mask_act, &mdone);
        if (rc || !mask || mdone) return rc;
    }
    if (rc || !mask || fused) return rc;
    return gs_act_bwd(y, mask, y, (int64_t)n * hb * wb * co, mask_act, dtype, stream);
}

extern "C" int gs_conv2d_fwd_bias_act_norm(const void* x, const float* w_hwio, const float* bias, void* z, void* y, int n, int h, int w, int ci, int co,
                                           int ksize, int stride, float alpha, int act, float eps, int dtype, int w_prepared, void* ws,
                                           size_t ws_bytes, void* stream) {
    GS_CHECK_ARG(y != nullptr, "conv2d_fwd_bias_act_norm: y is required (z is optional)");
    return conv2d_fwd_impl(x, w_hwio, bias, act, z, n, h, w, ci, co, ksize, stride, alpha, dtype, w_prepared, ws, ws_bytes, stream, y, eps);
}

extern "C" int gs_act_bwd(const void* g, const void* y, void* gx, int64_t numel, int act, int dtype, void* stream);

// gx = conv2d_bwd_data(gy, w) * mask_act'(.) through `mask` (the activation OUTPUT that was the conv's input; NULL: plain)
extern "C" int gs_conv2d_bwd_data_mask(const void* gy, const float* w_hwio, const void* mask, int mask_act, void* gx, int n, int h, int w, int ci, int co,
                                       int ksize, int stride, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream) {
    if (int e = check_conv_args(n, h, w, ci, co, ksize, stride, dtype)) return e;
    GS_CHECK_ARG(mask == nullptr || mask_act == GS_ACT_LRELU || mask_act == GS_ACT_TANH || mask_act == GS_ACT_LRELU_BITS, "conv2d_bwd_data_mask: bad activation %d", mask_act);
    const int bits_act = mask_act;   // (see gs_conv2d_fwd_mask)
    if (mask_act == GS_ACT_LRELU_BITS) mask_act = GS_ACT_LRELU;
    const float* bias = nullptr;
    const int act = GS_ACT_NONE;
    hipStream_t st = as_stream(stream);
    const int hb = h / stride, wb = w / stride;
    int rc;
    // In the epilogue the mask costs one more read of gx's size; with the mask vectors fetched ahead of the epilogue arithmetic
    // (conv_igemm.hip) that beats the separate in-place pass on every MFMA-path layer, the HBM-bound 32-channel top included
    // (+1.6 % on the step against fusing from 64 channels up).
    bool fused = mask != nullptr && ci >= mask_fuse_min_channels();
    const void* km = fused ? mask : nullptr;
    if (stride == 1) {  // flipped taps, roles of ci/co swapped
        if (ksize == 3 && igemm_supported(co, ci, dtype))
            rc = run_igemm(MODE_S1, 1, gy, w_hwio, gx, n, h, w, co, ci, ci, co, h, w, alpha, bias, act, dtype, w_prepared, ws, ws_bytes, st, km, bits_act);
        else { rc = run_direct(MODE_S1, ksize, 1, gy, w_hwio, gx, n, h, w, co, ci, ci, co, h, w, alpha, dtype, w_prepared, ws, ws_bytes, st); fused = false; }
    } else if (igemm_supported(co, ci, dtype)) {
        rc = run_igemm(MODE_T2, 2, gy, w_hwio, gx, n, hb, wb, co, ci, ci, co, hb, wb, alpha, bias, act, dtype, w_prepared, ws, ws_bytes, st, km, bits_act);
    } else {
        rc = run_direct(MODE_T2, 3, 2, gy, w_hwio, gx, n, hb, wb, co, ci, ci, co, h, w, alpha, dtype, w_prepared, ws, ws_bytes, st);
        fused = false;
    }
    if (rc || !mask || fused) return rc;
    return gs_act_bwd(gx, mask, gx, (int64_t)n * h * w * ci, mask_act, dtype, stream);   // shapes without the MFMA kernel: in place
}

extern "C" int gs_conv2d_bwd_data(const void* gy, const float* w_hwio, void* gx, int n, int h, int w, int ci, int co,
                                  int ksize, int stride, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream) {
    return gs_conv2d_bwd_data_mask(gy, w_hwio, nullptr, 0, gx, n, h, w, ci, co, ksize, stride, alpha, dtype, w_prepared, ws, ws_bytes, stream);
}

extern "C" size_t gs_channel_sum_workspace_bytes(int64_t p, int c);
extern "C" int gs_channel_sum(const void* g, float* out, int64_t p, int c, int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream);

// the (x, gy) pairs of one launch, as the kernels want them; for the transposed conv the roles of the two sides swap
static WgradSrcs make_srcs(const void* const* xs, const void* const* gys, int nsrc, int n_per, const int* ns, unsigned bias_mask, bool swap, int* total) {
    WgradSrcs s;
    memset(&s, 0, sizeof(s));
    int end = 0;
    for (int i = 0; i < GS_WGRAD_MAX_SRC; ++i) {
        if (i < nsrc) {
            s.x[i] = swap ? gys[i] : xs[i];
            s.gy[i] = swap ? xs[i] : gys[i];
            end += ns ? ns[i] : n_per;
        }
        s.n_end[i] = end;
    }
    s.bias_mask = bias_mask;
    *total = end;
    return s;
}

extern "C" int gs_conv2d_bwd_weight_bias_multi(const void* const* xs, const void* const* gys, const int* ns, int nsrc, unsigned bias_mask, float* gw_hwio,
                                               float* gb, int n, int h, int w, int ci, int co, int ksize, int stride, float alpha, int accumulate,
                                               int dtype, void* ws, size_t ws_bytes, GsWgradReduce* pending, void* stream) {
    if (int e = check_conv_args(n, h, w, ci, co, ksize, stride, dtype)) return e;
    GS_CHECK_ARG(xs && gys && nsrc >= 1 && nsrc <= GS_WGRAD_MAX_SRC, "conv2d_bwd_weight_bias_multi: %d sources (1..%d)", nsrc, GS_WGRAD_MAX_SRC);
    for (int i = 0; i < nsrc; ++i) GS_CHECK_ARG(xs[i] && gys[i], "conv2d_bwd_weight_bias_multi: null source %d", i);
    hipStream_t st = as_stream(stream);
    const int hb = h / stride, wb = w / stride;
    const int mode = stride == 2 ? MODE_S2 : MODE_S1;
    const bool mfma = ksize == 3 && wgrad_mfma_supported(ci, co, dtype);
    if (pending) memset(pending, 0, sizeof(*pending));
    if (!gb) bias_mask = 0;
    if (nsrc > 1 && !(mfma && (!gb || wgrad_mfma_has_bias(dtype)))) {
        // shapes without the multi-source kernels: one call per source (the first applies `accumulate`, the rest add)
        for (int i = 0; i < nsrc; ++i) {
            const int rc = gs_conv2d_bwd_weight_bias_multi(xs + i, gys + i, nullptr, 1, 1u, gw_hwio, ((bias_mask >> i) & 1u) ? gb : nullptr, ns ? ns[i] : n, h, w, ci, co,
                                                           ksize, stride, alpha, i == 0 ? accumulate : 1, dtype, ws, ws_bytes, nullptr, stream);
            if (rc) return rc;
        }
        return 0;
    }
    const bool fused_bias = bias_mask && mfma && wgrad_mfma_has_bias(dtype);
    // the channel-sum fallback of the bias gradient reuses ws: such calls cannot leave their partials pending
    GsWgradReduce* defer = (bias_mask && !fused_bias) ? nullptr : pending;
    int rc;
    const int n0 = ns ? ns[0] : n;   // (the single-source paths below)
    if (mfma) {
        int total = 0;
        const WgradSrcs srcs = make_srcs(xs, gys, nsrc, n, ns, bias_mask, false, &total);
        rc = run_wgrad_mfma(mode, srcs, nsrc, gw_hwio, fused_bias ? gb : nullptr, total, h, w, ci, co, hb, wb, alpha, 0, accumulate, dtype, ws, ws_bytes, st, defer);
    } else {
        rc = run_wgrad_direct(mode, ksize, xs[0], gys[0], gw_hwio, n0, h, w, ci, co, hb, wb, alpha, 0, accumulate, dtype, ws, ws_bytes, st, defer);
    }
    if (rc || !bias_mask || fused_bias) return rc;
    // shapes without the fused path: the plain channel sum (stream-ordered after the kernels above, same workspace)
    return gs_channel_sum(gys[0], gb, (int64_t)n0 * hb * wb, co, accumulate, dtype, ws, ws_bytes, stream);
}

extern "C" int gs_conv2d_bwd_weight_bias_partial(const void* x, const void* gy, float* gw_hwio, float* gb, int n, int h, int w, int ci, int co,
                                                 int ksize, int stride, float alpha, int accumulate, int dtype, void* ws, size_t ws_bytes,
                                                 GsWgradReduce* pending, void* stream) {
    return gs_conv2d_bwd_weight_bias_multi(&x, &gy, nullptr, 1, 1u, gw_hwio, gb, n, h, w, ci, co, ksize, stride, alpha, accumulate, dtype, ws, ws_bytes, pending, stream);
}

extern "C" int gs_conv2d_bwd_weight_bias(const void* x, const void* gy, float* gw_hwio, float* gb, int n, int h, int w, int ci, int co,
                                         int ksize, int stride, float alpha, int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream) {
    return gs_conv2d_bwd_weight_bias_partial(x, gy, gw_hwio, gb, n, h, w, ci, co, ksize, stride, alpha, accumulate, dtype, ws, ws_bytes, nullptr, stream);
}

// many pending slice reductions in a handful of launches: up to GS_REDUCE_BATCH entries per launch, a new launch whenever an
// entry adds into a gradient that the current launch already touches (list order = summation order)
extern "C" int gs_wgrad_reduce_batch(const GsWgradReduce* pending, int n, void* stream) {
    GS_CHECK_ARG(n >= 0 && (n == 0 || pending), "wgrad_reduce_batch: bad args");
    hipStream_t st = as_stream(stream);
    ReduceBatch b;
    int cnt = 0;
    long gx = 0;   // blocks of the entry that needs most (64 element quads per block with 4 slice lanes, 16 with 16: see the kernel)
    auto flush = [&]() {
        if (cnt == 0) return;
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)gx, (unsigned)cnt), dim3(256), 0, st, b);
        cnt = 0;
        gx = 0;
    };
    for (int i = 0; i < n; ++i) {
        const GsWgradReduce& d = pending[i];
        if (d.nslices <= 0) continue;   // nothing pending for this call
        GS_CHECK_ARG(d.partials && d.gw && d.taps > 0 && d.ic > 0 && d.oc > 0 && (((long)d.taps * d.ic * d.oc) & 3) == 0 && (!d.gb || (d.oc & 3) == 0),
                     "wgrad_reduce_batch: entry %d is not a pending reduction", i);
        bool clash = cnt == GS_REDUCE_BATCH;
        for (int j = 0; j < cnt && !clash; ++j) clash = b.e[j].gw == d.gw || (d.gb && b.e[j].gb == d.gb);
        if (clash) flush();
        b.e[cnt++] = d;
        const long pstride = (long)d.taps * d.ic * d.oc + (d.gb ? d.oc : 0);
        const long epb = d.nslices > 32 ? 16 : 64;
        const long blocks = (pstride / 4 + epb - 1) / epb;
        if (blocks > gx) gx = blocks;
    }
    flush();
    GS_CHECK_LAUNCH();
    return 0;
}

// ---- all weight gradients of a backward pass in one call (gs_conv_wgrad_jobs): the layers the 64 x 64-tile bf16 kernel takes are
// grouped by its instantiation (conv mode, tile width) and each group runs as ONE stream-K launch + one fold (conv_shared.h); every
// other layer goes through the per-layer entry points above, its slice reduction left pending, and one gs_wgrad_reduce_batch folds
// those at the end.  Workspace: the largest group (groups run one after the other on the stream) + the sum of the per-layer needs.
namespace gs {
struct JobPlan {
    std::vector<std::pair<int, SkGroup>> groups;   // (mode * 64 + tile width, group), in launch order
    std::vector<GsWgradJob> single;               // jobs on the per-layer path (one source each where the layer has no multi-source kernel)
    std::vector<size_t> single_off;               // their workspace offsets
    size_t group_bytes = 0, total_bytes = 0;
};
static int job_check(const GsWgradJob& jb, int idx) {
    GS_CHECK_ARG(jb.nsrc >= 1 && jb.nsrc <= GS_WGRAD_MAX_SRC && jb.gw, "conv_wgrad_jobs: job %d has %d sources (1..%d) / no output", idx, jb.nsrc, GS_WGRAD_MAX_SRC);
    for (int i = 0; i < jb.nsrc; ++i) GS_CHECK_ARG(jb.x[i] && jb.gy[i] && jb.n[i] > 0, "conv_wgrad_jobs: job %d, null or empty source %d", idx, i);
    GS_CHECK_ARG(!jb.transposed || (jb.ksize == 3 && jb.stride == 2 && !jb.gb), "conv_wgrad_jobs: job %d: the transposed conv is 3x3 stride 2 without bias", idx);
    GS_CHECK_ARG(jb.gw_ci_stride == 0 || (jb.gw_ci_stride >= jb.ci && !jb.transposed), "conv_wgrad_jobs: job %d: bad gw_ci_stride %d", idx, jb.gw_ci_stride);
    if (jb.transposed) return check_conv_args(jb.n[0], 2 * jb.h, 2 * jb.w, jb.co, jb.ci, 3, 2, jb.dtype);
    return check_conv_args(jb.n[0], jb.h, jb.w, jb.ci, jb.co, jb.ksize, jb.stride, jb.dtype);
}
static int job_total_images(const GsWgradJob& jb) {
    int t = 0;
    for (int i = 0; i < jb.nsrc; ++i) t += jb.n[i];
    return t;
}
static int plan_jobs(const GsWgradJob* jobs, int njobs, JobPlan& plan) {
    static const bool no_sk = getenv("GS_NO_WGRAD_GROUPS") != nullptr;   // measurement knob: everything on the per-layer path
    std::vector<int> open_idx(256, -1);   // key -> index of the group still accepting jobs
    for (int i = 0; i < njobs; ++i) {
        const GsWgradJob& jb = jobs[i];
        if (int e = job_check(jb, i)) return e;
        // kernel-role geometry: a transposed conv's gradient is the stride-2 one with the two sides swapped, stored transposed
        const int mode = jb.transposed ? MODE_S2 : (jb.stride == 2 ? MODE_S2 : MODE_S1);
        const int IC = jb.transposed ? jb.co : jb.ci, OC = jb.transposed ? jb.ci : jb.co;
        const int Hi = jb.transposed ? 2 * jb.h : jb.h, Wi = jb.transposed ? 2 * jb.w : jb.w;
        const int Hb = jb.transposed ? jb.h : jb.h / jb.stride, Wb = jb.transposed ? jb.w : jb.w / jb.stride;
        const int total = job_total_images(jb);
        if (!no_sk && jb.ksize == 3 && wgrad_sk_supported(mode, jb.dtype, IC, OC)) {
            const int tw = wgrad_sk_tile_width(Wb);
            const int key = mode * 64 + tw;
            int gi = open_idx[key];
            if (gi >= 0) {   // a full group, or one that already adds into this gradient, is closed (launch order = summation order)
                const SkGroup& og = plan.groups[gi].second;
                bool close = og.njobs == GS_SK_MAX_JOBS;
                for (int j = 0; j < og.njobs && !close; ++j) close = og.job[j].gw == jb.gw || (jb.gb && og.job[j].gb == jb.gb);
                if (close) gi = -1;
            }
            if (gi < 0) {
                SkGroup ng;
                memset(&ng, 0, sizeof(ng));
                plan.groups.push_back(std::make_pair(key, ng));
                gi = (int)plan.groups.size() - 1;
                open_idx[key] = gi;
            }
            SkGroup& g = plan.groups[gi].second;
            SkJob& q = g.job[g.njobs++];
            int tot = 0;
            q.srcs = make_srcs(jb.x, jb.gy, jb.nsrc, 0, jb.n, jb.gb ? jb.bias_mask : 0u, jb.transposed != 0, &tot);
            q.gw = jb.gw; q.gb = jb.gb; q.alpha = jb.alpha; q.transpose = jb.transposed ? 1 : 0; q.accumulate = jb.accumulate;
            q.Hi = Hi; q.Wi = Wi; q.IC = IC; q.OC = OC; q.Hb = Hb; q.Wb = Wb;
            q.ICld = jb.gw_ci_stride > 0 ? jb.gw_ci_stride : IC;
            wgrad_sk_job_geometry(mode, tw, total, q);
        } else {
            // layers without a multi-source kernel (direct / thin kernels, fp32 bias sums): one single-source job per pair, each with its own
            // partials so that every slice reduction can stay pending
            const bool mfma = jb.ksize == 3 && (jb.transposed ? wgrad_mfma_supported(jb.co, jb.ci, jb.dtype) : wgrad_mfma_supported(jb.ci, jb.co, jb.dtype));
            const bool multi = mfma && (!jb.gb || wgrad_mfma_has_bias(jb.dtype));
            if (jb.nsrc == 1 || multi) {
                plan.single.push_back(jb);
            } else {
                for (int sidx = 0; sidx < jb.nsrc; ++sidx) {
                    GsWgradJob one = jb;
                    one.nsrc = 1;
                    one.x[0] = jb.x[sidx]; one.gy[0] = jb.gy[sidx]; one.n[0] = jb.n[sidx];
                    one.bias_mask = (jb.bias_mask >> sidx) & 1u;
                    if (!one.bias_mask) one.gb = nullptr;
                    if (sidx > 0) one.accumulate = 1;
                    plan.single.push_back(one);
                }
            }
        }
    }
    for (auto& kg : plan.groups) {
        wgrad_sk_plan(kg.first / 64, kg.second);
        const size_t b = wgrad_sk_bytes(kg.second);
        if (b > plan.group_bytes) plan.group_bytes = b;
    }
    size_t off = plan.group_bytes;
    for (const GsWgradJob& jb : plan.single) {
        const int total = job_total_images(jb);
        plan.single_off.push_back(off);
        off += jb.transposed ? gs_conv2d_transpose_s2_workspace_bytes(GS_CONV_BWD_WEIGHT, total, jb.h, jb.w, jb.ci, jb.co, jb.dtype)
                             : gs_conv2d_workspace_bytes(GS_CONV_BWD_WEIGHT, total, jb.h, jb.w, jb.ci, jb.co, jb.ksize, jb.stride, jb.dtype);
    }
    plan.total_bytes = off;
    return 0;
}
}  // namespace gs

extern "C" size_t gs_conv_wgrad_jobs_workspace_bytes(const GsWgradJob* jobs, int njobs) {
    if (!jobs || njobs <= 0) return 0;
    JobPlan plan;
    if (plan_jobs(jobs, njobs, plan)) return 0;
    return plan.total_bytes;
}

extern "C" int gs_conv_wgrad_jobs(const GsWgradJob* jobs, int njobs, void* ws, size_t ws_bytes, void* stream) {
    GS_CHECK_ARG(njobs >= 0 && (njobs == 0 || jobs), "conv_wgrad_jobs: bad args");
    if (njobs == 0) return 0;
    JobPlan plan;
    if (int e = plan_jobs(jobs, njobs, plan)) return e;
    if (ws_bytes < plan.total_bytes) return fail(GS_ERR_WORKSPACE, "conv_wgrad_jobs: workspace %zu < %zu", ws_bytes, plan.total_bytes);
    hipStream_t st = as_stream(stream);
    for (auto& kg : plan.groups)
        if (int e = run_wgrad_sk(kg.first / 64, kg.first % 64, kg.second, ws, plan.group_bytes, st)) return e;
    std::vector<GsWgradReduce> pend;
    for (size_t k = 0; k < plan.single.size(); ++k) {
        const GsWgradJob& jb = plan.single[k];
        unsigned char* jws = reinterpret_cast<unsigned char*>(ws) + plan.single_off[k];
        const size_t jbytes = (k + 1 < plan.single.size() ? plan.single_off[k + 1] : plan.total_bytes) - plan.single_off[k];
        GsWgradReduce d;
        memset(&d, 0, sizeof(d));
        int rc;
        if (jb.transposed)
            rc = gs_conv2d_transpose_s2_bwd_weight_multi(jb.x, jb.gy, jb.n, jb.nsrc, jb.gw, jb.n[0], jb.h, jb.w, jb.ci, jb.co, jb.alpha, jb.accumulate, jb.dtype, jws, jbytes, &d, stream);
        else
            rc = gs_conv2d_bwd_weight_bias_multi(jb.x, jb.gy, jb.n, jb.nsrc, jb.bias_mask, jb.gw, jb.gb, jb.n[0], jb.h, jb.w, jb.ci, jb.co, jb.ksize, jb.stride, jb.alpha,
                                                 jb.accumulate, jb.dtype, jws, jbytes, &d, stream);
        if (rc) return rc;
        if (jb.gw_ci_stride > jb.ci) {   // a channel-slice target: only the pending (batched) fold knows the row stride
            if (d.nslices <= 0 || d.transpose) return fail(GS_ERR_UNSUPPORTED, "conv_wgrad_jobs: a %d -> %d layer cannot add into a channel slice", jb.ci, jb.co);
            d.ic_ld = jb.gw_ci_stride;
        }
        if (d.nslices > 0) pend.push_back(d);
    }
    if (!pend.empty()) return gs_wgrad_reduce_batch(pend.data(), (int)pend.size(), stream);
    return 0;
}

extern "C" int gs_conv2d_bwd_weight(const void* x, const void* gy, float* gw_hwio, int n, int h, int w, int ci, int co,
                                    int ksize, int stride, float alpha, int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream) {
    return gs_conv2d_bwd_weight_bias(x, gy, gw_hwio, nullptr, n, h, w, ci, co, ksize, stride, alpha, accumulate, dtype, ws, ws_bytes, stream);
}

// ---- conv2d_transpose 3x3 stride 2: re-labelings of the stride-2 maps (see include/gansynth_hip.h)
extern "C" size_t gs_conv2d_transpose_s2_workspace_bytes(int which, int n, int h, int w, int ci, int co, int dtype) {
    if (which == GS_CONV_BWD_WEIGHT) {
        if (wgrad_mfma_supported(co, ci, dtype)) return wgrad_mfma_bytes(MODE_S2, dtype, n, h, w, co, ci);
        return wgrad_direct_bytes(3, n, h, w, co, ci);
    }
    return align256((size_t)9 * ci * co * 4);
}

static int conv2d_transpose_fwd_impl(const void* x, const float* w_hwio, const float* bias, int act, void* y, int n, int h, int w, int ci,
                                     int co, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream, void* y2 = nullptr,
                                     float pn_eps = 0.f) {
    if (int e = check_conv_args(n, 2 * h, 2 * w, co, ci, 3, 2, dtype)) return e;
    GS_CHECK_ARG(y || y2, "conv2d_transpose: no output");
    hipStream_t st = as_stream(stream);
    // out[2i+k][co] += x[i][ci] * w[k][ci][co]: kernel roles ICk = ci, OCk = co, Wp[t][co][ci] (variant 0)
    if (igemm_supported(ci, co, dtype) && act != GS_ACT_TANH)
        return run_igemm(MODE_T2, 0, x, w_hwio, y, n, h, w, ci, co, ci, co, h, w, alpha, bias, act, dtype, w_prepared, ws, ws_bytes, st, nullptr, 0, y2, pn_eps);
    void* z = y ? y : y2;
    if (int e = run_direct(MODE_T2, 3, 0, x, w_hwio, z, n, h, w, ci, co, ci, co, 2 * h, 2 * w, alpha, dtype, w_prepared, ws, ws_bytes, st)) return e;
    if (bias || act != GS_ACT_NONE)
        if (int e = gs_bias_act_fwd(z, bias, z, (int64_t)n * 4 * h * w, co, act, dtype, stream)) return e;
    if (y2) return gs_pixel_norm_fwd(z, y2, (int64_t)n * 4 * h * w, co, pn_eps, dtype, stream);
    return 0;
}

extern "C" int gs_conv2d_transpose_s2_fwd(const void* x, const float* w_hwio, void* y, int n, int h, int w, int ci, int co,
                                          float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream) {
    return conv2d_transpose_fwd_impl(x, w_hwio, nullptr, GS_ACT_NONE, y, n, h, w, ci, co, alpha, dtype, w_prepared, ws, ws_bytes, stream);
}

extern "C" int gs_conv2d_transpose_s2_fwd_bias_act_norm(const void* x, const float* w_hwio, const float* bias, void* z, void* y, int n, int h, int w,
                                                        int ci, int co, float alpha, int act, float eps, int dtype, int w_prepared, void* ws,
                                                        size_t ws_bytes, void* stream) {
    GS_CHECK_ARG(y != nullptr, "conv2d_transpose_s2_fwd_bias_act_norm: y is required (z is optional)");
    return conv2d_transpose_fwd_impl(x, w_hwio, bias, act, z, n, h, w, ci, co, alpha, dtype, w_prepared, ws, ws_bytes, stream, y, eps);
}

extern "C" int gs_conv2d_transpose_s2_fwd_bias_act(const void* x, const float* w_hwio, const float* bias, void* y, int n, int h, int w, int ci,
                                                   int co, float alpha, int act, int dtype, int w_prepared, void* ws, size_t ws_bytes,
                                                   void* stream) {
    return conv2d_transpose_fwd_impl(x, w_hwio, bias, act, y, n, h, w, ci, co, alpha, dtype, w_prepared, ws, ws_bytes, stream);
}

extern "C" int gs_conv2d_transpose_s2_bwd_data(const void* gy, const float* w_hwio, void* gx, int n, int h, int w, int ci,
                                               int co, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream) {
    if (int e = check_conv_args(n, 2 * h, 2 * w, co, ci, 3, 2, dtype)) return e;
    const float* bias = nullptr;
    const int act = GS_ACT_NONE;
    hipStream_t st = as_stream(stream);
    // gx[i][ci] = sum gy[2i+k][co] * w[k][ci][co]: stride-2 conv, roles ICk = co, OCk = ci, Wp[t][ci][co] (variant 2)
    if (igemm_supported(co, ci, dtype))
        return run_igemm(MODE_S2, 2, gy, w_hwio, gx, n, 2 * h, 2 * w, co, ci, ci, co, h, w, alpha, bias, act, dtype, w_prepared, ws, ws_bytes, st);
    return run_direct(MODE_S2, 3, 2, gy, w_hwio, gx, n, 2 * h, 2 * w, co, ci, ci, co, h, w, alpha, dtype, w_prepared, ws, ws_bytes, st);
}

// Second-order pass (the mode-seeking term differentiates the generator's backward, models.py:60): the forward conv applied to a cotangent gives
// t = the gradient w.r.t. u = act'(z) pixel_norm_bwd(g, z), the first-order backward of the block whose activation z and incoming gradient g
// are given.  Both gradients of that node in the conv's epilogue (h = t act'(z)):
//   out_g = pixel_norm_bwd(h, z)                 (w.r.t. g)          out_z = d<h, pixel_norm_bwd(g, z)>/dz      (w.r.t. z)
// where a tile owns all channels of a pixel; otherwise the conv into out_g and gs_pixel_norm_bwd_bwd_fused (out_g in place).
extern "C" int gs_conv2d_fwd_pnbwdbwd(const void* x, const float* w_hwio, const void* g, const void* z, int act, float eps, void* out_g, void* out_z, int n, int h,
                                      int w, int ci, int co, int ksize, int stride, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream) {
    if (int e = check_conv_args(n, h, w, ci, co, ksize, stride, dtype)) return e;
    GS_CHECK_ARG(g && z && out_g && out_z && (act == GS_ACT_NONE || act == GS_ACT_LRELU), "conv2d_fwd_pnbwdbwd: g, z and both outputs are required, activation none or leaky relu (got %d)", act);
    hipStream_t st = as_stream(stream);
    const int hb = h / stride, wb = w / stride;
    if (ksize == 3 && igemm_supported(ci, co, dtype))
        return run_igemm(stride == 2 ? MODE_S2 : MODE_S1, 0, x, w_hwio, out_g, n, h, w, ci, co, ci, co, hb, wb, alpha, nullptr, GS_ACT_NONE, dtype, w_prepared, ws, ws_bytes, st,
                         z, act, out_z, eps, g, 2);
    if (int e = gs_conv2d_fwd(x, w_hwio, out_g, n, h, w, ci, co, ksize, stride, alpha, dtype, w_prepared, ws, ws_bytes, stream)) return e;
    return gs_pixel_norm_bwd_bwd_fused(out_g, g, z, out_z, out_g, (int64_t)n * hb * wb, co, eps, act, dtype, stream);
}
extern "C" int gs_conv2d_transpose_s2_fwd_pnbwdbwd(const void* x, const float* w_hwio, const void* g, const void* z, int act, float eps, void* out_g, void* out_z, int n,
                                                   int h, int w, int ci, int co, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream) {
    if (int e = check_conv_args(n, 2 * h, 2 * w, co, ci, 3, 2, dtype)) return e;
    GS_CHECK_ARG(g && z && out_g && out_z && (act == GS_ACT_NONE || act == GS_ACT_LRELU), "conv2d_transpose_s2_fwd_pnbwdbwd: g, z and both outputs are required, activation none or leaky relu (got %d)", act);
    hipStream_t st = as_stream(stream);
    if (igemm_supported(ci, co, dtype))
        return run_igemm(MODE_T2, 0, x, w_hwio, out_g, n, h, w, ci, co, ci, co, h, w, alpha, nullptr, GS_ACT_NONE, dtype, w_prepared, ws, ws_bytes, st, z, act, out_z, eps, g, 2);
    if (int e = gs_conv2d_transpose_s2_fwd(x, w_hwio, out_g, n, h, w, ci, co, alpha, dtype, w_prepared, ws, ws_bytes, stream)) return e;
    return gs_pixel_norm_bwd_bwd_fused(out_g, g, z, out_z, out_g, (int64_t)n * 4 * h * w, co, eps, act, dtype, stream);
}
// 1 when that call runs as one launch for the shape (n, h, w: the conv's input side)
extern "C" int gs_conv2d_fwd_pnbwdbwd_is_fused(int n, int h, int w, int ci, int co, int ksize, int stride, int transposed, int dtype) {
    if (ksize != 3 || !igemm_supported(ci, co, dtype)) return 0;
    if (transposed) return stride == 2 && igemm_normbwd_fused(MODE_T2, n, h, w, ci, co, dtype, 2) ? 1 : 0;
    return stride == 1 && igemm_normbwd_fused(MODE_S1, n, h, w, ci, co, dtype, 2) ? 1 : 0;
}

// Data gradient of a conv whose INPUT was y = pixel_norm(z), z = act(...) the previous block's activation (networks.py:41-93: every
// generator block ends conv -> leaky_relu -> pixel_norm), continued through that norm and activation in the conv's epilogue:
//   gx = (pixel_norm_bwd(B^T(gy, w), z) + addend) * act'(z)       (addend: optional second gradient into z, same shape)
// i.e. the gradient w.r.t. the previous block's pre-activation in ONE pass.  The epilogue form exists where a tile owns every channel of a
// pixel (the 32- / 64-channel layers -- where the bytes are); other shapes run the plain data gradient and gs_pixel_norm_bwd_fused in place.
extern "C" int gs_conv2d_bwd_data_pnbwd(const void* gy, const float* w_hwio, const void* z, const void* addend, int act, float eps, void* gx, int n, int h, int w,
                                        int ci, int co, int ksize, int stride, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream) {
    if (int e = check_conv_args(n, h, w, ci, co, ksize, stride, dtype)) return e;
    GS_CHECK_ARG(z && gx && (act == GS_ACT_NONE || act == GS_ACT_LRELU), "conv2d_bwd_data_pnbwd: z is required, activation none or leaky relu (got %d)", act);
    hipStream_t st = as_stream(stream);
    if (stride == 1 && ksize == 3 && igemm_supported(co, ci, dtype))
        return run_igemm(MODE_S1, 1, gy, w_hwio, gx, n, h, w, co, ci, ci, co, h, w, alpha, nullptr, GS_ACT_NONE, dtype, w_prepared, ws, ws_bytes, st, z, act, nullptr, eps, addend, 1);
    if (stride == 1 && thin_expand_pnbwd_ok(ksize, co, ci, dtype)) {   // the colour block (few -> many channels as a gradient): streaming, one pass
        const long total = (long)ci * co;
        if (ws_bytes < (size_t)total * 4) return fail(GS_ERR_WORKSPACE, "conv2d_bwd_data_pnbwd: workspace %zu < %zu", ws_bytes, (size_t)total * 4);
        float* wp = reinterpret_cast<float*>(ws);
        if (!w_prepared) {
            hipLaunchKernelGGL((weight_prep_kernel<float>), dim3(cdiv(total, 256)), dim3(256), 0, st, w_hwio, wp, 1, ci, co, 1);
            GS_CHECK_LAUNCH();
        }
        return run_thin_expand_pnbwd(gy, wp, z, addend, gx, (long)n * h * w, co, ci, alpha, eps, act, dtype, st);
    }
    if (int e = gs_conv2d_bwd_data_mask(gy, w_hwio, nullptr, 0, gx, n, h, w, ci, co, ksize, stride, alpha, dtype, w_prepared, ws, ws_bytes, stream)) return e;
    return gs_pixel_norm_bwd_fused(gx, z, addend, gx, (int64_t)n * h * w, ci, eps, GS_ACT_NONE, act, dtype, stream);
}
// 1 when the call above runs as ONE launch for this shape (the epilogue form), 0 when it is the conv + the norm's backward in place
extern "C" int gs_conv2d_bwd_data_pnbwd_is_fused(int n, int h, int w, int ci, int co, int ksize, int stride, int transposed, int dtype) {
    if (ksize == 1) return !transposed && stride == 1 && thin_expand_pnbwd_ok(1, co, ci, dtype) ? 1 : 0;
    if (ksize != 3) return 0;
    if (transposed) return stride == 2 && igemm_supported(co, ci, dtype) && igemm_normbwd_fused(MODE_S2, n, h, w, co, ci, dtype) ? 1 : 0;
    return stride == 1 && igemm_supported(co, ci, dtype) && igemm_normbwd_fused(MODE_S1, n, h, w, co, ci, dtype) ? 1 : 0;
}
extern "C" int gs_conv2d_transpose_s2_bwd_data_pnbwd(const void* gy, const float* w_hwio, const void* z, const void* addend, int act, float eps, void* gx, int n,
                                                     int h, int w, int ci, int co, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream) {
    if (int e = check_conv_args(n, 2 * h, 2 * w, co, ci, 3, 2, dtype)) return e;
    GS_CHECK_ARG(z && gx && (act == GS_ACT_NONE || act == GS_ACT_LRELU), "conv2d_transpose_s2_bwd_data_pnbwd: z is required, activation none or leaky relu (got %d)", act);
    hipStream_t st = as_stream(stream);
    if (igemm_supported(co, ci, dtype))
        return run_igemm(MODE_S2, 2, gy, w_hwio, gx, n, 2 * h, 2 * w, co, ci, ci, co, h, w, alpha, nullptr, GS_ACT_NONE, dtype, w_prepared, ws, ws_bytes, st, z, act, nullptr, eps, addend, 1);
    if (int e = gs_conv2d_transpose_s2_bwd_data(gy, w_hwio, gx, n, h, w, ci, co, alpha, dtype, w_prepared, ws, ws_bytes, stream)) return e;
    return gs_pixel_norm_bwd_fused(gx, z, addend, gx, (int64_t)n * h * w, ci, eps, GS_ACT_NONE, act, dtype, stream);
}

extern "C" int gs_conv2d_transpose_s2_bwd_weight_multi(const void* const* xs, const void* const* gys, const int* ns, int nsrc, float* gw_hwio, int n, int h,
                                                       int w, int ci, int co, float alpha, int accumulate, int dtype, void* ws, size_t ws_bytes,
                                                       GsWgradReduce* pending, void* stream) {
    if (int e = check_conv_args(n, 2 * h, 2 * w, co, ci, 3, 2, dtype)) return e;
    GS_CHECK_ARG(xs && gys && nsrc >= 1 && nsrc <= GS_WGRAD_MAX_SRC, "conv2d_transpose_s2_bwd_weight_multi: %d sources (1..%d)", nsrc, GS_WGRAD_MAX_SRC);
    for (int i = 0; i < nsrc; ++i) GS_CHECK_ARG(xs[i] && gys[i], "conv2d_transpose_s2_bwd_weight_multi: null source %d", i);
    hipStream_t st = as_stream(stream);
    if (pending) memset(pending, 0, sizeof(*pending));
    // gw[k][ci][co] = sum x[i][ci] * gy[2i+k][co]: stride-2 wgrad with (input side = gy, output side = x), transposed
    if (wgrad_mfma_supported(co, ci, dtype)) {
        int total = 0;
        const WgradSrcs srcs = make_srcs(xs, gys, nsrc, n, ns, 0u, true, &total);
        return run_wgrad_mfma(MODE_S2, srcs, nsrc, gw_hwio, nullptr, total, 2 * h, 2 * w, co, ci, h, w, alpha, 1, accumulate, dtype, ws, ws_bytes, st, pending);
    }
    for (int i = 0; i < nsrc; ++i) {   // (direct kernels: one call per source)
        const int rc = run_wgrad_direct(MODE_S2, 3, gys[i], xs[i], gw_hwio, ns ? ns[i] : n, 2 * h, 2 * w, co, ci, h, w, alpha, 1, i == 0 ? accumulate : 1, dtype, ws, ws_bytes, st,
                                        nsrc == 1 ? pending : nullptr);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int gs_conv2d_transpose_s2_bwd_weight_partial(const void* x, const void* gy, float* gw_hwio, int n, int h, int w, int ci,
                                                         int co, float alpha, int accumulate, int dtype, void* ws, size_t ws_bytes,
                                                         GsWgradReduce* pending, void* stream) {
    return gs_conv2d_transpose_s2_bwd_weight_multi(&x, &gy, nullptr, 1, gw_hwio, n, h, w, ci, co, alpha, accumulate, dtype, ws, ws_bytes, pending, stream);
}

extern "C" int gs_conv2d_transpose_s2_bwd_weight(const void* x, const void* gy, float* gw_hwio, int n, int h, int w, int ci,
                                                 int co, float alpha, int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream) {
    return gs_conv2d_transpose_s2_bwd_weight_partial(x, gy, gw_hwio, n, h, w, ci, co, alpha, accumulate, dtype, ws, ws_bytes, nullptr, stream);
}
