// waveform -> (log-mel magnitude, instantaneous frequency), wave-per-frame path for 2048-sample frames (gfx950).
// Reference: spectral_ops.py:45-94 (front pad, tf.signal.stft 2048/512 periodic Hann, drop DC, abs / angle, tensordot with the
// mel matrix on magnitude AND phase, log + normalise, unwrap / diff along time).
//
// The path is VALU-bound (1024-point FFT + hypot / atan2 / log per bin: ~1900 vector instructions per frame and lane), and a
// gfx950 wave issues at most one VALU instruction per ~5 cycles whatever its ILP while four waves per SIMD issue four
// (scripts/probe/valu_rate.hip).  So the design target is FOUR WAVES PER SIMD: <= 128 VGPRs and <= 10 KB of LDS per wave.
//
// One WAVEFRONT owns a frame; nothing is shared between the waves of a block except read-only tables, so there is no
// s_barrier after the prologue.  Per frame:
//   A  lane l loads z[n] = x[2n] + i x[2n+1], n = l + 64 j (16 coalesced 8-byte loads), times the periodic Hann window
//   B  16-point DFT over j in registers (radix 4 x 4), twiddle W_1024^(l k1)
//   D  one transpose through LDS (8.5 KB, padded rows: conflict-free both ways): lane (k1, l') holds l = l' + 4 j'
//   E  16-point DFT over j' in registers, twiddle W_64^(l' k2a)
//   G  the last radix-4 runs ACROSS the four lanes of a quad with DPP quad_perm moves (no LDS)
//   H  Z[k] to LDS in natural order;  I  real-FFT untangle on (k, 1024 - k) pairs: X[k] = E + W_2048^k O and
//      X[1024 - k] = conj(E - W_2048^k O) share one complex multiply; |.| and atan2 (polynomial) -> (mag, phase) back to LDS
//      in place (two halves ordered so that no unread Z is overwritten)
//   J  mel projection as a gather: the non-zeros of a mel column are one run of <= 6 linear bins (2042 non-zeros in the
//      1024 x 1024 matrix) -- never a dense GEMM;  K  log / normalise, IF, one 16-byte store per lane and 128-column block
// The reference's IF is diff(unwrap(phase)) / pi with unwrap = phase + cumsum(wrap(d) - d); in exact arithmetic that is
// wrap(d) / pi with d = phase[t] - phase[t-1], which needs only the previous frame.  A wave therefore walks a RUN of
// consecutive frames, keeps the previous frame's 16 mel phases per lane in registers and recomputes frame t0 - 1 once per
// run; fp32 deviation from the cumsum form <= 1e-4 (|unwrapped| reaches a few hundred rad).
#include "spectral_plan.h"

namespace gs {

#ifndef SW_WAVES
#define SW_WAVES 12         // waves per block = per CU.  Measured at batch 256: 12 waves (3 per SIMD, <= 168 VGPRs) 126 us, 16 waves
#endif                      // (128 VGPRs, Hann / run starts through L1) 149 us, 8 waves (256 VGPRs) 156 us: the LDS pipe is the bound
#if SW_WAVES > 12           // 16 waves x 8.5 KB leave 24 KB of LDS for tables: the Hann window and the mel run starts stay in L1 / L2
#define SW_TABLES_IN_LDS 0
#else
#define SW_TABLES_IN_LDS 1
#endif
#define SW_ROW 68           // float2 per transposed row: 64 + 4 pad (the k1 rows land on distinct bank groups)
#define SW_BUF (16 * SW_ROW)
#define SW_WTOT (22 * 128)  // floats of mel weights for SW_SHAPE

// run lengths of the mel columns per 128-column block in the reference configuration (1024 mel bins over 0..8 kHz at 16 kHz):
// compile-time, so the gather has no branch per bin.  Other mel shapes use the generic kernels of spectral.hip.
__device__ constexpr int SW_SHAPE[8] = {1, 1, 2, 2, 3, 3, 4, 6};
static const int SW_SHAPE_HOST[8] = {1, 1, 2, 2, 3, 3, 4, 6};

// 8-byte LDS read that the backend will not pair into ds_read2_b64: the paired form moves 128 B/clk, two ds_read_b64 256 B/clk
// (MI355X_MICROARCH.md, LDS table), and the LDS pipe is what bounds this kernel
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 lds_ld(const float2* p) {
    typedef const volatile __attribute__((address_space(3))) v2f_t* lds_ptr_t;   // (a volatile generic pointer would become a flat load)
    const v2f_t t = *(lds_ptr_t)(p);
    return make_float2(t.x, t.y);
}

// streaming store of one lane's (log-mel, IF) x 2 mel bins: the 1 MB per example of output must not evict the waveform lines the
// next frames re-read (75 % overlap) from the XCD's L2
__device__ __forceinline__ void st4_stream(float* p, const float (&o)[4]) {
    typedef float f4_t __attribute__((ext_vector_type(4)));
    const f4_t v = {o[0], o[1], o[2], o[3]};
    __builtin_nontemporal_store(v, reinterpret_cast<f4_t*>(p));
}
__device__ __forceinline__ void st4_stream(bf16_t* p, const float (&o)[4]) {
    typedef unsigned int u2_t __attribute__((ext_vector_type(2)));
    const u2_t v = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
    __builtin_nontemporal_store(v, reinterpret_cast<u2_t*>(p));
}

// (Tried and measured slower: complex numbers as 2-vectors compiled to v_pk_add / v_pk_fma_f32 -- 7 % fewer VALU instructions, but
//  aligned register pairs push the kernel over 168 VGPRs (14 spilled: 140 us) and without spills the hazard nops and longer
//  dependent chains cost more than the issue slots save: 127 us against 119 us.)
__device__ __forceinline__ float2 cmulw(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// forward 4-point DFT in place: (a, b, c, d) = x[0..3] -> X[0..3]
__device__ __forceinline__ void dft4(float2& a, float2& b, float2& c, float2& d) {
    const float2 s0 = make_float2(a.x + c.x, a.y + c.y), s1 = make_float2(a.x - c.x, a.y - c.y);
    const float2 s2 = make_float2(b.x + d.x, b.y + d.y), s3 = make_float2(b.x - d.x, b.y - d.y);
    a = make_float2(s0.x + s2.x, s0.y + s2.y);
    c = make_float2(s0.x - s2.x, s0.y - s2.y);
    b = make_float2(s1.x + s3.y, s1.y - s3.x);   // s1 - i s3
    d = make_float2(s1.x - s3.y, s1.y + s3.x);   // s1 + i s3
}

// forward 16-point DFT in place, input v[n] natural; output position p holds X[KPOS(p)], KPOS(p) = (p >> 2) + 4 (p & 3)
#define KPOS(p) (((p) >> 2) + 4 * ((p) & 3))
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
    const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, r = 0.70710678118654752f;
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4(v[n2], v[n2 + 4], v[n2 + 8], v[n2 + 12]);   // v[n2 + 4 k1]
    float2 t;
    t = v[5];  v[5]  = make_float2(t.x * c1 + t.y * s1, t.y * c1 - t.x * s1);       // W16^1 = (c1, -s1)
    t = v[9];  v[9]  = make_float2(r * (t.x + t.y), r * (t.y - t.x));                // W16^2 = (r, -r)
    t = v[13]; v[13] = make_float2(t.x * s1 + t.y * c1, t.y * s1 - t.x * c1);       // W16^3 = (s1, -c1)
    t = v[6];  v[6]  = make_float2(r * (t.x + t.y), r * (t.y - t.x));                // W16^2
    t = v[10]; v[10] = make_float2(t.y, -t.x);                                       // W16^4 = -i
    t = v[14]; v[14] = make_float2(r * (t.y - t.x), -r * (t.x + t.y));               // W16^6 = (-r, -r)
    t = v[7];  v[7]  = make_float2(t.x * s1 + t.y * c1, t.y * s1 - t.x * c1);       // W16^3
    t = v[11]; v[11] = make_float2(r * (t.y - t.x), -r * (t.x + t.y));               // W16^6
    t = v[15]; v[15] = make_float2(-t.x * c1 - t.y * s1, t.x * s1 - t.y * c1);       // W16^9 = (-c1, s1)
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}

template <int CTRL>
__device__ __forceinline__ float quad_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}

// atan2 with |error| <= 1.8e-6 rad (the contract on the phase is 1e-3; fp32 FFT round-off is larger on all but the strongest bins): odd
// minimax polynomial of degree 11 on [0, 1] (LP fit on 4000 points, error measured with the fp32 Horner evaluation) + octant reduction;
// (0, 0) -> 0 like np.angle.  Degree 15 (2e-7) cost two more FMAs per bin, 1024 bins per frame.
__device__ __forceinline__ float atan2_poly(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mn * __builtin_amdgcn_rcpf(fmaxf(mx, 1.0e-30f));   // (0, 0): t = 0 -> angle 0, like np.angle
    const float s = t * t;
    float r = -0.011708833277225494f;
    r = fmaf(r, s, 0.052617236971855164f);
    r = fmaf(r, s, -0.11639391630887985f);
    r = fmaf(r, s, 0.193524569272995f);
    r = fmaf(r, s, -0.3326195776462555f);
    r = fmaf(r, s, 0.9999770522117615f);
    r *= t;
    r = ay > ax ? 1.57079637050628662f - r : r;
    r = x < 0.f ? 3.14159274101257324f - r : r;
    return copysignf(r, y);
}

// Samples of one frame, lane l holding z[n] = (x[2n], x[2n+1]) for n = l + 64 j (zeros outside the waveform).
__device__ __forceinline__ void load_frame(const float* __restrict__ wv, int wave_len, int base, bool vec_ok, int lane, float2 (&xs)[16]) {
    if (base + 2048 <= 0 || base >= wave_len) {   // entirely inside the padding
#pragma unroll
        for (int j = 0; j < 16; ++j) xs[j] = make_float2(0.f, 0.f);
    } else if (vec_ok && base >= 0 && base + 2048 <= wave_len) {
#pragma unroll
        for (int j = 0; j < 16; ++j) xs[j] = *reinterpret_cast<const float2*>(wv + base + 2 * (lane + 64 * j));
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int s0 = base + 2 * (lane + 64 * j);
            xs[j].x = (s0 >= 0 && s0 < wave_len) ? wv[s0] : 0.f;
            xs[j].y = (s0 + 1 >= 0 && s0 + 1 < wave_len) ? wv[s0 + 1] : 0.f;
        }
    }
}

// 1024-point forward DFT of one wave: v[j] = z[lane + 64 j] (consumed) -> Z[k] in this wave's LDS buffer at P(k) = k + 4 (k >> 8).
//   s_twc: W_1024^(lane * KPOS(q)) as [q][lane];  s_twf: W_64^k (k < 64), both in LDS
__device__ __forceinline__ void fft1024_to_lds(float2 (&v)[16], const float2* s_twc, const float2* s_twf, float2* buf, int lane) {
    // B: DFT over j, twiddle W_1024^(l k1)
    dft16(v);
#pragma unroll
    for (int q = 1; q < 16; ++q) v[q] = cmulw(v[q], lds_ld(s_twc + 64 * q + lane));
    __builtin_amdgcn_sched_barrier(0);
    // D: transpose.  write Y[l][k1] at k1 * ROW + l; lane (k1 = lane >> 2, l' = lane & 3) reads l = l' + 4 j'
#pragma unroll
    for (int q = 0; q < 16; ++q) buf[KPOS(q) * SW_ROW + lane] = v[q];
    {
        const float2* src = buf + (lane >> 2) * SW_ROW + (lane & 3);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = lds_ld(src + 4 * j);
    }
    __builtin_amdgcn_sched_barrier(0);
    // E: DFT over j', twiddle W_64^(l' k2a)
    dft16(v);
    {
        const int lq = lane & 3;
#pragma unroll
        for (int q = 1; q < 16; ++q) v[q] = cmulw(v[q], lds_ld(s_twf + lq * KPOS(q)));
    }
    __builtin_amdgcn_sched_barrier(0);
    // G: radix 4 across the quad.  after it lane l' holds k2b = bitrev2(l')
    {
        const int lq = lane & 3;
        const float sa = (lq & 2) ? -1.f : 1.f;
        const float sb = (lq & 1) ? -1.f : 1.f;
        const bool rot = lq == 3;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float ux = fmaf(sa, v[q].x, quad_mov<0x4E>(v[q].x));   // own +- partner (lane ^ 2)
            const float uy = fmaf(sa, v[q].y, quad_mov<0x4E>(v[q].y));
            const float rx = rot ? uy : ux, ry = rot ? -ux : uy;         // lane 3: times -i
            v[q].x = fmaf(sb, rx, quad_mov<0xB1>(rx));                   // own +- partner (lane ^ 1)
            v[q].y = fmaf(sb, ry, quad_mov<0xB1>(ry));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // H: Z[k], k = k1 + 16 k2a + 256 k2b, at P(k) = k + 4 (k >> 8) (the four k2b planes on distinct bank groups)
    {
        const int lq = lane & 3;
        const int k2b = ((lq & 1) << 1) | (lq >> 1);
        float2* dst = buf + (lane >> 2) + k2b * 260;
#pragma unroll
        for (int q = 0; q < 16; ++q) dst[16 * KPOS(q)] = v[q];
    }
}

// One frame: windowed samples v (consumed) -> buf[k] = (|X[k]|, arg X[k]) for bins k = 1..1024 (this wave's LDS buffer).
//   s_twu: W_2048^k (k < 512) in LDS, the lane's eight in twu
__device__ __forceinline__ void frame_to_magphase(float2 (&v)[16], const float2* s_twc, const float2* s_twf, const float2 (&twu)[8],
                                                  float2* buf, int lane) {
    fft1024_to_lds(v, s_twc, s_twf, buf, lane);
    // I: untangle the packed real FFT on pairs (ka, 1024 - ka), ka = lane + 64 i; (mag, phase) of bin k goes to buf[k], in
    // place over Z.  Half A (i = 4..7) reads P in [260, 780] and writes [256, 511] + [513, 768]; half B (i = 0..3) reads
    // [0, 255] + [781, 1035] + P(0), untouched by A's writes, and writes [1, 255] + 512 + [769, 1024].  Within a half every
    // read precedes every write in program order (LDS executes a wave's accesses in order).
    const float2 z512 = buf[512 + 8];
#pragma unroll
    for (int half = 1; half >= 0; --half) {
        __builtin_amdgcn_sched_barrier(0);
        float2 z1[4], z2[4], oa[4], ob[4];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = 4 * half + ii;
            const int ka = lane + 64 * i;
            const int kb = (1024 - ka) & 1023;
            z1[ii] = lds_ld(buf + ka + 4 * (i >> 2));
            z2[ii] = lds_ld(buf + kb + 4 * (kb >> 8));
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = 4 * half + ii;
            // (the window table carries the factor 1/2 of the untangle: Z here is half the transform, exactly)
            const float2 e = make_float2(z1[ii].x + z2[ii].x, z1[ii].y - z2[ii].y);    // (Z[k] + conj Z[N-k]) / 2
            const float2 o = make_float2(z1[ii].y + z2[ii].y, z2[ii].x - z1[ii].x);    // (Z[k] - conj Z[N-k]) / 2i
            const float2 t = cmulw(twu[i], o);
            float2 xa = make_float2(e.x + t.x, e.y + t.y);          // X[ka]
            const float2 xb = make_float2(e.x - t.x, t.y - e.y);    // X[1024 - ka] = conj(E - T)
            if (i == 0 && lane == 0) xa = make_float2(2.f * z512.x, -2.f * z512.y);   // the ka = 0 slot carries bin 512 = conj(Z[512]); its xb is bin 1024
            oa[ii] = make_float2(__builtin_amdgcn_sqrtf(xa.x * xa.x + xa.y * xa.y), atan2_poly(xa.y, xa.x));
            ob[ii] = make_float2(__builtin_amdgcn_sqrtf(xb.x * xb.x + xb.y * xb.y), atan2_poly(xb.y, xb.x));
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = 4 * half + ii;
            const int ka = lane + 64 * i;
            buf[(i == 0 && lane == 0) ? 512 : ka] = oa[ii];
            buf[1024 - ka] = ob[ii];
        }
    }
}

// LDS carve (bytes): mel weights | mel run starts (u16) | W_64 | W_2048 (k < 512) | W_1024^(lane k1) | Hann half table | wave buffers
#define SW_LDS_W 0
#define SW_LDS_LO (SW_LDS_W + SW_WTOT * 4)
#define SW_LDS_TWF (SW_LDS_LO + (SW_TABLES_IN_LDS ? 1024 * 2 : 0))
#define SW_LDS_TWU (SW_LDS_TWF + 64 * 8)
#define SW_LDS_TWC (SW_LDS_TWU + 512 * 8)
#define SW_LDS_HANN (SW_LDS_TWC + 16 * 64 * 8)
#define SW_LDS_BUF (SW_LDS_HANN + (SW_TABLES_IN_LDS ? 1040 * 4 : 0))
#define SW_LDS_FLAG (SW_LDS_BUF + SW_WAVES * SW_BUF * 8)
#define SW_LDS_TOTAL (SW_LDS_FLAG + 64)

// MODE 1: images[b][T][1024][2] = (log-mel, IF);  MODE 0: o0 / o1 = magnitude / phase [b][T][1024]
template <typename T, int MODE>
__global__ __launch_bounds__(64 * SW_WAVES) void stft_wave_kernel(const float* __restrict__ hann, const float2* __restrict__ tw1k,
                                                                  const float2* __restrict__ twp, const int* __restrict__ mel_lo,
                                                                  const float* __restrict__ mel_w, int TT, int step,
                                                                  const float* __restrict__ wave, int wave_len, int front_pad,
                                                                  int batch, int runs, float* __restrict__ o0, float* __restrict__ o1,
                                                                  T* __restrict__ images, float* __restrict__ edge) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_w = reinterpret_cast<float*>(smem + SW_LDS_W);
    unsigned short* s_lo = reinterpret_cast<unsigned short*>(smem + SW_LDS_LO);
    float2* s_twf = reinterpret_cast<float2*>(smem + SW_LDS_TWF);
    float2* s_twu = reinterpret_cast<float2*>(smem + SW_LDS_TWU);
    float2* s_twc = reinterpret_cast<float2*>(smem + SW_LDS_TWC);   // [q][lane] = W_1024^(lane KPOS(q))
    float* s_hann = reinterpret_cast<float*>(smem + SW_LDS_HANN);   // w[0..1024]; w[i] = w[2048 - i] beyond
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the run bookkeeping and its branches stay on the scalar unit)
    float2* buf = reinterpret_cast<float2*>(smem + SW_LDS_BUF) + wid * SW_BUF;
    for (int i = threadIdx.x; i < 1024; i += 64 * SW_WAVES) {
        if (SW_TABLES_IN_LDS) {
            s_hann[i] = 0.5f * hann[i];   // (x 1/2: see the untangle)
            if (i == 0) s_hann[1024] = 0.5f * hann[1024];
            if (MODE == 1) s_lo[i] = (unsigned short)mel_lo[i];
        }
        if (i < 64) s_twf[i] = tw1k[16 * i];
        if (i < 512) s_twu[i] = twp[i];
        s_twc[i] = tw1k[(i & 63) * KPOS(i >> 6)];
    }
    if (MODE == 1)
        for (int k = threadIdx.x; k < SW_WTOT; k += 64 * SW_WAVES) s_w[k] = mel_w[k];
    volatile int* s_flag = reinterpret_cast<volatile int*>(smem + SW_LDS_FLAG);
    if (threadIdx.x < 16) s_flag[threadIdx.x] = 0;
    __syncthreads();   // the only block-level barrier
    const long wr = (long)blockIdx.x * SW_WAVES + wid;
    if (wr >= (long)batch * runs) return;
    // frames of an example split into `runs` runs as evenly as possible: the first (TT % runs) runs have one frame more
    const int b = (int)(wr / runs), r = (int)(wr % runs);
    const int q = TT / runs, rem = TT % runs;
    const int t0 = r * q + min(r, rem);
    const int t1 = t0 + q + (r < rem ? 1 : 0);
    const float* wv = wave + (long)b * wave_len;
    const bool vec_ok = ((step | front_pad | wave_len) & 1) == 0;

    float2 twu[8];   // W_2048^(lane + 64 i): lane constants, kept in registers
#pragma unroll
    for (int i = 0; i < 8; ++i) twu[i] = s_twu[lane + 64 * i];
    int2 lo[8];
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) lo[j] = *reinterpret_cast<const int2*>(mel_lo + 128 * j + 2 * lane);
    }
    float prev[16];
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 16; ++j) prev[j] = 0.f;
    }
    const float pi = 3.14159274101257324f;   // float32(np.pi), the constant the reference's unwrap uses
    const float two_pi = pi * 2.0f;

    // IF of a run's FIRST frame needs the mel phases of the frame before it, which the previous run owns.  When the runs of an
    // example fill whole blocks (runs % SW_WAVES == 0) wave w+1 publishes the phases of its first frame (global scratch `edge`,
    // an LDS flag behind a workgroup release) and wave w, done with its own last frame, writes that frame's IF; only the first
    // wave of a block recomputes frame t0 - 1 ("lead").  Otherwise every run recomputes its lead frame.
    const bool exch = MODE == 1 && edge != nullptr && runs % SW_WAVES == 0;
    const bool from_left = exch && wid != 0 && t0 > 0;                               // my first frame's IF is written by wave wid - 1
    const bool to_right = exch && wid != SW_WAVES - 1 && t1 < TT;                    // I write the IF of frame t1 (wave wid + 1's first)
    // spectral_ops.py:21-33 on one difference: floor-mod(d + pi, 2 pi) - pi, with -pi -> pi for d > 0.  Evaluated as d - 2 pi rint(d / 2 pi)
    // (one exact fma): the same value wherever the reference's own fp32 evaluation is more than an ulp of 2 pi away from the cut (rint's
    // ties -- d an odd multiple of pi -- give pi for d = pi and -pi for d = -pi like the reference); 3 instructions instead of 13,
    // 16 times per frame and lane.
    auto wrapped = [&](float d) __attribute__((always_inline)) {
        return fmaf(-__builtin_rintf(d * 0.15915494309189535f), two_pi, d);
    };
    const int tfirst = (MODE == 1 && t0 > 0 && !from_left) ? t0 - 1 : t0;
    for (int t = tfirst; t < t1; ++t) {
        const bool lead = t < t0;                       // frame t0 - 1: only its mel phases are needed
        const float inv_two_pi_live = t == 0 ? 0.f : 0.15915494309189535f;   // (scalar: rint(d * 0) = 0 leaves frame 0 unwrapped)
        const int base = t * step - front_pad;
        const bool silent = base + 2048 <= 0 || base >= wave_len;   // entirely inside the padding: spectrum exactly 0
        if (!silent) {
            float2 v[16], xs[16];
            load_frame(wv, wave_len, base, vec_ok, lane, xs);   // (a register prefetch of the next frame spills at 168 VGPRs: measured 150 us)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int n = lane + 64 * j;
                float2 h;
                if (!SW_TABLES_IN_LDS) { h = *reinterpret_cast<const float2*>(hann + 2 * n); h.x *= 0.5f; h.y *= 0.5f; }   // (8 KB table shared by every wave: L1)
                else if (j < 8) h = *reinterpret_cast<const float2*>(s_hann + 2 * n);        // w[2n], w[2n+1]
                else h = make_float2(s_hann[2048 - 2 * n], s_hann[2047 - 2 * n]);             // symmetric half
                v[j] = make_float2(xs[j].x * h.x, xs[j].y * h.y);
            }
            frame_to_magphase(v, s_twc, s_twf, twu, buf, lane);
        }
        const long row = ((long)b * TT + t) * 1024;
        if (MODE == 0) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int f = 4 * lane + 256 * qq;   // bins f+1 .. f+4
                float4 mg = make_float4(0.f, 0.f, 0.f, 0.f), ph = mg;
                if (!silent) {
                    const float2 a = buf[f + 1], c = buf[f + 2], d = buf[f + 3], e = buf[f + 4];
                    mg = make_float4(a.x, c.x, d.x, e.x);
                    ph = make_float4(a.y, c.y, d.y, e.y);
                }
                *reinterpret_cast<float4*>(o0 + row + f) = mg;
                *reinterpret_cast<float4*>(o1 + row + f) = ph;
            }
            continue;
        }
        // J: mel projection of magnitude and phase; lane owns columns m = 128 j + 2 lane + {0, 1}
        int off = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float am0 = 0.f, ap0 = 0.f, am1 = 0.f, ap1 = 0.f;
            if (!silent) {
                const float* wj = s_w + off + 2 * lane;
                const float2* ma = buf + 1 + lo[j].x;
                const float2* mb = buf + 1 + lo[j].y;
#pragma unroll
                for (int e = 0; e < SW_SHAPE[j]; ++e) {   // ascending bins: the order of the oracle's dot product over the non-zeros
                    const float2 w2 = lds_ld(reinterpret_cast<const float2*>(wj + 128 * e));
                    const float2 xa = lds_ld(ma + e), xb = lds_ld(mb + e);
                    if (e == 0) {   // (0 + w x = w x exactly: no zeroed accumulators)
                        am0 = w2.x * xa.x; ap0 = w2.x * xa.y;
                        am1 = w2.y * xb.x; ap1 = w2.y * xb.y;
                    } else {
                        am0 = fmaf(w2.x, xa.x, am0); ap0 = fmaf(w2.x, xa.y, ap0);
                        am1 = fmaf(w2.y, xb.x, am1); ap1 = fmaf(w2.y, xb.y, ap1);
                    }
                }
            }
            off += 128 * SW_SHAPE[j];
            // K: spectral_ops.py:88-92 and :21-44
            float vif[2];
            const float ap[2] = {ap0, ap1};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                // frame 0: the unwrapped phase itself.  prev is 0 there; `first` (wave-uniform, 0 or 1) cancels the wrap of that one frame
                // without a select per bin: ap - 2 pi rint(ap / 2 pi) * (1 - first)
                const float d = ap[e] - prev[2 * j + e];
                vif[e] = fmaf(-__builtin_rintf(d * inv_two_pi_live), two_pi, d) * 0.31830988618379069f;   // (a run's first frame in exchange mode: placeholder, see below)
                prev[2 * j + e] = ap[e];
            }
            if (from_left && t == t0)
                *reinterpret_cast<float2*>(edge + ((long)b * runs + r) * 1024 + 128 * j + 2 * lane) = make_float2(ap0, ap1);
            if (!lead) {
                // (ln(x) + 3.76) / 10.05 = log2(x) * (ln 2 / 10.05) + 3.76 / 10.05
                const float l0 = fmaf(__builtin_amdgcn_logf(am0 + 1.0e-6f), 0.068969869f, 0.37412935f);
                const float l1 = fmaf(__builtin_amdgcn_logf(am1 + 1.0e-6f), 0.068969869f, 0.37412935f);
                const float o[4] = {l0, vif[0], l1, vif[1]};
                st4_stream(images + (row + 128 * j + 2 * lane) * 2, o);
            }
        }
        if (from_left && t == t0) {   // this frame's image row and phases are out: tell wave wid - 1
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) s_flag[wid] = 1;
        }
    }
    if (MODE == 1 && to_right) {
        while (s_flag[wid + 1] == 0) __builtin_amdgcn_s_sleep(8);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const float* nxt = edge + ((long)b * runs + r + 1) * 1024;
        T* dst = images + ((long)b * TT + t1) * 2048;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float2 p2 = *reinterpret_cast<const float2*>(nxt + 128 * j + 2 * lane);
            DT<T>::st(dst + (128 * j + 2 * lane) * 2 + 1, wrapped(p2.x - prev[2 * j]) * 0.31830988618379069f);
            DT<T>::st(dst + (128 * j + 2 * lane) * 2 + 3, wrapped(p2.y - prev[2 * j + 1]) * 0.31830988618379069f);
        }
    }
}

// runs per example: enough wave-runs to fill SW_WAVES waves on every CU (a run of R frames costs R + 1 transforms), at most one per frame
static int runs_per_example(int batch, int time_steps) {
    int runs = (256 * SW_WAVES + batch - 1) / batch;
    if (runs > time_steps) runs = time_steps;
    if (runs < 1) runs = 1;
    if (runs >= SW_WAVES) runs -= runs % SW_WAVES;   // whole blocks per example: neighbouring runs exchange their edge phases in the block
    return runs;
}

template <typename KernT>
static int set_lds(KernT kern, size_t bytes) {
    static bool done = false;   // per kernel instantiation
    if (done) return 0;
    if (bytes > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        return fail(GS_ERR_HIP, "stft_wave: cannot reserve %zu bytes of LDS", bytes);
    done = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------ inverse
// (magnitude, phase) [frames][1024] -> windowed time frames [frames][2048] (spectral_ops.py:128-149 before the overlap-add): the same
// wave-per-frame machinery run backwards.  irfft of the 1025-bin spectrum X (DC = 0, X[k] = mag e^{i phase}) as a packed 1024-point
// complex transform: Z[k] = E[k] + i W_2048^{-k} D[k] with E / D the half sum / half difference of X[k] and conj X[1024 - k];
// z = IFFT(Z) = conj(FFT(conj Z)) reuses fft1024_to_lds; frame[2n] = Re z[n], frame[2n+1] = Im z[n], times 1/1024 and the inverse
// window.  A lane builds Z on the pairs (ka, 1024 - ka), ka = lane + 64 i -- its own FFT inputs for i < 8 -- and hands the upper
// half to the lanes that need it through the wave's LDS buffer.
//
// sin / cos of phases that reach ~1e3 rad: three-constant Cody-Waite reduction by pi/2 (each step one exact fma) and the Cephes
// single-precision polynomials on [-pi/4, pi/4]: 1e-7 absolute for |x| < 1e4, ~25 instructions for the pair.
__device__ __forceinline__ void sincos_reduced(float x, float& sn, float& cs) {
    const float k = __builtin_rintf(x * 0.63661977236758138f);
    float r = fmaf(-k, 1.57079637050628662f, x);       // fl(pi/2)
    r = fmaf(-k, -4.371138828673793e-08f, r);          // fl(pi/2 - fl(pi/2))
    r = fmaf(-k, -1.7151245100058819e-15f, r);         // and the next 24 bits
    const float s = r * r;
    float ps = -1.9515295891e-4f;
    ps = fmaf(ps, s, 8.3321608736e-3f);
    ps = fmaf(ps, s, -1.6666654611e-1f);
    const float sr = fmaf(r * s, ps, r);
    float pc = 2.443315711809948e-5f;
    pc = fmaf(pc, s, -1.388731625493765e-3f);
    pc = fmaf(pc, s, 4.166664568298827e-2f);
    pc = fmaf(pc, s, -0.5f);
    const float cr = fmaf(pc, s, 1.0f);
    const int q = (int)k & 3;
    const float a = (q & 1) ? cr : sr, b = (q & 1) ? sr : cr;   // q odd: sin <- cos, cos <- sin
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}

#define SWI_LDS_TWF 0
#define SWI_LDS_TWC (SWI_LDS_TWF + 64 * 8)
#define SWI_LDS_WIN (SWI_LDS_TWC + 16 * 64 * 8)
#define SWI_LDS_BUF (SWI_LDS_WIN + 2048 * 4)
#define SWI_LDS_TOTAL (SWI_LDS_BUF + SW_WAVES * SW_BUF * 8)

// OLA: the overlap-add (hop 512 = a quarter frame) and the front-padding crop happen here too, frames never reach memory.  Block =
// example, wave w = run w of its frames (SW_WAVES runs).  A lane keeps the running sum of the current 2048-sample window as 16
// sample pairs (n = lane + 64 i); a hop moves a pair from slot i to slot i - 4 of the SAME lane, so after adding frame t the four
// lowest slots are the finished segment t.  The first three segments of a run still miss the previous run's frames: they wait in
// registers until the wave on the left has parked the three unfinished segments of ITS window in its (now idle) LDS buffer -- a fixed
// two-term sum, no atomics, no pre-zeroed output.
#define SWI_LDS_FLAG SWI_LDS_TOTAL
template <bool OLA>
__global__ __launch_bounds__(64 * SW_WAVES) void istft_wave_kernel(const float2* __restrict__ tw1k, const float2* __restrict__ twp, const float* __restrict__ inv_window,
                                                                   const float* __restrict__ mag, const float* __restrict__ phase, float* __restrict__ frames,
                                                                   long nframes, int TT, float* __restrict__ wave, int wave_len, int front_pad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* s_twf = reinterpret_cast<float2*>(smem + SWI_LDS_TWF);
    float2* s_twc = reinterpret_cast<float2*>(smem + SWI_LDS_TWC);
    float* s_win = reinterpret_cast<float*>(smem + SWI_LDS_WIN);
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2* buf = reinterpret_cast<float2*>(smem + SWI_LDS_BUF) + wid * SW_BUF;
    for (int i = threadIdx.x; i < 2048; i += 64 * SW_WAVES) {
        s_win[i] = inv_window[i] * (1.0f / 1024.0f);   // (x 1/1024: the transform below is unnormalised)
        if (i < 64) s_twf[i] = tw1k[16 * i];
        if (i < 1024) s_twc[i] = tw1k[(i & 63) * KPOS(i >> 6)];
    }
    volatile int* s_flag = reinterpret_cast<volatile int*>(smem + SWI_LDS_FLAG);
    if (OLA && threadIdx.x < 16) s_flag[threadIdx.x] = 0;
    __syncthreads();   // the only block-level barrier
    float2 twu[8];     // e^{+2 pi i ka / 2048}, ka = lane + 64 i
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float2 t = twp[lane + 64 * i]; twu[i] = make_float2(t.x, -t.y); }
    // OLA: run wid of example blockIdx.x, frames [t0, t1) (the first TT % SW_WAVES runs have one frame more); else a grid-stride walk
    const int rq = TT / SW_WAVES, rrem = TT % SW_WAVES;
    const int t0 = OLA ? wid * rq + min(wid, rrem) : 0;
    const int t1 = OLA ? t0 + rq + (wid < rrem ? 1 : 0) : 0;
    float2 acc[16], head[12];
    if (OLA) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = make_float2(0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 12; ++i) head[i] = make_float2(0.f, 0.f);
    }
    float* const wv_out = OLA ? wave + (long)blockIdx.x * wave_len : nullptr;
    auto store_segment = [&](int h, const float2& a, const float2& b, const float2& c, const float2& d) __attribute__((always_inline)) {
        const float2 q[4] = {a, b, c, d};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sidx = h * 512 + 2 * (lane + 64 * i) - front_pad;   // (front_pad and wave_len are even: a pair never straddles the crop)
            if (sidx >= 0 && sidx < wave_len) *reinterpret_cast<float2*>(wv_out + sidx) = q[i];
        }
    };
    const long f_begin = OLA ? (long)blockIdx.x * TT + t0 : (long)blockIdx.x * SW_WAVES + wid;
    const long f_end = OLA ? (long)blockIdx.x * TT + t1 : nframes;
    const long f_step = OLA ? 1 : (long)gridDim.x * SW_WAVES;
    for (long f = f_begin; f < f_end; f += f_step) {
        const float* mg = mag + f * 1024;     // bin k at index k - 1
        const float* ph = phase + f * 1024;
        float2 v[16];
        float ma[8], pa[8], mb[8], pb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ka = lane + 64 * i;
            const bool dc = i == 0 && lane == 0;   // ka = 0: X[0] = 0, its partner is the Nyquist bin (real part only)
            ma[i] = dc ? 0.f : mg[ka - 1 + (dc ? 1 : 0)];
            pa[i] = dc ? 0.f : ph[ka - 1 + (dc ? 1 : 0)];
            mb[i] = mg[1023 - ka];
            pb[i] = ph[1023 - ka];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ka = lane + 64 * i;
            const bool dc = i == 0 && lane == 0;
            float sa, ca, sb, cb;
            sincos_reduced(pa[i], sa, ca);
            sincos_reduced(pb[i], sb, cb);
            const float2 xk = make_float2(ma[i] * ca, ma[i] * sa);                       // X[ka]
            const float2 xc = make_float2(mb[i] * cb, dc ? 0.f : mb[i] * sb);            // X[1024 - ka]
            const float2 w = twu[i];
            // Z[ka] = E + i W D,  E = (X[ka] + conj X[1024-ka]) / 2,  D = (X[ka] - conj X[1024-ka]) / 2
            const float2 e = make_float2(0.5f * (xk.x + xc.x), 0.5f * (xk.y - xc.y));
            const float2 d = make_float2(0.5f * (xk.x - xc.x), 0.5f * (xk.y + xc.y));
            const float2 o = cmulw(w, d);
            v[i] = make_float2(e.x - o.y, -(e.y + o.x));                                  // conj Z[ka]: this lane's input m = ka
            // Z[1024 - ka]: the roles of the two bins swap, W_2048^{-(1024 - ka)} = -conj(W_2048^{-ka})
            const float2 e2 = make_float2(e.x, -e.y);
            const float2 d2 = make_float2(-d.x, d.y);
            const float2 o2 = cmulw(make_float2(-w.x, w.y), d2);
            if (!dc) buf[1024 - ka] = make_float2(e2.x - o2.y, -(e2.y + o2.x));            // conj Z[kb], natural order (no bin 1024)
        }
        if (lane == 0) {   // Z[512] = conj X[512]: its own partner
            float s5, c5;
            sincos_reduced(ph[511], s5, c5);
            buf[512] = make_float2(mg[511] * c5, mg[511] * s5);                           // conj Z[512] = X[512]
        }
#pragma unroll
        for (int j = 8; j < 16; ++j) v[j] = lds_ld(buf + lane + 64 * j);
        __builtin_amdgcn_sched_barrier(0);
        fft1024_to_lds(v, s_twc, s_twf, buf, lane);                                      // U = FFT(conj Z) at P(k)
        float* fr = OLA ? nullptr : frames + f * 2048;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int n = lane + 64 * i;
            const float2 u = lds_ld(buf + n + 4 * (n >> 8));
            const float2 wn = *reinterpret_cast<const float2*>(s_win + 2 * n);
            const float2 val = make_float2(u.x * wn.x, -u.y * wn.y);   // z[n] = conj U[n]
            if (OLA) { acc[i].x += val.x; acc[i].y += val.y; }
            else *reinterpret_cast<float2*>(fr + 2 * n) = val;
        }
        if (OLA) {
            const int t = (int)(f - (long)blockIdx.x * TT), k = t - t0;
            if (t0 > 0 && k < 3) {   // still missing the left neighbour's frames
#pragma unroll
                for (int kk = 0; kk < 3; ++kk)
                    if (k == kk) { head[4 * kk] = acc[0]; head[4 * kk + 1] = acc[1]; head[4 * kk + 2] = acc[2]; head[4 * kk + 3] = acc[3]; }
            } else {
                store_segment(t, acc[0], acc[1], acc[2], acc[3]);
            }
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i] = acc[i + 4];
#pragma unroll
            for (int i = 12; i < 16; ++i) acc[i] = make_float2(0.f, 0.f);
        }
    }
    if (OLA) {
        // the three unfinished segments t1 .. t1 + 2 of this run's window: final for the last run, else parked for the wave on the right
        if (t1 >= TT) {
#pragma unroll
            for (int g = 0; g < 3; ++g) store_segment(t1 + g, acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        } else {
#pragma unroll
            for (int i = 0; i < 12; ++i) buf[64 * i + lane] = acc[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) s_flag[wid] = 1;
        }
        if (t0 > 0) {
            while (s_flag[wid - 1] == 0) __builtin_amdgcn_s_sleep(4);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const float2* left = reinterpret_cast<const float2*>(smem + SWI_LDS_BUF) + (wid - 1) * SW_BUF;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float2 q[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float2 l = left[64 * (4 * g + i) + lane];
                    q[i] = make_float2(head[4 * g + i].x + l.x, head[4 * g + i].y + l.y);
                }
                store_segment(t0 + g, q[0], q[1], q[2], q[3]);
            }
        }
    }
}

int launch_istft_wave(const gs_spectral_plan* p, const float* mag, const float* phase, float* frames, long nframes, hipStream_t st) {
    auto kern = istft_wave_kernel<false>;
    if (int e = set_lds(kern, SWI_LDS_TOTAL + 64)) return e;
    long blocks = (nframes + SW_WAVES - 1) / SW_WAVES;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * SW_WAVES), SWI_LDS_TOTAL + 64, st, (const float2*)p->tw1k, (const float2*)p->twp,
                       (const float*)p->inv_window, mag, phase, frames, nframes, 0, (float*)nullptr, 0, 0);
    GS_CHECK_LAUNCH();
    return 0;
}

// frames of a whole example per block, overlap-add and crop included: needs >= 3 frames per run and even crop offsets
bool istft_wave_ola_ok(const gs_spectral_plan* p, int wave_len, int front_pad) {
    return p->frame_length == 2048 && p->frame_step == 512 && p->time_steps >= 3 * SW_WAVES && (wave_len & 1) == 0 && (front_pad & 1) == 0;
}
int launch_istft_wave_ola(const gs_spectral_plan* p, const float* mag, const float* phase, float* wave, int batch, int wave_len, int front_pad, hipStream_t st) {
    auto kern = istft_wave_kernel<true>;
    if (int e = set_lds(kern, SWI_LDS_TOTAL + 64)) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(64 * SW_WAVES), SWI_LDS_TOTAL + 64, st, (const float2*)p->tw1k, (const float2*)p->twp,
                       (const float*)p->inv_window, mag, phase, (float*)nullptr, (long)batch * p->time_steps, p->time_steps, wave, wave_len, front_pad);
    GS_CHECK_LAUNCH();
    return 0;
}

bool stft_wave_shape_ok(const int* cnt) {
    for (int j = 0; j < 8; ++j) if (cnt[j] != SW_SHAPE_HOST[j]) return false;
    return true;
}

size_t stft_wave_edge_bytes(const gs_spectral_plan* p, int batch) {
    return (size_t)batch * runs_per_example(batch, p->time_steps) * 1024 * sizeof(float);
}

int launch_stft_wave_fused(const gs_spectral_plan* p, const float* wave, int batch, int wave_len, int front_pad, void* images, int dtype,
                           void* ws, size_t ws_bytes, hipStream_t st) {
    const int runs = runs_per_example(batch, p->time_steps);
    float* edge = ws_bytes >= stft_wave_edge_bytes(p, batch) ? reinterpret_cast<float*>(ws) : nullptr;   // (no scratch: every run recomputes its lead frame)
    const long nruns = (long)batch * runs;
    GS_DISPATCH_DTYPE(dtype, {
        auto kern = stft_wave_kernel<T, 1>;
        if (int e = set_lds(kern, SW_LDS_TOTAL)) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(nruns, SW_WAVES)), dim3(64 * SW_WAVES), SW_LDS_TOTAL, st, (const float*)p->hann,
                           (const float2*)p->tw1k, (const float2*)p->twp, (const int*)p->mel_lo, (const float*)p->mel_w, p->time_steps,
                           p->frame_step, wave, wave_len, front_pad, batch, runs, (float*)nullptr, (float*)nullptr, (T*)images, edge);
    });
    GS_CHECK_LAUNCH();
    return 0;
}

int launch_stft_wave_magphase(const gs_spectral_plan* p, const float* wave, int batch, int wave_len, int front_pad, float* mag, float* phase,
                              hipStream_t st) {
    const int runs = runs_per_example(batch, p->time_steps);
    const long nruns = (long)batch * runs;
    auto kern = stft_wave_kernel<float, 0>;
    if (int e = set_lds(kern, SW_LDS_TOTAL)) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(nruns, SW_WAVES)), dim3(64 * SW_WAVES), SW_LDS_TOTAL, st, (const float*)p->hann,
                       (const float2*)p->tw1k, (const float2*)p->twp, (const int*)p->mel_lo, (const float*)p->mel_w, p->time_steps, p->frame_step,
                       wave, wave_len, front_pad, batch, runs, mag, phase, (float*)nullptr, (float*)nullptr);
    GS_CHECK_LAUNCH();
    return 0;
}

}  // namespace gs
