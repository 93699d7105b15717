// waveform <-> (log-mel magnitude, instantaneous frequency) for gfx950
// (reference spectral_ops.py:8-149; TF ops replaced: tf.signal.stft / inverse_stft, tf.abs,
// tf.angle, tf.tensordot with the mel matrix and its tfp.math.pinv, unwrap/diff/cumsum).
//
//  stft_kernel     one workgroup per (frame, example): front-pad + framing + periodic Hann fused
//                  into the load, a 2048-point real FFT done as a 1024-point complex radix-2 FFT
//                  in LDS (packed even/odd trick), |.| and atan2 in the epilogue, and -- fused
//                  variant -- the mel projection straight out of LDS.  The mel matrix is 0.2 %
//                  dense (<= 6 non-zeros per column), so it is applied as an ELL gather, not a GEMM.
//  if kernels      threads across mel bins (coalesced), 128 sequential time steps per thread.
//  inverse         exp / cumsum prep, dense pinv(mel) contraction on fp32 MFMA (this one IS a
//                  dense GEMM), packed inverse FFT, windowed overlap-add.
#include "gs_common.h"
#include "gs_prof.h"
#include "spectral_plan.h"

#include <math.h>
#include <stdlib.h>
#include <vector>

namespace gs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// in-place radix-2 DIT FFT of H points held bit-reversed in LDS; INV conjugates the twiddles
template <bool INV>
__device__ inline void fft_lds(float2* z, const float2* __restrict__ tw, int H, int log2h) {
    for (int s = 1; s <= log2h; ++s) {
        const int half = 1 << (s - 1);
        const int tstep = (H >> 1) >> (s - 1);
        for (int bf = threadIdx.x; bf < (H >> 1); bf += blockDim.x) {
            const int pos = bf & (half - 1);
            const int i = ((bf >> (s - 1)) << s) + pos;
            float2 w = tw[pos * tstep];
            if (INV) w.y = -w.y;
            const float2 u = z[i];
            const float2 v = cmul(w, z[i + half]);
            z[i] = make_float2(u.x + v.x, u.y + v.y);
            z[i + half] = make_float2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
    }
}

__device__ inline int bitrev(int v, int bits) { return (int)(__brev((unsigned)v) >> (32 - bits)); }

// MODE 0: write magnitude/phase [b][T][H] (DC dropped).
// MODE 1: fused mel: images[b][T][H][2] channel 0 = (log(mel_mag + 1e-6) + 3.76)/10.05 ; mel_phase[b][T][H] fp32
// MZ: compile-time ELL width of the mel tables (0: run-time p.maxnz) -- with it the table loads of a mel bin are unrolled and
// in flight together instead of one dependent L1 round trip per entry
template <typename T, int MODE, int MZ>
__global__ __launch_bounds__(256) void stft_kernel(gs_spectral_plan p, const float* __restrict__ wave, int wave_len, int front_pad,
                                                   float* __restrict__ o0, float* __restrict__ o1, T* __restrict__ images) {
    __shared__ float2 z[1024];
    __shared__ float smag[MODE == 1 ? 1024 : 1];
    __shared__ float sph[MODE == 1 ? 1024 : 1];
    const int H = p.nbins, step = p.frame_step;
    const int t = blockIdx.x, b = blockIdx.y;
    const float* wv = wave + (long)b * wave_len;
    for (int n = threadIdx.x; n < H; n += blockDim.x) {
        const int s0 = t * step + 2 * n - front_pad;
        const float x0 = (s0 >= 0 && s0 < wave_len) ? wv[s0] * p.hann[2 * n] : 0.f;
        const float x1 = (s0 + 1 >= 0 && s0 + 1 < wave_len) ? wv[s0 + 1] * p.hann[2 * n + 1] : 0.f;
        z[bitrev(n, p.log2h)] = make_float2(x0, x1);
    }
    __syncthreads();
    fft_lds<false>(z, p.tw, H, p.log2h);
    const long row = ((long)b * p.time_steps + t) * H;
    for (int k = threadIdx.x + 1; k <= H; k += blockDim.x) {
        const float2 zk = z[k & (H - 1)];
        const float2 zc = z[(H - k) & (H - 1)];  // conj applied below
        const float2 e = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));   // (Zk + conj(Zc))/2
        const float2 o = make_float2(0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x));  // (Zk - conj(Zc))/(2i)
        const float2 wo = cmul(p.twp[k], o);
        const float re = e.x + wo.x, im = e.y + wo.y;
        const float mg = hypotf(re, im);
        const float ph = (re == 0.f && im == 0.f) ? 0.f : atan2f(im, re);
        if (MODE == 0) {
            o0[row + k - 1] = mg;
            o1[row + k - 1] = ph;
        } else {
            smag[k - 1] = mg;
            sph[k - 1] = ph;
        }
    }
    if (MODE == 1) {
        __syncthreads();
        for (int m = threadIdx.x; m < H; m += blockDim.x) {
            float am = 0.f, ap = 0.f;
            if (MZ > 0) {
                int f[MZ > 0 ? MZ : 1];
                float w[MZ > 0 ? MZ : 1];
#pragma unroll
                for (int j = 0; j < MZ; ++j) { f[j] = p.mel_idx[j * H + m]; w[j] = p.mel_val[j * H + m]; }
#pragma unroll
                for (int j = 0; j < MZ; ++j) { am += smag[f[j]] * w[j]; ap += sph[f[j]] * w[j]; }   // (ascending bins: the oracle's order)
            } else {
                for (int j = 0; j < p.maxnz; ++j) {
                    const int f = p.mel_idx[j * H + m];
                    const float w = p.mel_val[j * H + m];
                    am += smag[f] * w;
                    ap += sph[f] * w;
                }
            }
            DT<T>::st(images + (row + m) * 2, (logf(am + 1.0e-6f) + 3.76f) / 10.05f);
            o0[row + m] = ap;
        }
    }
}

// out[row][m] = sum_j in[row][idx[m][j]] * val[m][j]
static __global__ void mel_project_kernel(gs_spectral_plan p, const float* __restrict__ in, float* __restrict__ out, long rows) {
    const int H = p.nbins;
    const long total = rows * H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = i % H;
        const float* r = in + (i / H) * H;
        float a = 0.f;
        for (int j = 0; j < p.maxnz; ++j) a += r[p.mel_idx[j * H + m]] * p.mel_val[j * H + m];
        out[i] = a;
    }
}

// spectral_ops.py:21-44 along time: d = p[t]-p[t-1]; m = floormod(d+pi, 2pi)-pi; m = pi where (m==-pi & d>0);
// unwrapped = p + cumsum(m-d); IF = [unwrapped[0], diff(unwrapped)]/pi.  One thread per (example, mel bin).
template <typename T, int MODE>  // MODE 0: out fp32 [b][T][H]; MODE 1: images[b][T][H][2] channel 1
__global__ void if_unwrap_kernel(gs_spectral_plan p, const float* __restrict__ mel_phase, float* __restrict__ out, T* __restrict__ images, int batch) {
    const int H = p.nbins, TT = p.time_steps;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * H) return;
    const int b = i / H, m = i % H;
    const float pi = 3.14159274101257324f;  // float32(np.pi)
    const float two_pi = pi * 2.0f;
    float prev_p = 0.f, prev_u = 0.f, cum = 0.f;
    // the recurrence is sequential in t but its inputs are not: eight phases are fetched per trip (one load per step leaves
    // 128 dependent memory latencies per thread)
    for (int t0 = 0; t0 < TT; t0 += 8) {
        float phs[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) phs[k] = t0 + k < TT ? mel_phase[((long)b * TT + t0 + k) * H + m] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = t0 + k;
            if (t >= TT) break;
            const long o = ((long)b * TT + t) * H + m;
            const float ph = phs[k];
            float v;
            if (t == 0) {
                prev_u = ph;
                v = ph / pi;
            } else {
                const float d = ph - prev_p;
                float md = fmodf(d + pi, two_pi);
                if (md < 0.f) md += two_pi;
                md -= pi;
                if (md == -pi && d > 0.f) md = pi;
                cum += md - d;
                const float u = ph + cum;
                v = (u - prev_u) / pi;
                prev_u = u;
            }
            prev_p = ph;
            if (MODE == 0) out[o] = v;
            else DT<T>::st(images + o * 2 + 1, v);
        }
    }
}

// ------------------------------------------------------------------------------ inverse
// images -> mel_mag = exp(lm*10.05 - 3.76), mel_phase = cumsum(IF*pi) over time (spectral_ops.py:107-111)
// `split` (optional): the same two matrices stacked ([mel_mag; mel_phase], M = 2 x batch x time_steps rows) as three bf16 planes
// [3][M][nbins] whose sum is the fp32 value exactly -- the A operand of gemm_bf16x6_kernel
template <bool THIRD = true>
__device__ __forceinline__ void split3_store(unsigned short* __restrict__ split, long plane_elems, long idx, float v) {
    const unsigned a = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v);
    const float r1 = v - __uint_as_float(a << 16);
    const unsigned b = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)r1);
    const unsigned c = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)(r1 - __uint_as_float(b << 16)));
    split[idx] = (unsigned short)a;
    split[plane_elems + idx] = (unsigned short)b;
    if (THIRD) split[2 * plane_elems + idx] = (unsigned short)c;
}
template <typename T>
__global__ void inv_prep_kernel(gs_spectral_plan p, const T* __restrict__ images, float* __restrict__ mel_mag, float* __restrict__ mel_phase, int batch,
                                unsigned short* __restrict__ split, int mag_planes) {
    const int H = p.nbins, TT = p.time_steps;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * H) return;
    const int b = i / H, m = i % H;
    const float pi = 3.14159274101257324f;
    float cum = 0.f;
    for (int t = 0; t < TT; ++t) {
        const long o = ((long)b * TT + t) * H + m;
        const float lm = DT<T>::ld(images + o * 2) * 10.05f + (-3.76f);
        const float fi = DT<T>::ld(images + o * 2 + 1) * 1.0f + 0.0f;
        cum += fi * pi;
        if (split) {
            const long plane = 2L * batch * TT * H;
            if (mag_planes == 3) split3_store(split, plane, o, expf(lm));
            else split3_store<false>(split, plane, o, expf(lm));   // (the magnitude rows' contraction reads two planes)
            split3_store(split, plane, (long)batch * TT * H + o, cum);
        } else {
            mel_mag[o] = expf(lm);
            mel_phase[o] = cum;
        }
    }
}

// C[M][N] = A[M][K] @ B[K][N], fp32 MFMA (32x32x2), 64x64 block tile, 4 waves (2x2), K step 16.
static __global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    __shared__ float As[64][17];
    __shared__ float Bs[16][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int wm = (wv >> 1) * 32, wn = (wv & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        __syncthreads();
        for (int c = tid; c < 64 * 16; c += 256) {
            const int r = c >> 4, k = c & 15;
            As[r][k] = (m0 + r < M && k0 + k < K) ? A[(long)(m0 + r) * K + k0 + k] : 0.f;
        }
        for (int c = tid; c < 16 * 64; c += 256) {
            const int k = c >> 6, j = c & 63;
            Bs[k][j] = (k0 + k < K && n0 + j < N) ? B[(long)(k0 + k) * N + n0 + j] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[wm + l31][kk + hi], Bs[kk + hi][wn + l31], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int j = n0 + wn + l31;
        if (i < M && j < N) C[(long)i * N + j] = acc[r];
    }
}

// C[M][N] = A[M][K] @ B[K][N] for M, N multiples of 128 and K a multiple of 16: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, 157 TF/s
// peak), 128 x 128 block tile, 4 waves each owning 64 x 64 (2 x 2 accumulator tiles), K step 16.  Both operand tiles sit k-major in
// LDS (As[k][m], Bs[k][n], rows padded to 132) so that the one-float-per-lane MFMA operands are conflict-free ds_read_b32; A is
// transposed on its way in (coalesced 64-byte rows from global, 2-way conflicts on the LDS write).  Two blocks per CU (34 KB of LDS,
// ~110 VGPRs) overlap one block's staging with the other's MFMAs.  The pinv(mel) contraction of the inverse path: M = 2 x examples
// x 128 frames (magnitude and phase stacked), N = K = 1024; the 4 MB B matrix stays in L2.
static __global__ __launch_bounds__(256) void gemm_f32_128_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                                    int M, int N, int K) {
    __shared__ float As[16][132];
    __shared__ float Bs[16][132];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    // (blocks of one B column panel are adjacent: the 64 KB panel of B stays hot while A streams)
    const int nb = N / 128;
    const int m0 = (blockIdx.x / nb) * 128, n0 = (blockIdx.x % nb) * 128;
    const int wm = (wv >> 1) * 64, wn = (wv & 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // staging roles: A -- 4 lanes per 64-byte row, 64 rows per pass, 2 passes; B -- 32 lanes per 512-byte row, 8 rows per pass, 2 passes
    const int a_row = tid >> 2, a_kq = tid & 3;
    const int b_row = tid >> 5, b_col = (tid & 31) * 4;
    const float* ap = A + (long)(m0 + a_row) * K + 4 * a_kq;
    const float* bp = B + (long)b_row * N + n0 + b_col;
    typedef float f4_t __attribute__((ext_vector_type(4)));   // (native vectors: arrays of HIP's float4 struct stay in memory)
    f4_t ra[2], rb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ra[i] = *reinterpret_cast<const f4_t*>(ap + (long)(64 * i) * K);
        rb[i] = *reinterpret_cast<const f4_t*>(bp + (long)(8 * i) * N);
    }
    for (int k0 = 0; k0 < K; k0 += 16) {
        __syncthreads();   // the previous step's fragment reads are done
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            As[4 * a_kq + 0][a_row + 64 * i] = ra[i].x;
            As[4 * a_kq + 1][a_row + 64 * i] = ra[i].y;
            As[4 * a_kq + 2][a_row + 64 * i] = ra[i].z;
            As[4 * a_kq + 3][a_row + 64 * i] = ra[i].w;
            *reinterpret_cast<f4_t*>(&Bs[b_row + 8 * i][b_col]) = rb[i];
        }
        __syncthreads();
        if (k0 + 16 < K) {   // the next step's global loads fly under this step's MFMAs
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ra[i] = *reinterpret_cast<const f4_t*>(ap + (long)(64 * i) * K + k0 + 16);
                rb[i] = *reinterpret_cast<const f4_t*>(bp + (long)(k0 + 16 + 8 * i) * N);
            }
        }
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const float a0 = As[kk + hi][wm + l31], a1 = As[kk + hi][wm + 32 + l31];
            const float b0 = Bs[kk + hi][wn + l31], b1 = Bs[kk + hi][wn + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
                C[(long)row * N + n0 + wn + 32 * j + l31] = acc[i][j][r];
            }
}

// The same contraction on the bf16 MFMA with fp32-level accuracy: every fp32 operand is the exact sum of three bf16 numbers
// (a = a1 + a2 + a3 with a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2): 24 bits of mantissa), and the product keeps the six
// partial products down to 2^-16 of the leading one: a1 b1, a1 b2, a2 b1, a1 b3, a3 b1, a2 b2 -- each exact in the fp32 accumulator,
// the dropped ones (a2 b3, a3 b2, a3 b3) below 2^-24.  Six v_mfma_f32_32x32x16_bf16 (2.5 PF / 6 = 417 TF/s of fp32-grade work) against
// one fp32 MFMA (157 TF/s).  The unwrapped phases reach ~1e3 rad before cos / sin, so plain bf16 (or a two-term split) would not do.
// A comes pre-split from inv_prep_kernel (three planes [3][M][K] bf16), B from plan creation ([3][N][K], k contiguous): the GEMM itself
// only moves 16-byte chunks and issues MFMAs (splitting A inside the GEMM cost ~150 VALU instructions per K step and wave: measured
// slower than the fp32 kernel).
// 128 x 128 block tile, 4 waves of 64 x 64, K step 32; LDS rows of 32 k (64 B) padded to 80 B: conflict-free 16-byte fragment reads.
#define GX_ROW 80
// Block tile 128 x (64 NJ): NJ = 2 -> 128 x 128, two blocks per CU; NJ = 4 -> 128 x 256, one block per CU (92 KB of LDS) and half
// the passes over A, the big operand (6 bytes per element: with 128 x 128 tiles the kernel moved 6.4 GB through L2 in its 0.85 ms).
// NP = planes used: 3 -> all six partial products (fp32 accuracy: the phase rows); 2 -> a1 b1 + a1 b2 + a2 b1 (2^-16 relative per
// product: the magnitude rows, whose contract is 1e-3 relative) at half the MFMAs and two thirds of the operand traffic.
// KB = k blocks of 16 per staged step (2: 64-byte rows, the original; 4: 128-byte rows padded to 144 -- half the barriers per MFMA; fits two
// blocks per CU only with two planes)
template <int NJ, int NP, int KB = 2>
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((NJ == 4 || (NP == 3 && KB == 4)) ? 1 : 2, (NJ == 4 || (NP == 3 && KB == 4)) ? 1 : (NP == 2 && KB == 2 ? 3 : 2))))
void gemm_bf16x6_kernel(const unsigned short* __restrict__ Asplit, const unsigned short* __restrict__ Bsplit, float* __restrict__ C, int M, int N, int K, int m_begin) {
    constexpr int BN = 64 * NJ;            // block columns; a wave owns 64 rows x (32 NJ) columns
    constexpr int P = 2 * KB;              // 16-byte chunks per staged row
    constexpr int ROW = 32 * KB + 16;      // LDS row bytes (80 / 144: conflict-free 16-byte fragment reads)
    constexpr int KSTEP = 16 * KB;
    constexpr int ACH = NP * 128 * P / 256, BCH = NP * BN * P / 256;  // 16-byte chunks of A / B per thread and step
    __shared__ __attribute__((aligned(16))) unsigned char As[NP][128][ROW];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[NP][BN][ROW];
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int nb = N / BN;
    // XCD-aware tile order: block b runs on XCD b % 8, and the nb column tiles of one 128-row slab of A (the big operand: 0.8 MB per slab
    // and plane set) must meet in ONE XCD's L2 -- with the plain order they sat on eight XCDs and A crossed the fabric eight times
    // (1.9 GB fetched per launch for 0.2 GB of operands, 4.8 TB/s: the kernel was HBM-bound, profiles/r02_u_inverse_pmc_traffic.json)
    int mblk = blockIdx.x / nb, nblk = blockIdx.x % nb;
    if (((gridDim.x / nb) & 7) == 0) {
        const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        mblk = (loc / nb) * 8 + xcd;
        nblk = loc % nb;
    }
    const int m0 = m_begin + mblk * 128, n0 = nblk * BN;
    const int wm = (wv >> 1) * 64, wn = (wv & 1) * (32 * NJ);
    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // staging: three planes of [rows][32 k] bf16 per operand and step, as 16-byte chunks (chunk c: plane c / (4 rows), row (c % (4 rows)) >> 2,
    // part c & 3); the next step's chunks wait in registers
    // (plain unrolled code on purpose: pointer arrays and conditionally updated register arrays end up in scratch / promoted LDS)
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));   // (native vectors: arrays of HIP's uint4 struct stay in memory)
    u4_t ra[ACH], rb[BCH];
    const unsigned short* const abase = Asplit + (long)m0 * K;   // (plane stride M x K: M counts ALL rows of A)
    const unsigned short* const bbase = Bsplit + (long)n0 * K;
#define GX_AOFF(J) (((long)((tid + 256 * (J)) / (128 * P)) * M + (((tid + 256 * (J)) % (128 * P)) / P)) * K + ((tid + 256 * (J)) % P) * 8)
#define GX_BOFF(J) (((long)((tid + 256 * (J)) / (P * BN)) * N + (((tid + 256 * (J)) % (P * BN)) / P)) * K + ((tid + 256 * (J)) % P) * 8)
#pragma unroll
    for (int j = 0; j < ACH; ++j) ra[j] = *reinterpret_cast<const u4_t*>(abase + GX_AOFF(j));
#pragma unroll
    for (int j = 0; j < BCH; ++j) rb[j] = *reinterpret_cast<const u4_t*>(bbase + GX_BOFF(j));
    for (int k0 = 0; k0 < K; k0 += KSTEP) {
        __syncthreads();   // the previous step's fragment reads are done
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            const int c = tid + 256 * j;
            *reinterpret_cast<u4_t*>(&As[c / (128 * P)][(c % (128 * P)) / P][(c % P) * 16]) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < BCH; ++j) {
            const int c = tid + 256 * j;
            *reinterpret_cast<u4_t*>(&Bs[c / (P * BN)][(c % (P * BN)) / P][(c % P) * 16]) = rb[j];
        }
        __syncthreads();
        {   // the next step's global loads fly under this step's MFMAs (the last step re-reads its own: unconditional)
            const int kn = k0 + KSTEP < K ? k0 + KSTEP : k0;
#pragma unroll
            for (int j = 0; j < ACH; ++j) ra[j] = *reinterpret_cast<const u4_t*>(abase + GX_AOFF(j) + kn);
#pragma unroll
            for (int j = 0; j < BCH; ++j) rb[j] = *reinterpret_cast<const u4_t*>(bbase + GX_BOFF(j) + kn);
        }
        // fragments of k block 1 are read between the MFMAs of k block 0 (a wave issues in order: reads in front of the MFMAs of
        // their own block leave the pipe idle for an LDS round trip per block)
        constexpr int NF = 2 * NP + NP * NJ;   // fragments of a k block: A (row tile i, plane pl) = NP i + pl | B (column tile j, plane pl) = 2 NP + NP j + pl
        bf16x8_t fr[2][NF];
        auto rd = [&](int kb, int q) __attribute__((always_inline)) {
            const int t = q < 2 * NP ? q : q - 2 * NP, i = t / NP, pl = t % NP;
            fr[kb & 1][q] = q < 2 * NP ? *reinterpret_cast<const bf16x8_t*>(&As[pl][wm + 32 * i + l31][kb * 32 + hi * 16])
                                  : *reinterpret_cast<const bf16x8_t*>(&Bs[pl][wn + 32 * i + l31][kb * 32 + hi * 16]);
        };
#pragma unroll
        for (int q = 0; q < NF; ++q) rd(0, q);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            // (a1 + a2 + a3)(b1 + b2 + b3) down to 2^-16, small terms first; the term loop is the OUTER one so that consecutive MFMAs go to
            // different accumulators (six in a row into one accumulator wait for each other's result)
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
            constexpr int T0 = NP == 3 ? 0 : 3;   // two planes: the last three terms
#pragma unroll
            for (int t = T0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[kb & 1][NP * i + PA[t]], fr[kb & 1][2 * NP + NP * j + PB[t]], acc[i][j], 0, 0, 0);
                        constexpr int NM = (6 - T0) * 2 * NJ;           // MFMAs of the block: the NF reads of the next one spread evenly among them
                        const int slot = ((t - T0) * 2 + i) * NJ + j;
                        if (kb + 1 < KB)
                            for (int r = slot * NF / NM; r < (slot + 1) * NF / NM; ++r) rd(kb + 1, r);
                        __builtin_amdgcn_sched_barrier(0);
                    }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
                C[(long)row * N + n0 + wn + 32 * j + l31] = acc[i][j][r];
            }
#undef GX_AOFF
#undef GX_BOFF
}

// one workgroup per (frame, example): spectrum from (mag, phase), DC = 0 (spectral_ops.py:128-131), packed inverse
// real FFT, multiply by the inverse window; frames[b][T][L]
static __global__ __launch_bounds__(256) void istft_kernel(gs_spectral_plan p, const float* __restrict__ mag, const float* __restrict__ phase, float* __restrict__ frames) {
    __shared__ float2 x[1025];
    __shared__ float2 z[1024];
    const int H = p.nbins, L = p.frame_length;
    const int t = blockIdx.x, b = blockIdx.y;
    const long row = ((long)b * p.time_steps + t) * H;
    for (int k = threadIdx.x; k <= H; k += blockDim.x) {
        if (k == 0) x[0] = make_float2(0.f, 0.f);
        else {
            const float mg = mag[row + k - 1], ph = phase[row + k - 1];
            float sn, cs;
            sincosf(ph, &sn, &cs);
            x[k] = make_float2(mg * cs, mg * sn);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < H; k += blockDim.x) {
        float2 xk = x[k];
        float2 xc = x[H - k];
        if (k == 0) { xk.y = 0.f; xc.y = 0.f; }  // irfft uses only the real parts of DC and Nyquist
        const float2 e = make_float2(0.5f * (xk.x + xc.x), 0.5f * (xk.y - xc.y));  // (Xk + conj(X[H-k]))/2
        const float2 d = make_float2(0.5f * (xk.x - xc.x), 0.5f * (xk.y + xc.y));  // (Xk - conj(X[H-k]))/2
        float2 w = p.twp[k];
        w.y = -w.y;  // e^{+2 pi i k / N}
        const float2 o = cmul(w, d);
        // Z = Xe + i*Xo
        z[bitrev(k, p.log2h)] = make_float2(e.x - o.y, e.y + o.x);
    }
    __syncthreads();
    fft_lds<true>(z, p.tw, H, p.log2h);
    const float sc = 1.0f / (float)H;
    float* fr = frames + ((long)b * p.time_steps + t) * L;
    for (int n = threadIdx.x; n < H; n += blockDim.x) {
        fr[2 * n] = z[n].x * sc * p.inv_window[2 * n];
        fr[2 * n + 1] = z[n].y * sc * p.inv_window[2 * n + 1];
    }
}

// overlap-add (gather form) + drop the front padding: wave[b][s] = sum_f frames[b][f][s + front_pad - f*step]
static __global__ void overlap_add_kernel(gs_spectral_plan p, const float* __restrict__ frames, float* __restrict__ wave, int batch, int wave_len, int front_pad) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)batch * wave_len) return;
    const int b = i / wave_len;
    const int s = (int)(i % wave_len) + front_pad;
    const int L = p.frame_length, step = p.frame_step;
    int f1 = s / step;
    if (f1 > p.time_steps - 1) f1 = p.time_steps - 1;
    int f0 = (s - L + step) / step;
    if (s - L + 1 <= 0) f0 = 0;
    if (f0 < 0) f0 = 0;
    float a = 0.f;
    for (int f = f0; f <= f1; ++f) {
        const int off = s - f * step;
        if (off >= 0 && off < L) a += frames[((long)b * p.time_steps + f) * L + off];
    }
    wave[i] = a;
}

}  // namespace gs

using namespace gs;

#define GS_HIP_OK(expr)                                                                        \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) return gs::fail(GS_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

extern "C" int gs_spectral_plan_create(gs_spectral_plan** out, int frame_length, int frame_step, int time_steps,
                                       const float* mel_dense, const float* mel_pinv) {
    GS_CHECK_ARG(out && mel_dense, "plan_create: null argument");
    const int H = frame_length / 2;
    int log2h = 0;
    while ((1 << log2h) < H) ++log2h;
    GS_CHECK_ARG((1 << log2h) == H && H >= 64 && H <= 1024 && frame_length == 2 * H, "plan_create: frame_length %d must be 2*2^k, 128..2048", frame_length);
    GS_CHECK_ARG(frame_step > 0 && time_steps > 0, "plan_create: bad frame_step/time_steps");
    gs_spectral_plan* p = new gs_spectral_plan();
    memset(p, 0, sizeof(*p));
    p->frame_length = frame_length; p->frame_step = frame_step; p->time_steps = time_steps; p->nbins = H; p->log2h = log2h;
    std::vector<float> hann(frame_length), invw(frame_length);
    for (int n = 0; n < frame_length; ++n) hann[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * n / frame_length));
    {  // tf.signal.inverse_stft_window_fn: w / sum over overlaps of w^2
        const int overlaps = (frame_length + frame_step - 1) / frame_step;
        std::vector<double> den(overlaps * frame_step, 0.0), w2(overlaps * frame_step, 0.0);
        for (int n = 0; n < frame_length; ++n) { const double w = 0.5 - 0.5 * cos(2.0 * M_PI * n / frame_length); w2[n] = w * w; }
        for (int r = 0; r < frame_step; ++r) { double s = 0; for (int o = 0; o < overlaps; ++o) s += w2[o * frame_step + r]; for (int o = 0; o < overlaps; ++o) den[o * frame_step + r] = s; }
        for (int n = 0; n < frame_length; ++n) invw[n] = (float)((0.5 - 0.5 * cos(2.0 * M_PI * n / frame_length)) / den[n]);
    }
    std::vector<float2> tw(H / 2), twp(H + 1);
    for (int k = 0; k < H / 2; ++k) tw[k] = make_float2((float)cos(-2.0 * M_PI * k / H), (float)sin(-2.0 * M_PI * k / H));
    for (int k = 0; k <= H; ++k) twp[k] = make_float2((float)cos(-2.0 * M_PI * k / frame_length), (float)sin(-2.0 * M_PI * k / frame_length));
    int maxnz = 1;
    for (int m = 0; m < H; ++m) { int c = 0; for (int f = 0; f < H; ++f) if (mel_dense[(long)f * H + m] != 0.f) ++c; if (c > maxnz) maxnz = c; }
    if (maxnz <= 8) maxnz = (maxnz + 1) & ~1;   // even widths have an unrolled kernel instantiation (padding = weight 0 on bin 0)
    std::vector<int> idx((long)H * maxnz, 0);
    std::vector<float> val((long)H * maxnz, 0.f);
    for (int m = 0; m < H; ++m) { int c = 0; for (int f = 0; f < H; ++f) { const float w = mel_dense[(long)f * H + m]; if (w != 0.f) { idx[(long)c * H + m] = f; val[(long)c * H + m] = w; ++c; } } }
    p->maxnz = maxnz;
    GS_HIP_OK(hipMalloc(&p->hann, frame_length * sizeof(float)));
    GS_HIP_OK(hipMalloc(&p->inv_window, frame_length * sizeof(float)));
    GS_HIP_OK(hipMalloc(&p->tw, (H / 2) * sizeof(float2)));
    GS_HIP_OK(hipMalloc(&p->twp, (H + 1) * sizeof(float2)));
    GS_HIP_OK(hipMalloc(&p->mel_idx, idx.size() * sizeof(int)));
    GS_HIP_OK(hipMalloc(&p->mel_val, val.size() * sizeof(float)));
    GS_HIP_OK(hipMemcpy(p->hann, hann.data(), frame_length * sizeof(float), hipMemcpyHostToDevice));
    GS_HIP_OK(hipMemcpy(p->inv_window, invw.data(), frame_length * sizeof(float), hipMemcpyHostToDevice));
    GS_HIP_OK(hipMemcpy(p->tw, tw.data(), (H / 2) * sizeof(float2), hipMemcpyHostToDevice));
    GS_HIP_OK(hipMemcpy(p->twp, twp.data(), (H + 1) * sizeof(float2), hipMemcpyHostToDevice));
    GS_HIP_OK(hipMemcpy(p->mel_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
    GS_HIP_OK(hipMemcpy(p->mel_val, val.data(), val.size() * sizeof(float), hipMemcpyHostToDevice));
    if (H == 1024 && !getenv("GS_SPECTRAL_GENERIC")) {
        // wave-per-frame path (spectral_wave.hip): every mel column's non-zeros must be ONE run of linear bins, the longest run
        // of each 128-column block as in the reference configuration
        std::vector<int> lo(H, 0), len(H, 0);
        bool ok = true;
        for (int m = 0; m < H && ok; ++m) {
            int first = -1, last = -1, c = 0;
            for (int f = 0; f < H; ++f) if (mel_dense[(long)f * H + m] != 0.f) { if (first < 0) first = f; last = f; ++c; }
            if (c) { lo[m] = first; len[m] = last - first + 1; ok = len[m] <= 8; }
        }
        if (ok) {
            int tot = 0;
            for (int j = 0; j < 8; ++j) {
                int c = 1;
                for (int m = 128 * j; m < 128 * (j + 1); ++m) c = len[m] > c ? len[m] : c;
                p->mel_cnt[j] = c; p->mel_off[j] = tot; tot += c * 128;
            }
            std::vector<float> w(tot, 0.f);
            for (int j = 0; j < 8; ++j)
                for (int m = 128 * j; m < 128 * (j + 1); ++m) {
                    if (lo[m] + p->mel_cnt[j] > H) lo[m] = H - p->mel_cnt[j];   // keep every gathered bin inside the row (weight 0 there)
                    for (int e = 0; e < p->mel_cnt[j]; ++e) w[p->mel_off[j] + e * 128 + (m - 128 * j)] = mel_dense[(long)(lo[m] + e) * H + m];
                }
            std::vector<float2> t1k(1024);
            for (int k = 0; k < 1024; ++k) t1k[k] = make_float2((float)cos(-2.0 * M_PI * k / 1024.0), (float)sin(-2.0 * M_PI * k / 1024.0));
            p->mel_wtot = tot;
            GS_HIP_OK(hipMalloc(&p->tw1k, 1024 * sizeof(float2)));
            GS_HIP_OK(hipMalloc(&p->mel_lo, H * sizeof(int)));
            GS_HIP_OK(hipMalloc(&p->mel_w, (size_t)tot * sizeof(float)));
            GS_HIP_OK(hipMemcpy(p->tw1k, t1k.data(), 1024 * sizeof(float2), hipMemcpyHostToDevice));
            GS_HIP_OK(hipMemcpy(p->mel_lo, lo.data(), H * sizeof(int), hipMemcpyHostToDevice));
            GS_HIP_OK(hipMemcpy(p->mel_w, w.data(), (size_t)tot * sizeof(float), hipMemcpyHostToDevice));
            p->fast = stft_wave_shape_ok(p->mel_cnt) ? 1 : 0;   // (the kernel's gather is unrolled for the reference configuration's shape)
        }
    }
    if (mel_pinv) {
        GS_HIP_OK(hipMalloc(&p->pinv, (size_t)H * H * sizeof(float)));
        GS_HIP_OK(hipMemcpy(p->pinv, mel_pinv, (size_t)H * H * sizeof(float), hipMemcpyHostToDevice));
        // the three bf16 planes of pinv (exact: hi + mid + lo = the fp32 value), transposed to [n][k] for gemm_bf16x6_kernel
        auto rne = [](float v) -> unsigned short {
            unsigned u; memcpy(&u, &v, 4);
            if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);
            u += 0x7fffu + ((u >> 16) & 1u);
            return (unsigned short)(u >> 16);
        };
        auto widen = [](unsigned short h) -> float { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; };
        std::vector<unsigned short> planes((size_t)3 * H * H);
        for (int k = 0; k < H; ++k)
            for (int n = 0; n < H; ++n) {
                const float v = mel_pinv[(size_t)k * H + n];
                const unsigned short h1 = rne(v);
                const float r1 = v - widen(h1);
                const unsigned short h2 = rne(r1);
                const unsigned short h3 = rne(r1 - widen(h2));
                planes[((size_t)0 * H + n) * H + k] = h1;
                planes[((size_t)1 * H + n) * H + k] = h2;
                planes[((size_t)2 * H + n) * H + k] = h3;
            }
        GS_HIP_OK(hipMalloc(&p->pinv_split, planes.size() * sizeof(unsigned short)));
        GS_HIP_OK(hipMemcpy(p->pinv_split, planes.data(), planes.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    }
    *out = p;
    return 0;
}

extern "C" int gs_spectral_plan_destroy(gs_spectral_plan* p) {
    if (!p) return 0;
    (void)hipFree(p->hann); (void)hipFree(p->inv_window); (void)hipFree(p->tw); (void)hipFree(p->twp); (void)hipFree(p->mel_idx); (void)hipFree(p->mel_val);
    if (p->pinv) (void)hipFree(p->pinv);
    if (p->pinv_split) (void)hipFree(p->pinv_split);
    if (p->fast) { (void)hipFree(p->tw1k); (void)hipFree(p->mel_lo); (void)hipFree(p->mel_w); }
    delete p;
    return 0;
}

extern "C" int gs_stft_fwd(const gs_spectral_plan* p, const float* wave, int batch, int wave_len, int front_pad, float* magnitude,
                           float* phase, void* stream) {
    GS_CHECK_ARG(p && batch > 0 && wave_len > 0, "stft_fwd: bad args");
    if (p->fast) return launch_stft_wave_magphase(p, wave, batch, wave_len, front_pad, magnitude, phase, as_stream(stream));
    hipLaunchKernelGGL((stft_kernel<float, 0, 0>), dim3(p->time_steps, batch), dim3(256), 0, as_stream(stream), *p, wave, wave_len, front_pad,
                       magnitude, phase, (float*)nullptr);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_mel_project(const gs_spectral_plan* p, const float* in, float* out, int64_t rows, void* stream) {
    GS_CHECK_ARG(p && rows > 0, "mel_project: bad args");
    long g = ((long)rows * p->nbins + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(mel_project_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), *p, in, out, (long)rows);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_if_unwrap(const gs_spectral_plan* p, const float* mel_phase, float* mel_if, int batch, void* stream) {
    GS_CHECK_ARG(p && batch > 0, "if_unwrap: bad args");
    hipLaunchKernelGGL((if_unwrap_kernel<float, 0>), dim3(cdiv((long)batch * p->nbins, 256)), dim3(256), 0, as_stream(stream), *p, mel_phase, mel_if,
                       (float*)nullptr, batch);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t gs_stft_mel_if_workspace_bytes(const gs_spectral_plan* p, int batch) {
    if (p && p->fast) return stft_wave_edge_bytes(p, batch);   // the mel phases stay in registers; 4 KB per run for the run-edge exchange
    return p ? (size_t)batch * p->time_steps * p->nbins * sizeof(float) : 0;
}

extern "C" int gs_stft_mel_if_fwd(const gs_spectral_plan* p, const float* wave, int batch, int wave_len, int front_pad, void* images,
                                  int dtype, void* ws, size_t ws_bytes, void* stream) {
    GS_CHECK_ARG(p && batch > 0 && wave_len > 0, "stft_mel_if_fwd: bad args");
    if (p->fast) return launch_stft_wave_fused(p, wave, batch, wave_len, front_pad, images, dtype, ws, ws_bytes, as_stream(stream));
    if (ws_bytes < gs_stft_mel_if_workspace_bytes(p, batch)) return fail(GS_ERR_WORKSPACE, "stft_mel_if_fwd: workspace too small");
    hipStream_t st = as_stream(stream);
    float* mel_phase = (float*)ws;
#define GS_STFT(MZV) hipLaunchKernelGGL((stft_kernel<T, 1, MZV>), dim3(p->time_steps, batch), dim3(256), 0, st, *p, wave, wave_len, front_pad, mel_phase, (float*)nullptr, (T*)images)
    GS_DISPATCH_DTYPE(dtype, {
        if (p->maxnz == 2) GS_STFT(2); else if (p->maxnz == 4) GS_STFT(4); else if (p->maxnz == 6) GS_STFT(6); else if (p->maxnz == 8) GS_STFT(8); else GS_STFT(0);
        hipLaunchKernelGGL((if_unwrap_kernel<T, 1>), dim3(cdiv((long)batch * p->nbins, 256)), dim3(256), 0, st, *p, mel_phase, (float*)nullptr, (T*)images, batch);
    });
#undef GS_STFT
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t gs_mel_if_to_waveform_workspace_bytes(const gs_spectral_plan* p, int batch) {
    if (!p) return 0;
    const size_t rows = (size_t)batch * p->time_steps;
    // [mel_mag; mel_phase] (as two fp32 matrices, or as three bf16 planes of the stacked pair: 12 bytes per element pair), [mag; phase], frames
    return (5 * rows * p->nbins + rows * p->frame_length) * sizeof(float);
}

extern "C" int gs_mel_if_to_waveform(const gs_spectral_plan* p, const void* images, int batch, int wave_len, int front_pad, float* wave,
                                     int dtype, void* ws, size_t ws_bytes, void* stream) {
    GS_CHECK_ARG(p && batch > 0 && wave_len > 0, "mel_if_to_waveform: bad args");
    GS_CHECK_ARG(p->pinv != nullptr, "mel_if_to_waveform: the plan was created without mel_pinv");
    if (ws_bytes < gs_mel_if_to_waveform_workspace_bytes(p, batch)) return fail(GS_ERR_WORKSPACE, "mel_if_to_waveform: workspace too small");
    hipStream_t st = as_stream(stream);
    const long rows = (long)batch * p->time_steps;
    const int H = p->nbins;
    float* mel_mag = (float*)ws;
    float* mel_ph = mel_mag + rows * H;
    float* mag = mel_mag + 3 * rows * H;
    float* ph = mag + rows * H;
    float* frames = ph + rows * H;
    static const bool no_split = getenv("GS_INVERSE_FP32_GEMM") != nullptr;   // measurement knob: the exact-fp32 MFMA kernel
    const bool split = (2 * rows) % 128 == 0 && H % 128 == 0 && H % 32 == 0 && p->pinv_split && !no_split;
    // (the three bf16 planes of the stacked pair take 3 x 2 x rows x H x 2 bytes = the first 3 rows x H floats of the workspace)
    unsigned short* a_split = split ? reinterpret_cast<unsigned short*>(mel_mag) : nullptr;
    static const bool full_mag = getenv("GS_INVERSE_MAG_6TERMS") != nullptr;   // measurement knob: six terms for the magnitude rows too
    const bool two = split && rows % 128 == 0 && !full_mag;   // magnitude rows [0, rows): two planes, three terms
    // (two mel bins per thread with 4-byte plane stores and an unrolled time loop measured the same 121-124 us: 603 MB at ~5 TB/s, the chip's
    //  mixed read / write rate -- the one-bin form stays)
    GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((inv_prep_kernel<T>), dim3(cdiv((long)batch * H, 256)), dim3(256), 0, st, *p, (const T*)images, mel_mag, mel_ph, batch, a_split, two ? 2 : 3));
    GS_CHECK_LAUNCH();
    // [mel_mag; mel_phase] @ pinv(mel) -> [mag; phase]: the two contractions of spectral_ops.py:123,125 share the matrix and are
    // stacked in the workspace, so they are ONE GEMM with 2 x rows
    if (split) {
        // 128 x 128 tiles, two blocks per CU (GS_INVERSE_GEMM_256: 128 x 256 tiles, one block per CU -- half the passes over A, but
        // measured 1 % SLOWER: the kernel is not L2-bound.  PMC (profiles/r02_u_inverse_pmc.txt): the MFMA pipe is busy 54 % of the phase rows'
        // launch and 34 % of the magnitude rows'; the loop is 48 MFMAs + 24 ds_read_b128 + 12 global loads + 12 ds_write_b128 + 2 barriers, and
        // the two barriers of the single-buffered LDS tile are what the second block of the CU does not fully cover)
        static const bool wide = getenv("GS_INVERSE_GEMM_256") != nullptr;
        const int M2 = (int)(2 * rows);
        const unsigned gw = (unsigned)(H / (wide && H % 256 == 0 ? 256 : 128));
        const unsigned blocks_mag = (unsigned)(rows / 128) * gw, blocks_all = (unsigned)(2 * rows / 128) * gw;
        if (H % 256 == 0 && wide) {
            if (two) {
                hipLaunchKernelGGL((gemm_bf16x6_kernel<4, 2>), dim3(blocks_mag), dim3(256), 0, st, a_split, p->pinv_split, mag, M2, H, H, 0);
                hipLaunchKernelGGL((gemm_bf16x6_kernel<4, 3>), dim3(blocks_mag), dim3(256), 0, st, a_split, p->pinv_split, mag, M2, H, H, (int)rows);
            } else {
                hipLaunchKernelGGL((gemm_bf16x6_kernel<4, 3>), dim3(blocks_all), dim3(256), 0, st, a_split, p->pinv_split, mag, M2, H, H, 0);
            }
        } else if (two) {
            // (profiling records: kind 30 = magnitude rows, three bf16 products per multiply-add; 31 = phase rows, six; 32 = all rows, six.
            //  flops = EXECUTED bf16 MFMA flops, bytes = operand planes read once + fp32 result written once)
            const double half = 2.0 * (double)rows * H * H;
            {
                ProfScope ps(st, 3.0 * half, (double)rows * H * 4 + 2.0 * H * H * 2 + (double)rows * H * 4, 30, batch, p->time_steps, H, H, H, 0, 0);
                static const int kb_mag = getenv("GS_INVERSE_GEMM_KB") ? atoi(getenv("GS_INVERSE_GEMM_KB")) : 4;
                if (kb_mag == 4 && H % 64 == 0) hipLaunchKernelGGL((gemm_bf16x6_kernel<2, 2, 4>), dim3(blocks_mag), dim3(256), 0, st, a_split, p->pinv_split, mag, M2, H, H, 0);
                else hipLaunchKernelGGL((gemm_bf16x6_kernel<2, 2>), dim3(blocks_mag), dim3(256), 0, st, a_split, p->pinv_split, mag, M2, H, H, 0);
            }
            {
                ProfScope ps(st, 6.0 * half, (double)rows * H * 6 + 3.0 * H * H * 2 + (double)rows * H * 4, 31, batch, p->time_steps, H, H, H, 0, 0);
                static const int kb_ph = getenv("GS_INVERSE_GEMM_KB3") ? atoi(getenv("GS_INVERSE_GEMM_KB3")) : 2;
                if (kb_ph == 4 && H % 64 == 0) hipLaunchKernelGGL((gemm_bf16x6_kernel<2, 3, 4>), dim3(blocks_mag), dim3(256), 0, st, a_split, p->pinv_split, mag, M2, H, H, (int)rows);
                else hipLaunchKernelGGL((gemm_bf16x6_kernel<2, 3>), dim3(blocks_mag), dim3(256), 0, st, a_split, p->pinv_split, mag, M2, H, H, (int)rows);
            }
        } else {
            ProfScope ps(st, 12.0 * (double)rows * H * H, 2.0 * rows * H * 6 + 3.0 * H * H * 2 + 2.0 * rows * H * 4, 32, batch, p->time_steps, H, H, H, 0, 0);
            hipLaunchKernelGGL((gemm_bf16x6_kernel<2, 3>), dim3(blocks_all), dim3(256), 0, st, a_split, p->pinv_split, mag, M2, H, H, 0);
        }
    } else if ((2 * rows) % 128 == 0 && H % 128 == 0) {
        hipLaunchKernelGGL(gemm_f32_128_kernel, dim3((unsigned)((2 * rows / 128) * (H / 128))), dim3(256), 0, st, mel_mag, p->pinv, mag, (int)(2 * rows), H, H);
    } else {
        dim3 gg(cdiv(H, 64), cdiv(2 * rows, 64));
        hipLaunchKernelGGL(gemm_f32_kernel, gg, dim3(256), 0, st, mel_mag, p->pinv, mag, (int)(2 * rows), H, H);
    }
    GS_CHECK_LAUNCH();
    static const bool no_wave = getenv("GS_INVERSE_BLOCK_FFT") != nullptr;   // measurement knob: the block-per-frame radix-2 kernel
    static const bool no_ola = getenv("GS_INVERSE_SEPARATE_OLA") != nullptr;   // measurement knob: frames through memory + the gather kernel
    if (p->fast && H == 1024 && !no_wave && !no_ola && istft_wave_ola_ok(p, wave_len, front_pad))
        return launch_istft_wave_ola(p, mag, ph, wave, batch, wave_len, front_pad, st);   // overlap-add and crop inside: done
    if (p->fast && H == 1024 && !no_wave) {
        if (int e = launch_istft_wave(p, mag, ph, frames, rows, st)) return e;
    } else {
        hipLaunchKernelGGL(istft_kernel, dim3(p->time_steps, batch), dim3(256), 0, st, *p, mag, ph, frames);
        GS_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(overlap_add_kernel, dim3(cdiv((long)batch * wave_len, 256)), dim3(256), 0, st, *p, frames, wave, batch, wave_len, front_pad);
    GS_CHECK_LAUNCH();
    return 0;
}
