// MFMA implicit-GEMM 3x3 convolution family for gfx950 (channels-last activations).
//
// One LDS-tiled kernel template covers the three "gather-form" maps
//   MODE_S1 : 3x3 stride-1 SAME conv            (also its bwd-data, with flipped/transposed taps)
//   MODE_S2 : 3x3 stride-2 TF-SAME conv         (pad 0 before / 1 after on even inputs)
//   MODE_T2 : 3x3 stride-2 transposed conv      (= bwd-data of MODE_S2; 4 sub-pixel phases, no
//                                                zero insertion: 1+2+2+4 = 9 taps per 2x2 outputs)
// and a second template computes the weight gradient (K = pixels).
//
// GEMM orientation is "swapped": the MFMA A operand is the weight tile (rows = output channels),
// the B operand is the pixel tile (cols = pixels).  D[oc][pixel] then leaves each lane holding 4
// consecutive output channels of ONE pixel per accumulator quad, which is exactly a 16-byte
// channels-last store, and keeps the per-pixel channel reduction (pixel-norm) lane-local.
//
// Reference call sites replaced: tf.nn.conv2d ops.py:237-243 and tf.nn.conv2d_transpose
// ops.py:269-276 (plus the tf.gradients of both, models.py:47,60,81-89).
#include "conv_shared.h"

namespace gs {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));


// --------------------------------------------------------------------------- profiling hooks
struct ProfState {
    bool on = false;
    static constexpr int MAXEV = 8192;
    hipEvent_t ev[MAXEV][2];
    int created = 0;
    int used = 0;
    double flops = 0.0;
};
static ProfState g_prof;

struct ProfScope {
    hipStream_t s;
    int idx = -1;
    ProfScope(hipStream_t st, double flops) : s(st) {
        if (!g_prof.on || g_prof.used >= ProfState::MAXEV) return;
        idx = g_prof.used++;
        if (idx >= g_prof.created) {
            hipEventCreate(&g_prof.ev[idx][0]);
            hipEventCreate(&g_prof.ev[idx][1]);
            g_prof.created = idx + 1;
        }
        g_prof.flops += flops;
        hipEventRecord(g_prof.ev[idx][0], s);
    }
    ~ProfScope() {
        if (idx >= 0) hipEventRecord(g_prof.ev[idx][1], s);
    }
};

// ------------------------------------------------------------------------------- MFMA traits
template <typename T> struct Mma;
template <> struct Mma<float> {
    typedef f32x4 frag_t;  // 4 consecutive k for one row; substep e: lanes 0-31 carry k=e, 32-63 carry k=4+e
    __device__ static inline void mma(const frag_t& a, const frag_t& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c, 0, 0, 0);
    }
};
template <> struct Mma<bf16_t> {
    typedef bf16x8 frag_t;  // 8 consecutive k for one row
    __device__ static inline void mma(const frag_t& a, const frag_t& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

// -------------------------------------------------------------------------- mode geometry
template <int MODE> __host__ __device__ constexpr int patch_dim(int t) {
    return MODE == MODE_S1 ? t + 2 : (MODE == MODE_S2 ? 2 * t + 1 : t + 1);
}
// flat tap i in [0,9) -> kernel row/col, phase, LDS offset inside the patch
template <int MODE> __host__ __device__ constexpr int tap_ky(int i) {
    return MODE == MODE_T2 ? (i < 4 ? (i >> 1) * 2 : (i < 6 ? (i - 4) * 2 : 1)) : i / 3;
}
template <int MODE> __host__ __device__ constexpr int tap_kx(int i) {
    return MODE == MODE_T2 ? (i < 4 ? (i & 1) * 2 : (i < 6 ? 1 : (i < 8 ? (i - 6) * 2 : 1))) : i % 3;
}
template <int MODE> __host__ __device__ constexpr int tap_phase(int i) {
    return MODE == MODE_T2 ? (i < 4 ? 0 : (i < 6 ? 1 : (i < 8 ? 2 : 3))) : 0;
}
template <int MODE> __host__ __device__ constexpr int tap_off(int k) {  // patch offset for kernel index k
    return MODE == MODE_T2 ? (k == 2 ? 0 : 1) : k;
}

// ------------------------------------------------------------------------- implicit GEMM
// Block = 256 threads = 4 waves, every wave owns all 32*A output channels of the block and
// 32*B of its 128*B base pixels.  K loop: input-channel chunks of 64 bytes (16 f32 / 32 bf16),
// the halo'd input patch of a chunk is staged once and reused by all 9 taps.
template <typename T, int MODE, int A, int B, int TW, int TG>
__global__ __launch_bounds__(256) void conv_igemm_kernel(
    const T* __restrict__ x, const T* __restrict__ wp, T* __restrict__ y,
    int N, int Hi, int Wi, int IC, int OC, int Hb, int Wb, int tiles_x, int tiles_y, float alpha) {
    constexpr int NP = 128 * B;
    constexpr int TH = NP / TW;
    constexpr int PH = patch_dim<MODE>(TH), PW = patch_dim<MODE>(TW);
    constexpr int S = MODE == MODE_S2 ? 2 : 1;
    constexpr int NPH = MODE == MODE_T2 ? 4 : 1;
    constexpr int BKB = 64;
    constexpr int BK = BKB / (int)sizeof(T);
    constexpr int ROWB = BKB + 16;
    constexpr int CPP = BKB / 16;
    constexpr int OCT = 32 * A;
    typedef typename Mma<T>::frag_t frag_t;

    __shared__ __attribute__((aligned(16))) unsigned char lds[PH * PW * ROWB + TG * OCT * ROWB];
    unsigned char* lp = lds;
    unsigned char* lw = lds + PH * PW * ROWB;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    int bid = blockIdx.x;
    const int tile_x = bid % tiles_x;
    bid /= tiles_x;
    const int tile_y = bid % tiles_y;
    const int n = bid / tiles_y;
    const int oc0 = blockIdx.y * OCT;
    const int by = tile_y * TH, bx = tile_x * TW;
    const int oy0 = MODE == MODE_S2 ? 2 * by : by - 1;
    const int ox0 = MODE == MODE_S2 ? 2 * bx : bx - 1;

    f32x16 acc[NPH][A][B];
#pragma unroll
    for (int p = 0; p < NPH; ++p)
#pragma unroll
        for (int a = 0; a < A; ++a)
#pragma unroll
            for (int b = 0; b < B; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][a][b][r] = 0.f;

    int pixoff[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int p = (wv * B + b) * 32 + l31;
        const int ty = p / TW, tx = p % TW;
        pixoff[b] = ((ty * S) * PW + tx * S) * ROWB + hi * 16;
    }
    const int arow = l31 * ROWB + hi * 16;

    for (int ic0 = 0; ic0 < IC; ic0 += BK) {
        __syncthreads();
        // ---- stage the input patch chunk (zero outside the image)
        for (int c = tid; c < PH * PW * CPP; c += 256) {
            const int pix = c / CPP, part = c % CPP;
            const int ly = pix / PW, lx = pix % PW;
            const int iy = oy0 + ly, ix = ox0 + lx;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi) {
                const T* src = x + (((long)n * Hi + iy) * Wi + ix) * IC + ic0;
                v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(src) + part * 16);
            }
            *reinterpret_cast<uint4*>(lp + pix * ROWB + part * 16) = v;
        }
#pragma unroll
        for (int tg = 0; tg < 9; tg += TG) {
            if (tg > 0) __syncthreads();
            // ---- stage the weight rows of TG taps for this chunk
            for (int c = tid; c < TG * OCT * CPP; c += 256) {
                const int part = c % CPP;
                const int row = (c / CPP) % OCT;
                const int tt = c / (CPP * OCT);
                const int i = tg + tt;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (i < 9 && oc0 + row < OC) {
                    const int wt = tap_ky<MODE>(i) * 3 + tap_kx<MODE>(i);
                    const T* src = wp + ((long)wt * OC + oc0 + row) * IC + ic0;
                    v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(src) + part * 16);
                }
                *reinterpret_cast<uint4*>(lw + (tt * OCT + row) * ROWB + part * 16) = v;
            }
            __syncthreads();
#pragma unroll
            for (int tt = 0; tt < TG; ++tt) {
                const int i = tg + tt;
                if (i < 9) {
                    const int ph = tap_phase<MODE>(i);
                    const int toff = (tap_off<MODE>(tap_ky<MODE>(i)) * PW + tap_off<MODE>(tap_kx<MODE>(i))) * ROWB;
#pragma unroll
                    for (int ks = 0; ks < BKB / 32; ++ks) {
                        frag_t af[A], bf[B];
#pragma unroll
                        for (int a = 0; a < A; ++a)
                            af[a] = *reinterpret_cast<const frag_t*>(lw + (tt * OCT + a * 32) * ROWB + arow + ks * 32);
#pragma unroll
                        for (int b = 0; b < B; ++b)
                            bf[b] = *reinterpret_cast<const frag_t*>(lp + pixoff[b] + toff + ks * 32);
#pragma unroll
                        for (int a = 0; a < A; ++a)
#pragma unroll
                            for (int b = 0; b < B; ++b) Mma<T>::mma(af[a], bf[b], acc[ph][a][b]);
                    }
                }
            }
        }
    }

    // ---- epilogue: D[oc][pixel]; lane holds oc = 8q + 4hi + (0..3) of pixel l31 per accumulator quad
    const int Ho = MODE == MODE_T2 ? 2 * Hb : Hb, Wo = MODE == MODE_T2 ? 2 * Wb : Wb;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int p = (wv * B + b) * 32 + l31;
        const int gy = by + p / TW, gx = bx + p % TW;
        if (gy < Hb && gx < Wb) {
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) {
                const int oy = MODE == MODE_T2 ? 2 * gy + (ph >> 1) : gy;
                const int ox = MODE == MODE_T2 ? 2 * gx + (ph & 1) : gx;
                T* yp = y + (((long)n * Ho + oy) * Wo + ox) * OC + oc0;
#pragma unroll
                for (int a = 0; a < A; ++a) {
                    if (oc0 + a * 32 < OC) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float o[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = acc[ph][a][b][q * 4 + e] * alpha;
                            st4(yp + a * 32 + q * 8 + hi * 4, o);
                        }
                    }
                }
            }
        }
    }
}

// --------------------------------------------------------------------- weight gradient
// gw[tap][ic][oc] = sum_pixels x[in(pixel,tap)][ic] * gy[pixel][oc].  MFMA with K = pixels:
// A[i=ic][k=pixel], B[k=pixel][j=oc].  A block owns a 32x32 (ic,oc) tile for all 9 taps and
// strides over spatial tiles (`slice`); its 4 waves split each tile's pixels, then reduce through
// LDS and write one fp32 partial per slice (summed by wgrad_reduce_kernel -> deterministic).
template <typename T, int MODE, int TW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(
    const T* __restrict__ x, const T* __restrict__ gy, float* __restrict__ part,
    int N, int Hi, int Wi, int IC, int OC, int Hb, int Wb, int tiles_x, int tiles_y, int ntiles, int nslices) {
    // T = bf16: operands are widened to fp32 while staging (exact), the contraction runs on the fp32 MFMA.
    constexpr int NP = MODE == MODE_S2 ? 64 : 128;
    constexpr int TH = NP / TW;
    constexpr int PH = patch_dim<MODE>(TH), PW = patch_dim<MODE>(TW);
    constexpr int S = MODE == MODE_S2 ? 2 : 1;
    constexpr int ROWF = 32;  // floats per LDS row (32 channels)
    constexpr int LDS_MAIN = (PH * PW + NP) * ROWF;
    constexpr int LDS_RED = 4 * 1024;
    __shared__ __attribute__((aligned(16))) float lds[LDS_MAIN > LDS_RED ? LDS_MAIN : LDS_RED];
    float* lp = lds;
    float* lg = lds + PH * PW * ROWF;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int n_ict = IC / 32;
    const int ic0 = (blockIdx.x % n_ict) * 32, oc0 = (blockIdx.x / n_ict) * 32;
    const int slice = blockIdx.y;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int tile = slice; tile < ntiles; tile += nslices) {
        int b = tile;
        const int tile_x = b % tiles_x;
        b /= tiles_x;
        const int tile_y = b % tiles_y;
        const int n = b / tiles_y;
        const int by = tile_y * TH, bx = tile_x * TW;
        const int oy0 = MODE == MODE_S2 ? 2 * by : by - 1;
        const int ox0 = MODE == MODE_S2 ? 2 * bx : bx - 1;
        __syncthreads();
        for (int c = tid; c < PH * PW * 8; c += 256) {
            const int pix = c >> 3, part = c & 7;
            const int ly = pix / PW, lx = pix % PW;
            const int iy = oy0 + ly, ix = ox0 + lx;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi) ld4(x + (((long)n * Hi + iy) * Wi + ix) * IC + ic0 + part * 4, v);
            *reinterpret_cast<float4*>(lp + pix * ROWF + part * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
        for (int c = tid; c < NP * 8; c += 256) {
            const int pix = c >> 3, part = c & 7;
            const int gy_ = by + pix / TW, gx_ = bx + pix % TW;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (gy_ < Hb && gx_ < Wb) ld4(gy + (((long)n * Hb + gy_) * Wb + gx_) * OC + oc0 + part * 4, v);
            *reinterpret_cast<float4*>(lg + pix * ROWF + part * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
        __syncthreads();
#pragma unroll 2
        for (int pp = 0; pp < NP / 8; ++pp) {
            const int p = wv * (NP / 4) + 2 * pp + hi;
            const int ty = p / TW, tx = p % TW;
            const float bfrag = lg[p * ROWF + l31];
            const float* pbase = lp + ((ty * S) * PW + tx * S) * ROWF + l31;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float afrag = pbase[((t / 3) * PW + (t % 3)) * ROWF];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag, bfrag, acc[t], 0, 0, 0);
            }
        }
    }
    // ---- cross-wave reduction, one tap at a time: lds[wave][ic i][oc j]
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
            lds[wv * 1024 + i * 32 + l31] = acc[t][r];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = tid + 256 * k;
            const float s = lds[e] + lds[1024 + e] + lds[2048 + e] + lds[3072 + e];
            const int i = e >> 5, j = e & 31;
            part[(((long)slice * 9 + t) * IC + ic0 + i) * OC + oc0 + j] = s;
        }
    }
}

// bf16 weight gradient on the bf16 MFMA (32x32x16, K = 16 pixels per instruction).
// The contraction index is the PIXEL, but channels-last tiles keep channels contiguous, so each
// lane's 8 k-values live in 8 different LDS rows: they are fetched as 16-bit LDS reads and packed
// in registers.  The three horizontal taps of a kernel row read overlapping pixel windows
// (p+kx .. p+kx+7), so one row costs 10 reads (stride 1) / 17 reads (stride 2) for 3 MFMAs and the
// shifted fragments are rebuilt with v_alignbit -- 38 (59) LDS reads per 9 MFMAs, which balances
// the LDS pipe against the matrix pipe.
__device__ inline unsigned int pk16(unsigned short lo, unsigned short hi) { return (unsigned int)lo | ((unsigned int)hi << 16); }
__device__ inline unsigned int shr16(unsigned int hi, unsigned int lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }
__device__ inline bf16x8 mk_frag(unsigned int a, unsigned int b, unsigned int c, unsigned int d) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v = {a, b, c, d};
    return __builtin_bit_cast(bf16x8, v);
}

template <int MODE, int TW>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ gy, float* __restrict__ part,
    int N, int Hi, int Wi, int IC, int OC, int Hb, int Wb, int tiles_x, int tiles_y, int ntiles, int nslices) {
    constexpr int NP = MODE == MODE_S2 ? 128 : 256;
    constexpr int TH = NP / TW;
    constexpr int PH = patch_dim<MODE>(TH), PW = patch_dim<MODE>(TW);
    constexpr int S = MODE == MODE_S2 ? 2 : 1;
    constexpr int ROW = 32;  // bf16 per LDS row (32 channels = 64 B)
    constexpr int LDS_MAIN = (PH * PW + NP) * ROW * 2;
    constexpr int LDS_RED = 4 * 1024 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_MAIN > LDS_RED ? LDS_MAIN : LDS_RED];
    unsigned short* lp = reinterpret_cast<unsigned short*>(lds_raw);
    unsigned short* lg = lp + PH * PW * ROW;
    float* lred = reinterpret_cast<float*>(lds_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int n_ict = IC / 32;
    const int ic0 = (blockIdx.x % n_ict) * 32, oc0 = (blockIdx.x / n_ict) * 32;
    const int slice = blockIdx.y;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int tile = slice; tile < ntiles; tile += nslices) {
        int b = tile;
        const int tile_x = b % tiles_x;
        b /= tiles_x;
        const int tile_y = b % tiles_y;
        const int n = b / tiles_y;
        const int by = tile_y * TH, bx = tile_x * TW;
        const int oy0 = MODE == MODE_S2 ? 2 * by : by - 1;
        const int ox0 = MODE == MODE_S2 ? 2 * bx : bx - 1;
        __syncthreads();
        for (int c = tid; c < PH * PW * 4; c += 256) {
            const int pix = c >> 2, part4 = c & 3;
            const int ly = pix / PW, lx = pix % PW;
            const int iy = oy0 + ly, ix = ox0 + lx;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi)
                v = *reinterpret_cast<const uint4*>(x + (((long)n * Hi + iy) * Wi + ix) * IC + ic0 + part4 * 8);
            *reinterpret_cast<uint4*>(lp + pix * ROW + part4 * 8) = v;
        }
        for (int c = tid; c < NP * 4; c += 256) {
            const int pix = c >> 2, part4 = c & 3;
            const int gy_ = by + pix / TW, gx_ = bx + pix % TW;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (gy_ < Hb && gx_ < Wb)
                v = *reinterpret_cast<const uint4*>(gy + (((long)n * Hb + gy_) * Wb + gx_) * OC + oc0 + part4 * 8);
            *reinterpret_cast<uint4*>(lg + pix * ROW + part4 * 8) = v;
        }
        __syncthreads();
        for (int g = wv; g < NP / 16; g += 4) {
            const int ty = (g * 16) / TW, tx0 = (g * 16) % TW + 8 * hi;
            const unsigned short* gb = lg + (ty * TW + tx0) * ROW + l31;
            const bf16x8 bfrag = mk_frag(pk16(gb[0], gb[ROW]), pk16(gb[2 * ROW], gb[3 * ROW]), pk16(gb[4 * ROW], gb[5 * ROW]),
                                         pk16(gb[6 * ROW], gb[7 * ROW]));
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const unsigned short* pb = lp + ((ty * S + ky) * PW + tx0 * S) * ROW + l31;
                if (MODE == MODE_S2) {
                    unsigned int E[5], O[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        E[q] = pk16(pb[(4 * q) * ROW], pb[(4 * q + 2) * ROW]);
                        O[q] = pk16(pb[(4 * q + 1) * ROW], pb[(4 * q + 3) * ROW]);
                    }
                    E[4] = pb[16 * ROW];
                    acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(E[0], E[1], E[2], E[3]), bfrag, acc[ky * 3 + 0], 0, 0, 0);
                    acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(O[0], O[1], O[2], O[3]), bfrag, acc[ky * 3 + 1], 0, 0, 0);
                    acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        mk_frag(shr16(E[1], E[0]), shr16(E[2], E[1]), shr16(E[3], E[2]), shr16(E[4], E[3])), bfrag, acc[ky * 3 + 2], 0, 0, 0);
                } else {
                    unsigned int R[5];
#pragma unroll
                    for (int q = 0; q < 5; ++q) R[q] = pk16(pb[(2 * q) * ROW], pb[(2 * q + 1) * ROW]);
                    acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(R[0], R[1], R[2], R[3]), bfrag, acc[ky * 3 + 0], 0, 0, 0);
                    acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        mk_frag(shr16(R[1], R[0]), shr16(R[2], R[1]), shr16(R[3], R[2]), shr16(R[4], R[3])), bfrag, acc[ky * 3 + 1], 0, 0, 0);
                    acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(R[1], R[2], R[3], R[4]), bfrag, acc[ky * 3 + 2], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
            lred[wv * 1024 + i * 32 + l31] = acc[t][r];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = tid + 256 * k;
            const float s = lred[e] + lred[1024 + e] + lred[2048 + e] + lred[3072 + e];
            const int i = e >> 5, j = e & 31;
            part[(((long)slice * 9 + t) * IC + ic0 + i) * OC + oc0 + j] = s;
        }
    }
}

// ------------------------------------------------------------------------------ dispatch

template <typename T, int MODE, int A, int B, int TW, int TG>
static int launch_igemm(const T* x, const T* wp, T* y, int N, int Hi, int Wi, int IC, int OC, int Hb, int Wb,
                        float alpha, hipStream_t st) {
    constexpr int NP = 128 * B;
    constexpr int TH = NP / TW;
    const int tiles_x = cdiv(Wb, TW), tiles_y = cdiv(Hb, TH);
    dim3 grid((unsigned)((long)N * tiles_x * tiles_y), (unsigned)cdiv(OC, 32 * A));
    const double flops = 2.0 * 9.0 * (double)N * Hb * Wb * IC * OC * (MODE == MODE_T2 ? 1.0 : 1.0);
    ProfScope ps(st, flops);
    hipLaunchKernelGGL((conv_igemm_kernel<T, MODE, A, B, TW, TG>), grid, dim3(256), 0, st, x, wp, y, N, Hi, Wi, IC, OC,
                       Hb, Wb, tiles_x, tiles_y, alpha);
    return 0;
}

// choose the tile configuration from (OC, Wb)
template <typename T, int MODE>
static int dispatch_igemm(const T* x, const T* wp, T* y, int N, int Hi, int Wi, int IC, int OC, int Hb, int Wb,
                          float alpha, hipStream_t st) {
    if constexpr (MODE == MODE_T2) {
        if (OC == 32 && Wb >= 64) return launch_igemm<T, MODE, 1, 2, 64, 9>(x, wp, y, N, Hi, Wi, IC, OC, Hb, Wb, alpha, st);
        if (OC == 32) return launch_igemm<T, MODE, 1, 1, 32, 9>(x, wp, y, N, Hi, Wi, IC, OC, Hb, Wb, alpha, st);
        if (Wb >= 32) return launch_igemm<T, MODE, 2, 1, 32, 3>(x, wp, y, N, Hi, Wi, IC, OC, Hb, Wb, alpha, st);
        return launch_igemm<T, MODE, 2, 1, 16, 3>(x, wp, y, N, Hi, Wi, IC, OC, Hb, Wb, alpha, st);
    } else if constexpr (MODE == MODE_S2) {
        if (OC == 32) return launch_igemm<T, MODE, 1, 1, 32, 9>(x, wp, y, N, Hi, Wi, IC, OC, Hb, Wb, alpha, st);
        if (OC == 64 || OC % 128 != 0) {
            if (Wb >= 32) return launch_igemm<T, MODE, 2, 1, 32, 3>(x, wp, y, N, Hi, Wi, IC, OC, Hb, Wb, alpha, st);
            return launch_igemm<T, MODE, 2, 1, 16, 3>(x, wp, y, N, Hi, Wi, IC, OC, Hb, Wb, alpha, st);
        }
        if (Wb >= 32) return launch_igemm<T, MODE, 4, 1, 32, 3>(x, wp, y, N, Hi, Wi, IC, OC, Hb, Wb, alpha, st);
        return launch_igemm<T, MODE, 4, 1, 16, 3>(x, wp, y, N, Hi, Wi, IC, OC, Hb, Wb, alpha, st);
    } else {
    // MODE_S1
    if (OC == 32 && Wb >= 64) return launch_igemm<T, MODE, 1, 2, 64, 9>(x, wp, y, N, Hi, Wi, IC, OC, Hb, Wb, alpha, st);
    if (OC == 32) return launch_igemm<T, MODE, 1, 1, 32, 9>(x, wp, y, N, Hi, Wi, IC, OC, Hb, Wb, alpha, st);
    if ((OC == 64 || OC % 128 != 0) && Wb >= 64) return launch_igemm<T, MODE, 2, 2, 64, 3>(x, wp, y, N, Hi, Wi, IC, OC, Hb, Wb, alpha, st);
    if (OC == 64 || OC % 128 != 0) {
        if (Wb >= 32) return launch_igemm<T, MODE, 2, 1, 32, 3>(x, wp, y, N, Hi, Wi, IC, OC, Hb, Wb, alpha, st);
        return launch_igemm<T, MODE, 2, 1, 16, 3>(x, wp, y, N, Hi, Wi, IC, OC, Hb, Wb, alpha, st);
    }
    if (Wb >= 32) return launch_igemm<T, MODE, 4, 1, 32, 3>(x, wp, y, N, Hi, Wi, IC, OC, Hb, Wb, alpha, st);
    return launch_igemm<T, MODE, 4, 1, 16, 3>(x, wp, y, N, Hi, Wi, IC, OC, Hb, Wb, alpha, st);
    }
}

bool igemm_supported(int ic, int oc, int dtype) {
    const int bk = dtype == GS_F32 ? 16 : 32;
    return ic % bk == 0 && oc % 32 == 0;
}
bool wgrad_mfma_supported(int ic, int oc, int dtype) { return (dtype == GS_F32 || dtype == GS_BF16) && ic % 32 == 0 && oc % 32 == 0; }

size_t igemm_prep_bytes(int ic, int oc, int dtype) {
    return align256((size_t)9 * ic * oc * (dtype == GS_F32 ? 4 : 2));
}

// mode: MODE_*; variant: weight_prep variant; (ICk, OCk) are the kernel-role channel counts
template <typename T>
static int run_igemm_t(int mode, int variant, const void* x, const float* w_hwio, void* y, int N, int Hi, int Wi,
                       int ICk, int OCk, int w_ci, int w_co, int Hb, int Wb, float alpha, void* ws, size_t ws_bytes,
                       hipStream_t st) {
    const size_t need = (size_t)9 * w_ci * w_co * sizeof(T);
    if (ws_bytes < need) return fail(GS_ERR_WORKSPACE, "conv igemm: workspace %zu < %zu", ws_bytes, need);
    T* wp = reinterpret_cast<T*>(ws);
    const long total = 9L * w_ci * w_co;
    hipLaunchKernelGGL((weight_prep_kernel<T>), dim3(cdiv(total, 256)), dim3(256), 0, st, w_hwio, wp, 9, w_ci, w_co, variant);
    const T* xx = reinterpret_cast<const T*>(x);
    T* yy = reinterpret_cast<T*>(y);
    if (mode == MODE_S1) dispatch_igemm<T, MODE_S1>(xx, wp, yy, N, Hi, Wi, ICk, OCk, Hb, Wb, alpha, st);
    else if (mode == MODE_S2) dispatch_igemm<T, MODE_S2>(xx, wp, yy, N, Hi, Wi, ICk, OCk, Hb, Wb, alpha, st);
    else dispatch_igemm<T, MODE_T2>(xx, wp, yy, N, Hi, Wi, ICk, OCk, Hb, Wb, alpha, st);
    GS_CHECK_LAUNCH();
    return 0;
}

int run_igemm(int mode, int variant, const void* x, const float* w_hwio, void* y, int N, int Hi, int Wi, int ICk,
              int OCk, int w_ci, int w_co, int Hb, int Wb, float alpha, int dtype, void* ws, size_t ws_bytes,
              hipStream_t st) {
    GS_DISPATCH_DTYPE(dtype, return (run_igemm_t<T>(mode, variant, x, w_hwio, y, N, Hi, Wi, ICk, OCk, w_ci, w_co, Hb,
                                                    Wb, alpha, ws, ws_bytes, st)));
}

// ---- weight gradient (fp32 MFMA path)
static void wgrad_geometry(int mode, int dtype, int N, int Hb, int Wb, int IC, int OC, int* tw, int* tiles_x, int* tiles_y,
                           int* ntiles, int* nslices) {
    const int np = (mode == MODE_S2 ? 64 : 128) * (dtype == GS_BF16 ? 2 : 1);
    *tw = Wb >= 32 ? 32 : 16;
    const int th = np / *tw;
    *tiles_x = cdiv(Wb, *tw);
    *tiles_y = cdiv(Hb, th);
    *ntiles = N * *tiles_x * *tiles_y;
    const int pairs = (IC / 32) * (OC / 32);
    int ns = 512 / pairs;
    if (ns < 1) ns = 1;
    if (ns > *ntiles) ns = *ntiles;
    *nslices = ns;
}

size_t wgrad_mfma_bytes(int mode, int dtype, int N, int Hb, int Wb, int IC, int OC) {
    int tw, tx, ty, nt, ns;
    wgrad_geometry(mode, dtype, N, Hb, Wb, IC, OC, &tw, &tx, &ty, &nt, &ns);
    return align256((size_t)ns * 9 * IC * OC * sizeof(float));
}

// x: conv input side [N][Hi][Wi][IC]; gy: [N][Hb][Wb][OC]; gw[9][IC][OC] (or transposed)
int run_wgrad_mfma(int mode, const void* x, const void* gy, float* gw, int N, int Hi, int Wi, int IC, int OC, int Hb,
                   int Wb, float alpha, int transpose, int dtype, void* ws, size_t ws_bytes, hipStream_t st) {
    int tw, tiles_x, tiles_y, ntiles, nslices;
    wgrad_geometry(mode, dtype, N, Hb, Wb, IC, OC, &tw, &tiles_x, &tiles_y, &ntiles, &nslices);
    const size_t need = (size_t)nslices * 9 * IC * OC * sizeof(float);
    if (ws_bytes < need) return fail(GS_ERR_WORKSPACE, "conv wgrad: workspace %zu < %zu", ws_bytes, need);
    float* part = reinterpret_cast<float*>(ws);
    dim3 grid((IC / 32) * (OC / 32), nslices);
    {
        ProfScope ps(st, 2.0 * 9.0 * (double)N * Hb * Wb * IC * OC);
#define GS_WG(TT, M, TWV)                                                                                              \
    hipLaunchKernelGGL((conv_wgrad_kernel<TT, M, TWV>), grid, dim3(256), 0, st, reinterpret_cast<const TT*>(x),        \
                       reinterpret_cast<const TT*>(gy), part, N, Hi, Wi, IC, OC, Hb, Wb, tiles_x, tiles_y, ntiles, nslices)
#define GS_WG_ALL(TT)                                                                       \
    do {                                                                                    \
        if (mode == MODE_S1) { if (tw == 32) GS_WG(TT, MODE_S1, 32); else GS_WG(TT, MODE_S1, 16); } \
        else { if (tw == 32) GS_WG(TT, MODE_S2, 32); else GS_WG(TT, MODE_S2, 16); }          \
    } while (0)
        if (dtype == GS_F32) {
            GS_WG_ALL(float);
        } else {
#define GS_WGB(M, TWV)                                                                                                  \
    hipLaunchKernelGGL((conv_wgrad_bf16_kernel<M, TWV>), grid, dim3(256), 0, st, reinterpret_cast<const bf16_t*>(x),    \
                       reinterpret_cast<const bf16_t*>(gy), part, N, Hi, Wi, IC, OC, Hb, Wb, tiles_x, tiles_y, ntiles, nslices)
            if (mode == MODE_S1) { if (tw == 32) GS_WGB(MODE_S1, 32); else GS_WGB(MODE_S1, 16); }
            else { if (tw == 32) GS_WGB(MODE_S2, 32); else GS_WGB(MODE_S2, 16); }
#undef GS_WGB
        }
#undef GS_WG_ALL
#undef GS_WG
    }
    GS_CHECK_LAUNCH();
    const long total = 9L * IC * OC;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, 64)), dim3(256), 0, st, part, gw, nslices, 9, IC, OC, alpha, transpose);
    GS_CHECK_LAUNCH();
    return 0;
}

}  // namespace gs

extern "C" int gs_prof_enable(int on) {
    gs::g_prof.on = on != 0;
    gs::g_prof.used = 0;
    gs::g_prof.flops = 0.0;
    return 0;
}

extern "C" int gs_prof_collect(int* launches, double* total_ms, double* total_flops) {
    double ms = 0.0;
    for (int i = 0; i < gs::g_prof.used; ++i) {
        hipEventSynchronize(gs::g_prof.ev[i][1]);
        float t = 0.f;
        if (hipEventElapsedTime(&t, gs::g_prof.ev[i][0], gs::g_prof.ev[i][1]) == hipSuccess) ms += t;
    }
    if (launches) *launches = gs::g_prof.used;
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = gs::g_prof.flops;
    gs::g_prof.used = 0;
    gs::g_prof.flops = 0.0;
    return 0;
}
