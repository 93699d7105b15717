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
// Persistent, software-pipelined kernel.  Block = 256 threads = 4 waves; every wave owns all
// 32*A output channels of the block and 32*B of its 128*B base pixels.  A block walks a list of
// work items (spatial tile x output-channel tile; the list of an XCD is contiguous so that
// neighbouring tiles share halo rows in that XCD's L2).  The K loop runs over input-channel chunks of
// 64 bytes (16 f32 / 32 bf16) x tap groups; while the MFMAs of one stage run, the global loads of the
// next stage (next tap group / next chunk / next ITEM) are already in flight into registers, and are
// written to the other half of a double-buffered LDS ring after the MFMAs -- one barrier per stage.
// LDS rows are 64 bytes, the four 16-byte slots of a row are XOR-swizzled with bits 2..3 of the row
// index, which makes the 16-lane ds_read_b128 groups conflict-free without padding.
// TG == 9 is the "resident weights" mode for the thin top-of-pyramid layers (32 output channels):
// all taps of all chunks stay in LDS for the life of the block and only the input patch streams.
struct ConvP {
    const void* x;
    const void* wp;
    void* y;
    const float* bias;  // optional fused epilogue: y = act(alpha * conv + bias)
    int act;
    int N, Hi, Wi, IC, OC, Hb, Wb, tiles_x, tiles_y, nsp, noct, nch;
    float alpha;
};

template <typename T, int MODE, int A, int B, int TW, int TG, bool RESIDENT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvP p) {
    constexpr int NP = 128 * B;
    constexpr int TH = NP / TW;
    constexpr int PH = patch_dim<MODE>(TH), PW = patch_dim<MODE>(TW);
    constexpr int S = MODE == MODE_S2 ? 2 : 1;
    constexpr int NPH = MODE == MODE_T2 ? 4 : 1;
    constexpr int BK = 64 / (int)sizeof(T);
    constexpr int OCT = 32 * A;
    constexpr int NTG = 9 / TG;
    constexpr int PCH = PH * PW * 4;    // 16-byte slots of a patch chunk
    constexpr int WCH = TG * OCT * 4;   // 16-byte slots of one weight stage
    constexpr int PREG = (PCH + 255) / 256, WREG = (WCH + 255) / 256;
    constexpr int PBYTES = PH * PW * 64, WBYTES = TG * OCT * 64;
    typedef typename Mma<T>::frag_t frag_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const lpatch = lds;                // 2 x PBYTES
    unsigned char* const lwgt = lds + 2 * PBYTES;     // streamed: 2 x WBYTES ; resident: nch x WBYTES

    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ wp = reinterpret_cast<const T*>(p.wp);
    T* __restrict__ y = reinterpret_cast<T*>(p.y);
    const int Hi = p.Hi, Wi = p.Wi, IC = p.IC, OC = p.OC, Hb = p.Hb, Wb = p.Wb, NCH = p.nch;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, hi = lane >> 5, l31 = lane & 31;

    // ---- this block's item list (XCD-contiguous when the grid is a multiple of 8)
    const int total = p.nsp * p.noct;
    int first, stride, count;
    if ((gridDim.x & 7) == 0) {
        const int per_xcd = (total + 7) >> 3, gx = gridDim.x >> 3;
        const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        first = xcd * per_xcd + loc;
        stride = gx;
        int end = (xcd + 1) * per_xcd;
        if (end > total) end = total;
        count = first < end ? (end - first + gx - 1) / gx : 0;
    } else {
        first = blockIdx.x;
        stride = gridDim.x;
        count = first < total ? (total - first + stride - 1) / stride : 0;
    }
    if (count == 0) return;

    uint4 preg[PREG], wreg[WREG];

    auto item_coords = [&](int item, int& n, int& by, int& bx, int& oc0) {
        const int sp = item / p.noct;
        oc0 = (item - sp * p.noct) * OCT;
        const int tile_x = sp % p.tiles_x;
        const int r = sp / p.tiles_x;
        by = (r % p.tiles_y) * TH;
        bx = tile_x * TW;
        n = r / p.tiles_y;
    };
    auto load_patch = [&](int item, int ch) {
        int n, by, bx, oc0;
        item_coords(item, n, by, bx, oc0);
        const int oy0 = MODE == MODE_S2 ? 2 * by : by - 1;
        const int ox0 = MODE == MODE_S2 ? 2 * bx : bx - 1;
        const T* xb = x + (long)n * Hi * Wi * IC + ch * BK;
#pragma unroll
        for (int r = 0; r < PREG; ++r) {
            const int c = tid + 256 * r;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c < PCH) {
                const int pix = c >> 2, part = c & 3;
                const int ly = pix / PW, lx = pix - ly * PW;
                const int iy = oy0 + ly, ix = ox0 + lx;
                if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi)
                    v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(xb + ((long)iy * Wi + ix) * IC) + part * 16);
            }
            preg[r] = v;
        }
    };
    auto store_patch = [&](int buf) {
        unsigned char* dst = lpatch + buf * PBYTES;
#pragma unroll
        for (int r = 0; r < PREG; ++r) {
            const int c = tid + 256 * r;
            if (c < PCH) {
                const int pix = c >> 2, part = c & 3;
                *reinterpret_cast<uint4*>(dst + pix * 64 + ((part ^ ((pix >> 2) & 3)) << 4)) = preg[r];
            }
        }
    };
    auto load_weights = [&](int oc0, int ch, int tg) {
#pragma unroll
        for (int r = 0; r < WREG; ++r) {
            const int c = tid + 256 * r;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c < WCH) {
                const int part = c & 3;
                const int row = (c >> 2) % OCT;
                const int i = tg * TG + (c >> 2) / OCT;
                if (oc0 + row < OC) {
                    const int wt = tap_ky<MODE>(i) * 3 + tap_kx<MODE>(i);
                    const T* src = wp + ((long)wt * OC + oc0 + row) * IC + ch * BK;
                    v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(src) + part * 16);
                }
            }
            wreg[r] = v;
        }
    };
    auto store_weights = [&](unsigned char* dst) {
#pragma unroll
        for (int r = 0; r < WREG; ++r) {
            const int c = tid + 256 * r;
            if (c < WCH) {
                const int R = c >> 2, part = c & 3;
                *reinterpret_cast<uint4*>(dst + R * 64 + ((part ^ ((R >> 2) & 3)) << 4)) = wreg[r];
            }
        }
    };

    f32x16 acc[NPH][A][B];
    auto zero_acc = [&]() {
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
            for (int a = 0; a < A; ++a)
#pragma unroll
                for (int b = 0; b < B; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ph][a][b][r] = 0.f;
    };

    int pixbase[B];  // patch pixel index of this lane's base pixel (tap offset added per tap)
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int q = (wv * B + b) * 32 + l31;
        pixbase[b] = ((q / TW) * S) * PW + (q % TW) * S;
    }
    const int aswz = (l31 >> 2) & 3;

    // ---- prologue
    int item = first, done = 0;
    {
        int n, by, bx, oc0;
        item_coords(item, n, by, bx, oc0);
        load_patch(item, 0);
        store_patch(0);
        if (RESIDENT) {
            for (int ch = 0; ch < NCH; ++ch) {
                load_weights(oc0, ch, 0);
                store_weights(lwgt + ch * WBYTES);
            }
        } else {
            load_weights(oc0, 0, 0);
            store_weights(lwgt);
        }
    }
    __syncthreads();
    zero_acc();
    int pb = 0, wb = 0;

    while (true) {
        int n, by, bx, oc0;
        item_coords(item, n, by, bx, oc0);
        for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) {
                // ---- what comes next, and its global loads (in flight during the MFMAs below)
                const bool last_tg = tg == NTG - 1;
                const bool last_ch = ch == NCH - 1;
                const bool has_next_item = done + 1 < count;
                const bool need_patch = last_tg && (!last_ch || has_next_item);
                const bool need_w = !RESIDENT && (!last_tg || !last_ch || has_next_item);
                if (need_patch) {
                    if (!last_ch) load_patch(item, ch + 1);
                    else load_patch(item + stride, 0);
                }
                if (need_w) {
                    if (!last_tg) load_weights(oc0, ch, tg + 1);
                    else if (!last_ch) load_weights(oc0, ch + 1, 0);
                    else {
                        const int nit = item + stride;
                        load_weights((nit - (nit / p.noct) * p.noct) * OCT, 0, 0);
                    }
                }
                // ---- MFMAs of this stage
                const unsigned char* lp = lpatch + pb * PBYTES;
                const unsigned char* lw = RESIDENT ? lwgt + ch * WBYTES : lwgt + wb * WBYTES;
#pragma unroll
                for (int tt = 0; tt < TG; ++tt) {
                    const int i = tg * TG + tt;
                    const int ph = tap_phase<MODE>(i);
                    const int toff = tap_off<MODE>(tap_ky<MODE>(i)) * PW + tap_off<MODE>(tap_kx<MODE>(i));
                    int boff[B], bswz[B];
#pragma unroll
                    for (int b = 0; b < B; ++b) {
                        const int pidx = pixbase[b] + toff;
                        boff[b] = pidx * 64;
                        bswz[b] = (pidx >> 2) & 3;
                    }
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        frag_t af[A], bf[B];
#pragma unroll
                        for (int a = 0; a < A; ++a)
                            af[a] = *reinterpret_cast<const frag_t*>(lw + (tt * OCT + a * 32 + l31) * 64 + (((ks * 2 + hi) ^ aswz) << 4));
#pragma unroll
                        for (int b = 0; b < B; ++b)
                            bf[b] = *reinterpret_cast<const frag_t*>(lp + boff[b] + (((ks * 2 + hi) ^ bswz[b]) << 4));
#pragma unroll
                        for (int a = 0; a < A; ++a)
#pragma unroll
                            for (int b = 0; b < B; ++b) Mma<T>::mma(af[a], bf[b], acc[ph][a][b]);
                    }
                }
                // ---- epilogue of the item: D[oc][pixel]; a lane holds oc = 8q + 4hi + (0..3) of pixel l31 per quad
                if (last_tg && last_ch) {
                    const int Ho = MODE == MODE_T2 ? 2 * Hb : Hb, Wo = MODE == MODE_T2 ? 2 * Wb : Wb;
#pragma unroll
                    for (int b = 0; b < B; ++b) {
                        const int q = (wv * B + b) * 32 + l31;
                        const int gy = by + q / TW, gx = bx + q % TW;
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
                                        for (int qd = 0; qd < 4; ++qd) {
                                            float o[4];
#pragma unroll
                                            for (int e = 0; e < 4; ++e) o[e] = acc[ph][a][b][qd * 4 + e] * p.alpha;
                                            if (p.bias) {
                                                const float4 bv = *reinterpret_cast<const float4*>(p.bias + oc0 + a * 32 + qd * 8 + hi * 4);
                                                o[0] += bv.x; o[1] += bv.y; o[2] += bv.z; o[3] += bv.w;
                                            }
                                            if (p.act == GS_ACT_LRELU) {
#pragma unroll
                                                for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : 0.2f * o[e];
                                            }
                                            st4(yp + a * 32 + qd * 8 + hi * 4, o);
                                        }
                                    }
                                }
                            }
                        }
                    }
                    zero_acc();
                }
                // ---- publish the prefetched stage into the other LDS buffers
                if (need_patch) store_patch(pb ^ 1);
                if (need_w) store_weights(lwgt + (wb ^ 1) * WBYTES);
                if (need_patch || need_w) __syncthreads();
                if (need_patch) pb ^= 1;
                if (need_w) wb ^= 1;
            }
        }
        if (++done >= count) break;
        item += stride;
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
// The contraction index is the PIXEL, while channels-last global tiles keep channels contiguous, so the
// tiles are TRANSPOSED while they are staged: LDS holds [channel][row][pixel] planes (pixel contiguous).
// A lane's 8 k-values are then one aligned ds_read_b128, and the three horizontal taps of a kernel
// row come from one 10-pixel window (b128 + b32) shifted with v_alignbit: 7 LDS reads + 12 VALU per
// 9 MFMAs (stride 2: even/odd column planes, 13 reads).  Staging: a thread takes 2 (4) adjacent pixels x 8
// channels, transposes 16-bit pairs with v_perm and issues 32-bit LDS writes; channel planes are pitched
// an odd multiple of 16 bytes apart so that the 16-lane read groups hit 16 distinct bank slots.
__device__ inline unsigned int perm_lo(unsigned int hi_src, unsigned int lo_src) { return __builtin_amdgcn_perm(hi_src, lo_src, 0x05040100u); }  // (lo_src.lo | hi_src.lo << 16)
__device__ inline unsigned int perm_hi(unsigned int hi_src, unsigned int lo_src) { return __builtin_amdgcn_perm(hi_src, lo_src, 0x07060302u); }  // (lo_src.hi | hi_src.hi << 16)
__device__ inline unsigned int shr16(unsigned int hi, unsigned int lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }
__device__ inline bf16x8 mk_frag(unsigned int a, unsigned int b, unsigned int c, unsigned int d) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v = {a, b, c, d};
    return __builtin_bit_cast(bf16x8, v);
}
__host__ __device__ constexpr int odd16(int bytes) { return ((bytes + 15) / 16) % 2 ? ((bytes + 15) / 16) * 16 : ((bytes + 15) / 16 + 1) * 16; }

template <int MODE, int TW>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ gy, float* __restrict__ part,
    int N, int Hi, int Wi, int IC, int OC, int Hb, int Wb, int tiles_x, int tiles_y, int ntiles, int nslices) {
    constexpr bool S2 = MODE == MODE_S2;
    constexpr int NP = S2 ? 128 : 256;
    constexpr int TH = NP / TW;
    constexpr int PH = patch_dim<MODE>(TH), PW = patch_dim<MODE>(TW);
    constexpr int XP = S2 ? TW + 8 : ((TW + 2 + 7) / 8) * 8;  // entries per row of the (even) x plane
    constexpr int OP = TW;                                     // entries per row of the odd plane (stride 2 only)
    constexpr int XCP = odd16((PH * XP + (S2 ? PH * OP : 0)) * 2);  // bytes between channel planes of x
    constexpr int GCP = odd16(NP * 2);                               // bytes between channel planes of gy
    constexpr int LDS_MAIN = 32 * XCP + 32 * GCP;
    constexpr int LDS_RED = 4 * 1024 * 4;
    constexpr int PPT = S2 ? 4 : 2;                 // pixels handled per staging work item
    constexpr int XG = (PW + PPT - 1) / PPT;        // pixel groups per patch row
    constexpr int XITEMS = PH * XG * 4;             // x 4 channel groups of 8
    constexpr int GITEMS = (NP / 2) * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_MAIN > LDS_RED ? LDS_MAIN : LDS_RED];
    unsigned char* const lx_ = lds_raw;
    unsigned char* const lg_ = lds_raw + 32 * XCP;
    float* lred = reinterpret_cast<float*>(lds_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int n_ict = IC / 32;
    const int ic0 = (blockIdx.x % n_ict) * 32, oc0 = (blockIdx.x / n_ict) * 32;
    const int slice = blockIdx.y;
    // staging lane map inside a wave: 16 pixel groups x 4 channel groups (cg slow) -> coalesced 64-byte global
    // segments per pixel and at most 2-way (free) conflicts on the 32-bit LDS writes
    const int s_pg = lane & 15, s_cg = lane >> 4;

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
        const int oy0 = S2 ? 2 * by : by - 1;
        const int ox0 = S2 ? 2 * bx : bx - 1;
        __syncthreads();
        // ---- x patch: transpose into [ic][row][pixel]
        for (int base = wv * 16; base < PH * XG; base += 64) {
            const int g = base + s_pg;
            if (g < PH * XG) {
                const int ly = g / XG, lx = (g - ly * XG) * PPT;
                const int iy = oy0 + ly;
                uint4 v[PPT];
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const int ix = ox0 + lx + k;
                    v[k] = make_uint4(0, 0, 0, 0);
                    if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi && lx + k < PW)
                        v[k] = *reinterpret_cast<const uint4*>(x + (((long)n * Hi + iy) * Wi + ix) * IC + ic0 + s_cg * 8);
                }
                unsigned char* dst = lx_ + (s_cg * 8) * XCP;
                if (!S2) {
                    const unsigned int* a = reinterpret_cast<const unsigned int*>(&v[0]);
                    const unsigned int* c = reinterpret_cast<const unsigned int*>(&v[PPT - 1]);
                    unsigned char* d = dst + (ly * XP + lx) * 2;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        *reinterpret_cast<unsigned int*>(d + (2 * j) * XCP) = perm_lo(c[j], a[j]);
                        *reinterpret_cast<unsigned int*>(d + (2 * j + 1) * XCP) = perm_hi(c[j], a[j]);
                    }
                } else {
                    const unsigned int* p0 = reinterpret_cast<const unsigned int*>(&v[0]);
                    const unsigned int* p1 = reinterpret_cast<const unsigned int*>(&v[1]);
                    const unsigned int* p2 = reinterpret_cast<const unsigned int*>(&v[2]);
                    const unsigned int* p3 = reinterpret_cast<const unsigned int*>(&v[3]);
                    unsigned char* de = dst + (ly * XP + (lx >> 1)) * 2;            // even columns lx, lx+2
                    unsigned char* dq = dst + (PH * XP + ly * OP + (lx >> 1)) * 2;  // odd columns lx+1, lx+3
                    const bool odd_ok = (lx >> 1) + 1 < OP + 1 && (lx >> 1) < OP;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        *reinterpret_cast<unsigned int*>(de + (2 * j) * XCP) = perm_lo(p2[j], p0[j]);
                        *reinterpret_cast<unsigned int*>(de + (2 * j + 1) * XCP) = perm_hi(p2[j], p0[j]);
                        if (odd_ok) {
                            *reinterpret_cast<unsigned int*>(dq + (2 * j) * XCP) = perm_lo(p3[j], p1[j]);
                            *reinterpret_cast<unsigned int*>(dq + (2 * j + 1) * XCP) = perm_hi(p3[j], p1[j]);
                        }
                    }
                }
            }
        }
        // ---- gy tile: transpose into [oc][pixel]
        for (int base = wv * 16; base < NP / 2; base += 64) {
            const int pp = base + s_pg;  // pixel pair
            const int p0 = pp * 2;
            const int gy_ = by + p0 / TW, gx_ = bx + p0 % TW;
            uint4 v0 = make_uint4(0, 0, 0, 0), v1 = make_uint4(0, 0, 0, 0);
            if (gy_ < Hb) {
                const bf16_t* src = gy + (((long)n * Hb + gy_) * Wb + gx_) * OC + oc0 + s_cg * 8;
                if (gx_ < Wb) v0 = *reinterpret_cast<const uint4*>(src);
                if (gx_ + 1 < Wb) v1 = *reinterpret_cast<const uint4*>(src + OC);
            }
            const unsigned int* a = reinterpret_cast<const unsigned int*>(&v0);
            const unsigned int* c = reinterpret_cast<const unsigned int*>(&v1);
            unsigned char* d = lg_ + (s_cg * 8) * GCP + p0 * 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *reinterpret_cast<unsigned int*>(d + (2 * j) * GCP) = perm_lo(c[j], a[j]);
                *reinterpret_cast<unsigned int*>(d + (2 * j + 1) * GCP) = perm_hi(c[j], a[j]);
            }
        }
        __syncthreads();
        // ---- MFMAs: wave wv takes pixel groups wv, wv+4, ...
        for (int g = wv; g < NP / 16; g += 4) {
            const int ty = (g * 16) / TW, tx0 = (g * 16) % TW + 8 * hi;
            const bf16x8 bfrag = *reinterpret_cast<const bf16x8*>(lg_ + l31 * GCP + (ty * TW + tx0) * 2);
            const unsigned char* xb = lx_ + l31 * XCP;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                if (!S2) {
                    const unsigned char* rp = xb + ((ty + ky) * XP + tx0) * 2;
                    const uint4 d = *reinterpret_cast<const uint4*>(rp);
                    const unsigned int d4 = *reinterpret_cast<const unsigned int*>(rp + 16);
                    acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(d.x, d.y, d.z, d.w), bfrag, acc[ky * 3 + 0], 0, 0, 0);
                    acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        mk_frag(shr16(d.y, d.x), shr16(d.z, d.y), shr16(d.w, d.z), shr16(d4, d.w)), bfrag, acc[ky * 3 + 1], 0, 0, 0);
                    acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(d.y, d.z, d.w, d4), bfrag, acc[ky * 3 + 2], 0, 0, 0);
                } else {
                    const int r = 2 * ty + ky;
                    const unsigned char* ep = xb + (r * XP + tx0) * 2;
                    const unsigned char* op = xb + (PH * XP + r * OP + tx0) * 2;
                    const uint4 e = *reinterpret_cast<const uint4*>(ep);
                    const unsigned int e4 = *reinterpret_cast<const unsigned int*>(ep + 16);
                    const uint4 o = *reinterpret_cast<const uint4*>(op);
                    acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(e.x, e.y, e.z, e.w), bfrag, acc[ky * 3 + 0], 0, 0, 0);
                    acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(o.x, o.y, o.z, o.w), bfrag, acc[ky * 3 + 1], 0, 0, 0);
                    acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        mk_frag(shr16(e.y, e.x), shr16(e.z, e.y), shr16(e.w, e.z), shr16(e4, e.w)), bfrag, acc[ky * 3 + 2], 0, 0, 0);
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

static int g_num_cus = 0;
static int num_cus() {
    if (g_num_cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            g_num_cus = n;
        else
            g_num_cus = 256;
    }
    return g_num_cus;
}

template <typename T, int MODE, int A, int B, int TW, int TG, bool RESIDENT = false>
static int launch_igemm(ConvP p, hipStream_t st) {
    constexpr int NP = 128 * B;
    constexpr int TH = NP / TW;
    constexpr int PH = patch_dim<MODE>(TH), PW = patch_dim<MODE>(TW);
    constexpr int OCT = 32 * A;
    constexpr int BK = 64 / (int)sizeof(T);
    p.tiles_x = cdiv(p.Wb, TW);
    p.tiles_y = cdiv(p.Hb, TH);
    p.nsp = p.N * p.tiles_x * p.tiles_y;
    p.noct = cdiv(p.OC, OCT);
    p.nch = p.IC / BK;
    const int wbufs = RESIDENT ? p.nch : 2;
    const size_t lds = (size_t)2 * PH * PW * 64 + (size_t)wbufs * TG * OCT * 64;
    if (lds > 160 * 1024) return fail(GS_ERR_UNSUPPORTED, "conv igemm: %zu bytes of LDS needed", lds);
    auto kern = conv_igemm_kernel<T, MODE, A, B, TW, TG, RESIDENT>;
    static size_t max_set = 0;  // per template instantiation
    if (lds > max_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return fail(GS_ERR_HIP, "conv igemm: cannot reserve %zu bytes of dynamic LDS", lds);
        max_set = lds;
    }
    // resident blocks per CU: LDS-limited, and at most 2 (accumulator-heavy kernels hold 1-2 waves per SIMD)
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    const long total = (long)p.nsp * p.noct;
    long grid = (long)per_cu * num_cus();
    if (grid > total) grid = total;
    if (grid >= 8) grid &= ~7L;
    const double flops = 2.0 * 9.0 * (double)p.N * p.Hb * p.Wb * p.IC * p.OC;
    ProfScope ps(st, flops);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, p);
    return 0;
}

// choose the tile configuration from (OC, Wb, problem size)
template <typename T, int MODE>
static int dispatch_igemm(ConvP p, hipStream_t st) {
    constexpr int BK = 64 / (int)sizeof(T);
    const int OC = p.OC, Wb = p.Wb;
    const int nch = p.IC / BK;
    const bool resident_ok = OC == 32 && nch <= 2;
    // number of 128-pixel tiles: prefer 128-wide oc tiles only when they still give >= 2 blocks per CU
    const long tiles128 = (long)p.N * cdiv(p.Hb, Wb >= 32 ? 4 : 8) * cdiv(Wb, Wb >= 32 ? 32 : 16);
    const bool wide = OC % 128 == 0 && tiles128 * (OC / 128) >= 2L * num_cus();
    if constexpr (MODE == MODE_T2) {
        if (resident_ok && Wb >= 64) return launch_igemm<T, MODE, 1, 2, 64, 9, true>(p, st);
        if (OC == 32) return launch_igemm<T, MODE, 1, 1, 32, 3>(p, st);
        if (Wb >= 32) return launch_igemm<T, MODE, 2, 1, 32, 3>(p, st);
        return launch_igemm<T, MODE, 2, 1, 16, 3>(p, st);
    } else if constexpr (MODE == MODE_S2) {
        if (OC == 32) return launch_igemm<T, MODE, 1, 1, 32, 3>(p, st);
        if (!wide) {
            if (Wb >= 32) return launch_igemm<T, MODE, 2, 1, 32, 3>(p, st);
            return launch_igemm<T, MODE, 2, 1, 16, 3>(p, st);
        }
        if (Wb >= 32) return launch_igemm<T, MODE, 4, 1, 32, 3>(p, st);
        return launch_igemm<T, MODE, 4, 1, 16, 3>(p, st);
    } else {
        if (resident_ok && Wb >= 64) return launch_igemm<T, MODE, 1, 2, 64, 9, true>(p, st);
        if (OC == 32) return launch_igemm<T, MODE, 1, 1, 32, 3>(p, st);
        if (OC % 64 == 0 && Wb >= 32 && (long)p.N * cdiv(p.Hb, 8) * cdiv(Wb, 32) * (OC / 64) >= num_cus() / 2)
            return launch_igemm<T, MODE, 2, 2, 32, 9>(p, st);
        if (!wide) {
            if (Wb >= 32) return launch_igemm<T, MODE, 2, 1, 32, 3>(p, st);
            return launch_igemm<T, MODE, 2, 1, 16, 3>(p, st);
        }
        if (Wb >= 32) return launch_igemm<T, MODE, 4, 1, 32, 3>(p, st);
        return launch_igemm<T, MODE, 4, 1, 16, 3>(p, st);
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
                       int ICk, int OCk, int w_ci, int w_co, int Hb, int Wb, float alpha, const float* bias, int act,
                       void* ws, size_t ws_bytes, hipStream_t st) {
    const size_t need = (size_t)9 * w_ci * w_co * sizeof(T);
    if (ws_bytes < need) return fail(GS_ERR_WORKSPACE, "conv igemm: workspace %zu < %zu", ws_bytes, need);
    T* wp = reinterpret_cast<T*>(ws);
    const long total = 9L * w_ci * w_co;
    hipLaunchKernelGGL((weight_prep_kernel<T>), dim3(cdiv(total, 256)), dim3(256), 0, st, w_hwio, wp, 9, w_ci, w_co, variant);
    ConvP p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.wp = wp; p.y = y; p.bias = bias; p.act = act;
    p.N = N; p.Hi = Hi; p.Wi = Wi; p.IC = ICk; p.OC = OCk; p.Hb = Hb; p.Wb = Wb; p.alpha = alpha;
    int rc;
    if (mode == MODE_S1) rc = dispatch_igemm<T, MODE_S1>(p, st);
    else if (mode == MODE_S2) rc = dispatch_igemm<T, MODE_S2>(p, st);
    else rc = dispatch_igemm<T, MODE_T2>(p, st);
    if (rc) return rc;
    GS_CHECK_LAUNCH();
    return 0;
}

int run_igemm(int mode, int variant, const void* x, const float* w_hwio, void* y, int N, int Hi, int Wi, int ICk,
              int OCk, int w_ci, int w_co, int Hb, int Wb, float alpha, const float* bias, int act, int dtype, void* ws,
              size_t ws_bytes, hipStream_t st) {
    GS_DISPATCH_DTYPE(dtype, return (run_igemm_t<T>(mode, variant, x, w_hwio, y, N, Hi, Wi, ICk, OCk, w_ci, w_co, Hb,
                                                    Wb, alpha, bias, act, ws, ws_bytes, st)));
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
