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
    float* const lbias = reinterpret_cast<float*>(lwgt + (RESIDENT ? p.nch : 2) * WBYTES);  // OC floats (zeros without a bias)

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

    uint4 preg[PREG], wreg[WREG];  // (zero-initialised: conditionally-assigned struct arrays otherwise end up in scratch)
#pragma unroll
    for (int r = 0; r < WREG; ++r) wreg[r] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < PREG; ++r) preg[r] = make_uint4(0, 0, 0, 0);
    unsigned int pmask = 0;  // which patch slots of the prefetched stage lie inside the image

    // ---- per-thread staging descriptors, constant for the life of the block (keeps the per-stage address
    //      arithmetic down to one add + two unsigned compares per 16-byte slot)
    int p_ll[PREG], p_off[PREG], p_lds[PREG];
#pragma unroll
    for (int r = 0; r < PREG; ++r) {
        const int c = tid + 256 * r;
        const int pix = c >> 2, part = c & 3;
        const int ly = pix / PW, lx = pix - ly * PW;
        const bool ok = c < PCH;
        p_ll[r] = ok ? ((ly << 16) | lx) : 0x7fff0000;
        p_off[r] = ((ly * Wi + lx) * IC) * (int)sizeof(T) + part * 16;
        p_lds[r] = ok ? pix * 64 + ((part ^ ((lx >> 2) & 3)) << 4) : -1;
    }
    int w_off[WREG], w_lds[WREG];
#pragma unroll
    for (int r = 0; r < WREG; ++r) {
        const int c = tid + 256 * r;
        const int R = c >> 2, part = c & 3;
        const int row = R % OCT, tt = R / OCT;
        const bool ok = c < WCH;
        if (MODE == MODE_T2) w_off[r] = ok ? ((tt << 16) | row) : -1;  // tap order is a permutation: resolved per stage
        else w_off[r] = ok ? (tt * OC + row) * IC * (int)sizeof(T) + part * 16 : -1;
        w_lds[r] = ok ? R * 64 + ((part ^ ((R >> 2) & 3)) << 4) : -1;
    }

    auto item_coords = [&](int item, int& n, int& by, int& bx, int& oc0) __attribute__((always_inline)) {
        const int sp = item / p.noct;
        oc0 = (item - sp * p.noct) * OCT;
        const int tile_x = sp % p.tiles_x;
        const int r = sp / p.tiles_x;
        by = (r % p.tiles_y) * TH;
        bx = tile_x * TW;
        n = r / p.tiles_y;
    };
    auto load_patch = [&](int item, int ch) __attribute__((always_inline)) {
        int n, by, bx, oc0;
        item_coords(item, n, by, bx, oc0);
        const int oy0 = MODE == MODE_S2 ? 2 * by : by - 1;
        const int ox0 = MODE == MODE_S2 ? 2 * bx : bx - 1;
        const unsigned char* xb = reinterpret_cast<const unsigned char*>(x) +
                                  ((((long)n * Hi + oy0) * Wi + ox0) * IC + ch * BK) * (long)sizeof(T);
        // NOTE: every load is unconditional (out-of-image slots read the tensor base and are zeroed when they are
        // written to LDS).  A per-slot `if (inside) load` makes hipcc branch around each load and drain vmcnt(0)
        // per slot, which serialises the whole prefetch into dependent L2 round trips.
        pmask = 0;
#pragma unroll
        for (int r = 0; r < PREG; ++r) {
            const bool ok = (unsigned)(oy0 + (p_ll[r] >> 16)) < (unsigned)Hi && (unsigned)(ox0 + (p_ll[r] & 0xffff)) < (unsigned)Wi;
            pmask |= ok ? (1u << r) : 0u;
            const unsigned char* src = ok ? xb + p_off[r] : reinterpret_cast<const unsigned char*>(x);
            preg[r] = *reinterpret_cast<const uint4*>(src);
        }
    };
    auto store_patch = [&](int buf) __attribute__((always_inline)) {
        unsigned char* dst = lpatch + buf * PBYTES;
#pragma unroll
        for (int r = 0; r < PREG; ++r) {
            const uint4 v = (pmask >> r) & 1u ? preg[r] : make_uint4(0, 0, 0, 0);
            if (PCH % 256 == 0 || r < PREG - 1 || p_lds[r] >= 0) *reinterpret_cast<uint4*>(dst + (p_lds[r] >= 0 ? p_lds[r] : 0)) = v;
        }
    };
    auto load_weights = [&](int oc0, int ch, int tg) __attribute__((always_inline)) {
        if (MODE != MODE_T2) {
            const unsigned char* wb = reinterpret_cast<const unsigned char*>(wp) +
                                      (((long)tg * TG * OC + oc0) * IC + ch * BK) * (long)sizeof(T);
#pragma unroll
            for (int r = 0; r < WREG; ++r)  // unconditional (slots past the end re-read slot 0 and are not stored)
                wreg[r] = *reinterpret_cast<const uint4*>(wb + (w_off[r] >= 0 ? w_off[r] : 0));
        } else {
#pragma unroll
            for (int r = 0; r < WREG; ++r) {
                const int wo = w_off[r] >= 0 ? w_off[r] : 0;
                const int i = tg * TG + (wo >> 16), row = wo & 0xffff;
                const int wt = tap_ky<MODE>(i) * 3 + tap_kx<MODE>(i);
                const T* src = wp + ((long)wt * OC + oc0 + row) * IC + ch * BK;
                wreg[r] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(src) + ((tid + 256 * r) & 3) * 16);
            }
        }
    };
    auto store_weights = [&](unsigned char* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < WREG; ++r)
            if (WCH % 256 == 0 || r < WREG - 1 || w_lds[r] >= 0) *reinterpret_cast<uint4*>(dst + (w_lds[r] >= 0 ? w_lds[r] : 0)) = wreg[r];
    };

    f32x16 acc[NPH][A][B];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
            for (int a = 0; a < A; ++a)
#pragma unroll
                for (int b = 0; b < B; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ph][a][b][r] = 0.f;
    };

    // fragment byte offsets with the slot swizzle folded in: B side per (pixel group, horizontal tap offset, k step),
    // A side per k step; the vertical tap offset and the tap / channel-tile row are compile-time immediates
    int b_off[B][3][2], a_off[2];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int q = (wv * B + b) * 32 + l31;
        const int pb0 = ((q / TW) * S) * PW + (q % TW) * S;
#pragma unroll
        for (int ox = 0; ox < 3; ++ox) {
            const int key = (((q % TW) * S + ox) >> 2) & 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) b_off[b][ox][ks] = (pb0 + ox) * 64 + (((ks * 2 + hi) ^ key) << 4);
        }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a_off[ks] = l31 * 64 + (((ks * 2 + hi) ^ ((l31 >> 2) & 3)) << 4);

    for (int c = tid; c < OC; c += 256) lbias[c] = p.bias ? p.bias[c] : 0.f;

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
                // fragment reads are software-pipelined one (tap, k-step) ahead of the MFMAs that consume them: with one
                // or two waves per SIMD nothing else hides the LDS latency (the compiler issues them just-in-time)
                {
                    frag_t af[2][A], bf[2][B];
                    auto load_frags = [&](int step, int buf) __attribute__((always_inline)) {
                        const int tt = step >> 1, ks = step & 1;
                        const int i = tg * TG + tt;
                        const int oyv = tap_off<MODE>(tap_ky<MODE>(i)), oxv = tap_off<MODE>(tap_kx<MODE>(i));
#pragma unroll
                        for (int a = 0; a < A; ++a)
                            af[buf][a] = *reinterpret_cast<const frag_t*>(lw + (tt * OCT + a * 32) * 64 + a_off[ks]);
#pragma unroll
                        for (int b = 0; b < B; ++b)
                            bf[buf][b] = *reinterpret_cast<const frag_t*>(lp + oyv * PW * 64 + b_off[b][oxv][ks]);
                    };
                    load_frags(0, 0);
#pragma unroll
                    for (int step = 0; step < 2 * TG; ++step) {
                        if (step + 1 < 2 * TG) load_frags(step + 1, (step + 1) & 1);
                        const int ph = tap_phase<MODE>(tg * TG + (step >> 1));
#pragma unroll
                        for (int a = 0; a < A; ++a)
#pragma unroll
                            for (int b = 0; b < B; ++b) Mma<T>::mma(af[step & 1][a], bf[step & 1][b], acc[ph][a][b]);
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
                                            const float4 bv = *reinterpret_cast<const float4*>(lbias + oc0 + a * 32 + qd * 8 + hi * 4);
                                            o[0] += bv.x; o[1] += bv.y; o[2] += bv.z; o[3] += bv.w;
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
            float v[4];
            const bool ok = iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;
            ld4(ok ? x + (((long)n * Hi + iy) * Wi + ix) * IC + ic0 + part * 4 : x, v);  // unconditional load, zero-select after
            *reinterpret_cast<float4*>(lp + pix * ROWF + part * 4) = ok ? make_float4(v[0], v[1], v[2], v[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int c = tid; c < NP * 8; c += 256) {
            const int pix = c >> 3, part = c & 7;
            const int gy_ = by + pix / TW, gx_ = bx + pix % TW;
            float v[4];
            const bool ok = gy_ < Hb && gx_ < Wb;
            ld4(ok ? gy + (((long)n * Hb + gy_) * Wb + gx_) * OC + oc0 + part * 4 : gy, v);
            *reinterpret_cast<float4*>(lg + pix * ROWF + part * 4) = ok ? make_float4(v[0], v[1], v[2], v[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
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
// The contraction index is the PIXEL while channels-last tiles keep channels contiguous, i.e. the operands
// sit K-major in LDS ([pixel][32 channels], 64-byte rows, staged with plain 16-byte copies).  gfx950's
// transposing LDS read ds_read_b64_tr_b16 turns that into K-contiguous fragments for free.  Measured
// semantics (scripts/probe/tr_probe.hip): inside each 16-lane group, lane s supplies the address of 4
// consecutive b16 (8 bytes); lane i = 4m + pos receives, for j = 0..3, element `pos` of the data supplied
// by lane 4j + m.  With lane s pointing at (pixel row k0 + (s >> 2), channels 4(s & 3)..+3) every lane gets
// 4 consecutive pixels of its own channel; two reads = one MFMA operand.  The three horizontal taps of a
// kernel row use overlapping pixel windows, so 3 reads (12 pixels) + 4 v_alignbit feed 3 MFMAs.
// Block = 192 threads = 3 waves; wave w owns kernel row ky = w (3 taps, 48 fp32 accumulators) over all pixels
// of the tile: no cross-wave reduction, small register footprint, 3 blocks per CU.
__device__ inline unsigned int shr16(unsigned int hi, unsigned int lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }
__device__ inline bf16x8 mk_frag(unsigned int a, unsigned int b, unsigned int c, unsigned int d) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v = {a, b, c, d};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ inline uint2 lds_tr16(const unsigned char* p) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(reinterpret_cast<const s16x4*>(p)));
    return __builtin_bit_cast(uint2, v);
}

template <int MODE, int TW>
__global__ __launch_bounds__(192) void conv_wgrad_bf16_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ gy, float* __restrict__ part,
    int N, int Hi, int Wi, int IC, int OC, int Hb, int Wb, int tiles_x, int tiles_y, int ntiles, int nslices) {
    constexpr bool S2 = MODE == MODE_S2;
    constexpr int NP = S2 ? 128 : 256;
    constexpr int TH = NP / TW;
    constexpr int PH = patch_dim<MODE>(TH), PW = patch_dim<MODE>(TW);
    constexpr int S = S2 ? 2 : 1;
    constexpr int XCH = PH * PW * 4, GCH = NP * 4;          // 16-byte chunks to stage
    constexpr int XIT = (XCH + 191) / 192, GIT = (GCH + 191) / 192;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[(PH * PW + NP) * 64];
    unsigned char* const lx_ = lds_raw;
    unsigned char* const lg_ = lds_raw + PH * PW * 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int n_ict = IC / 32;
    const int ic0 = (blockIdx.x % n_ict) * 32, oc0 = (blockIdx.x / n_ict) * 32;
    const int slice = blockIdx.y;
    // transposing-read supplier role of this lane: pixel row (lane & 15) >> 2 of the 4-row block, channel quad
    const int t_row = (lane & 15) >> 2;
    const int t_col = (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;  // byte offset inside the 64-byte row

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
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
        uint4 xv[XIT], gv[GIT];
        unsigned int xok = 0, gok = 0;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int c = tid + 192 * it;
            const int pix = c >> 2, part4 = c & 3;
            const int ly = pix / PW, lx = pix - ly * PW;
            const int iy = oy0 + ly, ix = ox0 + lx;
            const bool ok = c < XCH && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
            xok |= ok ? (1u << it) : 0u;   // loads are unconditional; out-of-image slots are zeroed at the LDS store
            xv[it] = *reinterpret_cast<const uint4*>(ok ? x + (((long)n * Hi + iy) * Wi + ix) * IC + ic0 + part4 * 8 : x);
        }
#pragma unroll
        for (int it = 0; it < GIT; ++it) {
            const int c = tid + 192 * it;
            const int pix = c >> 2, part4 = c & 3;
            const int gy_ = by + pix / TW, gx_ = bx + pix % TW;
            const bool ok = c < GCH && gy_ < Hb && gx_ < Wb;
            gok |= ok ? (1u << it) : 0u;
            gv[it] = *reinterpret_cast<const uint4*>(ok ? gy + (((long)n * Hb + gy_) * Wb + gx_) * OC + oc0 + part4 * 8 : gy);
        }
        __syncthreads();  // every wave is done reading the previous tile
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int c = tid + 192 * it;
            if (c < XCH) *reinterpret_cast<uint4*>(lx_ + c * 16) = (xok >> it) & 1u ? xv[it] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < GIT; ++it) {
            const int c = tid + 192 * it;
            if (c < GCH) *reinterpret_cast<uint4*>(lg_ + c * 16) = (gok >> it) & 1u ? gv[it] : make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        // ---- MFMAs: this wave's kernel row (ky = wv) over every 16-pixel group of the tile
#pragma unroll 2
        for (int g = 0; g < NP / 16; ++g) {
            const int ty = (g * 16) / TW, tx0 = (g * 16) % TW + 8 * hi;
            const unsigned char* gp = lg_ + (ty * TW + tx0 + t_row) * 64 + t_col;
            const uint2 b0 = lds_tr16(gp), b1 = lds_tr16(gp + 4 * 64);
            const bf16x8 bfrag = mk_frag(b0.x, b0.y, b1.x, b1.y);
            const unsigned char* xp = lx_ + (((ty * S + wv) * PW + tx0 * S) + t_row * S) * 64 + t_col;
            if (!S2) {
                const uint2 d0 = lds_tr16(xp), d1 = lds_tr16(xp + 4 * 64), d2 = lds_tr16(xp + 8 * 64);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(d0.x, d0.y, d1.x, d1.y), bfrag, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(shr16(d0.y, d0.x), shr16(d1.x, d0.y), shr16(d1.y, d1.x), shr16(d2.x, d1.y)), bfrag, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(d0.y, d1.x, d1.y, d2.x), bfrag, acc[2], 0, 0, 0);
            } else {
                // even columns 2(p)+0 / +2 share a 9-pixel window; odd columns 2(p)+1 are their own 8-pixel window
                const uint2 e0 = lds_tr16(xp), e1 = lds_tr16(xp + 8 * 64), e2 = lds_tr16(xp + 16 * 64);
                const uint2 o0 = lds_tr16(xp + 64), o1 = lds_tr16(xp + 9 * 64);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(e0.x, e0.y, e1.x, e1.y), bfrag, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(o0.x, o0.y, o1.x, o1.y), bfrag, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(shr16(e0.y, e0.x), shr16(e1.x, e0.y), shr16(e1.y, e1.x), shr16(e2.x, e1.y)), bfrag, acc[2], 0, 0, 0);
            }
        }
    }
    // ---- each wave owns its 3 taps: D[ic i][oc j], lane = (j = l31, i = (r&3) + 8(r>>2) + 4hi)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        float* dst = part + (((long)slice * 9 + wv * 3 + kx) * IC + ic0) * OC + oc0 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(long)((r & 3) + 8 * (r >> 2) + 4 * hi) * OC] = acc[kx][r];
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
    const size_t lds = (size_t)2 * PH * PW * 64 + (size_t)wbufs * TG * OCT * 64 + (size_t)((p.OC + 3) / 4) * 16;
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
                       int w_prepared, void* ws, size_t ws_bytes, hipStream_t st) {
    const size_t need = (size_t)9 * w_ci * w_co * sizeof(T);
    if (ws_bytes < need) return fail(GS_ERR_WORKSPACE, "conv igemm: workspace %zu < %zu", ws_bytes, need);
    T* wp = reinterpret_cast<T*>(ws);
    const long total = 9L * w_ci * w_co;
    if (!w_prepared)
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
              int OCk, int w_ci, int w_co, int Hb, int Wb, float alpha, const float* bias, int act, int dtype, int w_prepared,
              void* ws, size_t ws_bytes, hipStream_t st) {
    GS_DISPATCH_DTYPE(dtype, return (run_igemm_t<T>(mode, variant, x, w_hwio, y, N, Hi, Wi, ICk, OCk, w_ci, w_co, Hb,
                                                    Wb, alpha, bias, act, w_prepared, ws, ws_bytes, st)));
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
    int ns = (dtype == GS_BF16 ? 768 : 512) / pairs;
    if (ns < 1) ns = 1;
    if (ns > *ntiles) ns = *ntiles;
    *nslices = ns;
}

size_t wgrad_mfma_bytes(int mode, int dtype, int N, int Hb, int Wb, int IC, int OC) {
    int tw, tx, ty, nt, ns;
    wgrad_geometry(mode, dtype, N, Hb, Wb, IC, OC, &tw, &tx, &ty, &nt, &ns);
    return align256(((size_t)ns * 9 * IC * OC + wgrad_reduce_extra(ns, 9L * IC * OC)) * sizeof(float));
}

// x: conv input side [N][Hi][Wi][IC]; gy: [N][Hb][Wb][OC]; gw[9][IC][OC] (or transposed)
int run_wgrad_mfma(int mode, const void* x, const void* gy, float* gw, int N, int Hi, int Wi, int IC, int OC, int Hb,
                   int Wb, float alpha, int transpose, int accumulate, int dtype, void* ws, size_t ws_bytes, hipStream_t st) {
    int tw, tiles_x, tiles_y, ntiles, nslices;
    wgrad_geometry(mode, dtype, N, Hb, Wb, IC, OC, &tw, &tiles_x, &tiles_y, &ntiles, &nslices);
    const size_t need = ((size_t)nslices * 9 * IC * OC + wgrad_reduce_extra(nslices, 9L * IC * OC)) * sizeof(float);
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
    hipLaunchKernelGGL((conv_wgrad_bf16_kernel<M, TWV>), grid, dim3(192), 0, st, reinterpret_cast<const bf16_t*>(x),    \
                       reinterpret_cast<const bf16_t*>(gy), part, N, Hi, Wi, IC, OC, Hb, Wb, tiles_x, tiles_y, ntiles, nslices)
            if (mode == MODE_S1) { if (tw == 32) GS_WGB(MODE_S1, 32); else GS_WGB(MODE_S1, 16); }
            else { if (tw == 32) GS_WGB(MODE_S2, 32); else GS_WGB(MODE_S2, 16); }
#undef GS_WGB
        }
#undef GS_WG_ALL
#undef GS_WG
    }
    GS_CHECK_LAUNCH();
    wgrad_reduce_launch(part, gw, nslices, 9, IC, OC, alpha, transpose, accumulate, st);
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
