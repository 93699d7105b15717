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
#include <type_traits>
#include "conv_shared.h"
#include "gs_prof.h"

#ifndef GS_WGRAD_THIN_PREFETCH
#define GS_WGRAD_THIN_PREFETCH 1   // conv_wgrad_bf16_kernel: loads of the next tile in flight under the MFMAs of this one (A/B: 0)
#endif

extern "C" int gs_pixel_norm_fwd(const void* x, void* y, int64_t p, int c, float eps, int dtype, void* stream);
extern "C" int gs_pack_act_bits(void* z, int64_t p, int c, int dtype, void* stream);
extern "C" int gs_pixel_norm_bwd_fused(const void* g, const void* x, const void* addend, void* gx, int64_t p, int c, float eps, int pre_act, int post_act, int dtype,
                                       void* stream);
extern "C" int gs_pixel_norm_bwd_bwd_fused(const void* gg, const void* g, const void* x, void* out, void* out_g, int64_t p, int c, float eps, int pre_act,
                                           int dtype, void* stream);

namespace gs {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));


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
// Persistent kernel fed by LDS-DMA.  Block = 256 threads = 4 waves; every wave owns all 32*A output
// channels of the block and 32*B of its 128*B base pixels.  A block walks a list of work items (spatial
// tile x output-channel tile; the list of an XCD is contiguous so that neighbouring tiles share halo rows
// in that XCD's L2).  The K loop runs over stages = input-channel chunks of 64 bytes (16 f32 / 32 bf16)
// x tap groups.
//
// Staging: nothing passes through registers.  Every operand row is 64 bytes = four 16-byte slots, and a
// `buffer_load_dwordx4 ... lds` wave-instruction deposits 64 slots (1 KiB, 16 rows) at M0 + 16*lane while
// each lane supplies its own source offset -- so the XOR slot swizzle that makes the 16-lane ds_read_b128
// groups conflict-free is applied on the SOURCE side (lane at slot position p fetches part p ^ key(row)).
// Out-of-image patch rows are free: the descriptor covers exactly one image, rows above / below it fall
// outside [0, num_records) and the hardware writes zeros (measured, scripts/probe/dma_probe.hip); columns
// left / right of the image are forced out of range per lane.
// The stages of the next D iterations are always in flight (ring of D+1 weight buffers, 1+ceil(D/NTG) patch
// buffers); a wave waits with a COUNTED s_waitcnt vmcnt(n) -- n = its DMA pieces of the stages that may
// still be in flight -- and one raw s_barrier per stage publishes the landed stage to the other waves.
// hipcc knows nothing about these loads (inline asm), so it adds no vmcnt(0) of its own; the main loop has
// no ordinary global loads (the bias lives in LDS), only the epilogue stores.
// TG == 9 with RESIDENT keeps all taps of all chunks in LDS for the life of the block (thin layers: 32 or
// 64 output channels) and only the input patches stream.
struct ConvP {
    const void* x;
    const void* wp;
    void* y;
    const float* bias;  // optional fused epilogue: y = act(alpha * conv + bias)
    int act;
    const void* mask;   // optional (data gradients): y *= mask_act'(.) expressed through the activation OUTPUT mask[..] (y's shape)
    int mask_act;
    // 1-bit leaky-relu masks (bf16, plain epilogues): the sign bits of an activation output, one dword per (pixel, 32-channel tile) -- bit
    // 8 (2 hi + qp) + k is channel 16 qp + 8 hi + k of the tile, the order the lanes store their 16-byte pieces in -- written by the forward
    // epilogue BEHIND the activation itself (same allocation: z [numel] then numel / 8 bytes) and read by the masked epilogues instead of z:
    // 1/16 of the mask bytes of a launch that is HBM-bound at the top of the pyramid.  (gs_common.h: GS_ACT_LRELU_BITS / GS_ACT_WRITE_BITS)
    const unsigned char* mask_bits;
    unsigned char* bits_out;
    void* y2;           // optional (NORM == 1 kernels): y2 = pixel_norm(y) over the channels, y itself optional then
    float pn_eps;
    // NORM == 2 kernels (data gradients): the conv result g is the gradient w.r.t. y = pixel_norm(z) of the PREVIOUS block; the epilogue turns it
    // into the gradient w.r.t. that block's pre-activation, y = (pixel_norm_bwd(g, z) + addend) * mask_act'(z), with z = `mask` (the
    // activation output, y's shape) and `addend` an optional second gradient into z (same shape): one pass instead of a conv + a 4-tensor
    // elementwise pass
    const void* addend;
    // NORM == 3 kernels (second-order pass): the conv result t is the gradient w.r.t. u = act'(z) pixel_norm_bwd(g, z) (the first-order backward
    // of a generator block, differentiated by the mode-seeking term).  With h = t act'(z): y = pixel_norm_bwd(h, z) (the gradient w.r.t. g) and
    // y2 = d<h, pixel_norm_bwd(g, z)>/dz (the gradient w.r.t. z); z = `mask`, g = `addend`'s slot.  Same normbwd flag, value 2.
    int normbwd;        // host side: the caller asks for the NORM == 2 epilogue; cleared (and *norm_pending = 2) when the chosen kernel has none
    int* norm_pending;  // host side: set to 1 when the chosen kernel did not fuse the norm
    int N, Hi, Wi, IC, OC, Hb, Wb, tiles_x, tiles_y, nsp, noct, nch;
    // ceil(2^32 / d) for d = noct, tiles_x, tiles_y (0 for d = 1): item -> (image, tile row, tile column, channel tile) with one s_mul_hi_u32 per
    // division instead of the ~20 scalar instructions of a signed division each -- on the 32-channel layers (36 MFMAs per tile) the tile's time
    // IS its instruction count (a wave issues one instruction per ~5 cycles), and an item is decoded twice (issue cursor, compute side)
    unsigned m_noct, m_tx, m_ty;
    float alpha;
#ifdef GS_IGEMM_TRACE
    unsigned long long* trace;  // [block][64] shader-clock stamps of wave 0 (scripts/probe/igemm_trace.hip)
#endif
};
#ifdef GS_IGEMM_TRACE
#define GS_TR(slot)                                                                                              \
    do {                                                                                                         \
        if (p.trace && threadIdx.x == 0 && (slot) < 64) {                                                        \
            p.trace[blockIdx.x * 64 + (slot)] = __builtin_amdgcn_s_memtime();                                    \
            if ((slot) == 0 || (slot) == 63) p.trace[4096 * 64 + blockIdx.x * 2 + ((slot) == 63)] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                                        \
    } while (0)
#else
#define GS_TR(slot) do { } while (0)
#endif

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>)
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

typedef int i32x4 __attribute__((ext_vector_type(4)));

// one LDS-DMA piece: 64 lanes x 16 bytes -> LDS[lds_addr + 16*lane].  M0 is written and consumed inside the statement and
// not restored; it is DECLARED as clobbered, so a compiler use of M0 (v_movrel / s_movrel indexing, LDS-direct) can never straddle a piece.
__device__ __forceinline__ void lds_dma16(unsigned lds_addr, unsigned voff, i32x4 rs) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rs)
                 : "memory", "m0");
}
// the same with a scalar offset: address = base + voff + soff, and soff takes part in the descriptor's range check (measured,
// scripts/probe/dma_probe.hip mode 2) -- so the per-piece part of a WEIGHT address that is uniform over the wave stays in an SGPR and the
// piece costs no VALU instruction at all (a wave issues one instruction per ~4-5 cycles: profiles/r05_c_igemm_mid_timeline.txt prices a
// DMA piece at ~31 cycles of a stage, i.e. at its instruction count).  Only for offsets that never go negative (weights: yes; the
// patch origin of a border tile: no -- a 33-bit sum of a negative soff would not wrap back into range).
__device__ __forceinline__ void lds_dma16_s(unsigned lds_addr, unsigned voff, i32x4 rs, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rs), "s"(__builtin_amdgcn_readfirstlane(soff))   // (uniform by construction; the compiler cannot always prove it)
                 : "memory", "m0");
}
// streamed-once operands (the weight-gradient inputs of the HBM-bound layers): non-temporal policy -- the lines are not kept in L2 for a
// second reader that never comes
#ifndef GS_THIN_DMA_NT
#define GS_THIN_DMA_NT 0
#endif
__device__ __forceinline__ void lds_dma16_stream(unsigned lds_addr, unsigned voff, i32x4 rs) {
#if GS_THIN_DMA_NT
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rs)
                 : "memory", "m0");
#else
    lds_dma16(lds_addr, voff, rs);
#endif
}
// raw buffer descriptor over [base, base + bytes): stride 0, 32-bit data format (gfx950)
__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    i32x4 rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    rs[1] = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
    rs[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    rs[3] = 0x00020000;
    return rs;
}
// s_waitcnt vmcnt(n) with n known only after unrolling (the asm immediate must be a literal)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define GS_VM(K) case K: asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory"); break;
    switch (n) {
        GS_VM(0) GS_VM(1) GS_VM(2) GS_VM(3) GS_VM(4) GS_VM(5) GS_VM(6) GS_VM(7) GS_VM(8) GS_VM(9) GS_VM(10) GS_VM(11)
        GS_VM(12) GS_VM(13) GS_VM(14) GS_VM(15) GS_VM(16) GS_VM(17) GS_VM(18) GS_VM(19) GS_VM(20) GS_VM(21) GS_VM(22)
        GS_VM(23) GS_VM(24) GS_VM(25) GS_VM(26) GS_VM(27) GS_VM(28) GS_VM(29) GS_VM(30) GS_VM(31) GS_VM(32)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef GS_VM
}
__device__ __forceinline__ void block_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// RB: bytes of an operand row in LDS = of a channel chunk (64: four 16-byte slots, the original layout; 128: eight slots, whole
// 128-byte cache lines per DMA row -- 53-60 instead of 30 B/cycle/CU through the L2 -> LDS path, scripts/probe/dma_rate.hip -- and
// half the stages, barriers and DMA round trips per item; bf16 only)
// SPEC: wave-specialised block of 8 waves -- waves 0-3 run the fragment reads, MFMAs and the epilogue exactly as before, waves 4-7
// (one on each SIMD beside its compute wave) issue every DMA piece and hold the counted waits.  A wave issues one instruction per
// ~5 cycles whatever it is (scripts/probe/valu_rate.hip), so in a 4-wave block the ~60 cycles of each DMA piece (offset arithmetic,
// M0, the load) come ON TOP of the MFMAs of the stage -- measured 2260 ticks per stage for 1152 ticks of MFMA on the few-block layers
// (scripts/probe/igemm_trace.hip); on their own wave they run under them.
#ifndef GS_NORM_EPI_PIPELINE
#define GS_NORM_EPI_PIPELINE 1   // (0: the fused norm-backward epilogues fetch z / addend where they use them -- the build to compare against)
#endif
// BITS (bf16, plain epilogues): the build of the kernel for launches with 1-bit leaky-relu masks -- its mask, if any, is the sign words behind
// an activation (p.mask_bits), and with p.bits_out it writes the sign words of its own result.  Its own instantiation, not a run-time branch: with
// both mask forms in one epilogue the compiler keeps 15-25 more VGPRs live and the larger tiles lose a wave of occupancy.
template <typename T, int MODE, int A, int B, int TW, int TG, bool RESIDENT, int D, int NORM, int RB = 64, bool SPEC = false, bool BITS = false>
__global__ __launch_bounds__(SPEC ? 512 : 256) void conv_igemm_kernel(const ConvP p) {
    static_assert(NORM >= 0 && NORM <= 3, "NORM: 0 plain, 1 pixel norm of the result (forward blocks), 2 pixel-norm backward of the result (data gradients), "
                                          "3 both gradients of a differentiated norm backward (second-order pass)");
    constexpr int NP = 128 * B;
    constexpr int TH = NP / TW;
    constexpr int PH = patch_dim<MODE>(TH), PW = patch_dim<MODE>(TW);
    constexpr int S = MODE == MODE_S2 ? 2 : 1;
    constexpr int NPH = MODE == MODE_T2 ? 4 : 1;
    constexpr int SZ = (int)sizeof(T);
    static_assert(RB == 64 || (RB == 128 && SZ == 2), "128-byte rows: bf16 only");
    constexpr int BK = RB / SZ;
    constexpr int SL = RB / 16;               // 16-byte slots per row
    constexpr int KS = RB / 32;               // MFMA k-steps per row (a k-step = two slots: one per lane half)
    constexpr int OCT = 32 * A;
    constexpr int NTG = 9 / TG;
    constexpr int PCH = PH * PW * SL;         // 16-byte slots of a patch chunk
    constexpr int NPP = (PCH + 63) / 64;      // its 1 KiB DMA pieces ...
    constexpr int PP = (NPP + 3) / 4;         // ... per wave
    constexpr int PBUF = PP * 4096;           // bytes of one patch buffer (whole pieces for every wave; the excess is zero-filled)
    constexpr int WBYTES = TG * OCT * RB;     // bytes of one weight stage
    constexpr int NWP = WBYTES / 1024;
    constexpr int WP = (NWP + 3) / 4;
    constexpr int WBUF = WP * 4096;           // bytes of one weight buffer
    constexpr int NPB = 1 + (D + NTG - 1) / NTG;  // patch ring
    constexpr int NWB = D + 1;                    // weight ring
    constexpr int NSTEP = KS * TG;                // (tap, k-step) MFMA steps of a stage
    typedef typename Mma<T>::frag_t frag_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const lpatch = lds;                // NPB x PBUF
    unsigned char* const lwgt = lds + NPB * PBUF;     // streamed: NWB x WBUF ; resident: nch x WBUF
    float* const lbias = reinterpret_cast<float*>(lwgt + (RESIDENT ? p.nch : NWB) * WBUF);  // OC floats (zeros without a bias)
    const unsigned a_patch = (unsigned)(uintptr_t)lds;  // low 32 bits of a flat LDS address = the LDS byte address
    const unsigned a_wgt = a_patch + NPB * PBUF;

    const T* __restrict__ wp = reinterpret_cast<const T*>(p.wp);
    T* __restrict__ y = reinterpret_cast<T*>(p.y);
    const int Hi = p.Hi, Wi = p.Wi, IC = p.IC, OC = p.OC, Hb = p.Hb, Wb = p.Wb, NCH = p.nch;

    const int tid = threadIdx.x;
    const int lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wv8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = SPEC && wv8 >= 4;     // issues the DMA; !loader computes
    const bool issuer = !SPEC || loader;
    const bool computer = !SPEC || !loader;
    const int wv = wv8 & 3;                   // index within the role: the wave's DMA pieces / its 32*B pixels

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
    GS_TR(0);

    // ---- per-lane DMA source descriptors, constant for the life of the block
    int p_voff[PP], p_lx[PP];
#pragma unroll
    for (int k = 0; k < PP; ++k) {
        const int slot = (wv + 4 * k) * 64 + lane;
        const int row = slot / SL, pos = slot % SL;
        const int ly = row / PW, lx = row - ly * PW;
        // slot swizzle: rows one bank row (256 bytes) apart must land on different slots -- every 4th row of 64 bytes, every 2nd of 128
        p_voff[k] = ((ly * Wi + lx) * IC) * SZ + ((pos ^ (RB == 64 ? (lx >> 2) & 3 : (lx >> 1) & 7)) << 4);
        p_lx[k] = slot < PCH ? lx : 0x40000000;  // never inside the image
    }
    // a 1 KiB weight piece = RPP rows of the stage; row r of a 32-row channel tile takes slot key (r >> 2) & 3 (64-byte rows) or
    // (r >> 1) & 7 (128-byte rows: pieces of 8 rows, so the key also depends on the parity of the piece: two lane offsets)
    constexpr int RPP = 1024 / RB;            // rows per piece
    constexpr int SUBS = OCT / RPP;           // pieces per tap
    const int w_lane0 = RB == 64 ? ((lane >> 2) * IC) * SZ + (((lane & 3) ^ ((lane >> 4) & 3)) << 4)
                                 : ((lane >> 3) * IC) * SZ + (((lane & 7) ^ (lane >> 4)) << 4);
    const int w_lane1 = RB == 64 ? w_lane0 : ((lane >> 3) * IC) * SZ + (((lane & 7) ^ (4 + (lane >> 4))) << 4);
    // block-constant scalar part of every weight piece: rows [tap tt][RPP*sub ..] of the stage (T2 walks the taps phase by phase)
    int w_soff[NTG][WP];
    bool w_odd[WP];
#pragma unroll
    for (int k = 0; k < WP; ++k) w_odd[k] = RB == 128 && (((wv + 4 * k) % SUBS) & 1);
#pragma unroll
    for (int tg = 0; tg < NTG; ++tg)
#pragma unroll
        for (int k = 0; k < WP; ++k) {
            const int j = wv + 4 * k;
            const int tt = j / SUBS, sub = j % SUBS;
            const int i = tg * TG + tt;
            const int wt = MODE == MODE_T2 ? (int)((0x453718620ULL >> (4 * (i < 9 ? i : 0))) & 15) : i;
            w_soff[tg][k] = (NWP % 4 == 0 || j < NWP) ? ((wt * OC + sub * RPP) * IC) * SZ : (int)0x80000000;
        }
    const unsigned img_bytes = (unsigned)Hi * Wi * IC * SZ;
    const unsigned w_bytes = 9u * IC * OC * SZ;
    i32x4 rs_w = make_rsrc(wp, w_bytes);
    const unsigned a_wave = wv * 1024;

    auto item_coords = [&](int item, int& n, int& by, int& bx, int& oc0) __attribute__((always_inline)) {
        // (exact: item < 2^21 and every divisor < 2^11, so item * (m d - 2^32) < 2^32)
        const unsigned it = (unsigned)item;
        const unsigned sp = p.m_noct ? __umulhi(it, p.m_noct) : it;
        oc0 = (int)(it - sp * (unsigned)p.noct) * OCT;
        const unsigned r = p.m_tx ? __umulhi(sp, p.m_tx) : sp;
        const int tile_x = (int)(sp - r * (unsigned)p.tiles_x);
        const unsigned nn = p.m_ty ? __umulhi(r, p.m_ty) : r;
        by = (int)(r - nn * (unsigned)p.tiles_y) * TH;
        bx = tile_x * TW;
        n = (int)nn;
    };

    // ---- issue side: cursor over (item, chunk); the tap group is a compile-time value at every call site
    int i_item = first, i_left = count, i_ch = 0, i_pb = 0, i_wb = 0;
    int i_n, i_by, i_bx, i_oc0;
    item_coords(i_item, i_n, i_by, i_bx, i_oc0);
    int i_oy0 = 0, i_ox0 = 0, i_org = 0;
    i32x4 rs_x = make_rsrc(p.x, img_bytes);

    // Past the end of the item list the stages are still "issued" (the counted waits stay static) but against empty
    // descriptors: every lane is out of range, nothing is fetched, zeros land in ring slots nobody reads again.
    int i_wbase = 0;
    auto issue_setup = [&](int tg) __attribute__((always_inline)) {  // scalars of the stage about to be issued
        const bool more = i_left > 0;
        if (tg == 0) {
            i_oy0 = MODE == MODE_S2 ? 2 * i_by : i_by - 1;
            i_ox0 = MODE == MODE_S2 ? 2 * i_bx : i_bx - 1;
            i_org = ((i_oy0 * Wi + i_ox0) * IC + i_ch * BK) * SZ;
            rs_x = make_rsrc(reinterpret_cast<const unsigned char*>(p.x) + (size_t)(more ? i_n : 0) * img_bytes, more ? img_bytes : 0u);
        }
        if (!RESIDENT) {
            rs_w[2] = more ? (int)w_bytes : 0;
            i_wbase = (i_oc0 * IC + i_ch * BK) * SZ;
        }
    };
    auto issue_patch_piece = [&](int k) __attribute__((always_inline)) {
        const unsigned voff = (unsigned)(i_ox0 + p_lx[k]) < (unsigned)Wi ? (unsigned)(i_org + p_voff[k]) : 0x80000000u;
        lds_dma16(a_patch + a_wave + i_pb * PBUF + k * 4096, voff, rs_x);
    };
    auto issue_weight_piece = [&](int k, int tg, int wbase, unsigned dst_base) __attribute__((always_inline)) {
        // (pieces past the end of the stage carry 0x80000000: out of range for any descriptor)
        // (the lane part is loop-invariant, the rest is wave-uniform and >= 0: scalar offset.  Pieces past the end of the stage carry
        //  0x80000000 in w_soff: base + 2^31 is out of range for any descriptor, with or without wbase on top)
        lds_dma16_s(dst_base + a_wave + k * 4096, (unsigned)(w_odd[k] ? w_lane1 : w_lane0), rs_w, (unsigned)(w_soff[tg][k] + wbase));
    };
    constexpr int NPW = RESIDENT ? 0 : WP;
    auto stage_pieces = [](int tg) { return (tg == 0 ? PP : 0) + NPW; };  // DMA pieces a wave issues for a stage
    // piece q of the stage being issued (patch pieces first)
    auto issue_piece = [&](int q, int tg) __attribute__((always_inline)) {
        const int np = tg == 0 ? PP : 0;
        if (q < np) issue_patch_piece(q);
        else issue_weight_piece(q - np, tg, i_wbase, a_wgt + i_wb * WBUF);
    };
    auto issue_advance = [&](int tg) __attribute__((always_inline)) {
        if (tg == 0) i_pb = i_pb + 1 == NPB ? 0 : i_pb + 1;
        if (!RESIDENT) i_wb = i_wb + 1 == NWB ? 0 : i_wb + 1;
        if (tg == NTG - 1) {
            if (++i_ch == NCH) {
                i_ch = 0;
                --i_left;
                i_item += stride;
                if (i_left > 0) item_coords(i_item, i_n, i_by, i_bx, i_oc0);
                else i_left = 0;
            }
        }
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
    int b_off[B][3][KS], a_off[KS];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int q = (wv * B + b) * 32 + l31;
        const int pb0 = ((q / TW) * S) * PW + (q % TW) * S;
#pragma unroll
        for (int ox = 0; ox < 3; ++ox) {
            const int lx = (q % TW) * S + ox;
            const int key = RB == 64 ? (lx >> 2) & 3 : (lx >> 1) & 7;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) b_off[b][ox][ks] = (pb0 + ox) * RB + (((ks * 2 + hi) ^ key) << 4);
        }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a_off[ks] = l31 * RB + (((ks * 2 + hi) ^ (RB == 64 ? (l31 >> 2) & 3 : (l31 >> 1) & 7)) << 4);

    // ---- prologue: resident weights and the first D stages go out first, the bias is staged while they fly (hipcc waits
    //      vmcnt(0) for the bias loads, which drains the DMAs too -- at this point that is exactly the wait that is needed)
    GS_TR(1);
    if (issuer) {
        if (RESIDENT) {
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int k = 0; k < WP; ++k) issue_weight_piece(k, 0, (i_oc0 * IC + ch * BK) * SZ, a_wgt + ch * WBUF);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            issue_setup(d % NTG);
#pragma unroll
            for (int q = 0; q < stage_pieces(d % NTG); ++q) issue_piece(q, d % NTG);
            issue_advance(d % NTG);
        }
    }
    GS_TR(2);
    for (int c = tid; c < OC; c += (SPEC ? 512 : 256)) lbias[c] = p.bias ? p.bias[c] : 0.f;
    // Everything the first MFMA needs besides the landed stage is computed HERE, in the shadow of the DMA round trip: left to itself hipcc
    // sinks the fragment-offset tables and the accumulator clears behind the barrier (their first use) -- ~340 instructions between "stage 0
    // landed" and the first MFMA of every block of every launch (profiles/r05_c_igemm_rb128_isa_slots.txt, slot 0); pinned here, ~180 of
    // them run before the barrier.  (Ordering the table arithmetic behind the first DMA statements as well -- opaque lane ids -- costs the
    // common subexpressions of the tables: +118 instructions, the last piece goes out later; measured in the ISA, not adopted.)
    zero_acc();
    if (computer) {
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int ox = 0; ox < 3; ++ox)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(b_off[b][ox][ks]));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(a_off[ks]));
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
            for (int a = 0; a < A; ++a)
#pragma unroll
                for (int b = 0; b < B; ++b) asm volatile("" : "+v"(acc[ph][a][b]));
    }
    wait_vmcnt(0);
    block_barrier();
    GS_TR(3);
#ifdef GS_IGEMM_TRACE
    int tr_stage = 0;
#endif

    int item = first, done = 0, c_pb = 0, c_wb = 0;
    while (true) {
        int n, by, bx, oc0;
        item_coords(item, n, by, bx, oc0);
        // BITS: the sign words of the lane's pixels, fetched at the START of the item's last stage -- one register per (pixel, 32-channel tile), so
        // they can wait through the MFMAs of the stage, where the 16-byte mask vectors (8-32 registers) cannot: a mask fetched in the epilogue
        // costs every item one exposed memory round trip (~0.9 us per 256-pixel item on the 32-channel layers: scripts/mask_bits_micro.py).
        unsigned mbw[BITS ? B * (MODE == MODE_T2 ? 4 : 1) : 1][A];
        // NORM 2 / 3 (the fused pixel-norm backward epilogues): the z vectors and the addend / g vectors of the lane's (pixel group, phase) steps, double
        // buffered -- step 0 is fetched at the start of the item's last stage like the sign words above, step i + 1 while step i is computed: fetched
        // where they are used, every step pays a memory round trip (1-4 per item).  Free registers: two blocks per CU leave a wave 256 VGPRs.
        // (not the 2 x 2 tiling: 64 more registers would take it past 256 and to one block per CU)
        constexpr bool NPIPE = (NORM == 2 || NORM == 3) && A * B <= 2 && GS_NORM_EPI_PIPELINE;
        typedef typename std::conditional<SZ == 4, float4, uint4>::type nvec_t;
        constexpr int NNV = SZ == 4 ? 4 : 2;
        nvec_t nz[NPIPE ? 2 : 1][A][NNV], nx[NPIPE ? 2 : 1][A][NNV];
        auto norm_fetch = [&](int i, nvec_t (&zq)[A][NNV], nvec_t (&xq)[A][NNV]) __attribute__((always_inline)) {
            constexpr int NPHB = MODE == MODE_T2 ? 4 : 1;
            const int b = i / NPHB, ph = i % NPHB;
            const int Ho = MODE == MODE_T2 ? 2 * Hb : Hb, Wo = MODE == MODE_T2 ? 2 * Wb : Wb;
            const int q = (wv * B + b) * 32 + l31;
            const int gy = by + q / TW, gx = bx + q % TW;
            const int oy = MODE == MODE_T2 ? 2 * gy + (ph >> 1) : gy;
            const int ox = MODE == MODE_T2 ? 2 * gx + (ph & 1) : gx;
            const long base = (gy < Hb && gx < Wb) ? (((long)n * Ho + oy) * Wo + ox) * OC + oc0 : 0;   // clamped: the loads stay unconditional
#ifdef GS_ABL_NOMASKLOAD
            const long zbase = base & 1023;
#else
            const long zbase = base;
#endif
#pragma unroll
            for (int a = 0; a < A; ++a)
#pragma unroll
                for (int v = 0; v < NNV; ++v) {
                    zq[a][v] = *reinterpret_cast<const nvec_t*>(reinterpret_cast<const T*>(p.mask) + zbase + a * 32 + v * (32 / NNV) + hi * (16 / NNV));
                    if (NORM == 3 || p.addend)
                        xq[a][v] = *reinterpret_cast<const nvec_t*>(reinterpret_cast<const T*>(p.addend) + base + a * 32 + v * (32 / NNV) + hi * (16 / NNV));
                }
        };
        for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) {
                const int itg = (tg + D) % NTG;            // tap group of the stage issued during this one
                const int npiece = stage_pieces(itg);
                if (issuer) issue_setup(itg);
                if constexpr (NPIPE) {
                    if (tg == NTG - 1 && ch == NCH - 1 && computer) norm_fetch(0, nz[0], nx[0]);
                }
                if constexpr (BITS) {
                    if (tg == NTG - 1 && ch == NCH - 1 && computer && p.mask_bits) {
                        constexpr int NPHB = MODE == MODE_T2 ? 4 : 1;
                        const int Ho = MODE == MODE_T2 ? 2 * Hb : Hb, Wo = MODE == MODE_T2 ? 2 * Wb : Wb;
#pragma unroll
                        for (int b = 0; b < B; ++b) {
                            const int q = (wv * B + b) * 32 + l31;
                            const int gy = by + q / TW, gx = bx + q % TW;
#pragma unroll
                            for (int ph = 0; ph < NPHB; ++ph) {
                                const int oy = MODE == MODE_T2 ? 2 * gy + (ph >> 1) : gy;
                                const int ox = MODE == MODE_T2 ? 2 * gx + (ph & 1) : gx;
                                const long off = (gy < Hb && gx < Wb) ? (((long)n * Ho + oy) * Wo + ox) * OC + oc0 : 0;   // clamped: unconditional loads
#pragma unroll
                                for (int a = 0; a < A; ++a) mbw[b * NPHB + ph][a] = *reinterpret_cast<const unsigned*>(p.mask_bits + ((off + a * 32) >> 3));
                            }
                        }
                    }
                }
                if (SPEC && loader) {   // the whole stage +D in one go, under the compute waves' MFMAs
#pragma unroll
                    for (int q = 0; q < npiece; ++q) issue_piece(q, itg);
                }
                // ---- MFMAs of this stage; fragment reads run one (tap, k-step) ahead, the DMA pieces of stage +D are
                //      spread over the steps
                const unsigned char* lp = lpatch + c_pb * PBUF;
                const unsigned char* lw = RESIDENT ? lwgt + ch * WBUF : lwgt + c_wb * WBUF;
                if (computer) {
                    constexpr int PF = NSTEP >= 6 ? 2 : 1;   // fragment reads run PF steps ahead of their MFMAs (LDS latency with 4
                                                             // waves on the pipe exceeds one step of 2-4 MFMAs)
                    frag_t af[PF + 1][A], bf[PF + 1][B];
                    // one fragment read of (step, r): r < A -> weight rows of channel tile r, else pixel group r - A
                    auto load_frag = [&](int step, int r) __attribute__((always_inline)) {
                        const int tt = step / KS, ks = step % KS, buf = step % (PF + 1);
                        const int i = tg * TG + tt;
                        const int oyv = tap_off<MODE>(tap_ky<MODE>(i)), oxv = tap_off<MODE>(tap_kx<MODE>(i));
                        if (r < A) af[buf][r] = *reinterpret_cast<const frag_t*>(lw + (tt * OCT + r * 32) * RB + a_off[ks]);
                        else bf[buf][r - A] = *reinterpret_cast<const frag_t*>(lp + oyv * PW * RB + b_off[r - A][oxv][ks]);
                    };
#if !defined(GS_ABL_NOMMA)
#pragma unroll
                    for (int st0 = 0; st0 < PF; ++st0)
#pragma unroll
                        for (int r = 0; r < A + B; ++r) load_frag(st0, r);
#endif
                    // Every MFMA is followed by its share of the other work of the step -- the fragment reads of step+1 and
                    // the DMA pieces of stage +D -- and the order is pinned: with one wave per SIMD only what is issued
                    // inside an MFMA's 32-cycle shadow is free, and left alone hipcc sinks the reads next to their consumers.
                    constexpr int NMMA = A * B;
                    constexpr int RPM = (A + B + NMMA - 1) / NMMA;            // reads per MFMA slot
                    const int ppm = (npiece + NSTEP * NMMA - 1) / (NSTEP * NMMA);  // DMA pieces per MFMA slot
#pragma unroll
                    for (int step = 0; step < NSTEP; ++step) {
                        const int ph = tap_phase<MODE>(tg * TG + step / KS);
#pragma unroll
                        for (int m = 0; m < NMMA; ++m) {
#ifndef GS_ABL_NOMMA
                            Mma<T>::mma(af[step % (PF + 1)][m / B], bf[step % (PF + 1)][m % B], acc[ph][m / B][m % B]);
#if !defined(GS_ABL_NOFRAG)
                            if (step + PF < NSTEP) {
#pragma unroll
                                for (int r = m * RPM; r < (m + 1) * RPM && r < A + B; ++r) load_frag(step + PF, r);
                            }
#endif
#endif
#ifndef GS_ABL_NODMA
                            if constexpr (!SPEC) {
                                const int slot = step * NMMA + m;
#pragma unroll
                                for (int q = slot * ppm; q < (slot + 1) * ppm && q < npiece; ++q) issue_piece(q, itg);
                            }
#endif
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                if (issuer) issue_advance(itg);
                // ---- the next stage must have landed before the barrier below: leave only the younger stages in flight
                if (issuer) {
                    int younger = 0;
#pragma unroll
                    for (int d = 2; d <= D; ++d) younger += stage_pieces((tg + d) % NTG);
#ifndef GS_ABL_NODMA
                    wait_vmcnt(younger);
#endif
                }
                // ---- epilogue of the item.  D[oc][pixel]: a lane holds oc = 8q + 4hi + (0..3) of pixel l31 per accumulator quad.
                //      The store path sustains ~7 B/cycle/CU with 8-byte stores and twice that with 16-byte ones (measured with
                //      the ablations of scripts/probe/igemm_trace.hip: the top-of-pyramid layers are bound by it), so a lane must
                //      leave with 16 bytes.  fp32: a quad is 16 bytes.  bf16: v_permlane32_swap exchanges the quads q / q+1
                //      between the two lane halves, after which lane (pixel, hi) owns 8 consecutive channels 16*(q/2) + 8*hi + ...
#ifndef GS_ABL_NOEPI
                if (tg == NTG - 1 && ch == NCH - 1 && computer) {
                    const int Ho = MODE == MODE_T2 ? 2 * Hb : Hb, Wo = MODE == MODE_T2 ? 2 * Wb : Wb;
                    const float slope = p.act == GS_ACT_LRELU ? 0.2f : 1.f;
                    auto mask_factor = [&](float z) __attribute__((always_inline)) {
                        return p.mask_act == GS_ACT_LRELU ? (z > 0.f ? 1.f : 0.2f) : (p.mask_act == GS_ACT_TANH ? 1.f - z * z : 1.f);
                    };
                    // the 32 channels of tile a of one pixel: act(alpha * acc + bias); o[qd][e] = channel 8 qd + 4 hi + e
                    auto finish = [&](int ph, int a, int b, float (&o)[4][4]) __attribute__((always_inline)) {
#pragma unroll
                        for (int qd = 0; qd < 4; ++qd) {
                            const float4 bv = *reinterpret_cast<const float4*>(lbias + oc0 + a * 32 + qd * 8 + hi * 4);
                            const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float v = acc[ph][a][b][qd * 4 + e] * p.alpha + bb[e];
                                o[qd][e] = fmaxf(v, slope * v);  // leaky relu (slope 1: identity)
                            }
                        }
                    };
                    // ... to dst[off + a * 32 + ...] (16 bytes per lane), optionally times mask_act'(.) through mask[off + ...]
                    // The mask vectors of a lane (its 16-byte pieces of the activation output, laid out like its stores) are fetched
                    // AHEAD of the arithmetic, several at a time: loaded where they are used, each costs the lane a full memory
                    // round trip (4-8 dependent trips per tile).
                    typedef typename std::conditional<SZ == 4, float4, uint4>::type mvec_t;
                    constexpr int NV = SZ == 4 ? 4 : 2;              // mask vectors per 32-channel tile of a pixel
                    constexpr bool use_mb = BITS;   // (1-bit masks: bf16, plain epilogues)
                    const bool emit_bits = BITS && p.bits_out != nullptr;
                    auto mask_fetch = [&](long off, bool inside, mvec_t (&mz)[A][NV]) __attribute__((always_inline)) {
#ifdef GS_ABL_NOMASKLOAD   // (ablation build only: every mask load hits the same few cache lines -- what would a mask of no bytes be worth?)
                        const long base = (inside ? off : 0) & 1023;
#else
                        const long base = inside ? off : 0;          // clamped: the loads stay unconditional
#endif
#pragma unroll
                        for (int a = 0; a < A; ++a)
#pragma unroll
                            for (int v = 0; v < NV; ++v)
                                mz[a][v] = *reinterpret_cast<const mvec_t*>(reinterpret_cast<const T*>(p.mask) + base + a * 32 + v * (32 / NV) + hi * (16 / NV));
                    };
                    auto store = [&](T* dst, long off, int a, float (&o)[4][4], bool inside, const mvec_t* mz, int mkind = 0,
                                     bool emit = false) __attribute__((always_inline)) {   // mkind: 0 no mask, 1 mask values in mz, 2 mask bits in mz[0].x
                        if constexpr (SZ == 4) {
#pragma unroll
                            for (int qd = 0; qd < 4; ++qd) {
                                if (mkind == 1) {
                                    const float4 zv = mz[qd];
                                    if (p.mask_act == GS_ACT_LRELU) {   // (the common case by itself: compare, scale, select per value)
                                        o[qd][0] = zv.x > 0.f ? o[qd][0] : 0.2f * o[qd][0]; o[qd][1] = zv.y > 0.f ? o[qd][1] : 0.2f * o[qd][1];
                                        o[qd][2] = zv.z > 0.f ? o[qd][2] : 0.2f * o[qd][2]; o[qd][3] = zv.w > 0.f ? o[qd][3] : 0.2f * o[qd][3];
                                    } else {
                                        o[qd][0] *= mask_factor(zv.x); o[qd][1] *= mask_factor(zv.y); o[qd][2] *= mask_factor(zv.z); o[qd][3] *= mask_factor(zv.w);
                                    }
                                }
                                if (inside) st4(reinterpret_cast<float*>(dst) + off + a * 32 + qd * 8 + hi * 4, o[qd]);
                            }
                        } else {
                            unsigned sign_bytes = 0u;
#pragma unroll
                            for (int qp = 0; qp < 2; ++qp) {
                                float lo[4], hi4[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[2 * qp][e]), __float_as_uint(o[2 * qp + 1][e]), false, false);
                                    lo[e] = __uint_as_float(r[0]);
                                    hi4[e] = __uint_as_float(r[1]);
                                }
                                if (BITS && mkind) {   // 1-bit mask: this lane's byte of the pixel's dword
                                    const unsigned byte = mz[0].x >> (8 * (2 * hi + qp));
#define GS_BR(V, K) V = (byte >> (K)) & 1u ? V : 0.2f * V
                                    GS_BR(lo[0], 0); GS_BR(lo[1], 1); GS_BR(lo[2], 2); GS_BR(lo[3], 3);
                                    GS_BR(hi4[0], 4); GS_BR(hi4[1], 5); GS_BR(hi4[2], 6); GS_BR(hi4[3], 7);
#undef GS_BR
                                } else if (!BITS && mkind) {   // the lane's 8 channels of the mask sit where its 16 bytes go
                                    const uint4 zv = mz[qp];
                                    if (p.mask_act == GS_ACT_LRELU) {
                                        // z > 0 on the packed pair: low half shifted up and compared as an integer, high half in place (>= 0x10000: sign
                                        // clear, magnitude bits not all zero); compare, scale, select per value
#define GS_LR(V, Z) V = (int)((Z) << 16) > 0 ? V : 0.2f * V
#define GS_HR(V, Z) V = (int)(Z) >= 0x10000 ? V : 0.2f * V
                                        GS_LR(lo[0], zv.x); GS_HR(lo[1], zv.x); GS_LR(lo[2], zv.y); GS_HR(lo[3], zv.y);
                                        GS_LR(hi4[0], zv.z); GS_HR(hi4[1], zv.z); GS_LR(hi4[2], zv.w); GS_HR(hi4[3], zv.w);
#undef GS_LR
#undef GS_HR
                                    } else {
                                        lo[0] *= mask_factor(__uint_as_float(zv.x << 16)); lo[1] *= mask_factor(__uint_as_float(zv.x & 0xffff0000u));
                                        lo[2] *= mask_factor(__uint_as_float(zv.y << 16)); lo[3] *= mask_factor(__uint_as_float(zv.y & 0xffff0000u));
                                        hi4[0] *= mask_factor(__uint_as_float(zv.z << 16)); hi4[1] *= mask_factor(__uint_as_float(zv.z & 0xffff0000u));
                                        hi4[2] *= mask_factor(__uint_as_float(zv.w << 16)); hi4[3] *= mask_factor(__uint_as_float(zv.w & 0xffff0000u));
                                    }
                                }
                                uint4 v;
                                v.x = pack_bf16x2(lo[0], lo[1]); v.y = pack_bf16x2(lo[2], lo[3]);
                                v.z = pack_bf16x2(hi4[0], hi4[1]); v.w = pack_bf16x2(hi4[2], hi4[3]);
                                if (inside) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(dst) + off + a * 32 + qp * 16 + hi * 8) = v;
                                if (BITS && emit) {   // sign bits of the STORED values (z > 0 exactly as the masked epilogues test it on the bf16 pair)
                                    const unsigned b8 = ((int)(v.x << 16) > 0 ? 1u : 0u) | ((int)v.x >= 0x10000 ? 2u : 0u) | ((int)(v.y << 16) > 0 ? 4u : 0u) |
                                                        ((int)v.y >= 0x10000 ? 8u : 0u) | ((int)(v.z << 16) > 0 ? 16u : 0u) | ((int)v.z >= 0x10000 ? 32u : 0u) |
                                                        ((int)(v.w << 16) > 0 ? 64u : 0u) | ((int)v.w >= 0x10000 ? 128u : 0u);
                                    sign_bytes |= b8 << (8 * qp);
                                }
                            }
                            if (BITS && emit && inside)   // the lane's two bytes of the pixel's dword: bytes 2 hi, 2 hi + 1
                                *reinterpret_cast<unsigned short*>(p.bits_out + ((off + a * 32) >> 3) + 2 * hi) = (unsigned short)sign_bytes;
                        }
                    };
                    // all mask vectors of the tile in one go when they fit in 32 registers, else one (pixel group, phase) at a time
                    constexpr bool MASK_ALL = !NORM && (BITS || B * NPH * A * NV <= 8);   // (sign words: one register per (pixel, tile), always ahead)
                    mvec_t mz_all[MASK_ALL ? B * NPH : 1][A][NV];
                    if constexpr (BITS) {
#pragma unroll
                        for (int i = 0; i < B * NPH; ++i)
#pragma unroll
                            for (int a = 0; a < A; ++a) mz_all[i][a][0].x = mbw[i][a];
                    } else if constexpr (MASK_ALL) {
                        if (p.mask) {
#pragma unroll
                            for (int b = 0; b < B; ++b) {
                                const int q = (wv * B + b) * 32 + l31;
                                const int gy = by + q / TW, gx = bx + q % TW;
#pragma unroll
                                for (int ph = 0; ph < NPH; ++ph) {
                                    const int oy = MODE == MODE_T2 ? 2 * gy + (ph >> 1) : gy;
                                    const int ox = MODE == MODE_T2 ? 2 * gx + (ph & 1) : gx;
                                    mask_fetch((((long)n * Ho + oy) * Wo + ox) * OC + oc0, gy < Hb && gx < Wb, mz_all[b * NPH + ph]);
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int b = 0; b < B; ++b) {
                        const int q = (wv * B + b) * 32 + l31;
                        const int gy = by + q / TW, gx = bx + q % TW;
#ifdef GS_ABL_NOSTORE
                        const bool inside = gy < -1000;
#else
                        const bool inside = gy < Hb && gx < Wb;
#endif
#pragma unroll
                        for (int ph = 0; ph < NPH; ++ph) {
                            const int oy = MODE == MODE_T2 ? 2 * gy + (ph >> 1) : gy;
                            const int ox = MODE == MODE_T2 ? 2 * gx + (ph & 1) : gx;
                            const long off = (((long)n * Ho + oy) * Wo + ox) * OC + oc0;
                            if constexpr (NORM == 2) {
                                // the block owns every channel of the pixel (OC == 32 A) and the accumulators are g = d L / d pixel_norm(z): finish
                                // the previous block's backward here.  With r = rsqrt(mean z^2 + eps): gx = r (g - z r^2 mean(z g)), then + addend
                                // and times act'(z).  Values are brought to the STORE layout first (bf16: the lane-half swap), where the lane's
                                // 16 channels per 32-channel tile sit exactly like the 16-byte vectors of z / addend it fetches.
                                constexpr int EV = 16 / NV;                    // values per 16-byte vector: 4 (fp32) or 8 (bf16)
                                const int cur = NPIPE ? ((b * NPH + ph) & 1) : 0;
                                if constexpr (NPIPE) {   // (this step's vectors are on their way since the last stage / the previous step: fetch the next one's)
                                    if (b * NPH + ph + 1 < B * NPH) norm_fetch(b * NPH + ph + 1, nz[(b * NPH + ph + 1) & 1], nx[(b * NPH + ph + 1) & 1]);
                                } else {
                                    mask_fetch(off, inside, nz[0]);
                                    if (p.addend) {
                                        const long base = inside ? off : 0;
#pragma unroll
                                        for (int a = 0; a < A; ++a)
#pragma unroll
                                            for (int v = 0; v < NV; ++v)
                                                nx[0][a][v] = *reinterpret_cast<const mvec_t*>(reinterpret_cast<const T*>(p.addend) + base + a * 32 + v * (32 / NV) + hi * (16 / NV));
                                    }
                                }
                                mvec_t (&zq)[A][NV] = nz[cur];
                                mvec_t (&aq)[A][NV] = nx[cur];
                                float gv[A][NV][EV], zv[A][NV][EV];
                                float ssq = 0.f, szg = 0.f;
#pragma unroll
                                for (int a = 0; a < A; ++a) {
                                    float o[4][4];
                                    finish(ph, a, b, o);   // (no bias, no activation on a data gradient: alpha * acc)
                                    if constexpr (SZ == 4) {
#pragma unroll
                                        for (int qd = 0; qd < 4; ++qd) {
                                            const float4 z4 = zq[a][qd];
                                            const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
                                            for (int e = 0; e < 4; ++e) { gv[a][qd][e] = o[qd][e]; zv[a][qd][e] = zz[e]; }
                                        }
                                    } else {
#pragma unroll
                                        for (int qp = 0; qp < 2; ++qp) {
#pragma unroll
                                            for (int e = 0; e < 4; ++e) {
                                                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[2 * qp][e]), __float_as_uint(o[2 * qp + 1][e]), false, false);
                                                gv[a][qp][e] = __uint_as_float(r[0]);
                                                gv[a][qp][4 + e] = __uint_as_float(r[1]);
                                            }
                                            const uint4 z4 = zq[a][qp];
                                            const unsigned zw[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
                                            for (int e = 0; e < 4; ++e) { zv[a][qp][2 * e] = __uint_as_float(zw[e] << 16); zv[a][qp][2 * e + 1] = __uint_as_float(zw[e] & 0xffff0000u); }
                                        }
                                    }
#pragma unroll
                                    for (int v = 0; v < NV; ++v)
#pragma unroll
                                        for (int e = 0; e < EV; ++e) { ssq += zv[a][v][e] * zv[a][v][e]; szg += zv[a][v][e] * gv[a][v][e]; }
                                }
                                ssq = swap32_sum(ssq);   // the partner lane of the other half holds the pixel's other channels
                                szg = swap32_sum(szg);
                                const float inv_c = 1.f / (float)(32 * A);
                                const float r = rsqrtf(ssq * inv_c + p.pn_eps);
                                const float m = szg * inv_c * r * r;
#pragma unroll
                                for (int a = 0; a < A; ++a)
#pragma unroll
                                    for (int v = 0; v < NV; ++v) {
                                        float out[EV], ad[EV];
#pragma unroll
                                        for (int e = 0; e < EV; ++e) ad[e] = 0.f;
                                        if (p.addend) {
                                            if constexpr (SZ == 4) {
                                                const float4 a4 = aq[a][v];
                                                ad[0] = a4.x; ad[1] = a4.y; ad[2] = a4.z; ad[3] = a4.w;
                                            } else {
                                                const uint4 a4 = aq[a][v];
                                                const unsigned aw[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                                                for (int e = 0; e < 4; ++e) { ad[2 * e] = __uint_as_float(aw[e] << 16); ad[2 * e + 1] = __uint_as_float(aw[e] & 0xffff0000u); }
                                            }
                                        }
#pragma unroll
                                        for (int e = 0; e < EV; ++e) out[e] = (r * (gv[a][v][e] - zv[a][v][e] * m) + ad[e]) * mask_factor(zv[a][v][e]);
                                        if (inside) {
                                            if constexpr (SZ == 4) {
                                                st4(reinterpret_cast<float*>(y) + off + a * 32 + v * 8 + hi * 4, out);
                                            } else {
                                                uint4 w4;
                                                w4.x = pack_bf16x2(out[0], out[1]); w4.y = pack_bf16x2(out[2], out[3]);
                                                w4.z = pack_bf16x2(out[4], out[5]); w4.w = pack_bf16x2(out[6], out[7]);
                                                *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(y) + off + a * 32 + v * 16 + hi * 8) = w4;
                                            }
                                        }
                                    }
                            } else if constexpr (NORM == 3) {
                                // accumulators t = gradient w.r.t. u = M J(z) g (M = act'(z), J the norm's Jacobian).  h = M t;
                                //   y  = J h                                   = r (h - z r^2 mean(h z))
                                //   y2 = d<h, J g>/dz = -r^3 (mean(h g) z + mean(z g) h + mean(h z) g) + 3 r^5 mean(h z) mean(z g) z
                                constexpr int EV = 16 / NV;
                                const int cur = NPIPE ? ((b * NPH + ph) & 1) : 0;
                                if constexpr (NPIPE) {
                                    if (b * NPH + ph + 1 < B * NPH) norm_fetch(b * NPH + ph + 1, nz[(b * NPH + ph + 1) & 1], nx[(b * NPH + ph + 1) & 1]);
                                } else {
                                    mask_fetch(off, inside, nz[0]);
                                    const long base = inside ? off : 0;
#pragma unroll
                                    for (int a = 0; a < A; ++a)
#pragma unroll
                                        for (int v = 0; v < NV; ++v)
                                            nx[0][a][v] = *reinterpret_cast<const mvec_t*>(reinterpret_cast<const T*>(p.addend) + base + a * 32 + v * (32 / NV) + hi * (16 / NV));
                                }
                                mvec_t (&zq)[A][NV] = nz[cur];
                                mvec_t (&gq)[A][NV] = nx[cur];
                                float hv[A][NV][EV], zv[A][NV][EV], gv[A][NV][EV];
                                float ssq = 0.f, shg = 0.f, shz = 0.f, szg = 0.f;
#pragma unroll
                                for (int a = 0; a < A; ++a) {
                                    float o[4][4];
                                    finish(ph, a, b, o);   // (alpha * acc: no bias, no activation)
                                    if constexpr (SZ == 4) {
#pragma unroll
                                        for (int qd = 0; qd < 4; ++qd) {
                                            const float4 z4 = zq[a][qd], g4 = gq[a][qd];
                                            const float zz[4] = {z4.x, z4.y, z4.z, z4.w}, g_[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                                            for (int e = 0; e < 4; ++e) { hv[a][qd][e] = o[qd][e]; zv[a][qd][e] = zz[e]; gv[a][qd][e] = g_[e]; }
                                        }
                                    } else {
#pragma unroll
                                        for (int qp = 0; qp < 2; ++qp) {
#pragma unroll
                                            for (int e = 0; e < 4; ++e) {
                                                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[2 * qp][e]), __float_as_uint(o[2 * qp + 1][e]), false, false);
                                                hv[a][qp][e] = __uint_as_float(r[0]);
                                                hv[a][qp][4 + e] = __uint_as_float(r[1]);
                                            }
                                            const uint4 z4 = zq[a][qp], g4 = gq[a][qp];
                                            const unsigned zw[4] = {z4.x, z4.y, z4.z, z4.w}, gw_[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                                            for (int e = 0; e < 4; ++e) {
                                                zv[a][qp][2 * e] = __uint_as_float(zw[e] << 16); zv[a][qp][2 * e + 1] = __uint_as_float(zw[e] & 0xffff0000u);
                                                gv[a][qp][2 * e] = __uint_as_float(gw_[e] << 16); gv[a][qp][2 * e + 1] = __uint_as_float(gw_[e] & 0xffff0000u);
                                            }
                                        }
                                    }
#pragma unroll
                                    for (int v = 0; v < NV; ++v)
#pragma unroll
                                        for (int e = 0; e < EV; ++e) {
                                            const float zc = zv[a][v][e], gc = gv[a][v][e];
                                            const float hc = hv[a][v][e] * mask_factor(zc);
                                            hv[a][v][e] = hc;
                                            ssq += zc * zc; shg += hc * gc; shz += hc * zc; szg += zc * gc;
                                        }
                                }
                                ssq = swap32_sum(ssq); shg = swap32_sum(shg);
                                shz = swap32_sum(shz); szg = swap32_sum(szg);
                                const float inv_c = 1.f / (float)(32 * A);
                                const float r = rsqrtf(ssq * inv_c + p.pn_eps);
                                const float r2 = r * r, r3 = r2 * r;
                                const float ma = shg * inv_c, mb = shz * inv_c, mm = szg * inv_c;
                                const float kz = 3.f * r3 * r2 * mb * mm - r3 * ma;   // coefficient of z in y2
#pragma unroll
                                for (int a = 0; a < A; ++a)
#pragma unroll
                                    for (int v = 0; v < NV; ++v) {
                                        float o1[EV], o2[EV];
#pragma unroll
                                        for (int e = 0; e < EV; ++e) {
                                            const float zc = zv[a][v][e], gc = gv[a][v][e], hc = hv[a][v][e];
                                            o1[e] = r * (hc - zc * r2 * mb);
                                            o2[e] = kz * zc - r3 * (mm * hc + mb * gc);
                                        }
                                        if (inside) {
                                            if constexpr (SZ == 4) {
                                                st4(reinterpret_cast<float*>(y) + off + a * 32 + v * 8 + hi * 4, o1);
                                                st4(reinterpret_cast<float*>(p.y2) + off + a * 32 + v * 8 + hi * 4, o2);
                                            } else {
                                                uint4 w4;
                                                w4.x = pack_bf16x2(o1[0], o1[1]); w4.y = pack_bf16x2(o1[2], o1[3]); w4.z = pack_bf16x2(o1[4], o1[5]); w4.w = pack_bf16x2(o1[6], o1[7]);
                                                *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(y) + off + a * 32 + v * 16 + hi * 8) = w4;
                                                w4.x = pack_bf16x2(o2[0], o2[1]); w4.y = pack_bf16x2(o2[2], o2[3]); w4.z = pack_bf16x2(o2[4], o2[5]); w4.w = pack_bf16x2(o2[6], o2[7]);
                                                *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.y2) + off + a * 32 + v * 16 + hi * 8) = w4;
                                            }
                                        }
                                    }
                            } else if constexpr (NORM == 1) {
                                // the block owns every channel of the pixel (OC == 32 A): pixel norm in the same pass.  The lane and its
                                // partner in the other half hold the pixel's channels between them.
                                float o[A][4][4], ssq = 0.f;
#pragma unroll
                                for (int a = 0; a < A; ++a) {
                                    finish(ph, a, b, o[a]);
#pragma unroll
                                    for (int k = 0; k < 16; ++k) ssq += o[a][k >> 2][k & 3] * o[a][k >> 2][k & 3];
                                }
                                ssq = swap32_sum(ssq);
                                const float r = rsqrtf(ssq * (1.f / (float)(32 * A)) + p.pn_eps);
#pragma unroll
                                for (int a = 0; a < A; ++a) {
                                    if (y) {   // the pre-norm activation, kept for the backward
                                        float oz[4][4];
#pragma unroll
                                        for (int k = 0; k < 16; ++k) oz[k >> 2][k & 3] = o[a][k >> 2][k & 3];
                                        store(y, off, a, oz, inside, static_cast<const mvec_t*>(nullptr));
                                    }
#pragma unroll
                                    for (int k = 0; k < 16; ++k) o[a][k >> 2][k & 3] *= r;
                                    store(reinterpret_cast<T*>(p.y2), off, a, o[a], inside, static_cast<const mvec_t*>(nullptr));
                                }
                            } else {
                                mvec_t mz_one[A][NV];
                                if constexpr (!MASK_ALL) {
                                    if (p.mask) mask_fetch(off, inside, mz_one);
                                }
#pragma unroll
                                for (int a = 0; a < A; ++a) {
                                    float o[4][4];
                                    finish(ph, a, b, o);
                                    const mvec_t* mz = MASK_ALL ? mz_all[MASK_ALL ? b * NPH + ph : 0][a] : mz_one[a];
                                    store(y, off, a, o, inside, mz, p.mask ? (use_mb ? 2 : 1) : 0, emit_bits);
                                }
                            }
                        }
                    }
                    zero_acc();
                }
#endif
                block_barrier();
#ifdef GS_IGEMM_TRACE
                GS_TR(4 + tr_stage);
                ++tr_stage;
#endif
                if (tg == NTG - 1) c_pb = c_pb + 1 == NPB ? 0 : c_pb + 1;
                if (!RESIDENT) c_wb = c_wb + 1 == NWB ? 0 : c_wb + 1;
            }
        }
        if (++done >= count) break;
        item += stride;
    }
    GS_TR(63);
    wait_vmcnt(0);  // the (empty) stages issued past the end must have retired before this block's LDS is handed on
}

// --------------------------------------------------------------------- weight gradient
// gw[tap][ic][oc] = sum_pixels x[in(pixel,tap)][ic] * gy[pixel][oc].  MFMA with K = pixels:
// A[i=ic][k=pixel], B[k=pixel][j=oc].  A block owns a 32x32 (ic,oc) tile for all 9 taps and
// strides over spatial tiles (`slice`); its 4 waves split each tile's pixels, then reduce through
// LDS and write one fp32 partial per slice (summed by wgrad_reduce_kernel -> deterministic).
template <typename T, int MODE, int TW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(
    const WgradSrcs srcs, float* __restrict__ part,
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
        int n;
        const int src = wgrad_source(srcs, b / tiles_y, n);
        const T* __restrict__ x = reinterpret_cast<const T*>(srcs.x[src]);
        const T* __restrict__ gy = reinterpret_cast<const T*>(srcs.gy[src]);
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

// acc += a.lo + a.hi for a packed bf16 pair (v_dot2c_f32_bf16 against (1, 1); hipcc has no builtin for it on gfx950)
__device__ inline void add_bf16_pair(float& acc, unsigned int a) {
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(0x3F803F80u));
}

// OT = 2: the block owns TWO 32-channel output tiles (a 32 x 64 pair): the patch is staged and its fragments are read once for both --
// the 32 -> 64 stride-2 layer of the top of the pyramid re-staged its (4-5x larger) patch for each of its two output tiles.
template <int MODE, int TW, int OT = 1>
__global__ __launch_bounds__(192) void conv_wgrad_bf16_kernel(
    const WgradSrcs srcs, float* __restrict__ part,
    int N, int Hi, int Wi, int IC, int OC, int Hb, int Wb, int tiles_x, int tiles_y, int ntiles, int nslices, int with_bias) {
    constexpr bool S2 = MODE == MODE_S2;
    constexpr int NP = S2 ? 128 : 256;
    constexpr int TH = NP / TW;
    constexpr int PH = patch_dim<MODE>(TH), PW = patch_dim<MODE>(TW);
    constexpr int S = S2 ? 2 : 1;
    constexpr int XCH = PH * PW * 4, GCH = NP * 4 * OT;     // 16-byte chunks to stage (gradient tile: OT planes of 32 channels)
    constexpr int XIT = (XCH + 191) / 192, GIT = (GCH + 191) / 192;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[(PH * PW + NP * OT) * 64];
    unsigned char* const lx_ = lds_raw;
    unsigned char* const lg_ = lds_raw + PH * PW * 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int n_ict = IC / 32;
    const int ic0 = (blockIdx.x % n_ict) * 32, oc0 = (blockIdx.x / n_ict) * 32 * OT;
    const int slice = blockIdx.y;
    // transposing-read supplier role of this lane: pixel row (lane & 15) >> 2 of the 4-row block, channel quad
    const int t_row = (lane & 15) >> 2;
    const int t_col = (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;  // byte offset inside the 64-byte row

    f32x16 acc[OT][3];
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[o][t][r] = 0.f;
    // bias gradient = sum over pixels of gy: the gradient fragment of a lane is 8 pixels of its output channel, four packed
    // dot-2 adds per pixel group fold them into one register; done by the first wave of the blocks of input-channel tile 0
    const bool bias_wave = with_bias && wv == 0 && ic0 == 0;
    float accb[OT];
#pragma unroll
    for (int o = 0; o < OT; ++o) accb[o] = 0.f;

    // Software pipeline over the block's tiles (GS_WGRAD_THIN_PREFETCH, default on): the global loads of tile t + 1 are issued -- into
    // registers -- BEFORE the MFMAs of tile t and stored to LDS after them, so that a block's memory latency runs under its own MFMAs instead
    // of only under those of the two other blocks of the CU.  These layers are HBM-bound (32 channels: 144 flop/byte): what counts is bytes in
    // flight per CU.
    uint4 xv[XIT], gv[GIT];
    unsigned int xok = 0, gok = 0;
    bool bias_fetched = false;
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        int b = tile;
        const int tile_x = b % tiles_x;
        b /= tiles_x;
        const int tile_y = b % tiles_y;
        int n;
        const int src = wgrad_source(srcs, b / tiles_y, n);
        const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(srcs.x[src]);
        const bf16_t* __restrict__ gy = reinterpret_cast<const bf16_t*>(srcs.gy[src]);
        bias_fetched = bias_wave && ((srcs.bias_mask >> src) & 1u);
        const int by = tile_y * TH, bx = tile_x * TW;
        const int oy0 = S2 ? 2 * by : by - 1;
        const int ox0 = S2 ? 2 * bx : bx - 1;
        xok = 0; gok = 0;
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
            const int c = tid + 192 * it;   // chunk c of LDS plane c / (NP * 4): pixel (c >> 2) % NP, channels 32 plane + 8 (c & 3)
            const int pix = (c >> 2) % NP, part4 = (c & 3) + 4 * (c / (NP * 4));
            const int gy_ = by + pix / TW, gx_ = bx + pix % TW;
            const bool ok = c < GCH && gy_ < Hb && gx_ < Wb;
            gok |= ok ? (1u << it) : 0u;
            gv[it] = *reinterpret_cast<const uint4*>(ok ? gy + (((long)n * Hb + gy_) * Wb + gx_) * OC + oc0 + part4 * 8 : gy);
        }
    };
    if (slice < ntiles) fetch(slice);
    for (int tile = slice; tile < ntiles; tile += nslices) {
        const bool do_bias = bias_fetched;
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
        if (GS_WGRAD_THIN_PREFETCH && tile + nslices < ntiles) fetch(tile + nslices);   // in flight under the MFMAs below
        __syncthreads();
        // ---- MFMAs: this wave's kernel row (ky = wv) over every 16-pixel group of the tile
#pragma unroll 2
        for (int g = 0; g < NP / 16; ++g) {
            const int ty = (g * 16) / TW, tx0 = (g * 16) % TW + 8 * hi;
            bf16x8 bfrag[OT];
#pragma unroll
            for (int o = 0; o < OT; ++o) {
                const unsigned char* gp = lg_ + o * NP * 64 + (ty * TW + tx0 + t_row) * 64 + t_col;
                const uint2 b0 = lds_tr16(gp), b1 = lds_tr16(gp + 4 * 64);
                bfrag[o] = mk_frag(b0.x, b0.y, b1.x, b1.y);
                if (do_bias) { add_bf16_pair(accb[o], b0.x); add_bf16_pair(accb[o], b0.y); add_bf16_pair(accb[o], b1.x); add_bf16_pair(accb[o], b1.y); }
            }
            const unsigned char* xp = lx_ + (((ty * S + wv) * PW + tx0 * S) + t_row * S) * 64 + t_col;
            if (!S2) {
                const uint2 d0 = lds_tr16(xp), d1 = lds_tr16(xp + 4 * 64), d2 = lds_tr16(xp + 8 * 64);
                const bf16x8 a0 = mk_frag(d0.x, d0.y, d1.x, d1.y), a1 = mk_frag(shr16(d0.y, d0.x), shr16(d1.x, d0.y), shr16(d1.y, d1.x), shr16(d2.x, d1.y)),
                             a2 = mk_frag(d0.y, d1.x, d1.y, d2.x);
#pragma unroll
                for (int o = 0; o < OT; ++o) {
                    acc[o][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bfrag[o], acc[o][0], 0, 0, 0);
                    acc[o][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bfrag[o], acc[o][1], 0, 0, 0);
                    acc[o][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bfrag[o], acc[o][2], 0, 0, 0);
                }
            } else {
                // even columns 2(p)+0 / +2 share a 9-pixel window; odd columns 2(p)+1 are their own 8-pixel window
                const uint2 e0 = lds_tr16(xp), e1 = lds_tr16(xp + 8 * 64), e2 = lds_tr16(xp + 16 * 64);
                const uint2 o0 = lds_tr16(xp + 64), o1 = lds_tr16(xp + 9 * 64);
                const bf16x8 a0 = mk_frag(e0.x, e0.y, e1.x, e1.y), a1 = mk_frag(o0.x, o0.y, o1.x, o1.y),
                             a2 = mk_frag(shr16(e0.y, e0.x), shr16(e1.x, e0.y), shr16(e1.y, e1.x), shr16(e2.x, e1.y));
#pragma unroll
                for (int o = 0; o < OT; ++o) {
                    acc[o][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bfrag[o], acc[o][0], 0, 0, 0);
                    acc[o][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bfrag[o], acc[o][1], 0, 0, 0);
                    acc[o][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bfrag[o], acc[o][2], 0, 0, 0);
                }
            }
        }
        if (!GS_WGRAD_THIN_PREFETCH && tile + nslices < ntiles) fetch(tile + nslices);
    }
    // ---- each wave owns its 3 taps: D[ic i][oc j], lane = (j = l31, i = (r&3) + 8(r>>2) + 4hi)
    const long pstride = 9L * IC * OC + (with_bias ? OC : 0);   // fp32 elements per slice: 9 taps (+ the bias row)
#pragma unroll
    for (int o = 0; o < OT; ++o) {
        if (bias_wave) {   // the two lane halves hold different pixels of the same channel
            const float tot = swap32_sum(accb[o]);
            if (hi == 0) part[(long)slice * pstride + 9L * IC * OC + oc0 + o * 32 + l31] = tot;
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            float* dst = part + (long)slice * pstride + (((long)wv * 3 + kx) * IC + ic0) * OC + oc0 + o * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(long)((r & 3) + 8 * (r >> 2) + 4 * hi) * OC] = acc[o][kx][r];
        }
    }
}

// The same contraction for the HBM-bound top of the pyramid (32 input channels: 144 flop / byte), staged by LDS-DMA and wave-specialised.
// SQ counters of conv_wgrad_bf16_kernel on these layers (profiles/r04_n_step_sq_pmc.txt): ~810 VALU instructions per tile and wave for 48 MFMAs --
// the per-thread address arithmetic of 14 16-byte loads, their border selects and LDS stores -- waves issuing 47 % of the time, MFMA pipe busy
// 17 %, 2.9 TB/s over x + gy where a streaming kernel reaches 4.5-6: the kernel is bound by its own instruction stream, not by HBM.  Here
//   * wave 3 is the LOADER: it owns the tile descriptors and issues every DMA piece of tile t + 1 (1 KiB = 16 pixel rows of 64 bytes each,
//     plain row-major: exactly the [pixel][32 channels] layout the transposing reads want; rows above / below the image fall outside the
//     per-image descriptor and arrive as zeros, columns outside are forced out of range) while
//   * waves 0-2 (wave = kernel row, 3 taps, no cross-wave reduction: as conv_wgrad_bf16_kernel) run the MFMAs of tile t from the other buffer;
//   * ONE barrier per tile: behind the loader's vmcnt(0).  It publishes tile t and, since the loader issues tile t + 1 only after it, also
//     says that every compute wave is done with the buffer tile t + 1 goes to.
// Two blocks per CU (2 x ~77 KiB of LDS): 77 KiB in flight per CU at any time.  TW = 32 only; stride 2 takes 64-pixel tiles (its patch is
// 4.6x the tile).  Same partial layout as conv_wgrad_bf16_kernel: the fold does not know which kernel ran.
template <int MODE, int OT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_thin_dma_kernel(   // (two waves per SIMD: two blocks per CU must fit the register file)
    const WgradSrcs srcs, float* __restrict__ part,
    int N, int Hi, int Wi, int IC, int OC, int Hb, int Wb, int tiles_x, int tiles_y, int ntiles, int nslices, int with_bias) {
    constexpr bool S2 = MODE == MODE_S2;
    constexpr int TW = 32;
    constexpr int NP = S2 ? 64 : 256;
    constexpr int TH = NP / TW;
    constexpr int PH = patch_dim<MODE>(TH), PW = patch_dim<MODE>(TW);
    constexpr int S = S2 ? 2 : 1;
    constexpr int XP = (PH * PW + 15) / 16;     // 1 KiB pieces (16 rows of 64 bytes) of the patch ...
    constexpr int GP = NP / 16;                 // ... and of one 32-channel plane of the gradient tile
    constexpr int XB = XP * 1024, GB = NP * 64;
    constexpr int BUF = XB + OT * GB;           // one staged tile; two of them
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const unsigned a_base = (unsigned)(uintptr_t)lds_raw;

    const int tid = threadIdx.x;
    const int lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wv == 3;
    const int oc0 = blockIdx.x * 32 * OT;       // (IC == 32: one input-channel tile)
    const int slice = blockIdx.y;
    const int t_row = (lane & 15) >> 2;
    const int t_col = (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;

    // ---- loader: piece j of the patch = rows 16 j + (lane >> 2) of its PH x PW pixel rows.  The (patch row, column) of a lane's row is
    //      walked incrementally from piece to piece (+16 columns, wrapping at PW) instead of being kept in 2 x XP registers: the register
    //      file is shared with the compute waves' accumulators, and the loader has instruction slots to spare
    const int x_lx0 = lane >> 2;                                 // row of piece 0: patch row 0, column lane >> 2 (PW > 16)
    const int x_voff0 = (x_lx0 * IC) * 2 + (lane & 3) * 16;
    // a gradient piece = 16 consecutive pixels of one tile row: pixel (j >> 1, 16 (j & 1) + (lane >> 2))
    const int g_lane = ((lane >> 2) * OC) * 2 + (lane & 3) * 16;
    const unsigned ximg = (unsigned)Hi * Wi * IC * 2, gimg = (unsigned)Hb * Wb * OC * 2;

    f32x16 acc[OT][3];
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[o][t][r] = 0.f;
    const bool bias_wave = with_bias && wv == 0;
    float accb[OT];
#pragma unroll
    for (int o = 0; o < OT; ++o) accb[o] = 0.f;

    auto tile_coords = [&](int tile, int& n, int& by, int& bx) __attribute__((always_inline)) {   // -> source index
        int b = tile;
        const int tile_x = b % tiles_x;
        b /= tiles_x;
        const int tile_y = b % tiles_y;
        const int src = wgrad_source(srcs, b / tiles_y, n);
        by = tile_y * TH;
        bx = tile_x * TW;
        return src;
    };
    auto issue_tile = [&](int tile, int bufi) __attribute__((always_inline)) {
        int n, by, bx;
        const int src = tile_coords(tile, n, by, bx);
        const int oy0 = S2 ? 2 * by : by - 1, ox0 = S2 ? 2 * bx : bx - 1;
        const i32x4 rs_x = make_rsrc(reinterpret_cast<const unsigned char*>(srcs.x[src]) + (size_t)n * ximg, ximg);
        const i32x4 rs_g = make_rsrc(reinterpret_cast<const unsigned char*>(srcs.gy[src]) + (size_t)n * gimg, gimg);
        const int xorg = ((oy0 * Wi + ox0) * IC) * 2;
        const unsigned a_x = a_base + bufi * BUF, a_g = a_x + XB;
        int lx = x_lx0, voff = xorg + x_voff0;
        asm volatile("" : "+v"(lx));   // (opaque per tile: or the compiler hoists the whole column walk out of the tile loop -- 2 x XP registers again)
        const int wrap = ((Wi - PW) * IC) * 2;                  // byte step from (ly, lx + PW) to (ly + 1, lx)
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            // (the last piece's rows past the patch: any column outside the image will do -- they are never read)
            const bool in = (unsigned)(ox0 + lx) < (unsigned)Wi && (j * 16 + 15 < PH * PW || j * 16 + (lane >> 2) < PH * PW);
            lds_dma16_stream(a_x + j * 1024, in ? (unsigned)voff : 0x80000000u, rs_x);
            lx += 16;
            voff += 16 * IC * 2;
            const bool w = lx >= PW;
            lx = w ? lx - PW : lx;
            voff = w ? voff + wrap : voff;
        }
        const bool in0 = bx + (lane >> 2) < Wb, in1 = bx + 16 + (lane >> 2) < Wb;
#pragma unroll
        for (int o = 0; o < OT; ++o)
#pragma unroll
            for (int j = 0; j < GP; ++j) {
                const int gy_ = by + (j >> 1);                        // (wave-uniform)
                const int gorg = ((gy_ * Wb + bx + 16 * (j & 1)) * OC + oc0 + 32 * o) * 2;
                const unsigned v = (gy_ < Hb && ((j & 1) ? in1 : in0)) ? (unsigned)(gorg + g_lane) : 0x80000000u;
                lds_dma16_stream(a_g + o * GB + j * 1024, v, rs_g);
            }
    };

    int buf = 0;
    if (loader && slice < ntiles) issue_tile(slice, 0);
    for (int tile = slice; tile < ntiles; tile += nslices) {
        if (loader) wait_vmcnt(0);   // this tile has landed ...
        block_barrier();             // ... for everybody; and everybody is done with the other buffer
        if (loader) {
            if (tile + nslices < ntiles) issue_tile(tile + nslices, buf ^ 1);
        } else {
            bool do_bias = false;
            if (bias_wave) {
                int n, by, bx;
                do_bias = (srcs.bias_mask >> tile_coords(tile, n, by, bx)) & 1u;
            }
            const unsigned char* const lx_ = lds_raw + buf * BUF;
            const unsigned char* const lg_ = lx_ + XB;
            auto group = [&](int g) __attribute__((always_inline)) {
                const int ty = (g * 16) / TW, tx0 = (g * 16) % TW + 8 * hi;
                bf16x8 bfrag[OT];
#pragma unroll
                for (int o = 0; o < OT; ++o) {
                    const unsigned char* gp = lg_ + o * GB + (ty * TW + tx0 + t_row) * 64 + t_col;
                    const uint2 b0 = lds_tr16(gp), b1 = lds_tr16(gp + 4 * 64);
                    bfrag[o] = mk_frag(b0.x, b0.y, b1.x, b1.y);
                    if (do_bias) { add_bf16_pair(accb[o], b0.x); add_bf16_pair(accb[o], b0.y); add_bf16_pair(accb[o], b1.x); add_bf16_pair(accb[o], b1.y); }
                }
                const unsigned char* xp = lx_ + (((ty * S + wv) * PW + tx0 * S) + t_row * S) * 64 + t_col;
                if (!S2) {
                    const uint2 d0 = lds_tr16(xp), d1 = lds_tr16(xp + 4 * 64), d2 = lds_tr16(xp + 8 * 64);
                    const bf16x8 a0 = mk_frag(d0.x, d0.y, d1.x, d1.y), a1 = mk_frag(shr16(d0.y, d0.x), shr16(d1.x, d0.y), shr16(d1.y, d1.x), shr16(d2.x, d1.y)),
                                 a2 = mk_frag(d0.y, d1.x, d1.y, d2.x);
#pragma unroll
                    for (int o = 0; o < OT; ++o) {
                        acc[o][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bfrag[o], acc[o][0], 0, 0, 0);
                        acc[o][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bfrag[o], acc[o][1], 0, 0, 0);
                        acc[o][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bfrag[o], acc[o][2], 0, 0, 0);
                    }
                } else {
                    // even columns 2(p)+0 / +2 share a 9-pixel window; odd columns 2(p)+1 are their own 8-pixel window
                    const uint2 e0 = lds_tr16(xp), e1 = lds_tr16(xp + 8 * 64), e2 = lds_tr16(xp + 16 * 64);
                    const uint2 o0 = lds_tr16(xp + 64), o1 = lds_tr16(xp + 9 * 64);
                    const bf16x8 a0 = mk_frag(e0.x, e0.y, e1.x, e1.y), a1 = mk_frag(o0.x, o0.y, o1.x, o1.y),
                                 a2 = mk_frag(shr16(e0.y, e0.x), shr16(e1.x, e0.y), shr16(e1.y, e1.x), shr16(e2.x, e1.y));
#pragma unroll
                    for (int o = 0; o < OT; ++o) {
                        acc[o][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bfrag[o], acc[o][0], 0, 0, 0);
                        acc[o][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bfrag[o], acc[o][1], 0, 0, 0);
                        acc[o][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bfrag[o], acc[o][2], 0, 0, 0);
                    }
                }
            };
            if constexpr (OT == 1) {
#pragma unroll 2
                for (int g = 0; g < NP / 16; ++g) group(g);
            } else {   // (two output tiles: 96 accumulators -- one group in flight keeps the wave within 256 registers, i.e. two blocks per CU)
#pragma unroll 1
                for (int g = 0; g < NP / 16; ++g) group(g);
            }
        }
        buf ^= 1;
    }
    if (loader) return;
    // ---- each compute wave owns its 3 taps: D[ic i][oc j], lane = (j = l31, i = (r&3) + 8(r>>2) + 4hi)
    const long pstride = 9L * IC * OC + (with_bias ? OC : 0);   // fp32 elements per slice: 9 taps (+ the bias row)
#pragma unroll
    for (int o = 0; o < OT; ++o) {
        if (bias_wave) {   // the two lane halves hold different pixels of the same channel
            const float tot = swap32_sum(accb[o]);
            if (hi == 0) part[(long)slice * pstride + 9L * IC * OC + oc0 + o * 32 + l31] = tot;
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            float* dst = part + (long)slice * pstride + (((long)wv * 3 + kx) * IC) * OC + oc0 + o * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(long)((r & 3) + 8 * (r >> 2) + 4 * hi) * OC] = acc[o][kx][r];
        }
    }
}

// One 16-pixel group of the 64 x 64-tile weight-gradient kernels: 9 MFMAs (3 kernel rows x 3 taps) against the gradient fragment, with the
// OTHER work of the wave interleaved between them -- the fragment reads of the next group (2 gradient + 9 / 15 input reads), the
// v_alignbit windows of the shifted taps, the DMA pieces of the next tile.  A wave issues in order and a 32x32x16 MFMA occupies the
// pipe for 32 cycles, so only what is issued right behind an MFMA runs in its shadow; with the reads and the DMA issue in front of the
// nine MFMAs of a group (round 1) the pipe idled half of the time (measured: 9.7 K cycles per 4.6 K-cycle unit).
//   needs in scope: fb[2][2], fx[2][3][XR], acc[9], accb, do_bias, S2, S, PW, TW, NG, XR, hi, t_row, t_col, issue_piece
#ifndef GS_WGABL_NOFRAG
#define GS_WGABL_NOFRAG 0   // timing ablations (results wrong): no fragment reads / no DMA of the next tile / no MFMAs
#endif
#ifndef GS_WGABL_NODMA
#define GS_WGABL_NODMA 0
#endif
#define GS_WG_GROUP_STEP(GI, MORE, XPL, GPL, NBUF)                                                                                  \
    do {                                                                                                                            \
        constexpr int cur_ = (GI) & 1, nxt_ = cur_ ^ 1;                                                                             \
        constexpr bool pre_ = (GI) + 1 < NG;                                                                                        \
        constexpr int NR_ = 2 + 3 * XR;                 /* fragment reads of the next group */                                      \
        constexpr int RPS_ = (NR_ + 8) / 9;             /* ... per MFMA slot */                                                     \
        constexpr int nty_ = (((GI) + 1) * 16) / TW, ntxc_ = (((GI) + 1) * 16) % TW;                                                \
        const unsigned char* const ngp_ = g_ptr_((GPL), nty_, ntxc_);                                                               \
        auto next_read_ = [&](int r) __attribute__((always_inline)) {                                                               \
            if (r < 2) { fb[nxt_][r] = lds_tr16(ngp_ + r * 4 * GROWB); return; }                                                    \
            const int ky = (r - 2) / XR, k = (r - 2) % XR;                                                                          \
            if (!S2) { fx[nxt_][ky][k] = lds_tr16(x_ptr_((XPL), nty_, ky, ntxc_, 0) + k * 4 * XROWB); return; }                     \
            /* stride 2: reads 0-2 = the even columns (rows +0, +8, +16), 3-4 = the odd ones (rows +1, +9) */                       \
            fx[nxt_][ky][k] = k < 3 ? lds_tr16(x_ptr_((XPL), nty_, ky, ntxc_, 0) + k * 8 * XROWB)                                   \
                                    : lds_tr16(x_ptr_((XPL), nty_, ky, ntxc_, 1) + (k - 3) * 8 * XROWB);                            \
        };                                                                                                                          \
        auto slot_ = [&](int m) __attribute__((always_inline)) {                                                                    \
            if (pre_ && !GS_WGABL_NOFRAG) {                                                                                         \
                _Pragma("unroll") for (int r = m * RPS_; r < (m + 1) * RPS_ && r < NR_; ++r) next_read_(r);                         \
            }                                                                                                                       \
            if ((MORE) && !GS_WGABL_NODMA) {                                                                                        \
                _Pragma("unroll") for (int q = (GI) * PPG; q < ((GI) + 1) * PPG && q < NPIECE; ++q)                                 \
                    if ((q - (GI) * PPG) * 9 / PPG == m) issue_piece(q, NBUF);                                                      \
            }                                                                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
        };                                                                                                                          \
        const uint2 b0_ = fb[cur_][0], b1_ = fb[cur_][1];                                                                           \
        const bf16x8 bfrag_ = mk_frag(b0_.x, b0_.y, b1_.x, b1_.y);                                                                  \
        if (do_bias) { add_bf16_pair(accb, b0_.x); add_bf16_pair(accb, b0_.y); add_bf16_pair(accb, b1_.x); add_bf16_pair(accb, b1_.y); } \
        _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {                                                                          \
            if (!S2) {                                                                                                              \
                const uint2 d0 = fx[cur_][ky][0], d1 = fx[cur_][ky][1], d2 = fx[cur_][ky][2];                                       \
                acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(d0.x, d0.y, d1.x, d1.y), bfrag_, acc[ky * 3 + 0], 0, 0, 0); \
                slot_(ky * 3 + 0);                                                                                                  \
                acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(shr16(d0.y, d0.x), shr16(d1.x, d0.y), shr16(d1.y, d1.x), shr16(d2.x, d1.y)), bfrag_, acc[ky * 3 + 1], 0, 0, 0); \
                slot_(ky * 3 + 1);                                                                                                  \
                acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(d0.y, d1.x, d1.y, d2.x), bfrag_, acc[ky * 3 + 2], 0, 0, 0); \
                slot_(ky * 3 + 2);                                                                                                  \
            } else {                                                                                                                \
                /* even columns 2(p)+0 / +2 share a 9-pixel window; odd columns 2(p)+1 are their own 8-pixel window */              \
                const uint2 e0 = fx[cur_][ky][0], e1 = fx[cur_][ky][1], e2 = fx[cur_][ky][2], o0 = fx[cur_][ky][3], o1 = fx[cur_][ky][4]; \
                acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(e0.x, e0.y, e1.x, e1.y), bfrag_, acc[ky * 3 + 0], 0, 0, 0); \
                slot_(ky * 3 + 0);                                                                                                  \
                acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(o0.x, o0.y, o1.x, o1.y), bfrag_, acc[ky * 3 + 1], 0, 0, 0); \
                slot_(ky * 3 + 1);                                                                                                  \
                acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk_frag(shr16(e0.y, e0.x), shr16(e1.x, e0.y), shr16(e1.y, e1.x), shr16(e2.x, e1.y)), bfrag_, acc[ky * 3 + 2], 0, 0, 0); \
                slot_(ky * 3 + 2);                                                                                                  \
            }                                                                                                                       \
        }                                                                                                                           \
    } while (0)

// 64 x 64 (input x output channel) tiles per block for layers with >= 64 channels on both sides: the four 32 x 32 pairs of
// the tile share ONE staged copy of the input patch and of the gradient tile (a 32 x 32 block re-stages the patch for every
// output tile and the gradients for every input tile: twice the L2 -> LDS stream per MFMA, and that stream is what bounds the
// kernel).  Block = 256 threads = 4 waves, wave w owns the pair (input tile w >> 1, output tile w & 1) for all 9 taps
// (144 fp32 accumulators); operands sit in LDS as two 32-channel planes per side so that the transposing reads keep their
// conflict-free 64-byte rows.
template <int MODE, int TW>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_2x2_kernel(
    const WgradSrcs srcs, float* __restrict__ part,
    int N, int Hi, int Wi, int IC, int OC, int Hb, int Wb, int tiles_x, int tiles_y, int ntiles, int nslices, int with_bias) {
    constexpr bool S2 = MODE == MODE_S2;
    constexpr int NP = S2 ? 64 : 256;   // stride 2: the patch is 4-5x the tile, 64 output pixels keep two staged tiles in LDS
    constexpr int TH = NP / TW;
    constexpr int PH = patch_dim<MODE>(TH), PW = patch_dim<MODE>(TW);
    constexpr int S = S2 ? 2 : 1;
    constexpr int XRG = (PH * PW + 15) / 16;          // 16-row groups (= 1 KiB LDS-DMA pieces) of a patch plane
    constexpr int GRG = NP / 16;
    constexpr int XK = (XRG + 3) / 4, GK = GRG / 4;   // row groups per wave (every wave issues the same number of pieces)
    constexpr int XPL = XK * 4096, GPL = NP * 64;     // bytes of one 32-channel plane
    constexpr int BUF = 2 * XPL + 2 * GPL;            // one staged tile; two of them: the DMA of tile t+1 runs under the MFMAs of tile t
    constexpr int NPIECE = 2 * XK + 2 * GK;           // DMA pieces a wave issues per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const unsigned a_base = (unsigned)(uintptr_t)lds_raw;

    const int tid = threadIdx.x;
    const int lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_ict = IC / 64;
    const int ic0 = (blockIdx.x % n_ict) * 64, oc0 = (blockIdx.x / n_ict) * 64;
    const int it = wv >> 1, ot = wv & 1;
    const int slice = blockIdx.y;
    const int t_row = (lane & 15) >> 2;
    const int t_col = (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;

    // staging = LDS-DMA (no registers, see the implicit-GEMM kernel): a piece is 16 rows x 64 bytes of one channel plane; rows
    // above / below the image fall outside the per-image descriptor (zero fill), columns outside are forced out of range.
    int x_voff[XK], x_lx[XK];
#pragma unroll
    for (int k = 0; k < XK; ++k) {
        const int row = (wv + 4 * k) * 16 + (lane >> 2);
        const int ly = row / PW, lx = row - ly * PW;
        x_voff[k] = ((ly * Wi + lx) * IC) * 2 + (lane & 3) * 16;
        x_lx[k] = row < PH * PW ? lx : 0x40000000;
    }
    const unsigned ximg = (unsigned)Hi * Wi * IC * 2, gimg = (unsigned)Hb * Wb * OC * 2;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // bias gradient (see conv_wgrad_bf16_kernel): the two waves of input tile 0 in the blocks of input-channel tile 0
    const bool bias_wave = with_bias && it == 0 && ic0 == 0;
    bool do_bias = false, bias_next = false;   // per tile: does the tile's source contribute to the bias gradient
    float accb = 0.f;

    // one DMA piece of a tile (q in [0, NPIECE)): patch plane 0 / 1 pieces first, then the gradient planes
    int n_t = 0, by_t = 0, bx_t = 0, ox0_t = 0, xorg_t = 0;
    i32x4 rs_xt = make_rsrc(srcs.x[0], ximg), rs_gt = make_rsrc(srcs.gy[0], gimg);
    auto tile_setup = [&](int tile) __attribute__((always_inline)) {
        int b = tile;
        const int tile_x = b % tiles_x;
        b /= tiles_x;
        const int tile_y = b % tiles_y;
        const int src = wgrad_source(srcs, b / tiles_y, n_t);
        bias_next = bias_wave && ((srcs.bias_mask >> src) & 1u);
        by_t = tile_y * TH;
        bx_t = tile_x * TW;
        const int oy0 = S2 ? 2 * by_t : by_t - 1;
        ox0_t = S2 ? 2 * bx_t : bx_t - 1;
        rs_xt = make_rsrc(reinterpret_cast<const unsigned char*>(srcs.x[src]) + (size_t)n_t * ximg, ximg);
        rs_gt = make_rsrc(reinterpret_cast<const unsigned char*>(srcs.gy[src]) + (size_t)n_t * gimg, gimg);
        xorg_t = ((oy0 * Wi + ox0_t) * IC + ic0) * 2;
    };
    auto issue_piece = [&](int q, int bufi) __attribute__((always_inline)) {
        const unsigned a_x = a_base + bufi * BUF, a_g = a_x + 2 * XPL;
        if (q < 2 * XK) {
            const int k = q >> 1, pl = q & 1;
            unsigned v = (unsigned)(ox0_t + x_lx[k]) < (unsigned)Wi ? (unsigned)(xorg_t + x_voff[k]) : 0x80000000u;
            if (pl && v != 0x80000000u) v += 64;
            lds_dma16(a_x + pl * XPL + (wv + 4 * k) * 1024, v, rs_xt);
        } else {
            const int k = (q - 2 * XK) >> 1, pl = (q - 2 * XK) & 1;
            const int pix = (wv + 4 * k) * 16 + (lane >> 2);
            const int gy_ = by_t + pix / TW, gx_ = bx_t + pix % TW;
            unsigned v = gy_ < Hb && gx_ < Wb ? (unsigned)(((gy_ * Wb + gx_) * OC + oc0) * 2 + (lane & 3) * 16) : 0x80000000u;
            if (pl && v != 0x80000000u) v += 64;
            lds_dma16(a_g + pl * GPL + (wv + 4 * k) * 1024, v, rs_gt);
        }
    };

    // fragments of one 16-pixel group: the gradient columns (2 transposing reads) and, per kernel row, the 3 (stride 1) or
    // 5 (stride 2) reads of the input window.  Two sets: the reads of group g+1 are issued before the MFMAs of group g.
    constexpr int XR = S2 ? 5 : 3;
    constexpr int NG = NP / 16;
    uint2 fb[2][2], fx[2][3][XR];
    // fragment addresses (GS_WG_GROUP_STEP): two 32-channel planes of 64-byte rows per side
    constexpr int XROWB = 64, GROWB = 64;
    auto x_ptr_ = [&](const unsigned char* xpl, int ty, int ky, int txc, int c) __attribute__((always_inline)) {
        return xpl + (((ty * S + ky) * PW + (txc + 8 * hi) * S) + t_row * S + c) * 64 + t_col;
    };
    auto g_ptr_ = [&](const unsigned char* gpl, int ty, int txc) __attribute__((always_inline)) {
        return gpl + (ty * TW + txc + 8 * hi + t_row) * 64 + t_col;
    };
    auto load_group = [&](int g, int fbuf, const unsigned char* xpl, const unsigned char* gpl) __attribute__((always_inline)) {
        const int ty = (g * 16) / TW, txc = (g * 16) % TW;
        const unsigned char* gp = g_ptr_(gpl, ty, txc);
        fb[fbuf][0] = lds_tr16(gp);
        fb[fbuf][1] = lds_tr16(gp + 4 * GROWB);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const unsigned char* xp = x_ptr_(xpl, ty, ky, txc, 0);
            if (!S2) {
                fx[fbuf][ky][0] = lds_tr16(xp); fx[fbuf][ky][1] = lds_tr16(xp + 4 * XROWB); fx[fbuf][ky][2] = lds_tr16(xp + 8 * XROWB);
            } else {
                const unsigned char* xo = x_ptr_(xpl, ty, ky, txc, 1);
                fx[fbuf][ky][0] = lds_tr16(xp); fx[fbuf][ky][1] = lds_tr16(xp + 8 * XROWB); fx[fbuf][ky][2] = lds_tr16(xp + 16 * XROWB);
                fx[fbuf][ky][3] = lds_tr16(xo); fx[fbuf][ky][4] = lds_tr16(xo + 8 * XROWB);
            }
        }
    };

    int buf = 0;
    if (slice < ntiles) {
        tile_setup(slice);
#pragma unroll
        for (int q = 0; q < NPIECE; ++q) issue_piece(q, 0);
    }
    constexpr int PPG = (NPIECE + NG - 1) / NG;   // DMA pieces of the next tile issued per pixel group of this one
    for (int tile = slice; tile < ntiles; tile += nslices) {
        const bool more = tile + nslices < ntiles;
        wait_vmcnt(0);    // this tile has landed (the next one is issued below, under the MFMAs)
        block_barrier();
        do_bias = bias_next;
        if (more) tile_setup(tile + nslices);
        const unsigned char* const xpl = lds_raw + buf * BUF + it * XPL;
        const unsigned char* const gpl = lds_raw + buf * BUF + 2 * XPL + ot * GPL;
        load_group(0, 0, xpl, gpl);
        __builtin_amdgcn_sched_barrier(0);
        static_for<NG>([&](auto gc) __attribute__((always_inline)) { GS_WG_GROUP_STEP(decltype(gc)::value, more, xpl, gpl, buf ^ 1); });
        block_barrier();  // every wave is done with this buffer: the next iteration may overwrite it
        buf ^= 1;
    }
    // ---- D[ic i][oc j], lane = (j = l31, i = (r&3) + 8(r>>2) + 4hi)
    const long pstride = 9L * IC * OC + (with_bias ? OC : 0);
    if (bias_wave) {
        const float tot = swap32_sum(accb);
        if (hi == 0) part[(long)slice * pstride + 9L * IC * OC + oc0 + ot * 32 + l31] = tot;
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* dst = part + (long)slice * pstride + ((long)t * IC + ic0 + it * 32) * OC + oc0 + ot * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(long)((r & 3) + 8 * (r >> 2) + 4 * hi) * OC] = acc[t][r];
    }
}

// The same kernel over a GROUP of layers (conv_shared.h, SkGroup): block b walks the units [b T / nb, (b + 1) T / nb) of the group's
// unit list; the pipeline (DMA of unit u+1 under the MFMAs of unit u) runs straight across run and layer boundaries -- the MFMA
// side only sees staged LDS tiles, whatever layer they came from -- and the accumulators are flushed to partial `b + run` where the
// block's range leaves a run.
// SPEC: 8 waves -- waves 0-3 multiply (fragment reads, MFMAs, flushes), waves 4-7 (one beside each on its SIMD) stage: they own the tile
// descriptors and issue every DMA piece.  A wave issues one instruction per ~5 cycles; a group of 9 MFMAs (288 pipe cycles) leaves ~57
// issue slots and the reads, v_alignbit windows and DMA pieces of a group need ~75: measured 145 us with everything on four waves,
// 105 us with neither reads nor DMA (scripts/bench_wgrad_group.py with the GS_WGABL_* builds).
#ifndef GS_SK_ROW128
#define GS_SK_ROW128 1
#endif
#ifndef GS_SK_FRONT_S1
#define GS_SK_FRONT_S1 0
#endif
#ifndef GS_SK_FRONT_S2
#define GS_SK_FRONT_S2 1   // measured same-box (scripts/ab_wgrad.sh, profiles/r04_c_ab_wgrad_front.txt): stride-2 flush 131 -> 121 / 183 -> 167 us (16 / 24 images); stride 1: no change at 4 or 8
#endif
template <int MODE, int TW, bool SPEC>
__global__ __launch_bounds__(SPEC ? 512 : 256) void conv_wgrad_bf16_2x2_sk_kernel(const SkGroup g, float* __restrict__ part) {
    constexpr bool S2 = MODE == MODE_S2;
    constexpr int NP = S2 ? 64 : 256;
    constexpr int TH = NP / TW;
    constexpr int PH = patch_dim<MODE>(TH), PW = patch_dim<MODE>(TW);
    constexpr int S = S2 ? 2 : 1;
    // Staged layout.  R128 (default): ONE plane per side with 128-byte rows = all 64 channels of a pixel, a DMA piece = 8 whole rows, i.e.
    // whole 128-byte cache lines: the wave issues a piece every ~77 cycles instead of ~134 with the half-line rows of the two-plane
    // layout (scripts/probe/dma_rate.hip), and a unit's time IS the issuing wave's serial sum -- MFMAs + ~13 cycles per transposing read
    // + the DMA issue (model and counters: DESIGN.md 6.4).  The 32-channel half h of row r sits at (h ^ (r >> 1 & 1)) * 64, applied on
    // the DMA's source side, so that the four rows of a transposing read (r .. r + 3) cover all 64 banks as the 64-byte rows did.
    constexpr bool R128 = GS_SK_ROW128 != 0;
    constexpr int XRG = R128 ? (PH * PW + 7) / 8 : (PH * PW + 15) / 16;   // DMA pieces (1 KiB) of the patch: of its one plane / of each of its two
    constexpr int GRG = R128 ? NP / 8 : NP / 16;
    constexpr int XK = (XRG + 3) / 4, GK = GRG / 4;
    constexpr int XPL = R128 ? 0 : XK * 4096, GPL = R128 ? 0 : NP * 64;   // plane stride (two-plane layout)
    constexpr int XT = R128 ? XK * 4096 : 2 * XK * 4096, GT = NP * 128;   // bytes of the staged patch / gradient tile
    constexpr int BUF = XT + GT;
    constexpr int NPIECE = R128 ? XK + GK : 2 * XK + 2 * GK;
    constexpr int XROWB = R128 ? 128 : 64, GROWB = XROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const unsigned a_base = (unsigned)(uintptr_t)lds_raw;

    const int tid = threadIdx.x;
    const int lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wv8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = SPEC && wv8 >= 4, issuer = !SPEC || loader, computer = !SPEC || !loader;
    const int wv = wv8 & 3;   // index within the role
    const int it = wv >> 1, ot = wv & 1;
    const int t_row = (lane & 15) >> 2;
    const int t_col = (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;
    const long nb = gridDim.x, total = g.total_units;
    const int U0 = (int)((blockIdx.x * total) / nb), U1 = (int)(((blockIdx.x + 1) * total) / nb);
    if (U0 >= U1) return;

    // ---- DMA side: the unit being staged (one ahead of the one being multiplied)
    constexpr int RPP = R128 ? 8 : 16;              // rows per piece
    const int p_row = R128 ? lane >> 3 : lane >> 2;   // row of the piece this lane fetches 16 bytes of
    int x_ly[XK], x_lx[XK], x_voff[XK], x_sw[XK];
#pragma unroll
    for (int k = 0; k < XK; ++k) {
        const int row = (wv + 4 * k) * RPP + p_row;
        x_ly[k] = row / PW;
        x_lx[k] = row - x_ly[k] * PW;
        x_voff[k] = 0;
        x_sw[k] = R128 ? (((lane & 7) ^ (((row >> 1) & 1) << 2)) * 16) : (lane & 3) * 16;   // source bytes of the lane's 16-byte slot within the channel row
    }
    int j_n = 0, ct_n = 0, tile_n = 0;                               // layer, channel tile, pixel tile
    int Hi = 0, Wi = 0, IC = 0, OC = 0, Hb = 0, Wb = 0, tiles_x = 1, tiles_y = 1, ntiles = 1, n_ict = 1, nct = 1;
    unsigned ximg = 0, gimg = 0;
    int ic0_n = 0, oc0_n = 0, run_n = 0;
    bool bias_wave_n = false;
    auto load_job = [&](int j) __attribute__((always_inline)) {
        const SkJob& q = g.job[j];
        Hi = q.Hi; Wi = q.Wi; IC = q.IC; OC = q.OC; Hb = q.Hb; Wb = q.Wb;
        tiles_x = q.tiles_x; tiles_y = q.tiles_y; ntiles = q.ntiles; n_ict = q.n_ict; nct = q.nct;
        ximg = (unsigned)Hi * Wi * IC * 2;
        gimg = (unsigned)Hb * Wb * OC * 2;
#pragma unroll
        for (int k = 0; k < XK; ++k) x_voff[k] = ((x_ly[k] * Wi + x_lx[k]) * IC) * 2 + x_sw[k];
    };
    auto set_ct = [&](int j, int ct) __attribute__((always_inline)) {
        ic0_n = (ct % n_ict) * 64;
        oc0_n = (ct / n_ict) * 64;
        run_n = g.job[j].run_base + ct;
        bias_wave_n = g.job[j].gb != nullptr && it == 0 && ic0_n == 0;
    };
    int n_t = 0, by_t = 0, bx_t = 0, ox0_t = 0, xorg_t = 0;
    bool bias_next = false;
    i32x4 rs_xt = make_rsrc(g.job[0].srcs.x[0], 0), rs_gt = rs_xt;
    // pixel-tile coordinates of the unit being staged, advanced incrementally (three runtime divisions per unit cost more issue
    // slots than a group of MFMAs leaves)
    int tx_n = 0, ty_n = 0, img_n = 0;
    auto tile_setup = [&](int j) __attribute__((always_inline)) {
        const SkJob& q = g.job[j];
        const int src = wgrad_source(q.srcs, img_n, n_t);
        bias_next = bias_wave_n && ((q.srcs.bias_mask >> src) & 1u);
        if (issuer) {
            by_t = ty_n * TH;
            bx_t = tx_n * TW;
            const int oy0 = S2 ? 2 * by_t : by_t - 1;
            ox0_t = S2 ? 2 * bx_t : bx_t - 1;
            rs_xt = make_rsrc(reinterpret_cast<const unsigned char*>(q.srcs.x[src]) + (size_t)n_t * ximg, ximg);
            rs_gt = make_rsrc(reinterpret_cast<const unsigned char*>(q.srcs.gy[src]) + (size_t)n_t * gimg, gimg);
            xorg_t = ((oy0 * Wi + ox0_t) * IC + ic0_n) * 2;
        }
    };
    auto issue_piece = [&](int q, int bufi) __attribute__((always_inline)) {
        const unsigned a_x = a_base + bufi * BUF, a_g = a_x + XT;
        if (R128) {
            if (q < XK) {
                const int k = q;
                const bool in = (wv + 4 * k) * 8 + p_row < PH * PW && (unsigned)(ox0_t + x_lx[k]) < (unsigned)Wi;
                const unsigned v = in ? (unsigned)(xorg_t + x_voff[k]) : 0x80000000u;
                lds_dma16(__builtin_amdgcn_readfirstlane(a_x + (wv + 4 * k) * 1024), v, rs_xt);
            } else {
                const int k = q - XK;
                const int pix = (wv + 4 * k) * 8 + p_row;
                const int gy_ = by_t + pix / TW, gx_ = bx_t + pix % TW;
                const bool in = gy_ < Hb && gx_ < Wb;
                const unsigned v = in ? (unsigned)(((gy_ * Wb + gx_) * OC + oc0_n) * 2 + (((lane & 7) ^ (((pix >> 1) & 1) << 2)) * 16)) : 0x80000000u;
                lds_dma16(__builtin_amdgcn_readfirstlane(a_g + (wv + 4 * k) * 1024), v, rs_gt);
            }
            return;
        }
        if (q < 2 * XK) {
            const int k = q >> 1, pl = q & 1;
            const bool in = (wv + 4 * k) * 16 + (lane >> 2) < PH * PW && (unsigned)(ox0_t + x_lx[k]) < (unsigned)Wi;
            unsigned v = in ? (unsigned)(xorg_t + x_voff[k]) : 0x80000000u;
            if (pl && in) v += 64;
            lds_dma16(__builtin_amdgcn_readfirstlane(a_x + pl * XPL + (wv + 4 * k) * 1024), v, rs_xt);
        } else {
            const int k = (q - 2 * XK) >> 1, pl = (q - 2 * XK) & 1;
            const int pix = (wv + 4 * k) * 16 + (lane >> 2);
            const int gy_ = by_t + pix / TW, gx_ = bx_t + pix % TW;
            const bool in = gy_ < Hb && gx_ < Wb;
            unsigned v = in ? (unsigned)(((gy_ * Wb + gx_) * OC + oc0_n) * 2 + (lane & 3) * 16) : 0x80000000u;
            if (pl && in) v += 64;
            lds_dma16(__builtin_amdgcn_readfirstlane(a_g + pl * GPL + (wv + 4 * k) * 1024), v, rs_gt);
        }
    };

    // ---- MFMA side (as conv_wgrad_bf16_2x2_kernel)
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float accb = 0.f;
    bool do_bias = false;
    constexpr int XR = S2 ? 5 : 3;
    constexpr int NG = NP / 16;
    uint2 fb[2][2], fx[2][3][XR];
    // fragment addresses (GS_WG_GROUP_STEP).  R128: row r of the tile at r * 128, this wave's 32-channel half at ((half ^ bit 1 of r) << 6).
    // Bit 1 of the row a lane reads is (a compile-time bit of the group / kernel row / column parity) ^ (a bit of t_row): everything that
    // depends on the lane is folded into two offsets per side, the rest into the instruction's immediate offset.
    //   stride 1: r = (ty + ky) PW + txc + 8 hi + t_row (+ 4 k),  PW = 34:  bit 1 = ((ty + ky) & 1) ^ (t_row >> 1)
    //   stride 2: r = (2 ty + ky) PW + 2 txc + 16 hi + 2 t_row + c (+ 8 k),  PW = 65:  bit 1 = (((2 ty + ky + c) >> 1) & 1) ^ (t_row & 1)
    //   gradient: r = ty TW + txc + 8 hi + t_row (+ 4 i):  bit 1 = t_row >> 1
    static_assert(!R128 || (S2 ? PW % 4 == 1 : PW % 4 == 2), "the swizzle algebra below assumes PW = 34 (stride 1) / 65 (stride 2)");
    const int xl_base = R128 ? (t_row * S + 8 * S * hi) * 128 + t_col : 0;
    const int xl_sw = S2 ? (t_row & 1) : (t_row >> 1);
    const int xlane[2] = {xl_base + (((it ^ xl_sw) & 1) << 6), xl_base + (((it ^ xl_sw ^ 1) & 1) << 6)};
    const int glane = R128 ? (t_row + 8 * hi) * 128 + t_col + (((ot ^ (t_row >> 1)) & 1) << 6) : 0;
    auto x_ptr_ = [&](const unsigned char* xt, int ty, int ky, int txc, int c) __attribute__((always_inline)) {
        if (R128) {
            const int a = S2 ? ((2 * ty + ky + c) >> 1) & 1 : (ty + ky) & 1;
            return xt + ((ty * S + ky) * PW + txc * S + c) * 128 + xlane[a];
        }
        return xt + it * XPL + (((ty * S + ky) * PW + (txc + 8 * hi) * S) + t_row * S + c) * 64 + t_col;
    };
    auto g_ptr_ = [&](const unsigned char* gt, int ty, int txc) __attribute__((always_inline)) {
        if (R128) return gt + (ty * TW + txc) * 128 + glane;
        return gt + ot * GPL + (ty * TW + txc + 8 * hi + t_row) * 64 + t_col;
    };
    auto load_group = [&](int gi, int fbuf, const unsigned char* xt, const unsigned char* gt) __attribute__((always_inline)) {
        const int ty = (gi * 16) / TW, txc = (gi * 16) % TW;
        const unsigned char* gp = g_ptr_(gt, ty, txc);
        fb[fbuf][0] = lds_tr16(gp);
        fb[fbuf][1] = lds_tr16(gp + 4 * GROWB);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const unsigned char* xp = x_ptr_(xt, ty, ky, txc, 0);
            if (!S2) {
                fx[fbuf][ky][0] = lds_tr16(xp); fx[fbuf][ky][1] = lds_tr16(xp + 4 * XROWB); fx[fbuf][ky][2] = lds_tr16(xp + 8 * XROWB);
            } else {
                const unsigned char* xo = x_ptr_(xt, ty, ky, txc, 1);
                fx[fbuf][ky][0] = lds_tr16(xp); fx[fbuf][ky][1] = lds_tr16(xp + 8 * XROWB); fx[fbuf][ky][2] = lds_tr16(xp + 16 * XROWB);
                fx[fbuf][ky][3] = lds_tr16(xo); fx[fbuf][ky][4] = lds_tr16(xo + 8 * XROWB);
            }
        }
    };

    // ---- first unit of the block
    while (j_n + 1 < g.njobs && U0 >= g.job[j_n + 1].unit_base) ++j_n;
    load_job(j_n);
    ct_n = (U0 - g.job[j_n].unit_base) / ntiles;
    tile_n = (U0 - g.job[j_n].unit_base) - ct_n * ntiles;
    tx_n = tile_n % tiles_x;
    ty_n = (tile_n / tiles_x) % tiles_y;
    img_n = tile_n / (tiles_x * tiles_y);
    set_ct(j_n, ct_n);
    tile_setup(j_n);
    if (issuer) {
#pragma unroll
        for (int q = 0; q < NPIECE; ++q) issue_piece(q, 0);
    }

    // DMA pieces of the next unit per pixel group of this one.  GS_SK_FRONT_S1 / _S2 = k > 0: all of them within the first k groups, so that
    // the last piece issued has the remaining groups' MFMAs between it and the wait at the top of the next unit (a piece issued in the
    // last group meets that wait ~300 cycles later and the unit pays its whole L2 / HBM latency: SQ_WAIT_ANY 45 % of the stride-2 kernel's
    // wave cycles, profiles/r04_a_wgrad_group_sq_pmc.txt)
    constexpr int FRONT = S2 ? GS_SK_FRONT_S2 : GS_SK_FRONT_S1;
    constexpr int PPG = FRONT > 0 ? (NPIECE + FRONT - 1) / FRONT : (NPIECE + NG - 1) / NG;
    int buf = 0;
    for (int u = U0; u < U1; ++u) {
        const bool more = u + 1 < U1;
        if (issuer) wait_vmcnt(0);
        block_barrier();
        // the unit being multiplied: what the DMA side was set to when it was issued
        do_bias = bias_next;
        const int run_c = run_n;
        const bool bias_wave_c = bias_wave_n;
        bool leave = !more;   // does the block's range leave the run after this unit?
        if (more) {
            if (++tx_n == tiles_x) {
                tx_n = 0;
                if (++ty_n == tiles_y) { ty_n = 0; ++img_n; }
            }
            if (++tile_n == ntiles) {
                tile_n = 0;
                tx_n = ty_n = img_n = 0;
                leave = true;
                if (++ct_n == nct) { ct_n = 0; ++j_n; load_job(j_n); }
                set_ct(j_n, ct_n);
            }
            tile_setup(j_n);
        }
        const unsigned char* const xpl = lds_raw + buf * BUF;          // the staged patch / gradient tile (x_ptr_ / g_ptr_ pick this wave's half)
        const unsigned char* const gpl = lds_raw + buf * BUF + XT;
        if (SPEC) {
            if (loader) {
                if (more) {
#pragma unroll
                    for (int q = 0; q < NPIECE; ++q) issue_piece(q, buf ^ 1);
                }
            } else {
                load_group(0, 0, xpl, gpl);
                __builtin_amdgcn_sched_barrier(0);
                static_for<NG>([&](auto gc) __attribute__((always_inline)) { GS_WG_GROUP_STEP(decltype(gc)::value, false, xpl, gpl, buf ^ 1); });
            }
        } else {
            load_group(0, 0, xpl, gpl);
            __builtin_amdgcn_sched_barrier(0);
            static_for<NG>([&](auto gc) __attribute__((always_inline)) { GS_WG_GROUP_STEP(decltype(gc)::value, more, xpl, gpl, buf ^ 1); });
        }
        block_barrier();
        buf ^= 1;
        if (leave && computer) {   // partial (block + run): [tap][ic 64][oc 64] + 64 bias sums; lane = (oc j = l31, ic i = (r&3) + 8(r>>2) + 4hi)
            float* const dst0 = part + (long)(blockIdx.x + run_c) * GS_SK_PSTRIDE;
            if (bias_wave_c) {
                const float tot = swap32_sum(accb);
                if (hi == 0) dst0[9 * 4096 + ot * 32 + l31] = tot;
            }
            accb = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float* dst = dst0 + (t * 64 + it * 32) * 64 + ot * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    dst[((r & 3) + 8 * (r >> 2) + 4 * hi) * 64] = acc[t][r];
                    acc[t][r] = 0.f;
                }
            }
        }
    }
}

// folds the partials of every run of a group into the gradients: grid (145, runs); block = 64 consecutive quads of the partial
// (1 KiB per partial: whole DRAM bursts) x 4 partial lanes, a thread keeps four partials in flight; fixed order -> deterministic
__global__ __launch_bounds__(256) void wgrad_sk_reduce_kernel(const SkGroup g, const float* __restrict__ part) {
    constexpr int L = 4, EPB = 64;
    __shared__ float4 red[256];
    const int r = blockIdx.y;
    int j = 0;
    while (j + 1 < g.njobs && r >= g.job[j + 1].run_base) ++j;
    const SkJob& q = g.job[j];
    const int ct = r - q.run_base;
    const int ic0 = (ct % q.n_ict) * 64, oc0 = (ct / q.n_ict) * 64;
    const long s0 = (long)q.unit_base + (long)ct * q.ntiles;
    const int b0 = sk_block_of(s0, g.nblocks, g.total_units), b1 = sk_block_of(s0 + q.ntiles - 1, g.nblocks, g.total_units);
    const int e = blockIdx.x * EPB + (threadIdx.x % EPB);   // quad index inside the partial
    const int sl = threadIdx.x / EPB;
    const bool live = e < GS_SK_PSTRIDE / 4;
    const bool is_bias = e >= 9 * 1024;
    if (blockIdx.x * EPB >= 9 * 1024 && !(q.gb && ic0 == 0)) return;   // (the bias quads are the last, whole block)
    const float* const base = part + (long)r * GS_SK_PSTRIDE + (long)e * 4;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    if (live) {
        int b = b0 + sl;
        for (; b + 3 * L <= b1; b += 4 * L) {
            const float4 v0 = *reinterpret_cast<const float4*>(base + (long)b * GS_SK_PSTRIDE);
            const float4 v1 = *reinterpret_cast<const float4*>(base + (long)(b + L) * GS_SK_PSTRIDE);
            const float4 v2 = *reinterpret_cast<const float4*>(base + (long)(b + 2 * L) * GS_SK_PSTRIDE);
            const float4 v3 = *reinterpret_cast<const float4*>(base + (long)(b + 3 * L) * GS_SK_PSTRIDE);
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
            a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
            a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
            a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
        }
        for (; b <= b1; b += L) {
            const float4 v0 = *reinterpret_cast<const float4*>(base + (long)b * GS_SK_PSTRIDE);
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        }
    }
    a0.x += a2.x; a0.y += a2.y; a0.z += a2.z; a0.w += a2.w;
    a1.x += a3.x; a1.y += a3.y; a1.z += a3.z; a1.w += a3.w;
    red[threadIdx.x] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
    __syncthreads();
    if (sl != 0 || !live) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const float4 v = red[threadIdx.x + k * EPB];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (is_bias) {
        if (!(q.gb && ic0 == 0)) return;
        float4* o = reinterpret_cast<float4*>(q.gb + oc0 + (e - 9 * 1024) * 4);
        const float4 old = q.accumulate ? *o : make_float4(0.f, 0.f, 0.f, 0.f);
        *o = make_float4(old.x + s.x, old.y + s.y, old.z + s.z, old.w + s.w);
        return;
    }
    const int t = e >> 10, i = (e >> 4) & 63, j4 = (e & 15) * 4;
    const float al = q.alpha;
    if (!q.transpose) {
        float4* o = reinterpret_cast<float4*>(q.gw + ((long)t * q.ICld + ic0 + i) * q.OC + oc0 + j4);
        const float4 old = q.accumulate ? *o : make_float4(0.f, 0.f, 0.f, 0.f);
        *o = make_float4(old.x + s.x * al, old.y + s.y * al, old.z + s.z * al, old.w + s.w * al);
    } else {
        const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float* o = q.gw + ((long)t * q.OC + oc0 + j4 + c) * q.IC + ic0 + i;
            *o = q.accumulate ? *o + v[c] * al : v[c] * al;
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
// weight-gradient launches sized for fewer CUs than the chip has (0: all): what runs beside a latency-bound chain on a forked branch of
// the run's hipGraph leaves that chain's few blocks somewhere to land (gs_wgrad_cu_cap; models.GANSynth._early_flush)
static int g_wgrad_cu_cap = 0;
static int wgrad_cus() {
    const int n = num_cus();
    return g_wgrad_cu_cap > 0 && g_wgrad_cu_cap < n ? g_wgrad_cu_cap : n;
}

#ifndef GS_MAX_BLOCKS_PER_CU
#define GS_MAX_BLOCKS_PER_CU 2
#endif
#ifndef GS_SMALL_D
#define GS_SMALL_D 2   // stages in flight for the few-block ("small") configurations
#endif
template <typename T, int MODE, int A, int B, int TW, int TG, bool RESIDENT = false, int D = 2, int NORM = 0, int RB = 64, bool SPEC = false, bool BITS = false>
static int launch_igemm(ConvP p, hipStream_t st) {
    if constexpr (!BITS && sizeof(T) == 2 && NORM == 0) {
        if (p.mask_bits || p.bits_out) return launch_igemm<T, MODE, A, B, TW, TG, RESIDENT, D, NORM, RB, SPEC, true>(p, st);
    }
    constexpr int NP = 128 * B;
    constexpr int TH = NP / TW;
    constexpr int PH = patch_dim<MODE>(TH), PW = patch_dim<MODE>(TW);
    constexpr int OCT = 32 * A;
    constexpr int BK = RB / (int)sizeof(T);
    constexpr int NTG = 9 / TG;
    constexpr int PBUF = ((PH * PW * (RB / 16) + 255) / 256) * 4096;
    constexpr int WBUF = ((TG * OCT * RB + 4095) / 4096) * 4096;
    constexpr int NPB = 1 + (D + NTG - 1) / NTG, NWB = D + 1;
    p.tiles_x = cdiv(p.Wb, TW);
    p.tiles_y = cdiv(p.Hb, TH);
    p.nsp = p.N * p.tiles_x * p.tiles_y;
    p.noct = cdiv(p.OC, OCT);
    p.nch = p.IC / BK;
    {   // reciprocals for the kernel's item decoding (see ConvP): exact while items < 2^21 and divisors < 2^11
        auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); };
        if (p.noct >= 2048 || p.tiles_x >= 2048 || p.tiles_y >= 2048)
            return fail(GS_ERR_UNSUPPORTED, "conv igemm: %d channel tiles / %d x %d spatial tiles exceed the item decoder's range", p.noct, p.tiles_x, p.tiles_y);
        static const long max_items = getenv("GS_IGEMM_MAX_ITEMS") ? atol(getenv("GS_IGEMM_MAX_ITEMS")) : (1L << 21);   // (tests lower it)
        if ((long)p.nsp * p.noct >= max_items) {
            // More work items than the reciprocal decoder is exact for (large-batch evaluation at full resolution): images are independent,
            // so the batch runs as several launches of as many images as fit the range -- same kernel, same results.
            const long per_image = (long)p.tiles_x * p.tiles_y * p.noct;
            const int chunk = (int)((max_items - 1) / per_image);
            if (chunk < 1) return fail(GS_ERR_UNSUPPORTED, "conv igemm: %ld work items per image exceed the item decoder's range", per_image);
            const size_t in_img = (size_t)p.Hi * p.Wi * p.IC * sizeof(T);
            const size_t out_img = (size_t)p.Hb * p.Wb * (MODE == MODE_T2 ? 4 : 1) * p.OC * sizeof(T);
            auto adv = [](const void* q, size_t b) -> const void* { return q ? (const char*)q + b : nullptr; };
            for (int n0 = 0; n0 < p.N; n0 += chunk) {
                ConvP q = p;
                q.N = p.N - n0 < chunk ? p.N - n0 : chunk;
                q.x = adv(p.x, n0 * in_img);
                q.y = const_cast<void*>(adv(p.y, n0 * out_img));
                q.y2 = const_cast<void*>(adv(p.y2, n0 * out_img));
                q.mask = adv(p.mask, n0 * out_img);
                q.mask_bits = reinterpret_cast<const unsigned char*>(adv(p.mask_bits, n0 * out_img / (8 * sizeof(T))));
                q.bits_out = reinterpret_cast<unsigned char*>(const_cast<void*>(adv(p.bits_out, n0 * out_img / (8 * sizeof(T)))));
                q.addend = adv(p.addend, n0 * out_img);
                int pending = 0;
                if (p.norm_pending) q.norm_pending = &pending;
                if (int e = launch_igemm<T, MODE, A, B, TW, TG, RESIDENT, D, NORM, RB, SPEC, BITS>(q, st)) return e;
                if (p.norm_pending && pending) *p.norm_pending = pending;
            }
            return 0;
        }
        p.m_noct = magic(p.noct); p.m_tx = magic(p.tiles_x); p.m_ty = magic(p.tiles_y);
    }
    const int wbufs = RESIDENT ? p.nch : NWB;
    const size_t lds = (size_t)NPB * PBUF + (size_t)wbufs * WBUF + (size_t)((p.OC + 3) / 4) * 16 + 0;
    if (p.OC % OCT != 0) return fail(GS_ERR_UNSUPPORTED, "conv igemm: %d output channels with %d-wide tiles", p.OC, OCT);
    if ((size_t)p.Hi * p.Wi * p.IC * sizeof(T) >= (1ull << 31)) return fail(GS_ERR_UNSUPPORTED, "conv igemm: one image exceeds 2 GiB");
    if (lds > 160 * 1024) return fail(GS_ERR_UNSUPPORTED, "conv igemm: %zu bytes of LDS needed", lds);
    if (NORM && p.OC != OCT) return fail(GS_ERR_UNSUPPORTED, "conv igemm: fused pixel norm needs the whole channel range in one tile");
    if ((NORM == 2 || NORM == 3) && p.OC != OCT) return fail(GS_ERR_UNSUPPORTED, "conv igemm: fused pixel-norm backward needs the whole channel range in one tile");
    if (!((NORM == 2 && p.normbwd == 1) || (NORM == 3 && p.normbwd == 2)) && p.normbwd) {
        // no fused form here: plain conv, the caller runs the norm's backward kernel(s) as their own pass
        if (p.norm_pending) *p.norm_pending = 1 + p.normbwd;   // 2: first-order form, 3: second-order form
        p.normbwd = 0;
        p.mask = nullptr;
        p.addend = nullptr;
        p.y2 = nullptr;
    }
    if (NORM != 1 && NORM != 3 && p.y2) {   // the norm stays a separate pass: this launch leaves the activation where that pass will read it
        if (!p.y) p.y = p.y2;
        p.y2 = nullptr;
        if (p.norm_pending) *p.norm_pending = 1;
    }
    if (p.IC % BK != 0) return fail(GS_ERR_UNSUPPORTED, "conv igemm: %d input channels with %d-channel chunks", p.IC, BK);
    auto kern = conv_igemm_kernel<T, MODE, A, B, TW, TG, RESIDENT, D, NORM, RB, SPEC, BITS>;
    static size_t max_set = 0;  // per template instantiation
    if (lds > max_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return fail(GS_ERR_HIP, "conv igemm: cannot reserve %zu bytes of dynamic LDS", lds);
        max_set = lds;
    }
    // resident blocks per CU: LDS-limited, and at most 2 (accumulator-heavy kernels hold 1-2 waves per SIMD)
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > GS_MAX_BLOCKS_PER_CU) per_cu = GS_MAX_BLOCKS_PER_CU;
    if (per_cu < 1 || SPEC) per_cu = 1;   // (8 waves of up to 256 VGPRs: one block per CU)
    const long total = (long)p.nsp * p.noct;
    long grid = (long)per_cu * num_cus();
    if (grid > total) grid = total;
    if (grid >= 8) grid &= ~7L;
    const double flops = 2.0 * 9.0 * (double)p.N * p.Hb * p.Wb * p.IC * p.OC;
    const double out_px = (double)p.N * p.Hb * p.Wb * (MODE == MODE_T2 ? 4 : 1);
    // algorithmic bytes: input + output + weights, + the activation mask a masked launch reads (1/16 of it as sign words), + the sign words a
    // forward launch writes, + the second output of a fused norm
    const double bytes = ((double)p.N * p.Hi * p.Wi * p.IC + out_px * p.OC * (1.0 + (p.mask ? (p.mask_bits ? 1.0 / 16.0 : 1.0) : 0.0) + (p.bits_out ? 1.0 / 16.0 : 0.0) + ((NORM == 1 && p.y) ? 1.0 : 0.0) + ((NORM == 2 && p.addend) ? 1.0 : 0.0) + (NORM == 3 ? 2.0 : 0.0)) + 9.0 * p.IC * p.OC) * sizeof(T);
    const int reps = prof_reps();   // (1 unless profiling in burst mode: the kernel is a pure function of its inputs)
    ProfScope ps(st, flops, bytes, MODE, p.N, p.Hb, p.Wb, p.IC, p.OC, p.mask ? 1 : 0, NORM ? 1 : 0, reps);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(SPEC ? 512 : 256), lds, st, p);
    return 0;
}

// choose the tile configuration from (OC, Wb, problem size)
// Which configuration families run wave-specialised (bit 0 few-block, 1 S1, 2 S2 / T2; GS_SPEC overrides for measurements).  Measured
// per iteration of configs[1] (scripts/run_spec.sh): few-block layers 1086 -> 1068 us; the larger S1 / S2 / T2 tiles LOSE 60-190 us
// (their stages are bound by the fragment reads and the barrier, not by the DMA issue) -- so only bit 0 is on.
static int spec_mask() {
    static const int m = getenv("GS_SPEC") ? atoi(getenv("GS_SPEC")) : 1;
    return m;
}

template <typename T, int MODE>
static int dispatch_igemm(ConvP p, hipStream_t st) {
    constexpr int BK = 64 / (int)sizeof(T);
    const int OC = p.OC, Wb = p.Wb;
    const int nch = p.IC / BK;
    const bool resident_ok = OC == 32 && nch <= 2;
    const bool resident64_ok = OC == 64 && nch <= 2;   // 9 taps x 64 x (<= 64 channels) <= 72 KiB stay in LDS
    const int a2 = OC % 64 == 0;
    // blocks the layer gives with 128*B-pixel x 64-channel tiles (TW = 32)
    auto items64 = [&](int B_) { return (long)p.N * cdiv(p.Hb, 4 * B_) * cdiv(Wb, 32) * (OC / 64); };
#ifdef GS_FORCE_CFG   // probes only: -DGS_FORCE_MODE=0 -DGS_FORCE_CFG=2,2,32,3,false,1
    if constexpr (MODE == GS_FORCE_MODE && sizeof(T) == 2) return launch_igemm<T, MODE, GS_FORCE_CFG>(p, st);
#endif
    // Measured on the layers of the fully grown networks (scripts/probe/run_variants.sh): two resident blocks per CU (<= 80 KiB
    // of LDS each, i.e. a one-stage-deep ring) beat one block with a deeper ring or a larger tile wherever the layer has at
    // least two blocks per CU to give -- the epilogue and DMA waits of one block run under the MFMAs of the other.
    const long cus = num_cus();
    static const bool no_small = getenv("GS_NO_SMALL_TILES") != nullptr;   // measurement knob
    const bool small = !no_small && items64(1) <= cus;   // no more blocks than CUs: a serial chain of stages per block
    // Fused pixel norm (p.y2): only the configurations whose tile owns every channel of a pixel (OC = 32 A) and that the generator's
    // 32- / 64-channel blocks actually use; everything else runs the plain kernel and the caller's separate norm pass (p.y2 = NULL
    // on return tells it so -- see run_igemm_t).
    const bool norm = p.y2 != nullptr && p.normbwd == 0;
    const bool nbwd = p.normbwd == 1;   // (the three data-gradient shapes of the generator's 32- / 64-channel blocks below; anything else falls back)
    const bool nbb = p.normbwd == 2;    // (second-order form: the forward-role shapes that have the NORM == 1 epilogue)
    if constexpr (MODE == MODE_T2) {
        if (resident_ok && Wb >= 64) {
            if (norm) return launch_igemm<T, MODE, 1, 1, 64, 9, true, 1, true>(p, st);
            if (nbb) return launch_igemm<T, MODE, 1, 1, 64, 9, true, 1, 3>(p, st);
            return launch_igemm<T, MODE, 1, 1, 64, 9, true, 1>(p, st);
        }
        // few blocks, each a serial chain of stages: 32-channel tiles double the number of busy CUs, 9-tap stages cut the
        // barriers and DMA round trips of the chain to a third
        if (small) {
            // 128-byte operand rows (whole cache lines per DMA row, half the stages): the few-block layers are bound by the L2 -> LDS
            // rate of a CU, not by its MFMAs
            if constexpr (sizeof(T) == 2) {
                static const bool rb128 = getenv("GS_NO_RB128") == nullptr;
                if (rb128 && p.IC % 64 == 0) {
                    if (spec_mask() & 1) {
                        if (Wb >= 32) return launch_igemm<T, MODE, 1, 1, 32, 9, false, 1, false, 128, true>(p, st);
                        return launch_igemm<T, MODE, 1, 1, 16, 9, false, 1, false, 128, true>(p, st);
                    }
                    if (Wb >= 32) return launch_igemm<T, MODE, 1, 1, 32, 9, false, 1, false, 128>(p, st);
                    return launch_igemm<T, MODE, 1, 1, 16, 9, false, 1, false, 128>(p, st);
                }
            }
            if (Wb >= 32) return launch_igemm<T, MODE, 1, 1, 32, 9, false, GS_SMALL_D>(p, st);
            return launch_igemm<T, MODE, 1, 1, 16, 9, false, GS_SMALL_D>(p, st);
        }
        if (!a2) { if (Wb >= 32) return launch_igemm<T, MODE, 1, 1, 32, 3>(p, st); return launch_igemm<T, MODE, 1, 1, 16, 3>(p, st); }
        if (Wb >= 32) {
            if (norm && OC == 64) return launch_igemm<T, MODE, 2, 1, 32, 3, false, 2, true>(p, st);
            if (nbb && OC == 64) return launch_igemm<T, MODE, 2, 1, 32, 3, false, 2, 3>(p, st);
            if (spec_mask() & 4) return launch_igemm<T, MODE, 2, 1, 32, 3, false, 2, false, 64, true>(p, st);
            return launch_igemm<T, MODE, 2, 1, 32, 3>(p, st);
        }
        return launch_igemm<T, MODE, 2, 1, 16, 3>(p, st);
    } else if constexpr (MODE == MODE_S2) {
        if (small) { if (Wb >= 32) return launch_igemm<T, MODE, 1, 1, 32, 9, false, 1>(p, st); return launch_igemm<T, MODE, 1, 1, 16, 9, false, 1>(p, st); }
        if (!a2) return launch_igemm<T, MODE, 1, 1, 32, 3>(p, st);
        // stride 2: the patch is 4.6x the output tile, so a 128-channel tile (the patch staged once for all of them) is worth
        // more than a second pixel tile as long as every CU still gets a block
        if (OC == 64 && nch == 1 && Wb >= 32 && items64(1) >= cus) {   // 36 KiB of weights: resident
            if (nbwd) return launch_igemm<T, MODE, 2, 1, 32, 9, true, 1, 2>(p, st);
            return launch_igemm<T, MODE, 2, 1, 32, 9, true, 1>(p, st);
        }
        if (OC % 128 == 0 && Wb >= 32 && (long)p.N * cdiv(p.Hb, 4) * cdiv(Wb, 32) * (OC / 128) >= cus) {
            if (spec_mask() & 4) return launch_igemm<T, MODE, 4, 1, 32, 3, false, 2, false, 64, true>(p, st);
            return launch_igemm<T, MODE, 4, 1, 32, 3>(p, st);
        }
        if (Wb >= 32) {
            if (spec_mask() & 4) return launch_igemm<T, MODE, 2, 1, 32, 3, false, 2, false, 64, true>(p, st);
            return launch_igemm<T, MODE, 2, 1, 32, 3>(p, st);
        }
        return launch_igemm<T, MODE, 2, 1, 16, 3>(p, st);
    } else {
        if (resident_ok && Wb >= 64) {
            if (norm) return launch_igemm<T, MODE, 1, 2, 64, 9, true, 1, true>(p, st);
            if (nbwd) return launch_igemm<T, MODE, 1, 2, 64, 9, true, 1, 2>(p, st);
            if (nbb) return launch_igemm<T, MODE, 1, 2, 64, 9, true, 1, 3>(p, st);
            return launch_igemm<T, MODE, 1, 2, 64, 9, true, 1>(p, st);
        }
        if (small) {
            // 128-byte operand rows (whole cache lines per DMA row, half the stages): the few-block layers are bound by the L2 -> LDS
            // rate of a CU, not by its MFMAs
            if constexpr (sizeof(T) == 2) {
                static const bool rb128 = getenv("GS_NO_RB128") == nullptr;
                if (rb128 && p.IC % 64 == 0) {
                    if (spec_mask() & 1) {
                        if (Wb >= 32) return launch_igemm<T, MODE, 1, 1, 32, 9, false, 1, false, 128, true>(p, st);
                        return launch_igemm<T, MODE, 1, 1, 16, 9, false, 1, false, 128, true>(p, st);
                    }
                    if (Wb >= 32) return launch_igemm<T, MODE, 1, 1, 32, 9, false, 1, false, 128>(p, st);
                    return launch_igemm<T, MODE, 1, 1, 16, 9, false, 1, false, 128>(p, st);
                }
            }
            if (Wb >= 32) return launch_igemm<T, MODE, 1, 1, 32, 9, false, GS_SMALL_D>(p, st);
            return launch_igemm<T, MODE, 1, 1, 16, 9, false, GS_SMALL_D>(p, st);
        }
        if (!a2) return launch_igemm<T, MODE, 1, 1, 32, 3>(p, st);
        if (Wb >= 32 && items64(2) >= 2 * cus) {
            if (norm && OC == 64) return launch_igemm<T, MODE, 2, 2, 32, 3, false, 1, true>(p, st);
            if (nbwd && OC == 64) return launch_igemm<T, MODE, 2, 2, 32, 3, false, 1, 2>(p, st);
            if (nbb && OC == 64) return launch_igemm<T, MODE, 2, 2, 32, 3, false, 1, 3>(p, st);
            if (spec_mask() & 2) return launch_igemm<T, MODE, 2, 2, 32, 3, false, 2, false, 64, true>(p, st);
            return launch_igemm<T, MODE, 2, 2, 32, 3, false, 1>(p, st);
        }
        if (resident64_ok && Wb >= 32 && items64(2) >= cus / 2) return launch_igemm<T, MODE, 2, 2, 32, 9, true>(p, st);
        // every block re-streams its 64 x IC x 9 weight slab from L2: the more pixels a block owns the smaller that
        // stream is per MFMA -- take the largest pixel tile that still gives every CU a block
        if (Wb >= 32 && items64(2) >= cus) {
            if (spec_mask() & 2) return launch_igemm<T, MODE, 2, 2, 32, 3, false, 2, false, 64, true>(p, st);
            // one block per CU: 128-byte rows halve the stages of the chain (22.0 -> 20.1 us on 256 -> 256 @ 16x128 x8, scripts/run_abl.sh)
            if constexpr (sizeof(T) == 2) {
                static const bool rb128 = getenv("GS_NO_RB128") == nullptr;
                if (rb128 && p.IC % 64 == 0) return launch_igemm<T, MODE, 2, 2, 32, 3, false, 1, false, 128>(p, st);
            }
            return launch_igemm<T, MODE, 2, 2, 32, 3>(p, st);
        }
        if (Wb >= 32) {
            if (spec_mask() & 2) return launch_igemm<T, MODE, 2, 1, 32, 3, false, 2, false, 64, true>(p, st);
            return launch_igemm<T, MODE, 2, 1, 32, 3>(p, st);
        }
        return launch_igemm<T, MODE, 2, 1, 16, 3>(p, st);
    }
}

// does dispatch_igemm have the NORM == 2 (pixel-norm backward) epilogue for this data-gradient shape?  (mirrors its three branches)
// form 1: NORM == 2 (first order, data-gradient roles); form 2: NORM == 3 (second order, forward roles)
bool igemm_normbwd_fused(int mode, int N, int Hb, int Wb, int IC, int OC, int dtype, int form) {
    const int bk = dtype == GS_F32 ? 16 : 32;
    if (IC % bk != 0 || OC % 32 != 0) return false;
    const int nch = IC / bk;
    const long cus = num_cus();
    auto items64 = [&](int B_) { return (long)N * cdiv(Hb, 4 * B_) * cdiv(Wb, 32) * (OC / 64); };
    if (form == 2) {   // mirrors the NORM == 1 branches of dispatch_igemm
        const bool small = items64(1) <= cus;
        if (mode == MODE_T2) {
            if (OC == 32 && nch <= 2 && Wb >= 64) return true;
            return !small && OC == 64 && Wb >= 32;
        }
        if (mode == MODE_S1) {
            if (OC == 32 && nch <= 2 && Wb >= 64) return true;
            return !small && OC == 64 && Wb >= 32 && items64(2) >= 2 * cus;
        }
        return false;
    }
    if (mode == MODE_S1) {
        if (OC == 32 && nch <= 2 && Wb >= 64) return true;
        const bool small = items64(1) <= cus;
        return !small && OC == 64 && Wb >= 32 && items64(2) >= 2 * cus;
    }
    if (mode == MODE_S2) {
        const bool small = items64(1) <= cus;
        return !small && OC == 64 && nch == 1 && Wb >= 32 && items64(1) >= cus;
    }
    return false;
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
                       int w_prepared, void* ws, size_t ws_bytes, hipStream_t st, const void* mask, int mask_act, void* y2, float pn_eps,
                       const void* addend, int normbwd) {
    const size_t need = (size_t)9 * w_ci * w_co * sizeof(T);
    if (ws_bytes < need) return fail(GS_ERR_WORKSPACE, "conv igemm: workspace %zu < %zu", ws_bytes, need);
    T* wp = reinterpret_cast<T*>(ws);
    const long total = 9L * w_ci * w_co;
    if (!w_prepared)
        hipLaunchKernelGGL((weight_prep_kernel<T>), dim3(cdiv(total, 256)), dim3(256), 0, st, w_hwio, wp, 9, w_ci, w_co, variant);
    ConvP p;
    memset(&p, 0, sizeof(p));
    const size_t out_numel = (size_t)N * Hb * Wb * (mode == MODE_T2 ? 4 : 1) * OCk;
    const bool plain = y2 == nullptr && !normbwd;   // (the fused-norm epilogues need z itself and write no bits)
    const bool write_bits = (act & GS_ACT_WRITE_BITS) != 0;
    act &= ~GS_ACT_WRITE_BITS;
    if (mask_act == GS_ACT_LRELU_BITS) {   // mask = an activation output with its sign bits behind it
        mask_act = GS_ACT_LRELU;
        if (sizeof(T) == 2 && plain && mask && OCk % 32 == 0) p.mask_bits = reinterpret_cast<const unsigned char*>(mask) + out_numel * sizeof(T);
    }
    bool bits_pending = write_bits;
    if (write_bits && sizeof(T) == 2 && plain && y && act == GS_ACT_LRELU && OCk % 32 == 0 && (!mask || p.mask_bits)) {   // (the BITS kernel reads no mask VALUES)
        p.bits_out = reinterpret_cast<unsigned char*>(y) + out_numel * sizeof(T);
        bits_pending = false;
    }
    p.x = x; p.wp = wp; p.y = y; p.bias = bias; p.act = act; p.mask = mask; p.mask_act = mask_act;
    p.N = N; p.Hi = Hi; p.Wi = Wi; p.IC = ICk; p.OC = OCk; p.Hb = Hb; p.Wb = Wb; p.alpha = alpha;
    int pending = 0;
    p.y2 = y2; p.pn_eps = pn_eps; p.norm_pending = &pending;
    p.addend = normbwd ? addend : nullptr; p.normbwd = normbwd;
    int rc;
    if (mode == MODE_S1) rc = dispatch_igemm<T, MODE_S1>(p, st);
    else if (mode == MODE_S2) rc = dispatch_igemm<T, MODE_S2>(p, st);
    else rc = dispatch_igemm<T, MODE_T2>(p, st);
    if (rc) return rc;
    GS_CHECK_LAUNCH();
    if (pending == 3) {   // second-order form without an epilogue: y holds t; both gradients from the norm's own kernel (y in place, y2)
        const long px = (long)N * Hb * Wb * (mode == MODE_T2 ? 4 : 1);
        return gs_pixel_norm_bwd_bwd_fused(y, addend, mask, y2, y, px, OCk, pn_eps, mask_act, sizeof(T) == 4 ? GS_F32 : GS_BF16, st);
    }
    if (pending == 2) {   // no fused pixel-norm backward for this shape: y holds the plain data gradient g; the norm's backward runs in place
        const long px = (long)N * Hb * Wb * (mode == MODE_T2 ? 4 : 1);
        return gs_pixel_norm_bwd_fused(y, mask, addend, y, px, OCk, pn_eps, GS_ACT_NONE, mask_act, sizeof(T) == 4 ? GS_F32 : GS_BF16, st);
    }
    if (pending) {   // y2 = pixel_norm(activation), the activation sitting in y (or in y2 itself when the caller keeps no copy)
        const long px = (long)N * Hb * Wb * (mode == MODE_T2 ? 4 : 1);
        if (int e = gs_pixel_norm_fwd(y ? y : y2, y2, px, OCk, pn_eps, sizeof(T) == 4 ? GS_F32 : GS_BF16, st)) return e;
    }
    if (bits_pending && y)   // (asked for the sign bits where this epilogue does not write them: the packing pass)
        return gs_pack_act_bits(y, (int64_t)(out_numel / OCk), OCk, sizeof(T) == 4 ? GS_F32 : GS_BF16, st);
    return 0;
}

int run_igemm(int mode, int variant, const void* x, const float* w_hwio, void* y, int N, int Hi, int Wi, int ICk,
              int OCk, int w_ci, int w_co, int Hb, int Wb, float alpha, const float* bias, int act, int dtype, int w_prepared,
              void* ws, size_t ws_bytes, hipStream_t st, const void* mask, int mask_act, void* y2, float pn_eps, const void* addend, int normbwd) {
    GS_DISPATCH_DTYPE(dtype, return (run_igemm_t<T>(mode, variant, x, w_hwio, y, N, Hi, Wi, ICk, OCk, w_ci, w_co, Hb,
                                                    Wb, alpha, bias, act, w_prepared, ws, ws_bytes, st, mask, mask_act, y2, pn_eps, addend, normbwd)));
}

static int patch_dim_rt(int mode, int t) { return mode == MODE_S1 ? t + 2 : (mode == MODE_S2 ? 2 * t + 1 : t + 1); }
// ---- weight gradient (fp32 MFMA path)
static bool wgrad_2x2(int mode, int dtype, int IC, int OC) { (void)mode; return dtype == GS_BF16 && IC % 64 == 0 && OC % 64 == 0; }
// thin bf16 layers whose output side has a multiple of 64 channels: 32 x 64 pairs per block (conv_wgrad_bf16_kernel<.., 2>)
static bool wgrad_thin_pairs(int mode, int dtype, int IC, int OC) {
    static const bool off = getenv("GS_NO_THIN_PAIRS") != nullptr;   // measurement knob
    return !off && dtype == GS_BF16 && !wgrad_2x2(mode, dtype, IC, OC) && OC % 64 == 0;
}
// the 32-input-channel bf16 layers (HBM-bound top of the pyramid): LDS-DMA staged, wave-specialised kernel (conv_wgrad_bf16_thin_dma_kernel)
static bool wgrad_thin_dma(int mode, int dtype, int IC, int OC, int Wb) {
    static const bool off = getenv("GS_NO_THIN_DMA") != nullptr;   // measurement knob: back to conv_wgrad_bf16_kernel
    (void)mode;
    return !off && dtype == GS_BF16 && IC == 32 && (OC == 32 || OC == 64) && Wb >= 32;
}
static void wgrad_geometry(int mode, int dtype, int N, int Hb, int Wb, int IC, int OC, int* tw, int* tiles_x, int* tiles_y,
                           int* ntiles, int* nslices) {
    int np = (mode == MODE_S2 ? 64 : 128) * (dtype == GS_BF16 ? 2 : 1);
    if (mode == MODE_S2 && wgrad_2x2(mode, dtype, IC, OC)) np = 64;
    if (wgrad_thin_dma(mode, dtype, IC, OC, Wb)) {   // 256-pixel tiles, 64 at stride 2; two double-buffered blocks per CU
        np = mode == MODE_S2 ? 64 : 256;
        *tw = 32;
        const int th = np / 32;
        *tiles_x = cdiv(Wb, 32);
        *tiles_y = cdiv(Hb, th);
        *ntiles = N * *tiles_x * *tiles_y;
        const int lds = 2 * (((patch_dim_rt(mode, th) * patch_dim_rt(mode, 32) + 15) / 16) * 1024 + (OC / 32) * np * 64);
        int per_cu = (160 * 1024) / lds;
        if (per_cu > 2) per_cu = 2;
        if (per_cu < 1) per_cu = 1;
        int ns = per_cu * wgrad_cus();
        if (ns > *ntiles) ns = *ntiles;
        *nslices = ns;
        return;
    }
    *tw = Wb >= 32 ? 32 : 16;
    const int th = np / *tw;
    *tiles_x = cdiv(Wb, *tw);
    *tiles_y = cdiv(Hb, th);
    *ntiles = N * *tiles_x * *tiles_y;
    int pairs = (IC / 32) * (OC / 32);
    int target = dtype == GS_BF16 ? 768 : 512;
    if (wgrad_2x2(mode, dtype, IC, OC)) { pairs /= 4; target = 256; }   // 64 x 64 tiles, double-buffered: one block of 4 waves per CU
    else if (wgrad_thin_pairs(mode, dtype, IC, OC)) pairs /= 2;
    int ns = target / pairs;
    if (ns < 1) ns = 1;
    if (ns > *ntiles) ns = *ntiles;
    *nslices = ns;
}

bool wgrad_mfma_has_bias(int dtype) { return dtype == GS_BF16; }   // the bf16 kernels produce the bias gradient on the side
size_t wgrad_mfma_bytes(int mode, int dtype, int N, int Hb, int Wb, int IC, int OC) {
    int tw, tx, ty, nt, ns;
    wgrad_geometry(mode, dtype, N, Hb, Wb, IC, OC, &tw, &tx, &ty, &nt, &ns);
    return align256((size_t)ns * (9 * (size_t)IC * OC + OC) * sizeof(float));
}

// x: conv input side [N][Hi][Wi][IC]; gy: [N][Hb][Wb][OC]; gw[9][IC][OC] (or transposed)
// gb (optional, bf16 only): bias gradient sum_pixels gy[.][oc], produced by the same two launches
int run_wgrad_mfma(int mode, const WgradSrcs& srcs, int nsrc, float* gw, float* gb, int N, int Hi, int Wi, int IC, int OC, int Hb,
                   int Wb, float alpha, int transpose, int accumulate, int dtype, void* ws, size_t ws_bytes, hipStream_t st,
                   GsWgradReduce* defer) {
    // N = images of ALL sources
    if (nsrc < 1 || nsrc > GS_WGRAD_MAX_SRC || srcs.n_end[nsrc - 1] != N) return fail(GS_ERR_ARG, "conv wgrad: %d sources ending at image %d for N=%d", nsrc, srcs.n_end[nsrc > 0 ? nsrc - 1 : 0], N);
    int tw, tiles_x, tiles_y, ntiles, nslices;
    wgrad_geometry(mode, dtype, N, Hb, Wb, IC, OC, &tw, &tiles_x, &tiles_y, &ntiles, &nslices);
    if (gb && !wgrad_mfma_has_bias(dtype)) return fail(GS_ERR_UNSUPPORTED, "conv wgrad: fused bias gradient needs the bf16 kernels");
    const int with_bias = gb != nullptr;
    const size_t need = (size_t)nslices * (9 * (size_t)IC * OC + (with_bias ? OC : 0)) * sizeof(float);
    if (ws_bytes < need) return fail(GS_ERR_WORKSPACE, "conv wgrad: workspace %zu < %zu", ws_bytes, need);
    float* part = reinterpret_cast<float*>(ws);
    dim3 grid((IC / 32) * (OC / 32), nslices);
    // algorithmic work of a weight gradient: the forward conv's FLOPs over all sources; bytes = x + gy read once, gw written once
    const double wg_flops = 2.0 * 9.0 * (double)N * Hb * Wb * IC * OC;
    const double wg_bytes = ((double)N * Hi * Wi * IC + (double)N * Hb * Wb * OC) * (dtype == GS_F32 ? 4.0 : 2.0) + 9.0 * IC * OC * 4.0;
    ProfScope ps(st, wg_flops, wg_bytes, 10 + mode, N, Hb, Wb, IC, OC, nsrc, defer ? 1 : 0);
    {
#define GS_WG(TT, M, TWV)                                                                                              \
    hipLaunchKernelGGL((conv_wgrad_kernel<TT, M, TWV>), grid, dim3(256), 0, st, srcs,                                  \
                       part, N, Hi, Wi, IC, OC, Hb, Wb, tiles_x, tiles_y, ntiles, nslices)
#define GS_WG_ALL(TT)                                                                       \
    do {                                                                                    \
        if (mode == MODE_S1) { if (tw == 32) GS_WG(TT, MODE_S1, 32); else GS_WG(TT, MODE_S1, 16); } \
        else { if (tw == 32) GS_WG(TT, MODE_S2, 32); else GS_WG(TT, MODE_S2, 16); }          \
    } while (0)
        if (dtype == GS_F32) {
            GS_WG_ALL(float);
        } else {
#define GS_WGB(M, TWV)                                                                                                  \
    do {                                                                                                                \
        if (wgrad_thin_pairs(mode, dtype, IC, OC))                                                                      \
            hipLaunchKernelGGL((conv_wgrad_bf16_kernel<M, TWV, 2>), dim3((IC / 32) * (OC / 64), nslices), dim3(192), 0, st, srcs, \
                               part, N, Hi, Wi, IC, OC, Hb, Wb, tiles_x, tiles_y, ntiles, nslices, with_bias);         \
        else                                                                                                            \
            hipLaunchKernelGGL((conv_wgrad_bf16_kernel<M, TWV>), grid, dim3(192), 0, st, srcs,                          \
                               part, N, Hi, Wi, IC, OC, Hb, Wb, tiles_x, tiles_y, ntiles, nslices, with_bias);         \
    } while (0)
#define GS_WGB2(M, TWV)                                                                                                 \
    do {                                                                                                                \
        constexpr int np_ = (M == MODE_S2 ? 64 : 256), th_ = np_ / TWV;                                                 \
        constexpr int rows_ = patch_dim<M>(th_) * patch_dim<M>(TWV);                                                    \
        constexpr int xt_ = GS_SK_ROW128 ? (((rows_ + 7) / 8 + 3) / 4) * 4096 : 2 * ((((rows_ + 15) / 16 + 3) / 4) * 4096);   \
        constexpr int lds_ = 2 * (xt_ + np_ * 128);                                                                     \
        auto kern_ = conv_wgrad_bf16_2x2_kernel<M, TWV>;                                                                \
        static bool set_ = false;                                                                                       \
        if (!set_) {                                                                                                    \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern_), hipFuncAttributeMaxDynamicSharedMemorySize, lds_) != hipSuccess) \
                return fail(GS_ERR_HIP, "conv wgrad: cannot reserve %d bytes of dynamic LDS", lds_);                    \
            set_ = true;                                                                                                \
        }                                                                                                               \
        hipLaunchKernelGGL(kern_, dim3((IC / 64) * (OC / 64), nslices), dim3(256), lds_, st, srcs,                          \
                           part, N, Hi, Wi, IC, OC, Hb, Wb, tiles_x, tiles_y, ntiles, nslices, with_bias);              \
    } while (0)
#define GS_WGT(M, OTV)                                                                                                  \
    do {                                                                                                                \
        constexpr int np_ = (M == MODE_S2 ? 64 : 256), th_ = np_ / 32;                                                  \
        constexpr int lds_ = 2 * (((patch_dim<M>(th_) * patch_dim<M>(32) + 15) / 16) * 1024 + OTV * np_ * 64);           \
        auto kern_ = conv_wgrad_bf16_thin_dma_kernel<M, OTV>;                                                           \
        static bool set_ = false;                                                                                       \
        if (!set_) {                                                                                                    \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern_), hipFuncAttributeMaxDynamicSharedMemorySize, lds_) != hipSuccess) \
                return fail(GS_ERR_HIP, "conv wgrad: cannot reserve %d bytes of dynamic LDS", lds_);                    \
            set_ = true;                                                                                                \
        }                                                                                                               \
        hipLaunchKernelGGL(kern_, dim3(1, nslices), dim3(256), lds_, st, srcs,                                          \
                           part, N, Hi, Wi, IC, OC, Hb, Wb, tiles_x, tiles_y, ntiles, nslices, with_bias);              \
    } while (0)
            if (wgrad_thin_dma(mode, dtype, IC, OC, Wb)) {
                if (mode == MODE_S1) { if (OC == 32) GS_WGT(MODE_S1, 1); else GS_WGT(MODE_S1, 2); }
                else { if (OC == 32) GS_WGT(MODE_S2, 1); else GS_WGT(MODE_S2, 2); }
            } else if (wgrad_2x2(mode, dtype, IC, OC)) {
                if (mode == MODE_S1) { if (tw == 32) GS_WGB2(MODE_S1, 32); else GS_WGB2(MODE_S1, 16); }
                else { if (tw == 32) GS_WGB2(MODE_S2, 32); else GS_WGB2(MODE_S2, 16); }
            } else {
                if (mode == MODE_S1) { if (tw == 32) GS_WGB(MODE_S1, 32); else GS_WGB(MODE_S1, 16); }
                else { if (tw == 32) GS_WGB(MODE_S2, 32); else GS_WGB(MODE_S2, 16); }
            }
#undef GS_WGT
#undef GS_WGB2
#undef GS_WGB
        }
#undef GS_WG_ALL
#undef GS_WG
    }
    GS_CHECK_LAUNCH();
    wgrad_reduce_launch(part, gw, gb, nslices, 9, IC, OC, alpha, transpose, accumulate, st, defer);
    GS_CHECK_LAUNCH();
    return 0;
}

// ---- grouped weight gradients: planning and launch of one (mode, tile width) group
bool wgrad_sk_supported(int mode, int dtype, int IC, int OC) { return wgrad_2x2(mode, dtype, IC, OC); }
// Always 32-wide tiles: a 16-wide image leaves half of a tile's columns empty either way (a 2 x 16 image fills 1/8 of a 16 x 16 tile
// and 1/8 of an 8 x 32 one), and with one width all layers of a conv mode share ONE group.  (GS_SK_TW16: separate groups, for measurements.)
int wgrad_sk_tile_width(int Wb) {
    static const bool tw16 = getenv("GS_SK_TW16") != nullptr;
    return (Wb >= 32 || !tw16) ? 32 : 16;
}
// fills the tiling of job q (its srcs / channel counts / image sizes already set; N = images of all sources)
void wgrad_sk_job_geometry(int mode, int tw, int N, SkJob& q) {
    const int np = mode == MODE_S2 ? 64 : 256;
    const int th = np / tw;
    q.tiles_x = cdiv(q.Wb, tw);
    q.tiles_y = cdiv(q.Hb, th);
    q.ntiles = N * q.tiles_x * q.tiles_y;
    q.n_ict = q.IC / 64;
    q.nct = q.n_ict * (q.OC / 64);
}
// unit / run numbering and the block count of a group whose jobs carry their geometry
void wgrad_sk_plan(int mode, SkGroup& g) {
    long units = 0;
    int runs = 0;
    for (int j = 0; j < g.njobs; ++j) {
        g.job[j].unit_base = (int)units;
        g.job[j].run_base = runs;
        units += (long)g.job[j].nct * g.job[j].ntiles;
        runs += g.job[j].nct;
    }
    g.total_units = (int)units;
    g.total_runs = runs;
    // a unit is 2.4 us of MFMAs (stride 1: 144 per wave) or ~2.5 us of patch staging (stride 2: 36 MFMAs under a 4-5x larger patch); a block
    // needs a few of them to amortise its prologue and its flush (measured: scripts/run_sk.sh)
    static const int upb_env = getenv("GS_SK_UNITS_PER_BLOCK") ? atoi(getenv("GS_SK_UNITS_PER_BLOCK")) : 0;
    static const int upb2_env = getenv("GS_SK_UNITS_PER_BLOCK_S2") ? atoi(getenv("GS_SK_UNITS_PER_BLOCK_S2")) : 0;
    const int upb = mode == MODE_S2 ? (upb2_env > 0 ? upb2_env : 4) : (upb_env > 0 ? upb_env : 2);
    long nb = units / upb;
    if (nb > wgrad_cus()) nb = wgrad_cus();
    if (nb < 1) nb = 1;
    g.nblocks = (int)nb;
}
size_t wgrad_sk_bytes(const SkGroup& g) { return align256((size_t)(g.nblocks + g.total_runs) * GS_SK_PSTRIDE * sizeof(float)); }

int run_wgrad_sk(int mode, int tw, const SkGroup& g, void* ws, size_t ws_bytes, hipStream_t st) {
    if (g.njobs < 1 || g.njobs > GS_SK_MAX_JOBS) return fail(GS_ERR_ARG, "conv wgrad group: %d jobs", g.njobs);
    if (ws_bytes < wgrad_sk_bytes(g)) return fail(GS_ERR_WORKSPACE, "conv wgrad group: workspace %zu < %zu", ws_bytes, wgrad_sk_bytes(g));
    if ((long)g.total_units <= 0) return 0;
    float* part = reinterpret_cast<float*>(ws);
    double flops = 0.0, bytes = 0.0;
    int images = 0;
    for (int j = 0; j < g.njobs; ++j) {
        const SkJob& q = g.job[j];
        const int N = q.srcs.n_end[GS_WGRAD_MAX_SRC - 1];
        images += N;
        flops += 2.0 * 9.0 * (double)N * q.Hb * q.Wb * q.IC * q.OC;
        bytes += ((double)N * q.Hi * q.Wi * q.IC + (double)N * q.Hb * q.Wb * q.OC) * 2.0 + 9.0 * q.IC * q.OC * 4.0;
    }
    {
        // kind 20 + mode: a GROUP of weight gradients (N = layers, Hb = tile width, Wb = blocks, IC = units, OC = runs)
        ProfScope ps(st, flops, bytes, 20 + mode, g.njobs, tw, g.nblocks, g.total_units, g.total_runs, images, 1);
#define GS_WGSK(M, TWV, SP)                                                                                              \
    do {                                                                                                                \
        constexpr int np_ = (M == MODE_S2 ? 64 : 256), th_ = np_ / TWV;                                                 \
        constexpr int lds_ = 2 * (2 * ((((patch_dim<M>(th_) * patch_dim<M>(TWV) + 15) / 16 + 3) / 4) * 4096 + np_ * 64));  \
        auto kern_ = conv_wgrad_bf16_2x2_sk_kernel<M, TWV, SP>;                                                         \
        static bool set_ = false;                                                                                       \
        if (!set_) {                                                                                                    \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern_), hipFuncAttributeMaxDynamicSharedMemorySize, lds_) != hipSuccess) \
                return fail(GS_ERR_HIP, "conv wgrad group: cannot reserve %d bytes of dynamic LDS", lds_);              \
            set_ = true;                                                                                                \
        }                                                                                                               \
        hipLaunchKernelGGL(kern_, dim3((unsigned)g.nblocks), dim3(SP ? 512 : 256), lds_, st, g, part);                  \
    } while (0)
        // Wave-specialised variant (4 compute + 4 staging waves): OFF by default (GS_SK_SPEC=1 / 2: on for both modes / stride 1 only).
        // Measured (scripts/bench_wgrad_group.py, the discriminator's layers, 16 images, launch + fold): stride 1 186 -> 175 us in isolation,
        // stride 2 (bound by the L2 -> LDS rate of its 4-5x larger patch; the staging waves start a tile's DMA only after the barrier)
        // 131 -> 168 us -- but the whole captured step LOSES 0.14 ms with the stride-1 groups specialised (6.83 -> 6.97 ms, three
        // alternating runs on one box) although their own eager timings improve by 7-12 us: kept for measurements only.
        static const int spec_env = getenv("GS_SK_SPEC") ? atoi(getenv("GS_SK_SPEC")) : 0;
        const bool spec = spec_env == 1 || (spec_env == 2 && mode == MODE_S1);
        if (spec) {
            if (mode == MODE_S1) { if (tw == 32) GS_WGSK(MODE_S1, 32, true); else GS_WGSK(MODE_S1, 16, true); }
            else { if (tw == 32) GS_WGSK(MODE_S2, 32, true); else GS_WGSK(MODE_S2, 16, true); }
        } else {
            if (mode == MODE_S1) { if (tw == 32) GS_WGSK(MODE_S1, 32, false); else GS_WGSK(MODE_S1, 16, false); }
            else { if (tw == 32) GS_WGSK(MODE_S2, 32, false); else GS_WGSK(MODE_S2, 16, false); }
        }
#undef GS_WGSK
    }
    GS_CHECK_LAUNCH();
    hipLaunchKernelGGL(wgrad_sk_reduce_kernel, dim3((GS_SK_PSTRIDE / 4 + 63) / 64, (unsigned)g.total_runs), dim3(256), 0, st, g, part);
    GS_CHECK_LAUNCH();
    return 0;
}

}  // namespace gs

extern "C" int gs_wgrad_cu_cap(int cap) {
    const int was = gs::g_wgrad_cu_cap;
    gs::g_wgrad_cu_cap = cap < 0 ? 0 : cap;
    return was;
}
extern "C" int gs_prof_enable(int on) {
    gs::g_prof.on = on != 0;
    gs::g_prof.burst = on > 1 ? on : 1;
    gs::g_prof.used = 0;
    gs::g_prof.flops = 0.0;
    return 0;
}

// Roofline accounting of the same launches: total algorithmic bytes, and the sum over launches of the time the binding roof
// (MFMA peak or HBM bandwidth, whichever is larger for that launch) allows.  Call BEFORE gs_prof_collect (which resets).
extern "C" int gs_prof_roofline(double peak_tflops, double peak_gbps, double* total_bytes, double* roof_ms, double* roof_ms_hbm_bound) {
    double bytes = 0.0, roof = 0.0, roof_hbm = 0.0;
    for (int i = 0; i < gs::g_prof.used; ++i) {
        if (gs::g_prof.ldesc[i][0] >= 10) continue;
        const double tf = gs::g_prof.lflops[i] / (peak_tflops * 1e12) * 1e3, tb = gs::g_prof.lbytes[i] / (peak_gbps * 1e9) * 1e3;
        bytes += gs::g_prof.lbytes[i];
        roof += tf > tb ? tf : tb;
        if (tb >= tf) roof_hbm += tb;
    }
    if (total_bytes) *total_bytes = bytes;
    if (roof_ms) *roof_ms = roof;
    if (roof_ms_hbm_bound) *roof_ms_hbm_bound = roof_hbm;
    return 0;
}

// Per-launch records of everything recorded since gs_prof_enable(1) (implicit-GEMM convs AND weight gradients): duration (ms),
// algorithmic FLOPs and bytes, and 8 ints {kind, N, Hb, Wb, IC, OC, masked | sources, fused norm | deferred}; kind = conv mode
// (0 stride 1, 1 stride 2, 2 transposed) or 10 + mode for the weight gradient of that conv.  Call BEFORE gs_prof_collect.
extern "C" int gs_prof_records(int max_records, int* n, double* ms, double* flops, double* bytes, int* desc) {
    int k = 0;
    for (int i = 0; i < gs::g_prof.used && k < max_records; ++i, ++k) {
        (void)hipEventSynchronize(gs::g_prof.ev[i][1]);
        float t = 0.f;
        if (hipEventElapsedTime(&t, gs::g_prof.ev[i][0], gs::g_prof.ev[i][1]) != hipSuccess) t = 0.f;
        t /= (float)gs::g_prof.lreps[i];
        ms[k] = t; flops[k] = gs::g_prof.lflops[i]; bytes[k] = gs::g_prof.lbytes[i];
        for (int j = 0; j < 8; ++j) desc[8 * k + j] = gs::g_prof.ldesc[i][j];
    }
    if (n) *n = k;
    return 0;
}

extern "C" int gs_prof_collect(int* launches, double* total_ms, double* total_flops) {
    double ms = 0.0;
    int count = 0;
    for (int i = 0; i < gs::g_prof.used; ++i) {
        if (gs::g_prof.ldesc[i][0] >= 10) continue;
        ++count;
        (void)hipEventSynchronize(gs::g_prof.ev[i][1]);
        float t = 0.f;
        if (hipEventElapsedTime(&t, gs::g_prof.ev[i][0], gs::g_prof.ev[i][1]) == hipSuccess) ms += t / (float)gs::g_prof.lreps[i];
    }
    if (launches) *launches = count;
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = gs::g_prof.flops;
    gs::g_prof.used = 0;
    gs::g_prof.flops = 0.0;
    return 0;
}
