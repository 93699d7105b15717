// Shared device/host helpers for libgansynth_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/gansynth_hip.h"

namespace gs {

// ------------------------------------------------------------------ error reporting
extern thread_local char g_err[512];
int fail(int code, const char* fmt, ...);

#define GS_CHECK_ARG(cond, ...)                         \
    do {                                                \
        if (!(cond)) return gs::fail(GS_ERR_ARG, __VA_ARGS__); \
    } while (0)

#define GS_CHECK_LAUNCH()                                                             \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) return gs::fail(GS_ERR_HIP, "%s:%d launch failed: %s", \
                                               __FILE__, __LINE__, hipGetErrorString(e__)); \
    } while (0)

// ---------------------------------------------------------------------- bf16 storage
struct bf16_t {
    unsigned short v;
};

__device__ __host__ inline float bf16_to_f32(bf16_t h) {
    union { unsigned int u; float f; } c;
    c.u = ((unsigned int)h.v) << 16;
    return c.f;
}
__device__ __host__ inline bf16_t f32_to_bf16(float f) {  // round-to-nearest-even
    union { unsigned int u; float f; } c;
    c.f = f;
    unsigned int u = c.u;
    bf16_t r;
    if ((u & 0x7fffffffu) > 0x7f800000u) { r.v = (unsigned short)((u >> 16) | 0x40); return r; }  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    r.v = (unsigned short)(u >> 16);
    return r;
}

template <typename T> struct DT;
template <> struct DT<float> {
    static constexpr int id = GS_F32;
    __device__ static inline float ld(const float* p) { return *p; }
    __device__ static inline void st(float* p, float v) { *p = v; }
};
template <> struct DT<bf16_t> {
    static constexpr int id = GS_BF16;
    __device__ static inline float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static inline void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 4-element vector access (16 B for f32, 8 B for bf16); pointers must be aligned.
__device__ inline void ld4(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ inline void st4(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}
__device__ inline void ld4(const bf16_t* p, float (&o)[4]) {
    uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}
// two floats -> packed bf16 pair, round-to-nearest-even: one v_cvt_pk_bf16_f32 on gfx950
__device__ inline unsigned int pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}
__device__ inline void st4(bf16_t* p, const float (&o)[4]) {
    uint2 v;
    v.x = pack_bf16x2(o[0], o[1]);
    v.y = pack_bf16x2(o[2], o[3]);
    *reinterpret_cast<uint2*>(p) = v;
}

// tanh for the streaming kernels: ~14 branch-free VALU instructions against ~50 branchy ones of the device library's tanhf (which the colour
// block of the generator evaluates at a quarter of the lanes: it WAS the kernel).  |x| >= 0.25: 1 - 2 / (exp(2|x|) + 1) with v_exp_f32 /
// v_rcp_f32 (relative error <= 5e-7 there, exactly 1 once exp overflows); below: the odd series to x^7 (next term 62/2835 x^9 < 1e-7 relative).
__device__ inline float fast_tanh(float x) {
    const float ax = fabsf(x);
    const float e = __expf(2.f * ax);
    const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);   // (v_rcp_f32: 1 ulp; __frcp_rn is a correctly rounded division, ~10 instructions)
    const float x2 = ax * ax;
    const float small = ax * (1.f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * -0.05396825f)));
    return copysignf(ax < 0.25f ? small : big, x);
}

// 16-byte accesses: 4 floats or 8 bf16 per lane (what the memory path wants from a streaming kernel)
template <typename T> struct Wide;   // 16 bytes of T
template <> struct Wide<float> { static constexpr int N = 4; };
template <> struct Wide<bf16_t> { static constexpr int N = 8; };
template <typename T> __device__ inline void ld_wide(const T* p, float* o);
template <> __device__ inline void ld_wide<float>(const float* p, float* o) { ld4(p, *reinterpret_cast<float(*)[4]>(o)); }
template <> __device__ inline void ld_wide<bf16_t>(const bf16_t* p, float* o) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
template <typename T> __device__ inline void st_wide(T* p, const float* o);
template <> __device__ inline void st_wide<float>(float* p, const float* o) { *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]); }
template <> __device__ inline void st_wide<bf16_t>(bf16_t* p, const float* o) {
    uint4 v;
    v.x = pack_bf16x2(o[0], o[1]); v.y = pack_bf16x2(o[2], o[3]); v.z = pack_bf16x2(o[4], o[5]); v.w = pack_bf16x2(o[6], o[7]);
    *reinterpret_cast<uint4*>(p) = v;
}

// ------------------------------------------------------------- wave64 / block reductions
// Cross-lane sums on the VALU: DPP moves inside a 16-lane row, v_permlane16_swap / v_permlane32_swap across rows.  (__shfl_xor is a
// ds_bpermute whatever its offset -- a round trip through the LDS pipe per step.  It also turned out NOT to be safe here: with another
// kernel's LDS-DMA traffic on the same CU -- a forked branch of the run's hipGraph -- three bpermutes in flight returned wrong sums in the
// upper half of the wave a few times per thousand rows, with s_waitcnt lgkmcnt(0) in front of every use: profiles/r05_e_bpermute_note.txt.)
template <int CTRL> __device__ inline float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
// v_permlane16_swap / v_permlane32_swap on a copy of the value, written out: with the builtin hipcc places the copy two wait states in front
// of the swap (v_mov, v_mov, v_mov, swap, swap in the NORM == 2 epilogue of conv_igemm) and the bf16 instantiation of that epilogue then
// summed wrong halves on the hardware (test_data_gradient_continued_through_the_previous_pixel_norm, 40 % of the tensor's scale) while the
// fp32 one (v_mov, v_mov, s_nop 0, swap, swap) was right -- five idle cycles on either side of the swap and both are.
// (Round 6, advisor: was an undeclared M0 write of the LDS-DMA asm the real cause?  No: with M0 declared clobbered there and the padding
//  removed here, 10 kernel tests fail again (pixel-norm sums, bias folds) -- the wait states are a data hazard of the swap itself, between the
//  VALU write of its operands and the swap and between the swap and its consumer, which the compiler's hazard recogniser does not insert
//  around inline asm.)
__device__ inline float swap16_sum(float v) {   // v[lane] + v[lane ^ 16]
    float a = v, b;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 4\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 4" : "+v"(a), "=&v"(b));
    return a + b;
}
__device__ inline float swap32_sum(float v) {   // v[lane] + v[lane ^ 32]
    float a = v, b;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 4\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 4" : "+v"(a), "=&v"(b));
    return a + b;
}
// Sum over every aligned group of L consecutive lanes (L = 1, 2, 4 ... 64, wave-uniform), result in every lane of the group.
__device__ inline float group_sum(float v, int L) {
    if (L > 1) v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
    if (L > 2) v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
    if (L > 4) v += dpp_mov<0x141>(v);    // row_half_mirror: the other quad of the 8 (every lane of a quad holds the quad's sum by now)
    if (L > 8) v += dpp_mov<0x140>(v);    // row_mirror: the other half of the row
    if (L > 16) v = swap16_sum(v);
    if (L > 32) v = swap32_sum(v);
    return v;
}
// Sum over the lanes with the same (lane % L): lane, lane + L, lane + 2L ... (L = 1, 2, 4 ... 64, wave-uniform), result in all of them.
__device__ inline float residue_sum(float v, int L) {
    if (L <= 1) v += dpp_mov<0x121>(v);   // row_ror:1 -- rotations inside the 16-lane row keep the residue for every power of two
    if (L <= 2) v += dpp_mov<0x122>(v);
    if (L <= 4) v += dpp_mov<0x124>(v);
    if (L <= 8) v += dpp_mov<0x128>(v);
    if (L <= 16) v = swap16_sum(v);
    if (L <= 32) v = swap32_sum(v);
    return v;
}
__device__ inline float wave_sum(float v) { return group_sum(v, 64); }
// Sum over a block of NT threads (NT multiple of 64, <= 1024); result valid in every thread.
template <int NT>
__device__ inline float block_sum(float v, float* smem /* >= NT/64 floats */) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smem[w] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += smem[i];
    return r;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// dispatch on dtype id
#define GS_DISPATCH_DTYPE(dtype, ...)                                   \
    do {                                                                \
        if ((dtype) == GS_F32) { using T = float; __VA_ARGS__; }        \
        else if ((dtype) == GS_BF16) { using T = gs::bf16_t; __VA_ARGS__; } \
        else return gs::fail(GS_ERR_ARG, "bad dtype %d", (int)(dtype)); \
    } while (0)

}  // namespace gs
