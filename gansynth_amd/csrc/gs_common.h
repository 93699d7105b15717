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
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
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
