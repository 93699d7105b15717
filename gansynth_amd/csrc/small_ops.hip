// Small-tensor kernels of the PGGAN hot path: dense (ops.py:183-201), embedding (ops.py:204-218)
// and minibatch stddev (ops.py:336-348) with their first/second-order gradients.  These are
// weight-bandwidth-bound (dense: 16.8 MB of fp32 weights for 8 rows) or tiny; none is MFMA work.
#include "gs_common.h"

namespace gs {

constexpr int DENSE_BT = 16;  // batch rows held in registers per pass

// epilogue of the fused forms (gs_dense_fwd_bias_act*): y = act(alpha * sum + bias[col]); bias may be NULL, act = GS_ACT_*
__device__ __forceinline__ float dense_epilogue(float sum, float alpha, const float* __restrict__ bias, int col, int act) {
    float v = sum * alpha;
    if (bias) v += bias[col];
    if (act == GS_ACT_LRELU) v = v > 0.f ? v : 0.2f * v;
    else if (act == GS_ACT_TANH) v = tanhf(v);
    return v;
}

// ---------------------------------------------------------------------------- dense fwd
// y[b][o] = alpha * sum_i x[b][i] * w[i][o].  Block = 64 output columns x 4 i-lanes; grid.y
// splits the reduction so that (out/64)*ksplit blocks cover the chip; partial sums go through
// `part[ks][b][o]` and a finalize pass (deterministic).
template <typename T>
__global__ __launch_bounds__(256) void dense_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, float* __restrict__ part,
                                                        int b0, int nb, int in, int out, int b_total, int ipb) {
    __shared__ float red[4][DENSE_BT][64];
    const int tid = threadIdx.x;
    const int col = blockIdx.x * 64 + (tid & 63);
    const int sub = tid >> 6;
    const int ks = blockIdx.y;
    const int i0 = ks * ipb;
    int i1 = i0 + ipb;
    if (i1 > in) i1 = in;
    float acc[DENSE_BT];
#pragma unroll
    for (int b = 0; b < DENSE_BT; ++b) acc[b] = 0.f;
    if (col < out) {
#pragma unroll 4
        for (int i = i0 + sub; i < i1; i += 4) {
            const float wv = w[(long)i * out + col];
            // (unconditional loads -- rows past nb re-read the last valid row, their sums are never written: a per-row `if (b < nb) load`
            //  makes hipcc branch around every load and drain vmcnt(0) each time, 13 us for the 256 -> 61 logits layer)
#pragma unroll
            for (int b = 0; b < DENSE_BT; ++b) acc[b] += DT<T>::ld(x + (long)(b0 + (b < nb ? b : nb - 1)) * in + i) * wv;
        }
    }
#pragma unroll
    for (int b = 0; b < DENSE_BT; ++b) red[sub][b][tid & 63] = acc[b];
    __syncthreads();
    if (sub == 0 && col < out) {
        for (int b = 0; b < nb; ++b)
            part[((long)ks * b_total + b0 + b) * out + col] = red[0][b][tid] + red[1][b][tid] + red[2][b][tid] + red[3][b][tid];
    }
}
// Small layers (the 256 -> 61 logits, ops.py:183-201 on networks.py:186): one block per (64 columns, batch row), 4 row lanes folded through
// LDS, the result written directly -- instead of 8 row-split blocks + a finalize launch (10 + 5 us for a 16 x 256 x 61 product).
template <typename T>
__global__ __launch_bounds__(256) void dense_fwd_small_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y, int in, int out, float alpha,
                                                              const float* __restrict__ bias, int act) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    const int colc = col < out ? col : out - 1;   // (unconditional loads)
    const T* xr = x + (long)blockIdx.y * in;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = sub;
    // 16 weight rows per trip, all 32 loads of the trip in flight together (a chain of 4-row trips was 16 dependent round trips: 10 us)
    for (; i + 60 < in; i += 64) {
        float wv[16], xv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { wv[k] = w[(long)(i + 4 * k) * out + colc]; xv[k] = DT<T>::ld(xr + i + 4 * k); }
#pragma unroll
        for (int k = 0; k < 16; k += 4) { a0 += xv[k] * wv[k]; a1 += xv[k + 1] * wv[k + 1]; a2 += xv[k + 2] * wv[k + 2]; a3 += xv[k + 3] * wv[k + 3]; }
    }
    for (; i < in; i += 4) a0 += DT<T>::ld(xr + i) * w[(long)i * out + colc];
    red[sub][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sub == 0 && col < out) DT<T>::st(y + (long)blockIdx.y * out + col, dense_epilogue(red[0][c] + red[1][c] + red[2][c] + red[3][c], alpha, bias, col, act));
}

// block = 64 elements x 4 split lanes (a narrow layer has few elements and up to 64 splits: a thread per element would walk them
// as one chain in a handful of blocks)
template <typename T>
__global__ __launch_bounds__(256) void dense_finalize_kernel(const float* __restrict__ part, T* __restrict__ y, long n, int ksplit, float alpha,
                                                             const float* __restrict__ bias = nullptr, int act = 0, int out = 1) {
    __shared__ float red[4][64];
    const int e = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + e;
    float s0 = 0.f, s1 = 0.f;
    if (i < n) {
        int k = kl;
        for (; k + 4 < ksplit; k += 8) { s0 += part[(long)k * n + i]; s1 += part[(long)(k + 4) * n + i]; }
        if (k < ksplit) s0 += part[(long)k * n + i];
    }
    red[kl][e] = s0 + s1;
    __syncthreads();
    if (kl == 0 && i < n) DT<T>::st(y + i, dense_epilogue(red[0][e] + red[1][e] + red[2][e] + red[3][e], alpha, bias, (int)(i % out), act));
}

// ------------------------------------------------------------------------ dense bwd data
// gx[b][i] = alpha * sum_o gy[b][o] * w[i][o] : one wave per weight row i (coalesced along o).
template <typename T>
__global__ __launch_bounds__(256) void dense_bwd_data_kernel(const T* __restrict__ gy, const float* __restrict__ w, T* __restrict__ gx,
                                                             int b0, int nb, int in, int out, float alpha) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= in) return;
    float acc[DENSE_BT];
#pragma unroll
    for (int b = 0; b < DENSE_BT; ++b) acc[b] = 0.f;
    const float* wr = w + (long)i * out;
#pragma unroll 2
    for (int o = lane; o < out; o += 64) {
        const float wv = wr[o];
#pragma unroll
        for (int b = 0; b < DENSE_BT; ++b) acc[b] += DT<T>::ld(gy + (long)(b0 + (b < nb ? b : nb - 1)) * out + o) * wv;   // (unconditional, see dense_fwd_kernel)
    }
#pragma unroll
    for (int b = 0; b < DENSE_BT; ++b) {
        if (b < nb) {
            const float s = wave_sum(acc[b]);
            if (lane == 0) DT<T>::st(gx + (long)(b0 + b) * in + i, s * alpha);
        }
    }
}

// ---------------------------------------------------------------------- dense bwd weight
// gw[i][o] = alpha * sum_b x[b][i] * gy[b][o]
template <typename T>
__global__ void dense_bwd_weight_kernel(const T* __restrict__ x, const T* __restrict__ gy, float* __restrict__ gw, int b, int in, int out, float alpha, int accumulate) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)in * out) return;
    const int o = e % out;
    const int i = e / out;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= b; k += 8) {   // eight batch rows per trip, their 16 loads in flight together
        float xv[8], gv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { xv[j] = DT<T>::ld(x + (long)(k + j) * in + i); gv[j] = DT<T>::ld(gy + (long)(k + j) * out + o); }
#pragma unroll
        for (int j = 0; j < 8; ++j) s += xv[j] * gv[j];
    }
    for (; k < b; ++k) s += DT<T>::ld(x + (long)k * in + i) * DT<T>::ld(gy + (long)k * out + o);
    gw[e] = accumulate ? gw[e] + s * alpha : s * alpha;
}

// wide rows (out >= 2048, e.g. the generator's 512 -> 8192 dense): one block per weight row
template <typename T>
__global__ __launch_bounds__(256) void dense_bwd_data_wide_kernel(const T* __restrict__ gy, const float* __restrict__ w, T* __restrict__ gx,
                                                                  int b0, int nb, int in, int out, float alpha) {
    __shared__ float red[4];
    const int i = blockIdx.x;
    float acc[DENSE_BT];
#pragma unroll
    for (int b = 0; b < DENSE_BT; ++b) acc[b] = 0.f;
    const float* wr = w + (long)i * out;
    for (int o = threadIdx.x; o < out; o += 256) {
        const float wv = wr[o];
#pragma unroll
        for (int b = 0; b < DENSE_BT; ++b)
            if (b < nb) acc[b] += DT<T>::ld(gy + (long)(b0 + b) * out + o) * wv;
    }
#pragma unroll
    for (int b = 0; b < DENSE_BT; ++b) {
        if (b < nb) {
            const float s = block_sum<256>(acc[b], red);
            if (threadIdx.x == 0) DT<T>::st(gx + (long)(b0 + b) * in + i, s * alpha);
        }
    }
}


// ---- fast paths (vector loads; the generic kernels above remain for odd shapes) ------------------------------------
// Input side in channels-last order (rc, rhw != 0): x / gx are the memory of a channels-last [b][hw][c] activation while the weight
// rows follow tf.layers.flatten of NCHW (row = c * hw + p, networks.py:185-186).  Weight rows are read and written whole, so the
// flatten is a ROW INDEX MAP inside the kernels: no NCHW copy of the activation, no copy of its gradient back.
__device__ __forceinline__ long dense_row(int r, int rc, int rhw) { return rc ? (long)(r % rc) * rhw + r / rc : (long)r; }

// batch rows per pass of the fast kernels: 8, 16 or 24 (FB template parameter) -- the discriminator's 16- and 24-row calls read the
// weight matrix once instead of two or three times
static int fast_rows(int b) { return b <= 8 ? 8 : (b <= 16 ? 16 : 24); }
// y[b][o] = alpha * sum_i x[b][i] w[i][o], in % 4 == 0, out % 32 == 0.  Block = 8 column threads (float4: 32 columns) x 32 row
// lanes; a row lane takes 4 consecutive rows per pass (one 8/16-byte load of x per batch row).  Row lanes are folded with
// xor-shuffles inside a wave and through LDS across the 4 waves; with ksplit == 1 the result is written directly.
template <typename T, int FB>
__global__ __launch_bounds__(256) void dense_fwd_fast_kernel(const T* __restrict__ x, const float* __restrict__ w, float* __restrict__ part,
                                                             T* __restrict__ y, int b0, int nb, int in, int out, int b_total, int ipb, float alpha, int rc, int rhw,
                                                             const float* __restrict__ bias, int act) {
    __shared__ float red[4][FB][32];
    const int tid = threadIdx.x;
    const int c = tid & 7, r = tid >> 3;           // column thread, row lane (0..31)
    const int col = blockIdx.x * 32 + c * 4;
    const int i0 = blockIdx.y * ipb;
    int i1 = i0 + ipb;
    if (i1 > in) i1 = in;
    float acc[FB][4];
#pragma unroll
    for (int b = 0; b < FB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[b][e] = 0.f;
#pragma unroll 2
    for (int i = i0 + 4 * r; i < i1; i += 128) {
        float4 wv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) wv[k] = *reinterpret_cast<const float4*>(w + dense_row(i + k, rc, rhw) * out + col);
        // (loads are unconditional -- rows past nb re-read the last valid row and their sums are never written: a per-row
        //  `if (b < nb) load` makes hipcc branch around every load and drain vmcnt(0) each time)
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            float xv[4];
            ld4(x + (long)(b0 + (b < nb ? b : nb - 1)) * in + i, xv);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[b][0] += xv[k] * wv[k].x; acc[b][1] += xv[k] * wv[k].y; acc[b][2] += xv[k] * wv[k].z; acc[b][3] += xv[k] * wv[k].w;
            }
        }
    }
    // fold the 8 row lanes of the wave (lane = r_local * 8 + c)
#pragma unroll
    for (int b = 0; b < FB; ++b) {
        if (b < nb) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[b][e];
                v = residue_sum(v, 8);   // lanes l, l + 8 ... l + 56
                acc[b][e] = v;
            }
        }
    }
    const int wv_ = tid >> 6;
    if ((tid & 63) < 8) {
#pragma unroll
        for (int b = 0; b < FB; ++b)
            if (b < nb) {
#pragma unroll
                for (int e = 0; e < 4; ++e) red[wv_][b][c * 4 + e] = acc[b][e];
            }
    }
    __syncthreads();
    for (int k = tid; k < nb * 32; k += 256) {
        const int b = k >> 5, cc = k & 31;
        const float s = red[0][b][cc] + red[1][b][cc] + red[2][b][cc] + red[3][b][cc];
        const int o = blockIdx.x * 32 + cc;
        if (part) part[((long)blockIdx.y * b_total + b0 + b) * out + o] = s;
        else DT<T>::st(y + (long)(b0 + b) * out + o, dense_epilogue(s, alpha, bias, o, act));
    }
}

// gx[b][i] = alpha * sum_o gy[b][o] w[i][o], out % 256 == 0: WPR waves per weight row (1: a wave per row; 4: the block's four
// waves split a long row -- the generator's 512 x 8192 dense has only 512 rows, a wave per row is 128 blocks of 32 serial
// trips), 16 bytes of w per lane per pass.
template <typename T, int WPR, int FB>
__global__ __launch_bounds__(256) void dense_bwd_data_fast_kernel(const T* __restrict__ gy, const float* __restrict__ w, T* __restrict__ gx,
                                                                  int b0, int nb, int in, int out, float alpha, int rc, int rhw) {
    __shared__ float red[4][FB];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * (4 / WPR) + wv / WPR;
    const int sub = wv % WPR;
    float acc[FB];
#pragma unroll
    for (int b = 0; b < FB; ++b) acc[b] = 0.f;
    const float* wr = w + dense_row(i < in ? i : in - 1, rc, rhw) * out;
#pragma unroll 4
    for (int o = (sub * 64 + lane) * 4; o < out; o += 256 * WPR) {
        const float4 wv = *reinterpret_cast<const float4*>(wr + o);
#pragma unroll
        for (int b = 0; b < FB; ++b) {   // unconditional loads, see dense_fwd_fast_kernel
            float gv[4];
            ld4(gy + (long)(b0 + (b < nb ? b : nb - 1)) * out + o, gv);
            acc[b] += gv[0] * wv.x + gv[1] * wv.y + gv[2] * wv.z + gv[3] * wv.w;
        }
    }
    if constexpr (WPR == 1) {
        if (i >= in) return;
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            if (b < nb) {
                const float sum = wave_sum(acc[b]);
                if (lane == 0) DT<T>::st(gx + (long)(b0 + b) * in + i, sum * alpha);
            }
        }
    } else {
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            const float sum = wave_sum(acc[b]);
            if (lane == 0) red[wv][b] = sum;
        }
        __syncthreads();
        if (threadIdx.x < nb && i < in)
            DT<T>::st(gx + (long)(b0 + threadIdx.x) * in + i, (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) * alpha);
    }
}

// The same map for MANY SHORT ROWS (the discriminator's 8192 x 256 dense: 8192 one-KiB weight rows) on the exact-fp32 MFMA
// (v_mfma_f32_16x16x4_f32: an fmaf chain, bitwise): a wave per tile of 16 weight rows x 16 batch rows, K = out.  The wave-per-row
// form above spends its time in cross-lane sums -- 16 wave reductions of 6 shuffle steps per KiB of weights, 21 us for 8.4 MB --
// where the matrix core reduces in hardware.  Lane (n = lane & 15, kq = lane >> 4) feeds B[k][n] = w[row n][k] and A[m][k] = gy[m][k] with
// m = lane & 15; step (j, e) of the K loop uses k = 16 j + 4 kq + e, so that a lane reads 16 contiguous bytes of its weight row per j
// (the four lanes of a row 64 contiguous bytes) and 4 consecutive gradients.  D[m = 4 (lane >> 4) + r][n = lane & 15] = acc[r].
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <typename T>
__global__ __launch_bounds__(256) void dense_bwd_data_mfma_kernel(const T* __restrict__ gy, const float* __restrict__ w, T* __restrict__ gx,
                                                                  int b, int in, int out, float alpha, int rc, int rhw) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile * 16 >= in) return;
    const int n = lane & 15, kq = lane >> 4;
    const int row = tile * 16 + n;
    const float* wr = w + dense_row(row < in ? row : in - 1, rc, rhw) * out + 4 * kq;
    for (int m0 = 0; m0 < b; m0 += 16) {
        const int m = m0 + n;   // (this lane's A row; rows past b re-read the last one, their results are never written)
        const T* ar = gy + (long)(m < b ? m : b - 1) * out + 4 * kq;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int k = 0; k < out; k += 16) {
            const float4 bv = *reinterpret_cast<const float4*>(wr + k);
            float av[4];
            ld4(ar + k, av);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv.w, acc, 0, 0, 0);
        }
        if (row < in) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mm = m0 + 4 * kq + r;
                if (mm < b) DT<T>::st(gx + (long)mm * in + row, acc[r] * alpha);
            }
        }
    }
}

// Forward of the same tall layer (in = 8192, out = 256, 8 or 16 batch rows): y[m][c] = sum_k x[m][k] w[k][c] on the exact-fp32 MFMA, K split
// over waves (64 input rows each: 512 waves for the discriminator's dense), partial sums through `part[ks][b][out]` and the existing
// finalize pass (fixed order: deterministic).  A wave owns 64 columns as four interleaved 16-column tiles -- tile e = columns 4 n + e --
// so that a lane's B operands of the four tiles are ONE 16-byte load of weight row k (the 16 lanes of a kq group read 256 contiguous
// bytes); lane (n, kq) takes k = k0 + 16 kq + 4 t + u.
template <typename T>
__global__ __launch_bounds__(256) void dense_fwd_mfma_kernel(const T* __restrict__ x, const float* __restrict__ w, float* __restrict__ part,
                                                             int b, int in, int out, int rc, int rhw) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int ncg = out / 64;
    const int cg = wave % ncg, ks = wave / ncg;
    if (ks * 64 >= in) return;
    const int n = lane & 15, kq = lane >> 4;
    const int kb = ks * 64 + 16 * kq;   // this lane's 16 input rows
    const float* wc = w + cg * 64 + 4 * n;
    // all 16 weight loads of the lane go out before the first MFMA (a `rc ? mapped : plain` branch per row made hipcc wait for every load
    // by itself: 16 serial memory latencies, 12-15 us); the row map is walked incrementally -- rc_ = "rows per column block", 1 << 30 when
    // there is no map, so that row % rc_ = row and row / rc_ = 0
    const int rc_ = rc ? rc : (1 << 30), rhw_ = rc ? rhw : 1;
    int rem = kb % rc_, quo = kb / rc_;
    float4 bv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        bv[i] = *reinterpret_cast<const float4*>(wc + ((long)rem * rhw_ + quo) * out);
        ++rem;
        const bool wrap = rem == rc_;
        rem = wrap ? 0 : rem;
        quo += wrap ? 1 : 0;
    }
    for (int m0 = 0; m0 < b; m0 += 16) {
        const int m = m0 + n;
        const T* ar = x + (long)(m < b ? m : b - 1) * in + kb;
        float av[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) ld4(ar + 4 * t, av[t]);
        f32x4_t acc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float a = av[i >> 2][i & 3];
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[i].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[i].y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[i].z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[i].w, acc[3], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mm = m0 + 4 * kq + r;
            if (mm < b) *reinterpret_cast<float4*>(part + ((long)ks * b + mm) * out + cg * 64 + 4 * n) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
        }
    }
}

// gw[i][o] (+)= alpha * sum_b x[b][i] gy[b][o], out % 256 == 0, b <= DENSE_BT: a thread keeps its 4 columns of gy for every
// batch row in registers and walks DENSE_WR rows of the weight matrix (16-byte stores).
constexpr int DENSE_WR = 16;
template <typename T>
__global__ __launch_bounds__(256) void dense_bwd_weight_fast_kernel(const T* __restrict__ x, const T* __restrict__ gy, float* __restrict__ gw,
                                                                    int b, int in, int out, float alpha, int accumulate, int rc, int rhw) {
    __shared__ float xs[DENSE_BT][DENSE_WR];
    const int tid = threadIdx.x;
    const int ct = tid & 63, rl = tid >> 6;
    const int col = blockIdx.x * 256 + ct * 4;
    const int r0 = blockIdx.y * DENSE_WR;
    for (int k = tid; k < DENSE_BT * DENSE_WR; k += 256) {
        const int bb = k / DENSE_WR, rr = k % DENSE_WR;
        xs[bb][rr] = (bb < b && r0 + rr < in) ? DT<T>::ld(x + (long)bb * in + r0 + rr) * alpha : 0.f;
    }
    float g[DENSE_BT][4];
#pragma unroll
    for (int bb = 0; bb < DENSE_BT; ++bb) {   // unconditional loads; rows past b are multiplied by the zeros staged in xs
        ld4(gy + (long)(bb < b ? bb : b - 1) * out + col, g[bb]);
    }
    __syncthreads();
    for (int rr = rl; rr < DENSE_WR && r0 + rr < in; rr += 4) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int bb = 0; bb < DENSE_BT; ++bb) {
            const float xv = xs[bb][rr];
            o[0] += xv * g[bb][0]; o[1] += xv * g[bb][1]; o[2] += xv * g[bb][2]; o[3] += xv * g[bb][3];
        }
        float4* dst = reinterpret_cast<float4*>(gw + dense_row(r0 + rr, rc, rhw) * out + col);
        if (accumulate) {
            const float4 old = *dst;
            o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w;
        }
        *dst = make_float4(o[0], o[1], o[2], o[3]);
    }
}

static bool dense_fwd_fast_ok(int in, int out) { return in % 4 == 0 && out % 32 == 0; }
static bool dense_fwd_mfma_ok(int in, int out) {
    static const bool no_mfma = getenv("GS_NO_DENSE_MFMA") != nullptr;   // A/B switch
    return !no_mfma && in >= 2048 && in % 64 == 0 && out % 64 == 0 && out <= 1024;
}
static void dense_split(int in, int out, int* ksplit, int* ipb) {
    if (dense_fwd_mfma_ok(in, out)) {   // tall layer on the fp32 MFMA: 64 input rows per wave
        *ipb = 64;
        *ksplit = in / 64;
        return;
    }
    if (dense_fwd_fast_ok(in, out)) {   // (out/32) x ksplit blocks should cover the chip twice; a split handles >= 128 rows
        const int tiles = out / 32;
        int ks = tiles >= 128 ? 1 : (512 + tiles - 1) / tiles;   // >= 128 column tiles: no split, no finalize launch
        const int maxks = in / 128 > 0 ? in / 128 : 1;
        if (ks > maxks) ks = maxks;
        if (ks < 1) ks = 1;
        int per = (in + ks - 1) / ks;
        per = (per + 3) & ~3;
        *ipb = per;
        *ksplit = (in + per - 1) / per;
        return;
    }
    const int tiles = cdiv(out, 64);
    int ks = 1024 / tiles;
    if (ks < 1) ks = 1;
    int maxks = in / 32;
    if (maxks < 1) maxks = 1;
    if (ks > maxks) ks = maxks;
    *ipb = cdiv(in, ks);
    *ksplit = cdiv(in, *ipb);
}

// ------------------------------------------------------------------------------ embedding
template <typename T>
__global__ void embedding_fwd_kernel(const int64_t* __restrict__ idx, const float* __restrict__ w, T* __restrict__ y, int b, int rows, int units, float alpha) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= b * units) return;
    const int r = e / units, u = e % units;
    long k = idx[r];
    if (k < 0) k = 0;
    if (k >= rows) k = rows - 1;
    DT<T>::st(y + e, w[k * units + u] * alpha);
}
// the same gather with the row index taken from the one-hot input itself (ops.py:204-218 gathers by tf.argmax of it): block b scans its
// label row for the first maximum, writes it to idx_out[b] (the backward's scatter index) and gathers -- no separate argmax launch
template <typename T>
__global__ __launch_bounds__(256) void embedding_onehot_fwd_kernel(const T* __restrict__ labels, const float* __restrict__ w, T* __restrict__ y, int64_t* __restrict__ idx_out,
                                                                    int rows, int units, float alpha) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int b = blockIdx.x, t = threadIdx.x;
    float best = -INFINITY;
    int bi = rows;
    for (int k = t; k < rows; k += 256) {   // (ascending k per thread: the first maximum wins)
        const float v = DT<T>::ld(labels + (long)b * rows + k);
        if (v > best) { best = v; bi = k; }
    }
    sv[t] = best; si[t] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            const float v = sv[t + o];
            const int i = si[t + o];
            if (v > sv[t] || (v == sv[t] && i < si[t])) { sv[t] = v; si[t] = i; }
        }
        __syncthreads();
    }
    int k = si[0];
    if (k >= rows) k = 0;
    if (t == 0) idx_out[b] = k;
    for (int u = t; u < units; u += 256) DT<T>::st(y + (long)b * units + u, w[(long)k * units + u] * alpha);
}
// gw[row][u] = alpha * sum_{b: idx[b]==row} gy[b][u]  (gather form: deterministic, no atomics)
template <typename T>
__global__ void embedding_bwd_kernel(const int64_t* __restrict__ idx, const T* __restrict__ gy, float* __restrict__ gw, int b, int rows, int units, float alpha) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * units) return;
    const int r = e / units, u = e % units;
    float s = 0.f;
    for (int k = 0; k < b; ++k)
        if (idx[k] == r) s += DT<T>::ld(gy + (long)k * units + u);
    gw[e] = s * alpha;
}

// -------------------------------------------------------------------------- batch stddev
// x [b][hw][c] (channels-last), groups of 4: column j in [0, M=b/4) holds samples j, j+M, j+2M, j+3M.
// One block per column j (M = 2 at batch 8: a latency chain, not a bandwidth problem): 1024 threads, VN = 16 bytes of
// positions per thread and member (VN = 1: scalar fallback for odd sizes), so that the 4 x 8192 values of the model's 2x16x256
// block are one trip of four independent 16-byte loads per thread.
template <typename T, int VN> __device__ inline void bs_ld(const T* p, float* o) {
    if constexpr (VN == 1) o[0] = DT<T>::ld(p); else ld_wide<T>(p, o);
}
template <typename T, int VN> __device__ inline void bs_st(T* p, const float* o) {
    if constexpr (VN == 1) DT<T>::st(p, o[0]); else st_wide<T>(p, o);
}
template <typename T, int VN>
__device__ inline void bs_load(const T* x, int M, int j, long pos, long npos, float (&v)[4][VN]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bs_ld<T, VN>(x + ((long)(i * M + j)) * npos + pos, v[i]);
}
constexpr int BS_NT = 1024;

template <typename T, int VN>
__global__ __launch_bounds__(BS_NT) void batch_stddev_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int M, int hw, int c, float eps) {
    __shared__ float red[BS_NT / 64];
    const int j = blockIdx.x;
    const long npos = (long)hw * c;
    float s = 0.f;
    for (long pos = (long)threadIdx.x * VN; pos < npos; pos += (long)BS_NT * VN) {
        float v[4][VN];
        bs_load<T, VN>(x, M, j, pos, npos, v);
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const float mu = 0.25f * (v[0][e] + v[1][e] + v[2][e] + v[3][e]);
            float var = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) var += (v[i][e] - mu) * (v[i][e] - mu);
            s += sqrtf(0.25f * var + eps);
        }
    }
    s = block_sum<BS_NT>(s, red) / (float)npos;
    for (int k = threadIdx.x; k < 4 * hw; k += BS_NT) {
        const int i = k / hw, p = k % hw;
        DT<T>::st(y + (long)(i * M + j) * hw + p, s);
    }
}

// gx_i = gs_j * d_i / (4 sigma N) (+ addend_i),  gs_j = sum over members/pixels of gy
// addend (optional): the OTHER gradient into x -- x feeds the statistic and the conv beside it (networks.py:174-176), the sum of the two
// gradients is formed here instead of by one more pass
template <typename T, int VN>
__global__ __launch_bounds__(BS_NT) void batch_stddev_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ x, const T* __restrict__ addend,
                                                                 T* __restrict__ gx, int M, int hw, int c, float eps) {
    __shared__ float red[BS_NT / 64];
    const int j = blockIdx.x;
    const long npos = (long)hw * c;
    float g = 0.f;
    for (int k = threadIdx.x; k < 4 * hw; k += BS_NT) g += DT<T>::ld(gy + (long)((k / hw) * M + j) * hw + (k % hw));
    g = block_sum<BS_NT>(g, red);
    const float coef = g / (4.f * (float)npos);
    for (long pos = (long)threadIdx.x * VN; pos < npos; pos += (long)BS_NT * VN) {
        float v[4][VN], o[4][VN];
        bs_load<T, VN>(x, M, j, pos, npos, v);
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const float mu = 0.25f * (v[0][e] + v[1][e] + v[2][e] + v[3][e]);
            float var = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) var += (v[i][e] - mu) * (v[i][e] - mu);
            const float inv = coef / sqrtf(0.25f * var + eps);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i][e] = (v[i][e] - mu) * inv;
        }
        if (addend) {
            bs_load<T, VN>(addend, M, j, pos, npos, v);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < VN; ++e) o[i][e] += v[i][e];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) bs_st<T, VN>(gx + ((long)(i * M + j)) * npos + pos, o[i]);
    }
}

// F = sum_pos gs_j/(4N) * S/sigma,  S = sum_i ggx_i d_i
//   ggy (every member / pixel of column j) = sum_pos S / (4 sigma N)
//   gx2_k = gs_j/(4N) * [ (ggx_k - mean_i ggx_i)/sigma - S d_k / (4 sigma^3) ]
template <typename T, int VN>
__global__ __launch_bounds__(BS_NT) void batch_stddev_bwd_bwd_kernel(const T* __restrict__ ggx, const T* __restrict__ gy, const T* __restrict__ x,
                                                                     T* __restrict__ ggy, T* __restrict__ gx2, int M, int hw, int c, float eps) {
    __shared__ float red[BS_NT / 64];
    const int j = blockIdx.x;
    const long npos = (long)hw * c;
    float g = 0.f;
    for (int k = threadIdx.x; k < 4 * hw; k += BS_NT) g += DT<T>::ld(gy + (long)((k / hw) * M + j) * hw + (k % hw));
    g = block_sum<BS_NT>(g, red);
    const float coef = g / (4.f * (float)npos);
    float t = 0.f;
    for (long pos = (long)threadIdx.x * VN; pos < npos; pos += (long)BS_NT * VN) {
        float v[4][VN], gg[4][VN], o[4][VN];
        bs_load<T, VN>(x, M, j, pos, npos, v);
        bs_load<T, VN>(ggx, M, j, pos, npos, gg);
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const float mu = 0.25f * (v[0][e] + v[1][e] + v[2][e] + v[3][e]);
            const float gm = 0.25f * (gg[0][e] + gg[1][e] + gg[2][e] + gg[3][e]);
            float var = 0.f, S = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { var += (v[i][e] - mu) * (v[i][e] - mu); S += gg[i][e] * (v[i][e] - mu); }
            const float sig = sqrtf(0.25f * var + eps);
            t += S / sig;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i][e] = coef * ((gg[i][e] - gm) / sig - S * (v[i][e] - mu) / (4.f * sig * sig * sig));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) bs_st<T, VN>(gx2 + ((long)(i * M + j)) * npos + pos, o[i]);
    }
    t = block_sum<BS_NT>(t, red) / (4.f * (float)npos);
    for (int k = threadIdx.x; k < 4 * hw; k += BS_NT) DT<T>::st(ggy + (long)((k / hw) * M + j) * hw + (k % hw), t);
}

// ------------------------------------------------------------------------------ GAN losses
// The loss algebra of models.py:39-65 on [N] / [N, C] tensors was ~50 torch launches per iteration (casts, products, sums,
// softplus, mean and their autograd mirror images); here each loss is ONE single-block launch that also writes the gradients of
// the mean w.r.t. its inputs (the backward of the autograd node then only scales them by the incoming scalar).
__device__ inline float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }       // tf.nn.softplus
__device__ inline float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// L_D = mean_i [ softplus(-r_i) + softplus(f_i) + pen_i ],  r_i / f_i = sum_c logits[i][c] * labels[i][c] (one-hot labels:
// tf.gather_nd(logits, tf.where(labels)), models.py:39-40).  g_real = d L_D / d real_logits etc.
template <typename T>
__global__ __launch_bounds__(256) void gan_d_loss_kernel(const T* __restrict__ rl, const T* __restrict__ fl, const T* __restrict__ lab,
                                                         const float* __restrict__ pen, float pen_w, int n, int c, float* __restrict__ loss,
                                                         T* __restrict__ g_real, T* __restrict__ g_fake, float* __restrict__ g_pen) {
    __shared__ float red[4];
    __shared__ float sr[1024], sf[1024];
    const float inv_n = 1.f / (float)n;
    float part = 0.f;
    // a WAVE per sample, its lanes over the classes (a thread per sample walked the 61 classes as one chain of dependent loads: 8-11 us for
    // 16 samples)
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x >> 6; i < n; i += 4) {
        float r = 0.f, f = 0.f;
        for (int k = lane; k < c; k += 64) {
            const float l = DT<T>::ld(lab + (long)i * c + k);
            if (rl) r += DT<T>::ld(rl + (long)i * c + k) * l;
            if (fl) f += DT<T>::ld(fl + (long)i * c + k) * l;
        }
        r = wave_sum(r);
        f = wave_sum(f);
        if (lane == 0) {
            // (either half of the sum may be absent: the two halves of a discriminator run that keeps its real and its fake pass on two
            //  streams are two launches, each with its own partial mean -- the gradients are the same numbers either way)
            part += (rl ? softplus_f(-r) : 0.f) + (fl ? softplus_f(f) : 0.f) + (pen ? pen_w * pen[i] : 0.f);
            if (g_pen) g_pen[i] = pen_w * inv_n;
            sr[i] = -sigmoid_f(-r) * inv_n;   // d mean / d r_i
            sf[i] = sigmoid_f(f) * inv_n;     // d mean / d f_i
        }
    }
    const float tot = block_sum<256>(part, red);
    if (threadIdx.x == 0) loss[0] = tot * inv_n;
    __syncthreads();
    for (int e = threadIdx.x; e < n * c; e += 256) {
        const float l = DT<T>::ld(lab + e);
        if (rl) DT<T>::st(g_real + e, sr[e / c] * l);
        if (fl) DT<T>::st(g_fake + e, sf[e / c] * l);
    }
}

// L_G = mean_i [ softplus(-f_i) + w / (s_i + eps) ],  s_i = sum((d sum(G(z)) / d z_i)^2) (models.py:57-64); g_s = d L_G / d s.
template <typename T>
__global__ __launch_bounds__(256) void gan_g_loss_kernel(const T* __restrict__ fl, const T* __restrict__ lab, const float* __restrict__ ssq, float w,
                                                         float eps, int n, int c, float* __restrict__ loss, T* __restrict__ g_fake,
                                                         float* __restrict__ g_ssq) {
    __shared__ float red[4];
    __shared__ float sf[1024];
    const float inv_n = 1.f / (float)n;
    float part = 0.f;
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x >> 6; i < n; i += 4) {   // a wave per sample (see gan_d_loss_kernel)
        float f = 0.f;
        if (fl) for (int k = lane; k < c; k += 64) f += DT<T>::ld(fl + (long)i * c + k) * DT<T>::ld(lab + (long)i * c + k);
        f = wave_sum(f);
        if (lane == 0) {
            if (fl) part += softplus_f(-f);   // (absent: the mode-seeking half alone, see gan_d_loss_kernel)
            sf[i] = -sigmoid_f(-f) * inv_n;
            if (ssq) {
                const float d = ssq[i] + eps;
                part += w / d;
                g_ssq[i] = -w / (d * d) * inv_n;
            }
        }
    }
    const float tot = block_sum<256>(part, red);
    if (threadIdx.x == 0) loss[0] = tot * inv_n;
    __syncthreads();
    if (fl) for (int e = threadIdx.x; e < n * c; e += 256) DT<T>::st(g_fake + e, sf[e / c] * DT<T>::ld(lab + e));
}

}  // namespace gs

using namespace gs;

extern "C" int gs_gan_d_loss(const void* real_logits, const void* fake_logits, const void* labels, const float* penalty, float penalty_weight, int n, int c,
                             float* loss, void* g_real, void* g_fake, float* g_penalty, int dtype, void* stream) {
    GS_CHECK_ARG(n > 0 && n <= 1024 && c > 0 && (real_logits || fake_logits) && labels && loss && (!real_logits || g_real) && (!fake_logits || g_fake),
                 "gan_d_loss: bad args (batch %d <= 1024)", n);
    GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gan_d_loss_kernel<T>), dim3(1), dim3(256), 0, as_stream(stream), (const T*)real_logits, (const T*)fake_logits,
                                                (const T*)labels, penalty, penalty_weight, n, c, loss, (T*)g_real, (T*)g_fake, penalty ? g_penalty : nullptr));
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_gan_g_loss(const void* fake_logits, const void* labels, const float* sumsq, float weight, float eps, int n, int c, float* loss,
                             void* g_fake, float* g_sumsq, int dtype, void* stream) {
    GS_CHECK_ARG(n > 0 && n <= 1024 && c > 0 && (fake_logits || sumsq) && (!fake_logits || (labels && g_fake)) && loss && (!sumsq || g_sumsq),
                 "gan_g_loss: bad args (batch %d <= 1024)", n);
    GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gan_g_loss_kernel<T>), dim3(1), dim3(256), 0, as_stream(stream), (const T*)fake_logits, (const T*)labels, sumsq,
                                                weight, eps, n, c, loss, (T*)g_fake, g_sumsq));
    GS_CHECK_LAUNCH();
    return 0;
}


extern "C" size_t gs_dense_fwd_workspace_bytes(int b, int in, int out) {
    int ks, ipb;
    dense_split(in, out, &ks, &ipb);
    return (size_t)ks * b * out * sizeof(float);
}

static int dense_fwd_impl(const void* x, const float* w, void* y, int b, int in, int out, float alpha, int dtype,
                          void* ws, size_t ws_bytes, void* stream, int rc, int rhw, const float* bias = nullptr, int act = 0) {
    GS_CHECK_ARG(act >= 0 && act <= 2, "dense_fwd: bad activation %d", act);
    GS_CHECK_ARG(b > 0 && in > 0 && out > 0, "dense_fwd: bad args");
    GS_CHECK_ARG(rc == 0 || (rc > 0 && rhw > 0 && rc * rhw == in), "dense_fwd: a %d x %d channels-last input is not %d wide", rc, rhw, in);
    if (rc && !dense_fwd_fast_ok(in, out)) return fail(GS_ERR_UNSUPPORTED, "dense_fwd: channels-last input needs in %% 4 == 0 and out %% 32 == 0");
    int ks, ipb;
    dense_split(in, out, &ks, &ipb);
    if (ws_bytes < (size_t)ks * b * out * sizeof(float)) return fail(GS_ERR_WORKSPACE, "dense_fwd: workspace too small");
    hipStream_t st = as_stream(stream);
    float* part = (float*)ws;
    if (!rc && in <= 1024 && out <= 256 && b <= 64) {   // small layer: direct, one launch
        GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dense_fwd_small_kernel<T>), dim3(cdiv(out, 64), b), dim3(256), 0, st, (const T*)x, w, (T*)y, in, out, alpha, bias, act));
        GS_CHECK_LAUNCH();
        return 0;
    }
    if (dense_fwd_mfma_ok(in, out)) {
        GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dense_fwd_mfma_kernel<T>), dim3(cdiv((out / 64) * ks, 4)), dim3(256), 0, st, (const T*)x, w, part, b, in, out, rc, rhw));
        GS_CHECK_LAUNCH();
        const long n = (long)b * out;
        GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dense_finalize_kernel<T>), dim3(cdiv(n, 64)), dim3(256), 0, st, part, (T*)y, n, ks, alpha, bias, act, out));
        GS_CHECK_LAUNCH();
        return 0;
    }
    const bool fast = dense_fwd_fast_ok(in, out);
    const int step = fast ? fast_rows(b) : DENSE_BT;
    for (int b0 = 0; b0 < b; b0 += step) {
        const int nb = b - b0 < step ? b - b0 : step;
        if (fast) {
#define GS_DFF(FBV) GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dense_fwd_fast_kernel<T, FBV>), dim3(out / 32, ks), dim3(256), 0, st, (const T*)x, w, \
                                                                ks > 1 ? part : nullptr, (T*)y, b0, nb, in, out, b, ipb, alpha, rc, rhw, bias, act))
            if (step == 8) GS_DFF(8); else if (step == 16) GS_DFF(16); else GS_DFF(24);
#undef GS_DFF
        } else {
            GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dense_fwd_kernel<T>), dim3(cdiv(out, 64), ks), dim3(256), 0, st, (const T*)x, w, part, b0, nb, in, out, b, ipb));
        }
        GS_CHECK_LAUNCH();
    }
    if (fast && ks == 1) return 0;
    const long n = (long)b * out;
    GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dense_finalize_kernel<T>), dim3(cdiv(n, 64)), dim3(256), 0, st, part, (T*)y, n, ks, alpha, bias, act, out));
    GS_CHECK_LAUNCH();
    return 0;
}
extern "C" int gs_dense_fwd_bias_act(const void* x, const float* w, const float* bias, void* y, int b, int in, int out, float alpha, int act, int dtype,
                                     void* ws, size_t ws_bytes, void* stream) {
    return dense_fwd_impl(x, w, y, b, in, out, alpha, dtype, ws, ws_bytes, stream, 0, 0, bias, act);
}
extern "C" int gs_dense_fwd_bias_act_nhwc(const void* x, const float* w, const float* bias, void* y, int b, int c, int hw, int out, float alpha, int act, int dtype,
                                          void* ws, size_t ws_bytes, void* stream) {
    return dense_fwd_impl(x, w, y, b, c * hw, out, alpha, dtype, ws, ws_bytes, stream, c, hw, bias, act);
}
extern "C" int gs_dense_fwd(const void* x, const float* w, void* y, int b, int in, int out, float alpha, int dtype,
                            void* ws, size_t ws_bytes, void* stream) {
    return dense_fwd_impl(x, w, y, b, in, out, alpha, dtype, ws, ws_bytes, stream, 0, 0);
}
extern "C" int gs_dense_fwd_nhwc(const void* x, const float* w, void* y, int b, int c, int hw, int out, float alpha, int dtype,
                                 void* ws, size_t ws_bytes, void* stream) {
    return dense_fwd_impl(x, w, y, b, c * hw, out, alpha, dtype, ws, ws_bytes, stream, c, hw);
}

static int dense_bwd_data_impl(const void* gy, const float* w, void* gx, int b, int in, int out, float alpha, int dtype, void* stream, int rc, int rhw) {
    GS_CHECK_ARG(b > 0 && in > 0 && out > 0, "dense_bwd_data: bad args");
    GS_CHECK_ARG(rc == 0 || (rc > 0 && rhw > 0 && rc * rhw == in), "dense_bwd_data: a %d x %d channels-last input is not %d wide", rc, rhw, in);
    if (rc && out % 256 != 0) return fail(GS_ERR_UNSUPPORTED, "dense_bwd_data: channels-last input needs out %% 256 == 0");
    hipStream_t st = as_stream(stream);
    static const bool no_mfma = getenv("GS_NO_DENSE_MFMA") != nullptr;   // A/B switch
    if (!no_mfma && in >= 1024 && out % 16 == 0 && out <= 4096) {   // many short rows: a wave per 16 weight rows on the fp32 MFMA
        GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dense_bwd_data_mfma_kernel<T>), dim3(cdiv(cdiv(in, 16), 4)), dim3(256), 0, st, (const T*)gy, w, (T*)gx,
                                                    b, in, out, alpha, rc, rhw));
        GS_CHECK_LAUNCH();
        return 0;
    }
    const int step = out % 256 == 0 ? fast_rows(b) : DENSE_BT;
    for (int b0 = 0; b0 < b; b0 += step) {
        const int nb = b - b0 < step ? b - b0 : step;
        if (out % 256 == 0) {
#define GS_DBF(WPRV, FBV, GRID) GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dense_bwd_data_fast_kernel<T, WPRV, FBV>), dim3(GRID), dim3(256), 0, st, (const T*)gy, w, (T*)gx, b0, nb, in, out, alpha, rc, rhw))
            if (in <= 1024 && out >= 1024) {   // few long rows: the whole block on one row
                if (step == 8) GS_DBF(4, 8, in); else if (step == 16) GS_DBF(4, 16, in); else GS_DBF(4, 24, in);
            } else {
                if (step == 8) GS_DBF(1, 8, cdiv(in, 4)); else if (step == 16) GS_DBF(1, 16, cdiv(in, 4)); else GS_DBF(1, 24, cdiv(in, 4));
            }
#undef GS_DBF
        } else if (out >= 2048) {
            GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dense_bwd_data_wide_kernel<T>), dim3(in), dim3(256), 0, st, (const T*)gy, w, (T*)gx, b0, nb, in, out, alpha));
        } else {
            GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dense_bwd_data_kernel<T>), dim3(cdiv(in, 4)), dim3(256), 0, st, (const T*)gy, w, (T*)gx, b0, nb, in, out, alpha));
        }
        GS_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int gs_dense_bwd_data(const void* gy, const float* w, void* gx, int b, int in, int out, float alpha, int dtype, void* stream) {
    return dense_bwd_data_impl(gy, w, gx, b, in, out, alpha, dtype, stream, 0, 0);
}
extern "C" int gs_dense_bwd_data_nhwc(const void* gy, const float* w, void* gx, int b, int c, int hw, int out, float alpha, int dtype, void* stream) {
    return dense_bwd_data_impl(gy, w, gx, b, c * hw, out, alpha, dtype, stream, c, hw);
}

static int dense_bwd_weight_impl(const void* x, const void* gy, float* gw, int b, int in, int out, float alpha, int accumulate, int dtype, void* stream, int rc, int rhw) {
    GS_CHECK_ARG(b > 0 && in > 0 && out > 0, "dense_bwd_weight: bad args");
    GS_CHECK_ARG(rc == 0 || (rc > 0 && rhw > 0 && rc * rhw == in), "dense_bwd_weight: a %d x %d channels-last input is not %d wide", rc, rhw, in);
    if (rc && !(out % 256 == 0 && b <= DENSE_BT)) return fail(GS_ERR_UNSUPPORTED, "dense_bwd_weight: channels-last input needs out %% 256 == 0 and batch <= %d", DENSE_BT);
    const long n = (long)in * out;
    if (out % 256 == 0 && b <= DENSE_BT) {
        GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dense_bwd_weight_fast_kernel<T>), dim3(out / 256, cdiv(in, DENSE_WR)), dim3(256), 0, as_stream(stream),
                                                    (const T*)x, (const T*)gy, gw, b, in, out, alpha, accumulate, rc, rhw));
        GS_CHECK_LAUNCH();
        return 0;
    }
    GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dense_bwd_weight_kernel<T>), dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), (const T*)x, (const T*)gy, gw, b, in, out, alpha, accumulate));
    GS_CHECK_LAUNCH();
    return 0;
}
extern "C" int gs_dense_bwd_weight(const void* x, const void* gy, float* gw, int b, int in, int out, float alpha, int accumulate, int dtype, void* stream) {
    return dense_bwd_weight_impl(x, gy, gw, b, in, out, alpha, accumulate, dtype, stream, 0, 0);
}
extern "C" int gs_dense_bwd_weight_nhwc(const void* x, const void* gy, float* gw, int b, int c, int hw, int out, float alpha, int accumulate, int dtype, void* stream) {
    return dense_bwd_weight_impl(x, gy, gw, b, c * hw, out, alpha, accumulate, dtype, stream, c, hw);
}

extern "C" int gs_embedding_fwd(const int64_t* idx, const float* w, void* y, int b, int rows, int units, float alpha, int dtype, void* stream) {
    GS_CHECK_ARG(b > 0 && rows > 0 && units > 0, "embedding_fwd: bad args");
    GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((embedding_fwd_kernel<T>), dim3(cdiv((long)b * units, 256)), dim3(256), 0, as_stream(stream), idx, w, (T*)y, b, rows, units, alpha));
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_embedding_onehot_fwd(const void* labels, const float* w, void* y, int64_t* idx_out, int b, int rows, int units, float alpha, int dtype, void* stream) {
    GS_CHECK_ARG(b > 0 && rows > 0 && units > 0 && labels && w && y && idx_out, "embedding_onehot_fwd: bad args");
    GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((embedding_onehot_fwd_kernel<T>), dim3(b), dim3(256), 0, as_stream(stream), (const T*)labels, w, (T*)y, idx_out, rows, units, alpha));
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_embedding_bwd(const int64_t* idx, const void* gy, float* gw, int b, int rows, int units, float alpha, int dtype, void* stream) {
    GS_CHECK_ARG(b > 0 && rows > 0 && units > 0, "embedding_bwd: bad args");
    GS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((embedding_bwd_kernel<T>), dim3(cdiv((long)rows * units, 256)), dim3(256), 0, as_stream(stream), idx, (const T*)gy, gw, b, rows, units, alpha));
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_batch_stddev_fwd(const void* x, void* y, int b, int hw, int c, float eps, int dtype, void* stream) {
    GS_CHECK_ARG(b > 0 && b % 4 == 0 && hw > 0 && c > 0, "batch_stddev: batch %d must be a positive multiple of 4 (ops.py:341)", b);
    GS_DISPATCH_DTYPE(dtype, {
        if (((long)hw * c) % Wide<T>::N == 0) hipLaunchKernelGGL((batch_stddev_fwd_kernel<T, Wide<T>::N>), dim3(b / 4), dim3(BS_NT), 0, as_stream(stream), (const T*)x, (T*)y, b / 4, hw, c, eps);
        else hipLaunchKernelGGL((batch_stddev_fwd_kernel<T, 1>), dim3(b / 4), dim3(BS_NT), 0, as_stream(stream), (const T*)x, (T*)y, b / 4, hw, c, eps);
    });
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_batch_stddev_bwd(const void* gy, const void* x, const void* addend, void* gx, int b, int hw, int c, float eps, int dtype, void* stream) {
    GS_CHECK_ARG(b > 0 && b % 4 == 0 && hw > 0 && c > 0, "batch_stddev_bwd: bad args");
    GS_DISPATCH_DTYPE(dtype, {
        if (((long)hw * c) % Wide<T>::N == 0) hipLaunchKernelGGL((batch_stddev_bwd_kernel<T, Wide<T>::N>), dim3(b / 4), dim3(BS_NT), 0, as_stream(stream), (const T*)gy, (const T*)x, (const T*)addend, (T*)gx, b / 4, hw, c, eps);
        else hipLaunchKernelGGL((batch_stddev_bwd_kernel<T, 1>), dim3(b / 4), dim3(BS_NT), 0, as_stream(stream), (const T*)gy, (const T*)x, (const T*)addend, (T*)gx, b / 4, hw, c, eps);
    });
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_batch_stddev_bwd_bwd(const void* ggx, const void* gy, const void* x, void* ggy, void* gx2, int b, int hw, int c, float eps, int dtype, void* stream) {
    GS_CHECK_ARG(b > 0 && b % 4 == 0 && hw > 0 && c > 0, "batch_stddev_bwd_bwd: bad args");
    GS_DISPATCH_DTYPE(dtype, {
        if (((long)hw * c) % Wide<T>::N == 0) hipLaunchKernelGGL((batch_stddev_bwd_bwd_kernel<T, Wide<T>::N>), dim3(b / 4), dim3(BS_NT), 0, as_stream(stream), (const T*)ggx, (const T*)gy, (const T*)x, (T*)ggy, (T*)gx2, b / 4, hw, c, eps);
        else hipLaunchKernelGGL((batch_stddev_bwd_bwd_kernel<T, 1>), dim3(b / 4), dim3(BS_NT), 0, as_stream(stream), (const T*)ggx, (const T*)gy, (const T*)x, (T*)ggy, (T*)gx2, b / 4, hw, c, eps);
    });
    GS_CHECK_LAUNCH();
    return 0;
}
