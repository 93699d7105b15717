// Device plan of the spectral front end (immutable after gs_spectral_plan_create), shared by spectral.hip (generic kernels,
// inverse path) and spectral_wave.hip (the wave-per-frame forward path for 2048-sample frames).
#pragma once
#include "gs_common.h"

struct gs_spectral_plan {
    int frame_length, frame_step, time_steps, nbins, log2h, maxnz;
    float* hann;        // [frame_length]
    float2* tw;         // [nbins/2]  exp(-2 pi i k / nbins)
    float2* twp;        // [nbins+1]  exp(-2 pi i k / frame_length)
    int* mel_idx;       // [maxnz][nbins] ELL by mel column, entry-major (lanes = consecutive mel bins read consecutive words)
    float* mel_val;     // [maxnz][nbins]
    float* pinv;        // [nbins][nbins] or nullptr
    unsigned short* pinv_split;   // [3][nbins (n)][nbins (k)] bf16: pinv = plane 0 + plane 1 + plane 2 exactly (gemm_bf16x6_kernel), or nullptr
    float* inv_window;  // [frame_length]
    // wave-per-frame path (nbins == 1024, every mel column's non-zeros form one run, run lengths per block as the kernel expects), else fast == 0
    int fast;
    float2* tw1k;       // [1024] exp(-2 pi i k / 1024)
    int* mel_lo;        // [nbins] first linear bin of the column's run (clamped so that lo + cnt <= nbins)
    float* mel_w;       // per 128-column block j: [cnt[j]][128] weights of bins lo[m] + e, columns in natural order
    int mel_cnt[8];     // run length used for block j (max over its 128 columns)
    int mel_off[8];     // float offset of block j in mel_w
    int mel_wtot;       // floats in mel_w
};

namespace gs {
// spectral_wave.hip
bool stft_wave_shape_ok(const int* mel_cnt);
size_t stft_wave_edge_bytes(const gs_spectral_plan* p, int batch);
int launch_stft_wave_fused(const gs_spectral_plan* p, const float* wave, int batch, int wave_len, int front_pad, void* images, int dtype,
                           void* ws, size_t ws_bytes, hipStream_t st);
int launch_istft_wave(const gs_spectral_plan* p, const float* mag, const float* phase, float* frames, long nframes, hipStream_t st);
bool istft_wave_ola_ok(const gs_spectral_plan* p, int wave_len, int front_pad);
int launch_istft_wave_ola(const gs_spectral_plan* p, const float* mag, const float* phase, float* wave, int batch, int wave_len, int front_pad, hipStream_t st);
int launch_stft_wave_magphase(const gs_spectral_plan* p, const float* wave, int batch, int wave_len, int front_pad, float* mag, float* phase,
                              hipStream_t st);
}  // namespace gs
