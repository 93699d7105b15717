/* libgansynth_hip.so -- C ABI of the MI355X (gfx950) GANSynth hot path.
 *
 * The reference (skmhrk1209/GANSynth) has no FFI layer: its boundary is the Python call
 * surface of ops.py / networks.py / spectral_ops.py / models.py, whose arithmetic runs inside
 * TensorFlow kernels.  Each entry point below replaces the TF kernel(s) behind one of those
 * reference call sites (cited as file:line relative to the reference tree); the Python host in
 * gansynth_amd/ re-exposes the reference's own function names on top of them.
 *
 * Conventions
 *  - every function returns 0 on success, a negative GS_ERR_* code otherwise;
 *    gs_last_error() returns a thread-local message for the last failure;
 *  - all pointers are DEVICE pointers owned by the caller (torch); the library never allocates,
 *    frees or synchronises on the hot path and is hipGraph-capture safe; scratch space is passed
 *    in as `ws` and sized by the matching *_workspace_bytes query;
 *  - activations are channels-last: a logical NCHW tensor [n,c,h,w] is stored [n][h][w][c]
 *    ("NHWC"); 2-D tensors are [rows][cols] row-major.  `dtype` is the storage type of
 *    activations (GS_F32 or GS_BF16); accumulation is always fp32;
 *  - parameters (weights, biases), their gradients and optimizer state are always fp32 in the
 *    reference's own layouts: conv HWIO [kh][kw][Cin][Cout], dense [in][out];
 *  - `alpha` is the equalized-learning-rate runtime scale sqrt(variance_scale / fan_in) of
 *    ops.py:154-160, applied inside the kernel;
 *  - `stream` is a hipStream_t passed as void*.
 */
#ifndef GANSYNTH_HIP_H
#define GANSYNTH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { GS_F32 = 0, GS_BF16 = 1 };
enum { GS_OK = 0, GS_ERR_ARG = -1, GS_ERR_HIP = -2, GS_ERR_UNSUPPORTED = -3, GS_ERR_WORKSPACE = -4 };
enum { GS_ACT_NONE = 0, GS_ACT_LRELU = 1, GS_ACT_TANH = 2 };
/* 1-bit leaky-relu masks (bf16 activations whose channel count is a multiple of 32).  A forward conv called with `act = GS_ACT_LRELU |
 * GS_ACT_WRITE_BITS` also leaves the SIGN BITS of its result behind it: the caller's buffer holds the activation (numel values) followed by
 * numel / 8 bytes, one 32-bit word per (pixel, 32-channel tile), bit 8 (2 h + q) + k = "channel 16 q + 8 h + k of the tile is > 0".  A masked
 * conv (gs_conv2d_fwd_mask / gs_conv2d_bwd_data_mask) called with `mask_act = GS_ACT_LRELU_BITS` is promised such a buffer as `mask` and reads
 * the words instead of the values where its epilogue can (1 / 16 of the mask bytes); everywhere else it reads the values as with GS_ACT_LRELU.
 * gs_pack_act_bits writes the words for an activation that some other kernel produced. */
enum { GS_ACT_LRELU_BITS = 5, GS_ACT_WRITE_BITS = 16 };
/* which of the three bilinear conv maps a workspace query is for */
enum { GS_CONV_FWD = 0, GS_CONV_BWD_DATA = 1, GS_CONV_BWD_WEIGHT = 2 };

const char* gs_last_error(void);
int gs_version(void);
/* number of CUs etc. are queried lazily; this forces it (and checks the device is gfx950) */
int gs_init(void);
/* n plain (normal priority, non-blocking) streams, made and destroyed around the instantiation of a hipGraph WITH PARALLEL BRANCHES.
 * The runtime hands every new stream the least-used of its GPU_MAX_HW_QUEUES hardware queues; ROCm 7.0.2's hip::GraphExec makes one
 * stream more than the branches need and hip::Graph::UpdateStreams, at every launch, skips those that share the LAUNCH stream's hardware
 * queue -- without a bounds check: two of them on that queue and hipGraphLaunch reads past the end of the list (segmentation fault,
 * profiles/r05_e_graph_replay_crash.txt).  Two consecutive new streams only land on one queue when it is at least two users short of
 * every other one; a hundred-odd throw-away streams level the pool first (each goes to the least-used queue), so the exec's streams
 * land on different queues and at most one is skipped.  The reference has no such object (one TF session, models.py:189-194). */
int gs_streams_create(int n, void** streams);
int gs_streams_destroy(int n, void** streams);

/* ---------------------------------------------------------------- profiling hooks (bench.py)
 * When enabled, every launch of conv_igemm_kernel (the MFMA implicit-GEMM conv) is bracketed by a pair of
 * HIP events on its own stream.  gs_prof_collect synchronises those events and returns the number
 * of launches, their summed duration (ms) and summed algorithmic FLOPs.
 * gs_prof_enable(n) with n > 1 = burst mode: a conv launch (a pure function of its inputs) is issued n times back to back inside
 * its event pair and the elapsed time divided by n -- the steady-state launch-to-launch time (one launch boundary included), free
 * of the host's eager-launch latency that an event pair around a single few-microsecond launch also measures. */
int gs_prof_enable(int on);
int gs_prof_collect(int* launches, double* total_ms, double* total_flops);
/* roofline accounting of the launches recorded since gs_prof_enable(1): algorithmic bytes (every operand read once, the result
 * written once) and the time the binding roof allows, summed per launch (max of flops / peak_tflops and bytes / peak_gbps);
 * roof_ms_hbm_bound = the part of it that comes from HBM-bound launches.  Call before gs_prof_collect (which resets). */
int gs_prof_roofline(double peak_tflops, double peak_gbps, double* total_bytes, double* roof_ms, double* roof_ms_hbm_bound);
/* per-launch records (implicit-GEMM convs and their weight gradients) for a per-stage roofline: duration, algorithmic FLOPs and
 * bytes, and desc[8 * i .. +8] = {kind, N, Hb, Wb, IC, OC, masked | sources, fused norm | deferred}; kind = 0 / 1 / 2 for the
 * stride-1 / stride-2 / transposed conv map (ops.py:237-243, 269-276), 10 + that for its weight gradient.  Before gs_prof_collect. */
int gs_prof_records(int max_records, int* n, double* ms, double* flops, double* bytes, int* desc);

/* ------------------------------------------------------------------------ data parallelism (new: the reference is single GPU,
 * gan_synth_main.py:91-98).  One process per GPU.  A communicator wraps ncclCommInitRank of RCCL (resolved at run
 * time from the librccl.so.1 the process already has, none needed on one GPU); the collectives run ON THE CALLER'S STREAM, i.e.
 * ordered behind the backward that produced the gradients and ahead of gs_adam_tf_step, with no cross-stream event.
 *   gs_comm_available   0 when librccl can be resolved in this process (no communicator is created): lets every rank agree that the
 *                       blocking gs_comm_init will be entered by ALL of them before any of them enters it
 *   gs_comm_unique_id   rank 0 fills 128 bytes (ncclGetUniqueId) and ships them to the other ranks by any means
 *   gs_comm_init        every rank, same id; binds to the current HIP device
 *   gs_allreduce_sum_f32 / gs_broadcast_f32   in place, fp32 (the flat gradient / parameter buffers of models.py:67-89's two
 *                       optimizers; the 1 / world averaging is gs_adam_tf_step's grad_scale) */
#define GS_COMM_ID_BYTES 128
typedef struct gs_comm gs_comm;
int gs_comm_available(void);
int gs_comm_unique_id(void* id128);
int gs_comm_init(gs_comm** out, int rank, int world, const void* id128);
int gs_comm_destroy(gs_comm* comm);
int gs_comm_count(gs_comm* comm, int* ranks);   /* ncclCommCount: the ranks the communicator was built over (bench.py reports it as rccl_ranks) */
int gs_allreduce_sum_f32(gs_comm* comm, float* data, int64_t count, void* stream);
/* test hook, ONE-rank communicators only (RCCL short-cuts their all-reduce to nothing): from now on gs_allreduce_sum_f32 launches a one-block
 * kernel that holds the stream for `us` microseconds instead (us < 0: off again) -- where a collective sits in a captured graph then shows as time */
int gs_comm_set_marker_us(gs_comm* comm, double us);
int gs_broadcast_f32(gs_comm* comm, float* data, int64_t count, int root, void* stream);

/* ------------------------------------------------------------------------------- conv2d
 * tf.nn.conv2d NCHW/HWIO padding=SAME (ops.py:237-243) with ksize in {1,3}, stride in {1,2}
 * (stride 2 only with ksize 3; TF SAME on an even input pads 0 before / 1 after).
 *   x [n][h][w][ci]  w [k][k][ci][co] fp32  y [n][h/stride][w/stride][co]
 *   y = alpha * conv(x, w)                       (bias / activation are separate entry points)
 * The three maps are closed under differentiation (each one's gradient is another one), which is
 * how the Python host gets the second-order terms of models.py:47,60.
 *   bwd_data  : gx[n][h][w][ci] = alpha * d<gy, conv(x,w)>/dx
 *   bwd_weight: gw[k][k][ci][co] = alpha * d<gy, conv(x,w)>/dw   (fp32 out)
 * (n,h,w) are always the dims of x (the conv INPUT side), for all three.
 * fwd / bwd_data first re-lay the weight into the kernel operand at the start of `ws`; `w_prepared` != 0 says that
 * `ws` still holds that operand from an earlier call with the same weight values (same map, dtype), so the
 * re-layout is skipped -- the caller keeps one persistent ws per (weight, map) between optimizer steps.
 * Every gradient-of-a-parameter entry point (bwd_weight, dense_bwd_weight, channel_sum, act_bwd_bias) takes
 * `accumulate`: 0 overwrites the output, 1 adds into it (tf.gradients sums the contributions of a variable used
 * several times; accumulating in the producing kernel replaces one read-modify-write pass per contribution). */
size_t gs_conv2d_workspace_bytes(int which, int n, int h, int w, int ci, int co, int ksize, int stride, int dtype);
int gs_conv2d_fwd(const void* x, const float* w_hwio, void* y, int n, int h, int w, int ci, int co,
                  int ksize, int stride, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream);
/* same with the bias add + activation the reference applies right after (ops.py:244-246 + tf.nn.leaky_relu /
 * tf.nn.tanh in networks.py) fused into the GEMM epilogue: y = act(alpha * conv(x, w) + bias); bias may be NULL */
int gs_conv2d_fwd_bias_act(const void* x, const float* w_hwio, const float* bias, void* y, int n, int h, int w, int ci, int co,
                           int ksize, int stride, float alpha, int act, int dtype, int w_prepared, void* ws, size_t ws_bytes,
                           void* stream);
int gs_conv2d_bwd_data(const void* gy, const float* w_hwio, void* gx, int n, int h, int w, int ci, int co,
                       int ksize, int stride, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream);
/* A generator block in one call (networks.py:44-61: conv -> leaky_relu -> pixel_norm): z = act(alpha * conv(x, w) + bias),
 * y = pixel_norm(z, eps) over the channels (ops.py:330).  The norm is fused into the conv epilogue where a tile owns every
 * channel of a pixel (co = 32 / 64 on the MFMA path), a separate pass otherwise; z may be NULL when the caller keeps no copy
 * of the activation (inference, the no-grad generator pass of the D run).  gs_conv2d_transpose_s2_fwd_bias_act_norm: same
 * for the upscaling conv. */
int gs_conv2d_fwd_bias_act_norm(const void* x, const float* w_hwio, const float* bias, void* z, void* y, int n, int h, int w, int ci, int co,
                                int ksize, int stride, float alpha, int act, float eps, int dtype, int w_prepared, void* ws,
                                size_t ws_bytes, void* stream);
int gs_conv2d_transpose_s2_fwd_bias_act_norm(const void* x, const float* w_hwio, const float* bias, void* z, void* y, int n, int h, int w,
                                             int ci, int co, float alpha, int act, float eps, int dtype, int w_prepared, void* ws,
                                             size_t ws_bytes, void* stream);
/* bwd_data whose result is multiplied by the derivative of the activation that PRODUCED the conv's input, expressed through
 * that input itself: gx = conv2d_bwd_data(gy, w) * act'(.)|mask  (mask = x of the forward conv, same shape as gx; act = LRELU
 * or TANH).  It is the data gradient w.r.t. the previous layer's pre-activation in one pass (the separate act_bwd pass of the
 * previous layer disappears).  mask NULL = gs_conv2d_bwd_data. */
int gs_conv2d_bwd_data_mask(const void* gy, const float* w_hwio, const void* mask, int mask_act, void* gx, int n, int h, int w, int ci, int co,
                            int ksize, int stride, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream);
/* The same on the forward map: y = conv2d(x, w) * act'(.)|mask with mask of y's shape (the second-order pass of a gradient penalty
 * runs the convs forward on cotangents; each result meets the derivative of the activation that follows that conv). */
int gs_pack_act_bits(void* z, int64_t p, int c, int dtype, void* stream);   /* z: [p][c] values followed by p * c / 8 bytes (written here); c % 32 == 0 */
int gs_conv2d_fwd_mask(const void* x, const float* w_hwio, const void* mask, int mask_act, void* y, int n, int h, int w, int ci, int co,
                       int ksize, int stride, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream);
int gs_conv2d_bwd_weight(const void* x, const void* gy, float* gw_hwio, int n, int h, int w, int ci, int co,
                         int ksize, int stride, float alpha, int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream);
/* bwd_weight that also returns the bias gradient of the block, gb[co] (+)= sum_{n,h,w} gy (fp32; no alpha): for bf16 3x3 convs
 * the sum rides along in the same two launches (one extra MFMA per 16 pixels against an all-ones operand), other shapes fall
 * back to gs_channel_sum inside.  gb may be NULL (= gs_conv2d_bwd_weight).  `accumulate` applies to gw and gb alike. */
int gs_conv2d_bwd_weight_bias(const void* x, const void* gy, float* gw_hwio, float* gb, int n, int h, int w, int ci, int co,
                              int ksize, int stride, float alpha, int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream);

/* tf.nn.conv2d_transpose NCHW, 3x3, stride 2, SAME, output = 2h x 2w (ops.py:266-276): the
 * gradient-of-conv definition out[2i+k] += x[i] * w[k][ci][co], cropped at the end.
 *   x [n][h][w][ci]  w [3][3][ci][co] fp32 (the STORED variable of ops.py:259-265)  y [n][2h][2w][co]
 * These are thin re-labelings of the stride-2 conv2d maps above (same kernels):
 *   transpose_fwd(x,w)        = conv2d_bwd_data (gy:=x, w^T)     with conv input side = y
 *   transpose_bwd_data(gy,w)  = conv2d_fwd      (x:=gy, w^T)
 *   transpose_bwd_weight(x,gy)= conv2d_bwd_weight(x:=gy, gy:=x)^T
 * (n,h,w) are the dims of x (the LOW resolution side). */
size_t gs_conv2d_transpose_s2_workspace_bytes(int which, int n, int h, int w, int ci, int co, int dtype);
int gs_conv2d_transpose_s2_fwd(const void* x, const float* w_hwio, void* y, int n, int h, int w, int ci, int co,
                               float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream);
int gs_conv2d_transpose_s2_fwd_bias_act(const void* x, const float* w_hwio, const float* bias, void* y, int n, int h, int w,
                                        int ci, int co, float alpha, int act, int dtype, int w_prepared, void* ws,
                                        size_t ws_bytes, void* stream);
int gs_conv2d_transpose_s2_bwd_data(const void* gy, const float* w_hwio, void* gx, int n, int h, int w, int ci, int co,
                                    float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream);
/* Second-order pass of the mode-seeking term (models.py:57-64: tf.gradients of tf.gradients(fake_images, [latents])): the conv applied to a cotangent
 * yields t = the gradient w.r.t. u = act'(z) pixel_norm_bwd(g, z) (a block's first-order backward); with h = t act'(z) the epilogue writes
 *   out_g = pixel_norm_bwd(h, z)   and   out_z = d<h, pixel_norm_bwd(g, z)>/dz      (g, z, out_g, out_z: the conv's output shape)
 * in one pass where a tile owns all channels of a pixel (gs_conv2d_fwd_pnbwdbwd_is_fused), else as the conv + gs_pixel_norm_bwd_bwd_fused. */
int gs_conv2d_fwd_pnbwdbwd(const void* x, const float* w_hwio, const void* g, const void* z, int act, float eps, void* out_g, void* out_z, int n, int h, int w,
                           int ci, int co, int ksize, int stride, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream);
int gs_conv2d_transpose_s2_fwd_pnbwdbwd(const void* x, const float* w_hwio, const void* g, const void* z, int act, float eps, void* out_g, void* out_z, int n,
                                        int h, int w, int ci, int co, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream);
int gs_conv2d_fwd_pnbwdbwd_is_fused(int n, int h, int w, int ci, int co, int ksize, int stride, int transposed, int dtype);
/* Data gradient continued through the PREVIOUS block's pixel norm and activation (networks.py:41-93: conv -> leaky_relu -> pixel_norm;
 * the backward tf.gradients builds for ops.py:330-333 behind ops.py:237-243 / 269-276), one pass where a tile owns all channels of a pixel:
 *   gx = (pixel_norm_bwd(B^T(gy, w), z) + addend) * act'(z);  z: the previous block's activation output (gx's shape), addend: optional */
int gs_conv2d_bwd_data_pnbwd(const void* gy, const float* w_hwio, const void* z, const void* addend, int act, float eps, void* gx, int n, int h, int w,
                             int ci, int co, int ksize, int stride, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream);
/* 1 when that call is one launch for the shape (n, h, w: the conv's INPUT side, i.e. gx / z; transposed: the s2 transposed conv), 0 when it
 * runs as the plain data gradient followed by gs_pixel_norm_bwd_fused in place */
int gs_conv2d_bwd_data_pnbwd_is_fused(int n, int h, int w, int ci, int co, int ksize, int stride, int transposed, int dtype);
int gs_conv2d_transpose_s2_bwd_data_pnbwd(const void* gy, const float* w_hwio, const void* z, const void* addend, int act, float eps, void* gx, int n,
                                          int h, int w, int ci, int co, float alpha, int dtype, int w_prepared, void* ws, size_t ws_bytes, void* stream);
int gs_conv2d_transpose_s2_bwd_weight(const void* x, const void* gy, float* gw_hwio, int n, int h, int w, int ci, int co,
                                      float alpha, int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream);

/* Deferred slice reduction of the weight gradients.  Every bwd_weight call is two phases: block-partial sums over pixel slices
 * (in ws), then a reduction over the slices into gw (+ gb).  The `_partial` entry points run phase 1 only and describe phase 2 in
 * `pending`; gs_wgrad_reduce_batch then folds many pending reductions in a handful of launches (a backward pass has ~70 of them;
 * each is a 10 us launch on its own).  Contract: the call's ws must stay untouched until gs_wgrad_reduce_batch has been enqueued
 * on the same stream; pending->nslices == 0 on return means nothing is pending (shapes without the vector reduce ran both
 * phases at once).  Entries that add into the same gw are applied in list order (the sum stays deterministic). */
typedef struct GsWgradReduce {
    const float* partials; /* [nslices][taps*ic*oc (+ oc when gb)] fp32, inside the call's ws */
    float* gw;             /* [taps][ic][oc], or [taps][oc][ic] when transpose */
    float* gb;             /* optional [oc] */
    int32_t nslices, taps, ic, oc;
    float alpha;
    int32_t transpose, accumulate;
    int32_t ic_ld;         /* 0, or the input-channel rows of the stored variable when gw is a channel slice of a wider one (not with transpose) */
} GsWgradReduce;
int gs_conv2d_bwd_weight_bias_partial(const void* x, const void* gy, float* gw_hwio, float* gb, int n, int h, int w, int ci, int co,
                                      int ksize, int stride, float alpha, int accumulate, int dtype, void* ws, size_t ws_bytes,
                                      GsWgradReduce* pending, void* stream);
int gs_conv2d_transpose_s2_bwd_weight_partial(const void* x, const void* gy, float* gw_hwio, int n, int h, int w, int ci, int co,
                                              float alpha, int accumulate, int dtype, void* ws, size_t ws_bytes,
                                              GsWgradReduce* pending, void* stream);
int gs_wgrad_reduce_batch(const GsWgradReduce* pending, int n, void* stream);   /* `pending`: host array */
/* Several (x, gy) pairs of ONE layer in one launch: gw (+)= sum_s bwd_weight(xs[s], gys[s]) -- the real and the fake pass of the
 * discriminator, the second-order contribution of a gradient penalty.  All pairs share (h, w, ci, co); pair s has ns[s] images
 * (ns NULL: n each).  The layer then costs one set of block partials and one slice reduction instead of one per pair.  nsrc <= GS_WGRAD_MAX_SOURCES; bit s of bias_mask
 * says whether pair s contributes to gb (ignored when gb is NULL).  `pending` as above (NULL = reduce at once); ws sized by
 * gs_conv2d_workspace_bytes(GS_CONV_BWD_WEIGHT, total images, ...) / gs_conv2d_transpose_s2_workspace_bytes(.., total images, ..). */
#define GS_WGRAD_MAX_SOURCES 4
int gs_conv2d_bwd_weight_bias_multi(const void* const* xs, const void* const* gys, const int* ns, int nsrc, unsigned bias_mask, float* gw_hwio,
                                    float* gb, int n, int h, int w, int ci, int co, int ksize, int stride, float alpha, int accumulate,
                                    int dtype, void* ws, size_t ws_bytes, GsWgradReduce* pending, void* stream);
int gs_conv2d_transpose_s2_bwd_weight_multi(const void* const* xs, const void* const* gys, const int* ns, int nsrc, float* gw_hwio, int n, int h,
                                            int w, int ci, int co, float alpha, int accumulate, int dtype, void* ws, size_t ws_bytes,
                                            GsWgradReduce* pending, void* stream);

/* Every conv weight gradient of a backward pass in ONE call -- where `minimize` asks for the gradients of all variables of a
 * network at once (models.py:81-89).  A job is one layer: up to GS_WGRAD_MAX_SOURCES (x, gy) pairs with n[s] images each, as in
 * gs_conv2d_bwd_weight_bias_multi (conv: h, w = input size) / gs_conv2d_transpose_s2_bwd_weight_multi (transposed = 1: h, w = the
 * transposed conv's input size; no bias).  bf16 layers with >= 64 channels on both sides are grouped by kernel instantiation and each
 * group runs as one stream-K launch over the pixel tiles of all its layers + one fold: partials = blocks + (layer, channel tile)
 * runs per GROUP instead of blocks per layer, summed in a fixed order (deterministic).  Other layers take the per-layer path with
 * their reductions batched at the end.  Jobs that add into the same gw are applied in list order.  `jobs` is a host array. */
typedef struct GsWgradJob {
    const void* x[GS_WGRAD_MAX_SOURCES];
    const void* gy[GS_WGRAD_MAX_SOURCES];
    int32_t n[GS_WGRAD_MAX_SOURCES];
    int32_t nsrc;
    uint32_t bias_mask;     /* which pairs contribute to gb */
    float* gw;              /* [k][k][ci][co] fp32 */
    float* gb;              /* optional [co] */
    int32_t h, w, ci, co, ksize, stride, transposed;
    float alpha;
    int32_t accumulate, dtype;
    int32_t gw_ci_stride;   /* 0, or the input-channel count of the stored variable when gw is the slice [:, :, lo:lo+ci, :] of a
                             * wider one (the 257-channel conv of the last discriminator block, networks.py:174-176): element
                             * (t, i, o) lives at gw[(t * gw_ci_stride + i) * co + o].  Grouped (stream-K) layers and layers whose slice reduction stays
                             * pending (the 1-channel direct kernel). */
} GsWgradJob;
size_t gs_conv_wgrad_jobs_workspace_bytes(const GsWgradJob* jobs, int njobs);
int gs_conv_wgrad_jobs(const GsWgradJob* jobs, int njobs, void* ws, size_t ws_bytes, void* stream);
/* Size the weight-gradient launches of the calls that follow for `cap` CUs instead of the whole chip (0: the whole chip); returns the
 * previous setting.  For a call whose launches run on a forked branch of a hipGraph beside a latency-bound chain of few-block kernels,
 * which then finds CUs to land on.  Host-side state, read when a launch is planned (workspace query and launch alike). */
int gs_wgrad_cu_cap(int cap);

/* Refreshing many prepared weight operands in one launch (after an optimizer step: ~60 conv maps, one kernel instead of
 * one re-layout launch in front of each conv).  A descriptor names the fp32 HWIO master weight, the persistent workspace
 * of one (weight, map) pair and the map: GS_PREP_* below, (ci, co, ksize, stride) of the variable, activation dtype.
 * gs_weight_prep_batch writes exactly what the map's entry point would write with w_prepared = 0, so the next call of that
 * entry point may pass w_prepared = 1.  `descs` lives in DEVICE memory (n descriptors). */
enum { GS_PREP_CONV_FWD = 0, GS_PREP_CONV_BWD_DATA = 1, GS_PREP_CONVT_FWD = 2, GS_PREP_CONVT_BWD_DATA = 3 };
typedef struct GsPrepDesc {
    const float* w_hwio; /* master weight [k][k][ci][co] */
    void* ws;            /* the map's workspace (operand at its start) */
    int32_t map, ci, co, ksize, stride, dtype;
} GsPrepDesc;
int gs_weight_prep_batch(const GsPrepDesc* descs, int n, void* stream);

/* -------------------------------------------------------------------------------- dense
 * tf.matmul (ops.py:197): y[b][out] = alpha * x[b][in] @ w[in][out] (split-K partials live in ws).
 * bwd_data: gx = alpha * gy @ w^T ; bwd_weight: gw = alpha * x^T @ gy (fp32). */
size_t gs_dense_fwd_workspace_bytes(int b, int in, int out);
int gs_dense_fwd(const void* x, const float* w, void* y, int b, int in, int out, float alpha, int dtype,
                 void* ws, size_t ws_bytes, void* stream);
int gs_dense_bwd_data(const void* gy, const float* w, void* gx, int b, int in, int out, float alpha, int dtype, void* stream);
int gs_dense_bwd_weight(const void* x, const void* gy, float* gw, int b, int in, int out, float alpha, int accumulate, int dtype,
                        void* stream);
/* ops.py:183-201 as the reference calls it -- dense, bias_add, activation (networks.py:185-187: the discriminator's 8192 -> 256 dense
 * + leaky_relu and its 256 -> 61 logits layer): y = act(alpha * x @ w + bias) with bias (may be NULL) and activation applied where the
 * forward writes its result (the split-K finalize pass or the direct store): one or two launches instead of three.  fp32: the same
 * operations in the same order as gs_dense_fwd + gs_bias_act_fwd (bit-identical); bf16: one rounding instead of two. */
int gs_dense_fwd_bias_act(const void* x, const float* w, const float* bias, void* y, int b, int in, int out, float alpha, int act,
                          int dtype, void* ws, size_t ws_bytes, void* stream);
/* The same three maps with the INPUT side in channels-last memory: x / gx are the [b][hw][c] memory of an activation whose
 * tf.layers.flatten (NCHW: column c * hw + p, networks.py:185) feeds the layer, w stays [c * hw][out] as stored.  The flatten is a
 * row-index map inside the kernels -- no NCHW copy of the activation, no copy of its gradient back.  Vector-path shapes only
 * (out % 256 == 0, batch <= 16 for bwd_weight; GS_ERR_UNSUPPORTED otherwise). */
int gs_dense_fwd_nhwc(const void* x, const float* w, void* y, int b, int c, int hw, int out, float alpha, int dtype,
                      void* ws, size_t ws_bytes, void* stream);
int gs_dense_fwd_bias_act_nhwc(const void* x, const float* w, const float* bias, void* y, int b, int c, int hw, int out, float alpha,
                               int act, int dtype, void* ws, size_t ws_bytes, void* stream);
int gs_dense_bwd_data_nhwc(const void* gy, const float* w, void* gx, int b, int c, int hw, int out, float alpha, int dtype, void* stream);
int gs_dense_bwd_weight_nhwc(const void* x, const void* gy, float* gw, int b, int c, int hw, int out, float alpha, int accumulate,
                             int dtype, void* stream);

/* tf.nn.embedding_lookup(w*alpha, argmax(labels,1)) (ops.py:217): idx[b] are the argmax indices.
 * fwd: y[b][units] = alpha * w[idx[b]][:]  ; bwd: gw[rows][units] = alpha * scatter_add(gy) (gw zero-filled here). */
int gs_embedding_fwd(const int64_t* idx, const float* w, void* y, int b, int rows, int units, float alpha, int dtype, void* stream);
/* ... with the row index taken from the one-hot input itself (ops.py:207: tf.argmax of the inputs; first maximum), written to idx_out[b]
 * for gs_embedding_bwd: labels [b][rows] of `dtype`, y [b][units] of `dtype` */
int gs_embedding_onehot_fwd(const void* labels, const float* w, void* y, int64_t* idx_out, int b, int rows, int units, float alpha, int dtype, void* stream);
int gs_embedding_bwd(const int64_t* idx, const void* gy, float* gw, int b, int rows, int units, float alpha, int dtype, void* stream);

/* ----------------------------------------------------- bias / activations (channels-last)
 * tf.nn.bias_add + tf.nn.leaky_relu(alpha=0.2) / tf.nn.tanh (ops.py:244-246, networks.py:55,66,106 ...).
 *   y[p][c] = act(x[p][c] + bias[c])        (bias may be NULL)
 *   act_bwd : gx = g * act'(.) expressed through the activation OUTPUT y
 *   tanh_bwd_bwd: second-order term d/dy [g*(1-y^2)] . gg = -2*y*g*gg  (lrelu has none)
 *   channel_sum: out[c] = sum_p g[p][c] (bias gradient, fp32 out) */
int gs_bias_act_fwd(const void* x, const float* bias, void* y, int64_t p, int c, int act, int dtype, void* stream);
int gs_act_bwd(const void* g, const void* y, void* gx, int64_t numel, int act, int dtype, void* stream);
/* The generator's first block, networks.py:41-56 (dense -> tf.reshape to [n, c, h, w] -> leaky_relu): the dense layer's units are channel-major
 * (unit u = ch * hw + p), activations here are channels-last -- the reorder rides in the bias / activation pass.
 *   gs_units_bias_act_to_nhwc: z[n][p][ch] = act(y[n][u] + bias[u]) (mask NULL), or y[n][u] * act'(mask[n][p][ch]) (second-order pass)
 *   gs_nhwc_act_bwd_to_units:  gu[n][u] = g[n][p][ch] * act'(z[n][p][ch])  (the backward, in the units' order for gs_dense_bwd_*) */
int gs_units_bias_act_to_nhwc(const void* y, const float* bias, const void* mask, void* z, int n, int c, int hw, int act, int dtype, void* stream);
int gs_nhwc_act_bwd_to_units(const void* g, const void* z, void* gu, int n, int c, int hw, int act, int dtype, void* stream);

/* act_bwd and the bias gradient in one pass: gx = g*act'(y), gb[c] = sum_p gx[p][c] (ws: gs_channel_sum_workspace_bytes) */
int gs_act_bwd_bias(const void* g, const void* y, void* gx, float* gb, int64_t p, int c, int act, int accumulate, int dtype,
                    void* ws, size_t ws_bytes, void* stream);
int gs_tanh_bwd_bwd(const void* gg, const void* g, const void* y, void* out, int64_t numel, int dtype, void* stream);
size_t gs_channel_sum_workspace_bytes(int64_t p, int c);
int gs_channel_sum(const void* g, float* out, int64_t p, int c, int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream);
/* Deferred folds of the bias gradients (the variable gradients of models.py:81-89 are only read after the whole backward).  With
 * GS_SUM_PARTIALS or-ed into `accumulate`, gs_channel_sum / gs_act_bwd_bias / gs_pixel_norm_bwd_fused_bias leave their per-block
 * partial rows in `ws` (the caller's own buffer of gs_bias_partial_rows(...) x c floats, alive until the fold) and do not touch
 * gb; gs_channel_fold_batch then folds every pending gradient in ONE launch (two when a producer left more than 64 rows), in
 * the order the immediate fold uses (bit-identical).  gs_bias_partial_rows == 0: that shape is summed directly whatever the flag. */
#define GS_SUM_PARTIALS 2
enum { GS_BIAS_FROM_CHANNEL_SUM = 0, GS_BIAS_FROM_ACT_BWD = 1, GS_BIAS_FROM_PIXEL_NORM_BWD = 2 };
typedef struct GsFoldJob {
    const float* part; /* [nparts][c] partial rows */
    float* out;        /* [c] */
    int32_t nparts, c;
    int32_t accumulate; /* 1: out += sum */
    int32_t reserved;
} GsFoldJob;
int gs_bias_partial_rows(int producer, int64_t p, int c, int dtype);
size_t gs_channel_fold_batch_workspace_bytes(const GsFoldJob* jobs, int njobs);
int gs_channel_fold_batch(const GsFoldJob* jobs, int njobs, void* ws, size_t ws_bytes, void* stream);

/* pixel_normalization (ops.py:330-333): y = x / sqrt(mean_c(x^2) + eps), per row p over c.
 *   bwd      : gx = r*(g - y*mean_c(y*g)),  r = rsqrt(mean_c(x^2)+eps)
 *   bwd_bwd_x: d<gg, bwd(g,x)>/dx = (r^2/C) * (-(gg.g) y - (y.g) gg - (y.gg) g + 3 (y.gg)(y.g) y / C)
 *   (d<gg, bwd(g,x)>/dg = bwd(gg, x): the Jacobian is symmetric) */
int gs_pixel_norm_fwd(const void* x, void* y, int64_t p, int c, float eps, int dtype, void* stream);
int gs_pixel_norm_bwd(const void* g, const void* x, void* gx, int64_t p, int c, float eps, int dtype, void* stream);
/* The generator blocks are conv -> activation -> pixel norm, i.e. the norm's input x is an activation OUTPUT.  The passes
 * that surround the norm's gradients fold into them (act'(.) is expressed through x; act = NONE / LRELU / TANH):
 *   bwd_fused    : gx = (pixel_norm_bwd(g * pre_act'(x), x) + addend) * post_act'(x)      (addend may be NULL)
 *     post_act  -> gradient w.r.t. the pre-activation (replaces the act_bwd pass that follows);
 *     addend    -> a second gradient arriving at x (from the second-order graph), summed in the same pass;
 *     pre_act   -> the transposed form, used when this chain is itself differentiated (mode-seeking term)
 *   bwd_bwd_fused: out = pixel_norm_bwd_bwd(gg', g, x) with gg' = gg * pre_act'(x); out_g (may be NULL) = pixel_norm_bwd(gg', x)
 *                  -- both gradients of a differentiated norm-backward node from one pass over gg, g, x */
int gs_pixel_norm_bwd_fused(const void* g, const void* x, const void* addend, void* gx, int64_t p, int c, float eps, int pre_act, int post_act,
                            int dtype, void* stream);
/* ... the same pass also sums its result over the pixels: gb[c] (+)= sum_p gx[p][c] -- the bias gradient of the block
 * z = act(conv + bias) whose output x is (networks.py:80-87: conv_transpose -> bias -> leaky_relu -> pixel_norm); one partial row per block
 * in ws, folded in a fixed order */
size_t gs_pixel_norm_bwd_bias_workspace_bytes(int64_t p, int c, int dtype);
int gs_pixel_norm_bwd_fused_bias(const void* g, const void* x, const void* addend, void* gx, float* gb, int64_t p, int c, float eps, int pre_act,
                                 int post_act, int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream);
int gs_pixel_norm_bwd_bwd_fused(const void* gg, const void* g, const void* x, void* out, void* out_g, int64_t p, int c, float eps, int pre_act,
                                int dtype, void* stream);
int gs_pixel_norm_bwd_bwd(const void* gg, const void* g, const void* x, void* out, int64_t p, int c, float eps, int dtype, void* stream);

/* upscale2d / downscale2d (ops.py:283-305).
 *   upscale : y[n][h*fy][w*fx][c] = x[n][h][w][c]                       (bit-exact copy)
 *   blocksum: y[n][h/fy][w/fx][c] = scale * sum_{fy x fx block} x        (scale = 1/(fy*fx) is avg_pool;
 *             scale = 1 is the adjoint of upscale) */
int gs_upscale2d(const void* x, void* y, int n, int h, int w, int c, int fy, int fx, float scale, int dtype, void* stream);
int gs_blocksum2d(const void* x, void* y, int n, int h, int w, int c, int fy, int fx, float scale, int dtype, void* stream);

/* batch_stddev (ops.py:336-348), groups = 4: x [b][h][w][c] -> y [b][h][w][1], b % 4 == 0.
 *   bwd: gx from gy ; bwd_bwd: (ggy, gx2) = gradients of <ggx, bwd(gy, x)> w.r.t. gy and x. */
int gs_batch_stddev_fwd(const void* x, void* y, int b, int hw, int c, float eps, int dtype, void* stream);
/* (addend, optional: the other gradient into x -- that of the conv beside the statistic, networks.py:174-176 -- added in the same pass) */
int gs_batch_stddev_bwd(const void* gy, const void* x, const void* addend, void* gx, int b, int hw, int c, float eps, int dtype, void* stream);
int gs_batch_stddev_bwd_bwd(const void* ggx, const void* gy, const void* x, void* ggy, void* gx2,
                            int b, int hw, int c, float eps, int dtype, void* stream);

/* lerp (networks.py:10-11) and generic fused axpby: out = ca*a + cb*b. */
int gs_axpby(const void* a, const void* b, void* out, int64_t numel, float ca, float cb, int dtype, void* stream);
/* The same with the two coefficients read from a device table (coef[ia], coef[ib]): a fade-in weight that changes every step
 * must not be frozen into a captured hipGraph as a by-value scalar. */
int gs_axpby_dev(const void* a, const void* b, void* out, int64_t numel, const float* coef, int ia, int ib, int dtype, void* stream);

/* per-sample sum of squares (the R1 penalty reduction, models.py:48): out[r] = sum_j x[r][j]^2 (fp32 out);
 * row_scale: out[r][j] = s[r] * x[r][j]  (its gradient, s fp32). */
size_t gs_sumsq_rows_workspace_bytes(int rows);
int gs_sumsq_rows(const void* x, float* out, int rows, int64_t cols, int dtype, void* ws, size_t ws_bytes, void* stream);
int gs_row_scale(const void* x, const float* s, float alpha, void* out, int rows, int64_t cols, int dtype, void* stream);   /* out[r][:] = alpha s[r] x[r][:] */

/* The two GAN losses (models.py:39-65) with their gradients, one launch each.  logits / labels [n][c] in the activation dtype,
 * labels one-hot (real_logit_i = sum_c logits[i][c] * labels[i][c] = tf.gather_nd(logits, tf.where(labels))):
 *   L_D = mean_i [ softplus(-r_i) + softplus(f_i) + penalty_weight penalty_i ]     penalty: optional (R1 term: sum of squared gradients per example), fp32 [n]
 *   L_G = mean_i [ softplus(-f_i) + weight / (sumsq_i + eps) ]       sumsq: optional, sum((d sum(G(z)) / d z_i)^2), fp32 [n]
 * loss: fp32 scalar; g_*: d loss / d logits ([n][c], activation dtype), g_sumsq: d L_G / d sumsq (fp32 [n]); g_penalty: d L_D / d penalty = penalty_weight / n (fp32 [n], written when penalty is given).
 * Either half of a sum may be ABSENT (real_logits or fake_logits NULL with its g_*; for L_G fake_logits NULL with sumsq given, c = 1): a run that
 * keeps two independent passes on two streams takes the loss as two launches, each the partial mean over its own terms -- the gradients are the same
 * numbers, and no pass waits for the other's forward before its backward starts. */
int gs_gan_d_loss(const void* real_logits, const void* fake_logits, const void* labels, const float* penalty, float penalty_weight, int n, int c,
                  float* loss, void* g_real, void* g_fake, float* g_penalty, int dtype, void* stream);
int gs_gan_g_loss(const void* fake_logits, const void* labels, const float* sumsq, float weight, float eps, int n, int c, float* loss,
                  void* g_fake, float* g_sumsq, int dtype, void* stream);

/* tf.train.AdamOptimizer step (models.py:67-89), TF form, fused over one flat fp32 buffer:
 *   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= lr_t * m / (sqrt(v) + eps),
 *   lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller; grad_scale multiplies g first (1/world). */
int gs_adam_tf_step(float* p, const float* g, float* m, float* v, int64_t numel, float lr_t, float beta1,
                    float beta2, float eps, float grad_scale, void* stream);
/* the same step, and g is cleared behind it: tf.gradients starts every run from zero (models.py:81-89 builds the sums anew), the
 * flat gradient buffers are accumulated into across a run, so the update hands the next run a zeroed buffer without a fill pass */
int gs_adam_tf_step_zero_grad(float* p, float* g, float* m, float* v, int64_t numel, float lr_t, float beta1,
                              float beta2, float eps, float grad_scale, void* stream);
/* the same step with lr_t READ FROM DEVICE MEMORY at execution time (`lr_t_dev[0]`, written by the caller stream-ordered ahead of the
 * launch): a by-value scalar would be frozen into a captured hipGraph, and with the optimizer step inside the iteration's graph no eager
 * launch is left between the two runs of models.py:191-192.  A NEGATIVE value means "no step pending": every buffer is left untouched
 * (the first replay of a graph that starts with the PREVIOUS iteration's update).  zero_grad != 0: g is cleared behind the update. */
int gs_adam_tf_step_dev(float* p, float* g, float* m, float* v, int64_t numel, const float* lr_t_dev, float beta1,
                        float beta2, float eps, float grad_scale, int zero_grad, void* stream);

/* ------------------------------------------------------------------------------ spectral
 * spectral_ops.py:45-94.  Plan = immutable per-device tables (Hann window, twiddles, CSR mel matrix
 * supplied by the caller as built by linear_to_mel_weight_matrix, dense pinv for the inverse). */
typedef struct gs_spectral_plan gs_spectral_plan;
int gs_spectral_plan_create(gs_spectral_plan** plan, int frame_length, int frame_step, int time_steps,
                            const float* mel_dense /* host [nbins][nbins] */, const float* mel_pinv /* host, may be NULL */);
int gs_spectral_plan_destroy(gs_spectral_plan* plan);
/* stage-wise entry points (parity tests) */
int gs_stft_fwd(const gs_spectral_plan* plan, const float* wave, int batch, int wave_len, int front_pad,
                float* magnitude, float* phase, void* stream);           /* [b][T][nbins] each, DC dropped */
int gs_mel_project(const gs_spectral_plan* plan, const float* in, float* out, int64_t rows, void* stream);
int gs_if_unwrap(const gs_spectral_plan* plan, const float* mel_phase, float* mel_if, int batch, void* stream);
/* fused: waveform -> (log-mel, IF) written as one channels-last image [b][T][nbins][2] */
int gs_stft_mel_if_fwd(const gs_spectral_plan* plan, const float* wave, int batch, int wave_len, int front_pad,
                       void* images, int dtype, void* ws, size_t ws_bytes, void* stream);
size_t gs_stft_mel_if_workspace_bytes(const gs_spectral_plan* plan, int batch);
/* inverse (spectral_ops.py:97-149): images [b][T][nbins][2] -> wave [b][wave_len] */
int gs_mel_if_to_waveform(const gs_spectral_plan* plan, const void* images, int batch, int wave_len, int front_pad,
                          float* wave, int dtype, void* ws, size_t ws_bytes, void* stream);
size_t gs_mel_if_to_waveform_workspace_bytes(const gs_spectral_plan* plan, int batch);

#ifdef __cplusplus
}
#endif
#endif /* GANSYNTH_HIP_H */
