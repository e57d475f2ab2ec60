/*
 * rgfm.h -- C ABI of librgfm_hip.so: the MI355X (gfx950) implementation of the
 * ratio-guided flow-matching SAMPLER of foubari/ratio_guided_Multimodal_FM.
 *
 * The reference has no FFI; its boundary for this path is a Python module API.
 * Each entry point below replaces the reference call cited next to it, and is
 * what a ctypes / cffi / pybind stub on the reference side would bind
 * (INTEGRATION.md shows that stub).
 *
 * Conventions
 *   - every pointer named *_dev / x / y / out / ws is a raw DEVICE pointer to
 *     contiguous fp32 data that the CALLER owns (PyTorch: tensor.data_ptr());
 *     the library owns only what *_create allocates and frees it in *_destroy;
 *   - image tensors are NCHW fp32, exactly as the reference passes them;
 *   - `stream` is a hipStream_t (PyTorch: torch.cuda.current_stream().cuda_stream);
 *     all work is enqueued on it; forward/sample calls neither synchronise nor
 *     allocate nor create streams/events (per-device resources are created by the
 *     first *_create on that device; rgfm_profile_reserve pre-creates the bench
 *     timers' events) and read the RGFM_* environment switches once on entry;
 *   - return value: 0 on success, a negative RGFM_E* code otherwise; nothing is
 *     thrown across the ABI; rgfm_last_error() returns text for the calling
 *     thread's last failure;
 *   - one host thread per handle; distinct handles are independent;
 *   - arithmetic: fp32 tensors, fp32 accumulation.  By default (RGFM_CONV_HX2) the 3x3 /
 *     transposed convolutions EMULATE each fp32 product on the f16 matrix cores: both
 *     (power-of-two scaled) operands are held as two fp16 planes -- 22 significant bits -- and
 *     three of the four plane products are accumulated in fp32 (product
 *     error measured 2^-24.8 median, 2^-20.3 at the 99.9th percentile; one whole U-Net
 *     evaluation against float64: 2.6e-6, the reference's own fp32 run 1.7e-6).  The
 *     representation has a window: activations 2^-8 <= max|a| per wave block and |a| < 2048,
 *     GroupNorm parameters and weights of ordinary magnitude.  Convs whose weights or norm
 *     parameters are outside it are routed to the split-bf16 kernel when the handle is
 *     created; activations outside it raise the handle's range flag (rgfm_unet_range_flag:
 *     bit 0 too large, bit 1 too small), on which the caller repeats the call with the handle
 *     set to RGFM_CONV_BX3 / RGFM_CONV_F32 (rgfm_unet_set_conv_mode) -- the Python host does both.
 *     RGFM_CONV_BX3: exact three-way bf16 split, six products, fp32 exponent range.
 *     RGFM_CONV_F32: v_mfma_f32_32x32x2_f32 for every convolution.
 */
#ifndef RGFM_H_
#define RGFM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGFM_ABI_VERSION 3

#define RGFM_OK 0
#define RGFM_EINVAL (-1)    /* bad argument / unsupported shape          */
#define RGFM_ENOMEM (-2)    /* hipMalloc failed or workspace too small   */
#define RGFM_EHIP (-3)      /* a HIP runtime call failed                 */
#define RGFM_ENODEVICE (-4) /* no gfx950 device visible                  */

#define RGFM_MAX_LEVELS 4

typedef void* rgfm_stream_t; /* hipStream_t */

/* ------------------------------------------------------------------------
 * Velocity U-Net.  Replaces FlexibleUNet / FlowMatchingUNetMNIST /
 * FlowMatchingUNetSVHN (reference src/models/unet_flexible.py:111-291) and
 * UNetMNIST / FlowMatchingUNet (src/models/unet.py:122-305): same
 * architecture parameters, same parameter tensors.
 * ---------------------------------------------------------------------- */
typedef struct rgfm_unet_desc {
  int32_t in_channels;    /* 1 (MNIST) or 3 (SVHN)                                 */
  int32_t img_size;       /* 28 or 32 (square)                                     */
  int32_t model_channels; /* 32 / 64                                               */
  int32_t num_levels;     /* len(channel_mult)                                     */
  int32_t channel_mult[RGFM_MAX_LEVELS];
  int32_t num_res_blocks; /* 2                                                     */
} rgfm_unet_desc;

typedef struct rgfm_unet rgfm_unet;

/* Number of fp32 values in the parameter blob for `desc`: the tensors of the
 * reference module's state_dict(), in state_dict() order, each flattened
 * row-major and concatenated (unet_flexible.py:146-201 registration order). */
int rgfm_unet_param_floats(const rgfm_unet_desc* desc, size_t* n_floats);

/* Builds the device-side packed weights from the state_dict-order blob
 * (replaces module construction + load_state_dict, src/utils/__init__.py:25-51).
 * The blob may be freed once `stream` has been synchronised. */
int rgfm_unet_create(const rgfm_unet_desc* desc, const float* params_dev, size_t n_floats,
                     rgfm_stream_t stream, rgfm_unet** out);
void rgfm_unet_destroy(rgfm_unet* h);

/* Scratch bytes one forward / sample call needs for `batch` rows. */
int rgfm_unet_workspace_bytes(const rgfm_unet* h, int batch, size_t* bytes);

/* v_out[B,C,H,W] = model(x[B,C,H,W], t)   (FlexibleUNet.forward,
 * unet_flexible.py:203-261).  t_dev holds t_count in {1, batch} timesteps
 * (t_count == 1: one t shared by every row, as inside the samplers). */
int rgfm_unet_forward(rgfm_unet* h, const float* x, const float* t_dev, int t_count, float* v_out,
                      int batch, void* ws, size_t ws_bytes, rgfm_stream_t stream);

/* Debug / parity hook: after a forward, copy activation `index` (the order the
 * tensors are produced in; see rgfm_unet_num_activations) as NCHW fp32 into
 * out_dev.  Only valid when the handle was put in trace mode, which keeps every
 * activation in its own buffer. */
int rgfm_unet_set_trace(rgfm_unet* h, int enable);
/* Parity hook: emb_out[t_count][model_channels] = timestep_embedding(t, model_channels) exactly as the
 * device evaluates it in front of the time MLPs (unet_flexible.py:16-36: cos half first, t unscaled).
 * ws: at least rgfm_unet_workspace_bytes(h, t_count) bytes. */
int rgfm_unet_time_embedding(rgfm_unet* h, const float* t_dev, int t_count, float* emb_out, void* ws,
                             size_t ws_bytes, rgfm_stream_t stream);
int rgfm_unet_num_activations(const rgfm_unet* h, int* n);
/* Debug / test hook: how many ResBlocks of the handle's LATEST network walk handed conv1's output to conv2 in the
 * pre-normalised pre-split "P format" (conv_mfma_hx2d.hip) instead of as an fp32 map.  Lets a
 * test see that the hand-over is really taken (its results are the fp32 hand-over's to the last bit or two). */
int rgfm_unet_p_handovers(const rgfm_unet* h, int* blocks);
/* Debug / test hook: how many convs of the handle's LATEST network walk were described for the Winograd F(2x2, 3x3) kernel
 * (conv_mfma_hx2w.hip; opt-in: RGFM_WINO=1). */
int rgfm_unet_wino_convs(const rgfm_unet* h, int* convs);
int rgfm_unet_activation_shape(const rgfm_unet* h, int index, int* channels, int* height, int* width);
int rgfm_unet_read_activation(rgfm_unet* h, int index, int batch, const void* ws, float* out_dev,
                              rgfm_stream_t stream);

/* ------------------------------------------------------------------------
 * Density-ratio estimators.  Replaces RatioEstimatorMNISTSVHN
 * (src/models/ratio_flexible.py:305-385) and RatioEstimator
 * (src/models/ratio_estimator.py:96-191), eval mode.
 * ---------------------------------------------------------------------- */
#define RGFM_RATIO_MNIST_SVHN 0 /* x[N,1,32,32], y[N,3,32,32], BatchNorm encoders */
#define RGFM_RATIO_MNIST28 1    /* x[N,1,28,28], y[N,1,28,28], GroupNorm encoders */

#define RGFM_LOSS_DISC 0
#define RGFM_LOSS_RULSIF 1

#define RGFM_RATIO_OUT_SCORE 0     /* forward(x, y)                      */
#define RGFM_RATIO_OUT_LOG_RATIO 1 /* log_ratio(x, y)                    */
#define RGFM_RATIO_OUT_RATIO 2     /* log_ratio(x, y).exp()  (mc_ratios) */

typedef struct rgfm_ratio_desc {
  int32_t kind;        /* RGFM_RATIO_*      */
  int32_t feature_dim; /* 256               */
  int32_t hidden_dim;  /* 512               */
  int32_t loss_type;   /* RGFM_LOSS_*       */
} rgfm_ratio_desc;

typedef struct rgfm_ratio rgfm_ratio;

int rgfm_ratio_param_floats(const rgfm_ratio_desc* desc, size_t* n_floats);
/* Blob = state_dict() order, fp32; BatchNorm `num_batches_tracked` entries are
 * carried as one fp32 each (value ignored) so that offsets follow the keys. */
int rgfm_ratio_create(const rgfm_ratio_desc* desc, const float* params_dev, size_t n_floats,
                      rgfm_stream_t stream, rgfm_ratio** out);
void rgfm_ratio_destroy(rgfm_ratio* h);
int rgfm_ratio_workspace_bytes(const rgfm_ratio* h, int n, size_t* bytes);
/* out[n] per `what` (RGFM_RATIO_OUT_*): forward ratio_flexible.py:347-364,
 * log_ratio :366-385, and the .exp() of sample_mnist_svhn.py:110-111. */
int rgfm_ratio_eval(rgfm_ratio* h, const float* x, const float* y, float* out, int n, int what,
                    void* ws, size_t ws_bytes, rgfm_stream_t stream);

/* Gradient of the log-ratio, d log_ratio(x, y) / d(x, y): what torch.autograd.grad(model.log_ratio(x, y).sum(),
 * (x, y)) returns for the reference module in eval mode (ratio_flexible.py:347-385, ratio_estimator.py:137-191;
 * hand-written reverse pass).  RGFM_RATIO_MNIST_SVHN: gx[n,1,32,32], gy[n,3,32,32]; RGFM_RATIO_MNIST28: gx, gy
 * [n,1,28,28]; log_ratio_out (optional) [n]. */
int rgfm_ratio_grad_workspace_bytes(const rgfm_ratio* h, int n, size_t* bytes);
int rgfm_ratio_grad_log_ratio(rgfm_ratio* h, const float* x, const float* y, float* gx, float* gy,
                              float* log_ratio_out, int n, void* ws, size_t ws_bytes, rgfm_stream_t stream);

/* ------------------------------------------------------------------------
 * Samplers (the Euler/ODE loops).
 * ---------------------------------------------------------------------- */

/* Unguided Euler integration of one model, in place on x_inout[B,C,H,W]:
 *   for step in [step_begin, step_end): t = step*(1/num_steps);
 *       x <- x + model(x, t) * dt
 * Replaces CFMSchedule.sample (src/utils/flow_utils.py:69-100) and each MC
 * pre-phase loop (src/sample_mnist_svhn.py:90-95, :99-104;
 * flow_utils.py:236-241, :245-250). */
int rgfm_sample_single_workspace_bytes(const rgfm_unet* h, int batch, size_t* bytes);
int rgfm_sample_single(rgfm_unet* h, float* x_inout, int batch, int num_steps, int step_begin,
                       int step_end, void* ws, size_t ws_bytes, rgfm_stream_t stream);

/* Paired Euler loop with optional MC importance-weighted guidance, in place on
 * x_inout[B,Cx,H,W], y_inout[B,Cy,H,W].  Replaces the main loop of
 * sample_bimodal_guided_mnist_svhn (src/sample_mnist_svhn.py:114-175) and of
 * sample_bimodal_guided (src/utils/flow_utils.py:263-373).
 *   mc_x1[N,..], mc_y1[N,..], mc_ratios[N]: terminal MC set; pass n_mc = 0 (and
 *   null pointers) for guidance_method == 'none'.
 *   gamma = guidance_strength (not clamped).  Guidance is applied on steps with
 *   t = step/num_steps > 1e-3, as in the reference.
 *   The two networks of a step are independent; the second one is enqueued on a
 *   library-owned side stream that is forked from and joined back into `stream`
 *   with events every step (RGFM_OVERLAP=0 keeps everything on `stream`), so the
 *   call is still ordered with respect to `stream` on entry and on return. */
int rgfm_sample_pair_workspace_bytes(const rgfm_unet* hx, const rgfm_unet* hy, int batch, int n_mc,
                                     size_t* bytes);
int rgfm_sample_pair(rgfm_unet* hx, rgfm_unet* hy, float* x_inout, float* y_inout,
                     const float* mc_x1, const float* mc_y1, const float* mc_ratios, int n_mc,
                     int batch, int num_steps, double gamma, int step_begin, int step_end, void* ws,
                     size_t ws_bytes, rgfm_stream_t stream);

/* Paired Euler loop with GRADIENT LOG-RATIO guidance, in place: every step
 *     x <- x + (v_x(x, t) + gamma * d log r(x, y)/dx) dt,   y likewise
 * (reference README.md:159-164, "v_guided = v_ind + gamma * grad log r(x_t, y_t)").  The reference ships no code
 * for this mode, so the composition above is this library's reading of that line; the gradient itself is the
 * autograd gradient of the reference module (rgfm_ratio_grad_log_ratio).  The pair must be the estimator's:
 * 1x32x32 + 3x32x32 (RGFM_RATIO_MNIST_SVHN) or 1x28x28 + 1x28x28 (RGFM_RATIO_MNIST28). */
int rgfm_sample_pair_grad_workspace_bytes(const rgfm_unet* hx, const rgfm_unet* hy, const rgfm_ratio* hr,
                                          int batch, size_t* bytes);
int rgfm_sample_pair_grad(rgfm_unet* hx, rgfm_unet* hy, rgfm_ratio* hr, float* x_inout, float* y_inout,
                          int batch, int num_steps, double gamma, int step_begin, int step_end, void* ws,
                          size_t ws_bytes, rgfm_stream_t stream);

/* ------------------------------------------------------------------ FlowMatchingModel ("--model original")
 * The reference's encoder-decoder velocity net (src/models/flow_matching.py:127-173;
 * built by src/sample.py:152-154 and src/evaluate.py:144-146) for 1x28x28 images:
 * ImageEncoder (:34-72: four 3x3 convs, two of them stride 2, GroupNorm(8)+SiLU,
 * Linear 12544->feature_dim), SinusoidalPositionEmbeddings (:11-31) and
 * VelocityDecoder (:75-124: Linear, two ConvTranspose2d(k4,s2,p1), 3x3 convs).
 * Same conventions as rgfm_unet_*: blob = state_dict() order, NCHW boundary tensors,
 * caller-owned workspace, stream-ordered, no synchronisation. */
typedef struct rgfm_fmnet_desc {
  int32_t img_channels; /* 1   (FlowMatchingModel.__init__ argument, :138) */
  int32_t feature_dim;  /* 256 */
  int32_t time_emb_dim; /* 128 */
} rgfm_fmnet_desc;

typedef struct rgfm_fmnet rgfm_fmnet;

int rgfm_fmnet_param_floats(const rgfm_fmnet_desc* desc, size_t* n_floats);
int rgfm_fmnet_create(const rgfm_fmnet_desc* desc, const float* params_dev, size_t n_floats,
                      rgfm_stream_t stream, rgfm_fmnet** out);
void rgfm_fmnet_destroy(rgfm_fmnet* h);
/* bytes for one forward or one rgfm_fmnet_sample_single call at this batch */
int rgfm_fmnet_workspace_bytes(const rgfm_fmnet* h, int batch, size_t* bytes);
/* v_out[B,1,28,28] = model(x, t)  (FlowMatchingModel.forward, :153-173); t_count in {1, batch} */
int rgfm_fmnet_forward(rgfm_fmnet* h, const float* x, const float* t_dev, int t_count, float* v_out,
                       int batch, void* ws, size_t ws_bytes, rgfm_stream_t stream);
/* CFMSchedule.sample / the unguided Euler loops of flow_utils.py:69-100, :186-278 with this net */
int rgfm_fmnet_sample_single(rgfm_fmnet* h, float* x_inout, int batch, int num_steps, int step_begin,
                             int step_end, void* ws, size_t ws_bytes, rgfm_stream_t stream);
/* paired_sampler (flow_utils.py:186-278) with two FlowMatchingModel nets: as rgfm_sample_pair */
int rgfm_fmnet_sample_pair_workspace_bytes(const rgfm_fmnet* hx, const rgfm_fmnet* hy, int batch,
                                           int n_mc, size_t* bytes);
int rgfm_fmnet_sample_pair(rgfm_fmnet* hx, rgfm_fmnet* hy, float* x_inout, float* y_inout,
                           const float* mc_x1, const float* mc_y1, const float* mc_ratios, int n_mc,
                           int batch, int num_steps, double gamma, int step_begin, int step_end,
                           void* ws, size_t ws_bytes, rgfm_stream_t stream);

/* One guidance evaluation on its own (parity hook for sample_mnist_svhn.py:124-171):
 * vx/vy are overwritten with (1-gamma)*v + gamma*g at time t; weights_out[B,N]
 * (optional, may be null) receives the normalised importance weights. */
int rgfm_guidance_workspace_bytes(int batch, int n_mc, size_t* bytes);
int rgfm_guidance_apply(const float* x, const float* y, float* vx, float* vy, const float* mc_x1,
                        const float* mc_y1, const float* mc_ratios, int batch, int n_mc, int dim_x,
                        int dim_y, double t, double gamma, float* weights_out, void* ws,
                        size_t ws_bytes, rgfm_stream_t stream);

/* ------------------------------------------------------------------------
 * Bench support: per-kernel-class device time measured with hipEvents on the
 * launch stream (bench.py's roofline line).  Timing is off by default.
 * ---------------------------------------------------------------------- */
#define RGFM_KCLASS_CONV_MFMA 0  /* conv3x3/1x1 implicit GEMM on the matrix cores: work = FLOPs (2*MAC)      */
#define RGFM_KCLASS_OTHER 1      /* everything not listed here (work = 0)                                     */
/* HBM-bound kernels of the U-Net pair step: work = ALGORITHMIC HBM bytes of the launch */
#define RGFM_KCLASS_CONV_IN1 2   /* input_conv, 1-channel image (conv_in_kernel<1>)                          */
#define RGFM_KCLASS_CONV_IN3 3   /* input_conv, 3-channel image (conv_in_kernel<3>)                          */
#define RGFM_KCLASS_CONV_OUT1 4  /* out_norm + SiLU + out_conv (+ fused Euler), 1 channel                    */
#define RGFM_KCLASS_CONV_OUT3 5  /* ... 3 channels                                                           */
#define RGFM_KCLASS_GUID_LOGP 6  /* guidance: squared distances to the MC set (guid_logp_kernel)             */
#define RGFM_KCLASS_GUID_APPLY 7 /* guidance: weights, weighted velocity, blend, Euler (guid_apply_kernel x2) */
#define RGFM_KCLASS_COUNT 8
int rgfm_profile_enable(int enable);
int rgfm_profile_reset(void);
/* Waits for the recorded events, then returns for the class, since the last reset:
 *   busy_ms  = length of the UNION of the launches' [start, stop] intervals (the two velocity nets
 *              of a step run on two streams, so launches of one class may overlap in time);
 *   sum_ms   = plain sum of the launch durations (== busy_ms when nothing overlaps);
 *   launches, flops = launch count and the class's algorithmic work (FLOPs or bytes, see RGFM_KCLASS_*). */
int rgfm_profile_read(int kclass, double* busy_ms, double* sum_ms, int64_t* launches, double* flops);

/* Pre-creates the hipEvents of `launches` timed launches, so that none is created inside a timed region. */
int rgfm_profile_reserve(int64_t launches);

/* Measured ceilings of the current device, for the roofline line (each call takes ~0.2-0.5 s and synchronises):
 *   rgfm_ubench_mfma_f16: sustained dense f16 MFMA rate (v_mfma_f32_32x32x16_f16, operands in registers, random
 *     data, two waves per SIMD on every CU, >= 0.2 s) in TFLOP/s -- what the chip holds under DVFS, as opposed to
 *     the 2500 TFLOP/s nominal peak;
 *   rgfm_ubench_hbm_copy: float4 copy of `bytes` (>= 256 MiB recommended) in GB/s of read + written bytes. */
int rgfm_ubench_mfma_f16(double* tflops);
int rgfm_ubench_hbm_copy(size_t bytes, double* gbps);

/* Conv arithmetic of ONE handle (see "arithmetic" above); RGFM_CONV_DEFAULT follows the RGFM_CONV environment
 * variable (hx2 | bx3 | f32, read once per API call; unset = hx2).  A handle setting, not process state: two host
 * threads driving two handles never change each other's arithmetic. */
#define RGFM_CONV_DEFAULT (-1)
#define RGFM_CONV_HX2 0
#define RGFM_CONV_BX3 1
#define RGFM_CONV_F32 2
int rgfm_unet_set_conv_mode(rgfm_unet* h, int mode);
int rgfm_fmnet_set_conv_mode(rgfm_fmnet* h, int mode);

/* Range flag of the default fp16 conv path, one word per handle: waits for `stream`, then *flagged = the bits raised
 * by the handle's launches since the last reset -- 1: a staged activation reached |a| >= 2048 (fp16 overflow);
 * 2: an output that a later conv stages without a GroupNorm in front (the residual stream: reference
 * src/models/unet_flexible.py:85,96,107-108 consume it in fp32 at any magnitude) had a 64-pixel wave block whose
 * largest |value| was below 2^-8, where the two fp16 planes lose bits.  Non-zero: the results of those calls are not
 * fp32-class; repeat them with rgfm_*_set_conv_mode(h, RGFM_CONV_BX3) (bit 0) or RGFM_CONV_F32 (bit 1: the exact fp32
 * convs are the reference's arithmetic at any magnitude).  reset != 0 clears the word. */
int rgfm_unet_range_flag(rgfm_unet* h, int* flagged, int reset, rgfm_stream_t stream);
int rgfm_fmnet_range_flag(rgfm_fmnet* h, int* flagged, int reset, rgfm_stream_t stream);

int rgfm_abi_version(void);
const char* rgfm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RGFM_H_ */
