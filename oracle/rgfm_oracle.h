/*
 * rgfm_oracle.h -- CPU restatement (plain C, fp32, NCHW) of the reference's
 * ratio-guided flow-matching sampler path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the shipped path
 * (ratio_guided_multimodal_fm_amd/ -> csrc/librgfm_hip.so) never does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * below against golden vectors produced by importing the reference itself in
 * the build container (tests/golden/make_golden.py).
 *
 * Parameter blobs are the reference module's state_dict() tensors, flattened
 * and concatenated in state_dict() order (same convention as include/rgfm.h).
 */
#ifndef RGFM_ORACLE_H_
#define RGFM_ORACLE_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ro_unet_desc {
  int32_t in_channels, img_size, model_channels, num_levels;
  int32_t channel_mult[4];
  int32_t num_res_blocks;
} ro_unet_desc;

size_t ro_unet_param_floats(const ro_unet_desc* d);
/* activations in production order: input_conv; per ResBlock {conv1+temb, block out};
 * Downsample / Upsample outputs; network output (last). */
int ro_unet_num_activations(const ro_unet_desc* d);
void ro_unet_activation_shape(const ro_unet_desc* d, int idx, int* c, int* h, int* w);
/* out[B,C,H,W] = model(x, t); t has t_count in {1,B} entries; acts (optional)
 * is an array of ro_unet_num_activations() pointers to [B,c,h,w] buffers. */
void ro_unet_forward(const ro_unet_desc* d, const float* params, const float* x, const float* t,
                     int t_count, float* out, int B, float** acts);
void ro_timestep_embedding(const float* t, int n, int dim, float* out);

#define RO_RATIO_MNIST_SVHN 0
#define RO_RATIO_MNIST28 1
size_t ro_ratio_param_floats(int kind, int feature_dim, int hidden_dim);
/* what: 0 score, 1 log_ratio, 2 exp(log_ratio); loss: 0 disc, 1 rulsif.
 * feat (optional): [n, 2*feature_dim] encoder features (x then y). */
void ro_ratio_eval(int kind, int feature_dim, int hidden_dim, int loss, const float* params,
                   const float* x, const float* y, float* out, int n, int what, float* feat);

/* d log_ratio / d(x, y) (eval mode) of RatioEstimatorMNISTSVHN (kind 0: gx [n,1,32,32], gy [n,3,32,32]) or
 * RatioEstimator (kind 1: gx, gy [n,1,28,28]); log_ratio optional [n] */
void ro_ratio_grad(int kind, int feature_dim, int hidden_dim, int loss, const float* params, const float* x, const float* y,
                   float* gx, float* gy, float* log_ratio, int n);
/* paired Euler loop with gradient log-ratio guidance: x <- x + (v_x + gamma dlogr/dx) dt (README.md:159-164) */
void ro_sample_pair_grad(const ro_unet_desc* dx, const float* px, const ro_unet_desc* dy, const float* py, int kind,
                         int feature_dim, int hidden_dim, int loss, const float* pr, float* x, float* y, int B,
                         int num_steps, double gamma, int step_begin, int step_end);

/* vx, vy <- (1-gamma) v + gamma g at time t (sample_mnist_svhn.py:124-171);
 * weights (optional) [B,N]. */
void ro_guidance_apply(const float* x, const float* y, float* vx, float* vy, const float* mc_x1,
                       const float* mc_y1, const float* mc_ratios, int B, int N, int dx, int dy,
                       double t, double gamma, float* weights);

void ro_sample_single(const ro_unet_desc* d, const float* params, float* x, int B, int num_steps,
                      int step_begin, int step_end);
void ro_sample_pair(const ro_unet_desc* dx, const float* px, const ro_unet_desc* dy, const float* py,
                    float* x, float* y, const float* mc_x1, const float* mc_y1,
                    const float* mc_ratios, int n_mc, int B, int num_steps, double gamma,
                    int step_begin, int step_end);
/* FlowMatchingModel ("--model original", src/models/flow_matching.py), 1x28x28, feature 256, time 128 */
size_t ro_fm_param_floats(void);
void ro_fm_time_embedding(const float* t, int n, int dim, float* out);
void ro_fm_forward(const float* params, const float* x, const float* t, int t_count, float* out, int B);
int ro_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
