/*
 * tem_hip.h -- C ABI of libtem_hip.so, the MI355X (gfx950) implementation of the
 * arithmetic that transfer_em's CycleGAN hot path delegates to TensorFlow.
 *
 * The reference (janelia-flyem/transfer_em) has no FFI of its own: its hot path is
 * Python calling tf.keras layers.  Each entry point below replaces
 * the TF kernels behind the cited reference call sites (paths relative to the
 * reference root).  Conventions for every function:
 *
 *   - plain C types only; all pointers are DEVICE pointers owned by the caller;
 *   - activations are float32, channels-last (N,D,H,W,C); 2-D data is D == 1, kd == 1;
 *   - work is enqueued on `stream` (a hipStream_t); nothing allocates, frees or
 *     synchronises, so every call can be captured into a hipGraph;
 *   - returns 0 on success, a negative TEM_E* code for a rejected argument, or a
 *     positive hipError_t from the launch.
 *
 * INTEGRATION.md shows the ctypes binding the reference-side Python would add.
 */
#ifndef TEM_HIP_H
#define TEM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TEM_ABI_VERSION 1

#define TEM_OK            0
#define TEM_EINVAL       -1   /* malformed descriptor (NULL pointer, negative dim, ...) */
#define TEM_EUNSUPPORTED -2   /* channel count / kernel size outside the compiled set   */
#define TEM_ESHAPE       -3   /* tensor extents inconsistent with the operator geometry  */

typedef void *tem_stream_t;   /* hipStream_t */

/* Strided view of a float32 NDHWC tensor.  Strides are in elements; the channel
 * stride is 1.  A crop is a view (ptr offset + parent strides); a channel slice is
 * a view with C smaller than sW. */
typedef struct tem_view {
  float  *ptr;
  int32_t N, D, H, W, C;
  int64_t sN, sD, sH, sW;
} tem_view;

/* Fused epilogue shared by the convolution entry points.  With acc the convolution
 * result for one output element of out0 (bias already added):
 *
 *     v = acc + add[element - add_off]        (if add.ptr; zero outside add's window)
 *     v = gate > 0 ? v : gate_slope * v       (if gate.ptr: LeakyReLU *gradient*,
 *                                              gated on the saved forward output)
 *     v = keep(element) ? 2 v : 0             (if dropout: Philox4x32-10 stream
 *                                              (seed, site, step); element = dense
 *                                              NDHWC index in out0, or in the
 *                                              full tensor out0 is a window of)
 *     v = v > 0 ? v : slope * v               (forward LeakyReLU; slope 1 = linear)
 *
 * Channels written to out1 (the second half of a split output) skip add/gate/
 * dropout/slope and receive the raw accumulator.
 *
 * Replaces: tf.keras.layers.LeakyReLU (models/utils.py:77,83,126,135;
 * generator.py:57,99,109; discriminator.py:53,74,94), Dropout(0.5)
 * (models/utils.py:134) and their gradients, Cropping/Concatenate gradients
 * (generator.py:74-86).
 */
typedef struct tem_epilogue {
  const float *bias;          /* [C_out] or NULL (discriminator.py:97-99 only)        */
  float        slope;
  tem_view     gate;          /* same extents as out0, or ptr == NULL                 */
  float        gate_slope;
  tem_view     add;           /* window placed at add_off inside out0, or ptr == NULL */
  int32_t      add_off[3];
  int32_t      dropout;       /* 0 / 1                                                */
  uint64_t     seed;
  uint32_t     site;
  uint32_t     step;
  const uint32_t *step_dev;   /* if non-NULL the step is read from device memory
                                 (graph replay) and `step` is ignored               */
  /* When out0 is a window of a larger logical tensor (region-restricted execution of the cycle
   * path), the dropout element index is taken in the frame of that full tensor: voxel
   * (z,y,x) of out0 is voxel (z,y,x) + drop_org of a tensor with spatial extents drop_dims.
   * drop_dims[0] == 0: out0 itself is the full tensor. */
  int32_t      drop_org[3];
  int32_t      drop_dims[3];
  /* Optional keep mask of the dropout stream: one bit per element of the dropout frame, bit (e & 7) of byte
   * (e >> 3) for dense element index e (C_out a multiple of 8: a voxel's channels start on a byte).
   *   keep_mode 1: the kernel also WRITES the bits it draws (the forward pass of Dropout);
   *   keep_mode 2: the kernel MAY read the bits instead of re-running Philox (the matching input-gradient; a
   *                pure speed hint -- the bits are the same either way);
   *   keep_mode 0 / keep_mask NULL: no mask.
   * Mode 1 is honoured or the call returns TEM_EUNSUPPORTED. */
  uint8_t     *keep_mask;
  int32_t      keep_mode;
} tem_epilogue;

/* Weight addressing for tem_conv: element (tap, c_in, c_out) of the operator's
 * kernel is w[tap' * C_a * C_b + ...] with
 *   TEM_W_TAP_CI_CO : w[tap][c_in][c_out]                  (Keras Conv kernel, forward;
 *                                                           Keras ConvTranspose kernel
 *                                                           used as its own input-grad)
 *   TEM_W_FLIP_CO_CI: w[ntap-1-tap][c_out][c_in]           (input-gradient of a
 *                                                           stride-1 Keras Conv)       */
#define TEM_W_TAP_CI_CO  0
#define TEM_W_FLIP_CO_CI 1
/* TEM_W_WINOGRAD: `w` is the layer's kernel in the Winograd F(2x2, 3x3) domain of the (y, x) axes as tem_winograd_weights wrote it
 * (k 3, s 1, operator channel pairs C_in -> C_out in {8->8, 8->16, 16->8, 16->16, 16->32, 32->16, 32->32}, epilogues LeakyReLU /
 * LeakyReLU' gate, and for 16->16 and 32->32 gate + keep bits + split output; the call returns TEM_EUNSUPPORTED
 * otherwise -- ask tem_conv_is_tiled first).  Same operator and epilogue; the result differs from the direct form by fp32 rounding. */
#define TEM_W_WINOGRAD   2

typedef struct tem_conv_args {
  tem_view in0, in1;          /* logical input = concat(in0, in1) on C; in1.ptr may be NULL */
  const float *w;
  int32_t w_layout;
  int32_t kd, kh, kw;         /* kernel extents, each in {1,3,4}                       */
  int32_t sd, sh, sw;         /* strides, each in {1,2}                                */
  int32_t pd, ph, pw;         /* zero padding; negative == crop                        */
  tem_view out0, out1;        /* output channels [0,out0.C) and [out0.C, out0.C+out1.C) */
  tem_epilogue ep;
} tem_conv_args;

/* out[n,o,co] = sum_{tap,ci} in[n, o*s + tap - p, ci] * W(tap,ci,co)   (cross-correlation)
 *
 * Replaces the TF kernels behind tf.keras.layers.Conv2D/Conv3D forward
 * (models/utils.py:73,80,122; generator.py:54,96,108,110; discriminator.py:45,78,97),
 * Conv3DBackpropInput for the stride-1 layers (TEM_W_FLIP_CO_CI, p = k-1-p_fwd), and
 * Conv3DTranspose's input-gradient (a stride-2 convolution, models/utils.py:129). */
int tem_conv(const tem_conv_args *a, tem_stream_t stream);

/* out[n,o,co] = sum over (j,tap) with o == j*s + tap - p of in[n,j,ci] * w[tap][co][ci]
 *
 * Replaces tf.keras.layers.Conv3DTranspose forward (models/utils.py:129-130; 'same'
 * padding with k 4, s 2 is p = 1) and Conv3DBackpropInput of the stride-2 layers
 * (models/utils.py:80; p = 0, Keras kernel (tap,C_in,C_out) read as [tap][co][ci]).
 * w_layout is ignored. */
int tem_conv_transpose(const tem_conv_args *a, tem_stream_t stream);

/* The shape-generic VALU implementations behind tem_conv / tem_conv_transpose, exported so
 * that tests can compare them with the LDS/MFMA-tiled kernels the dispatcher prefers. */
int tem_conv_direct(const tem_conv_args *a, tem_stream_t stream);
int tem_conv_transpose_direct(const tem_conv_args *a, tem_stream_t stream);

/* 1 if tem_conv / tem_conv_transpose would run these arguments on the LDS/MFMA-tiled kernel
 * (then `name`, if non-NULL, receives the kernel's template name as rocprofv3 prints it), 0 if
 * on the direct kernel.  Used by bench.py to label its per-kernel timings. */
int tem_conv_is_tiled(const tem_conv_args *a, int32_t transposed, char *name, int32_t name_len);

typedef struct tem_bww_args {
  tem_view in0, in1;          /* forward input (concat on C)                           */
  tem_view dout;              /* gradient w.r.t. the pre-activation output             */
  int32_t kd, kh, kw, sd, sh, sw, pd, ph, pw;
  float  *slabs;              /* partial sums: slab s starts at slabs + s*slab_stride   */
  int64_t slab_stride;        /* elements between slabs; 0 == ntap*C_in*C_out (dense).
                                 A whole network shares one [nslab][n_params] workspace
                                 by passing slabs = ws + param_offset, stride = n_params */
  int32_t nslab;              /* number of partial slabs the voxel range is split into */
  int32_t accumulate;         /* 0: slabs are overwritten, 1: added to                 */
} tem_bww_args;

/* slab[s][tap][ci][co] (+)= sum over the s-th share of (n,o) of
 *                           in[n, o*s + tap - p, ci] * dout[n, o, co]
 *
 * Replaces Conv3DBackpropFilter (all Conv layers) and, with in := upstream gradient
 * and dout := the layer input, the kernel gradient of Conv3DTranspose in its Keras
 * layout (tap, C_out, C_in).  The split over slabs is deterministic; tem_reduce_slabs
 * finishes the sum. */
int tem_conv_bwd_weight(const tem_bww_args *a, tem_stream_t stream);

/* Workspace sizing: with a->nslab = the largest split the caller is willing to hold, returns the
 * number of slabs (<= a->nslab) the launch should be given for best occupancy of the 256 CUs
 * (the LDS-tiled kernel writes one slab per workgroup).  Pass the returned value as nslab to
 * tem_conv_bwd_weight.  a->slabs may be NULL here.  Negative: TEM_E*. */
int tem_conv_bwd_weight_nslab(const tem_bww_args *a);

/* Same as tem_conv_is_tiled for tem_conv_bwd_weight (a->nslab must be the value the launch will use). */
int tem_bww_is_tiled(const tem_bww_args *a, char *name, int32_t name_len);

/* out[i] = (accumulate ? out[i] : 0) + scale * sum_s slabs[s*slab_stride + i]
 * (slab_stride 0 == n) */
int tem_reduce_slabs(const float *slabs, int32_t nslab, int64_t n, int64_t slab_stride,
                     float *out, int32_t accumulate, float scale, tem_stream_t stream);

/* One work item of tem_reduce_slabs_multi: out[i] = scale * sum_{s < nslab} slabs[s*stride + i]
 * for i < count (count <= 64). */
typedef struct tem_reduce_item {
  const float *slabs;
  int64_t      stride;
  int32_t      nslab;
  int32_t      count;
  float       *out;
} tem_reduce_item;

/* Finishes the split-K sums of a whole network in ONE launch: `items_dev` is a device array of
 * `nitems` work items (one workgroup each) covering every kernel of the network. */
int tem_reduce_slabs_multi(const tem_reduce_item *items_dev, int32_t nitems, float scale,
                           tem_stream_t stream);

/* The discriminator's 1x1x1 head as one launch per direction (reference discriminator.py:78-99; the launch plans of
 * models/discriminator.py use it in place of two tem_conv, two tem_conv_bwd_weight, one tem_channel_sum and two
 * input-gradient launches on the 8^3 logits map):
 *   forward : p1[v][co] = LeakyReLU_slope(sum_ci e6[v][ci] w1[ci][co]);  z[v] = sum_co p1[v][co] w2[co] + bias[0]
 *   backward: g_p1 = (p1 > 0 ? 1 : slope_p1) dz w2;  g_e6 = (e6 > 0 ? 1 : slope_e6) (g_p1 . w1^T)   (g_e6 may be NULL
 *             only with slabs); with slab_w1 != NULL every workgroup b < nslab writes its partial kernel gradients
 *             slab_w1[b][32*32] (sum_v e6 (x) g_p1), slab_w2[b][32] (sum_v p1 dz), slab_b[b] (sum_v dz) -- rows of the
 *             layers' ordinary slab sets (tem_reduce_slabs_multi sums them); nslab = tem_disc_head_nslab(nvox).
 * All tensors dense float32, channels last, 32 channels (the reference hard-codes them, discriminator.py:60,72,78). */
typedef struct tem_head_bwd_args {
  const float *dz, *e6, *p1, *w1, *w2;
  float *g_e6;
  float slope_p1, slope_e6;
  float *slab_w1, *slab_w2, *slab_b;
  int32_t nslab;
  int64_t nvox;
} tem_head_bwd_args;
int tem_disc_head_nslab(int64_t nvox);
int tem_disc_head_fwd(const float *e6, const float *w1, const float *w2, const float *bias, float *p1, float *z,
                      int64_t nvox, float slope, tem_stream_t stream);
int tem_disc_head_bwd(const tem_head_bwd_args *a, tem_stream_t stream);

/* out[c] = (accumulate ? out[c] : 0) + sum over all (n,d,h,w) of g[...,c]   (bias gradient,
 * discriminator.py:97-99) */
int tem_channel_sum(const tem_view *g, float *out, int32_t accumulate, tem_stream_t stream);

/* tfa.losses.SigmoidFocalCrossEntropy(from_logits=True, alpha=0.5, gamma) with
 * Reduction.AUTO, as used by generator_loss / discriminator_loss (cgan.py:78-79,110-120).
 *   L = mean over z of 0.5 * (1-p_t)^gamma * BCEWithLogits(target, z)
 * losses[k] += loss_scale * L for every bit k set in slot_mask (losses is double[8]);
 * if dz.ptr: dz = grad_scale * dL/dz. */
int tem_focal_logits(const tem_view *z, int32_t target, float gamma,
                     double *losses, uint32_t slot_mask, float loss_scale,
                     const tem_view *dz, float grad_scale, tem_stream_t stream);

/* identity_loss / calc_cycle_loss (cgan.py:122-142): t = 1 - |a-b|/2,
 *   L = mean of 0.5 * (1-t)^gamma * -log(clip(t,eps,1-eps)+eps)   (from_logits=False form)
 * losses[k] += loss_scale * L; if db.ptr: db = grad_scale * dL/db (db has b's extents). */
int tem_focal_match(const tem_view *a, const tem_view *b, float gamma,
                    double *losses, uint32_t slot_mask, float loss_scale,
                    const tem_view *db, float grad_scale, tem_stream_t stream);

/* tf.keras.optimizers.Adam dense update (cgan.py:69-73,218-228), Keras/TF2 form:
 *   t = *step_dev + 1;  lr_t = lr*sqrt(1-b2^t)/(1-b1^t)
 *   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  theta -= lr_t * m / (sqrt(v) + eps)
 * g = grad_scale * grad.  step_dev is NOT incremented here (see tem_step_tick). */
int tem_adam_keras(float *theta, const float *grad, float *m, float *v, int64_t n,
                   float lr, float beta1, float beta2, float eps, float grad_scale,
                   const uint32_t *step_dev, tem_stream_t stream);

/* *step_dev += 1 (one thread). */
int tem_step_tick(uint32_t *step_dev, tem_stream_t stream);

/* Keep bits of (up to two) Dropout(0.5) layers (models/utils.py:134) for one step, written ahead of the transposed
 * convolutions that apply them: bit e of mask = keep(element e) of the Philox4x32-10 stream (seed, site, step) that
 * tem_epilogue.dropout draws from -- the layout tem_epilogue.keep_mask (keep_mode 2) reads.  nbytes = ceil(elements / 128)
 * * 16 (whole Philox blocks; 16-byte aligned); mask1 may be NULL.  step_dev (device counter) overrides step when non-NULL. */
int tem_dropout_masks(uint8_t *mask0, int64_t nbytes0, uint32_t site0, uint8_t *mask1, int64_t nbytes1, uint32_t site1,
                      uint64_t seed, const uint32_t *step_dev, uint32_t step, tem_stream_t stream);

/* datasets.py:193-202 + 157-163: out = ((float)in / 127.5 - 1 - mean) / std */
int tem_u8_to_f32_std(const uint8_t *in, float *out, int64_t n, float mean, float std,
                      tem_stream_t stream);

/* utils.py:109,118: out = (uint8) rint((y*std + mean + 1) * 127.5)  (wraps mod 256).
 * `y` is a view (tile interior), `out` a dense-by-strides uint8 block with element
 * strides oD,oH,oW (batch 1, channel 0). */
int tem_f32_unstd_to_u8(const tem_view *y, uint8_t *out, int64_t oD, int64_t oH, int64_t oW,
                        float mean, float std, tem_stream_t stream);

/* Tiled inference, input side (utils.py:77-89 with the cloud fetch replaced by a resident uint8 volume
 * vol[Z][Y][X]): tile t of out[ntile][edge][edge][edge] = the window of `vol` at origin
 * (z,y,x) = origins_dev[3t..3t+2] -- voxels outside the volume read as 0 -- converted as tem_u8_to_f32_std. */
int tem_u8_tiles_to_f32_std(const uint8_t *vol, int32_t Z, int32_t Y, int32_t X, const int32_t *origins_dev,
                            int32_t ntile, int32_t edge, float *out, float mean, float std, tem_stream_t stream);

/* Tiled inference, output side (utils.py:107-126): the interior of tile t of y[ntile][yedge]^3 (`tpad` voxels
 * stripped per face, utils.py:113-116) is converted as tem_f32_unstd_to_u8 and written into the uint8 volume
 * out[OZ][OY][OX] at (z,y,x) = index_dev[3t..3t+2].  The caller guarantees the blocks lie inside `out`. */
int tem_f32_tiles_unstd_to_u8(const float *y, int32_t ntile, int32_t yedge, int32_t tpad, const int32_t *index_dev,
                              uint8_t *out, int32_t OZ, int32_t OY, int32_t OX, float mean, float std,
                              tem_stream_t stream);

/* Random augmentation of one cached sample (datasets.py:123-155) with host-drawn parameters:
 *   dst = reverse(transpose(src, perm = (p0,p1,p2)), dims with f_k != 0) * scale + shift
 * src is a dense single-channel (D,H,W) volume (2-D: D == 1, p0 must be 0); dst has extents
 * (dim[p0], dim[p1], dim[p2]) and must not alias src. */
int tem_augment_f32(const float *src, int32_t D, int32_t H, int32_t W, int32_t p0, int32_t p1, int32_t p2,
                    int32_t f0, int32_t f1, int32_t f2, float scale, float shift, float *dst, tem_stream_t stream);

/* debug.warp_tensor (debug.py:7-63) on the device: dst = box blur of src (3 wide per non-unit axis, zero SAME
 * padding); hole seeds ~ Bernoulli(rate) per voxel (Philox stream of `seed`, written to `seeds`, one byte per
 * voxel), dilated by a 4-wide box (SAME: 1 before, 2 after); holes := mean(blurred).  `sum`: one double of
 * workspace.  dst must not alias src. */
int tem_warp_f32(const float *src, int32_t D, int32_t H, int32_t W, float rate, uint64_t seed, float *dst,
                 uint8_t *seeds, double *sum, tem_stream_t stream);

/* dst[i] = value */
int tem_fill_f32(float *dst, int64_t n, float value, tem_stream_t stream);

/* dst(view) = src(view) elementwise; same extents (crop / pad / channel-slice copies) */
int tem_copy_view(const tem_view *src, const tem_view *dst, tem_stream_t stream);

/* dst(view) += src(view) */
int tem_add_view(const tem_view *src, const tem_view *dst, tem_stream_t stream);

/* One convolution kernel inside a flat parameter vector: `ntap` taps of a [ci][co] block at element `offset`. */
typedef struct tem_wlayer {
  int64_t offset;
  int32_t ntap, ci, co;
} tem_wlayer;

/* theta_t := theta with every listed kernel's taps reversed and (ci, co) transposed,
 *   theta_t[off + ((ntap-1-t)*co + b)*ci + a] = theta[off + (t*ci + a)*co + b];
 * elements outside the table (biases) are copied.  The input-gradient of a stride-1 convolution is then a plain
 * TEM_W_TAP_CI_CO convolution over theta_t -- same numbers as TEM_W_FLIP_CO_CI over theta, but the kernel-tap
 * fragments are read as contiguous 64-byte runs (measured 27-33 % faster on the LDS-tiled kernels).  One launch
 * per network and step, after the optimizer update (reference: the transposed filter cuDNN/Eigen build inside
 * Conv3DBackpropInputV2).  `layers_dev` is device memory. */
int tem_flip_transpose(const float *theta, float *theta_t, const tem_wlayer *layers_dev, int32_t nlayers,
                       int64_t total, tem_stream_t stream);

/* One kernel of tem_winograd_weights: the 27 taps of a [ci][co] block at theta + src_off (flip = 0: the operator's
 * kernel is theta[tap][ci][co], a forward Conv layer; flip = 1: theta[26-tap][co][ci], its input-gradient) are
 * transformed to U[kz] = G g[kz] G^T on the (y, x) axes (16 points per z tap) and written at u + dst_off in the fragment
 * order of the Winograd kernel: (ci / 8) * ceil(co / 16) * 6144 floats per layer (8 -> 8: 8192), ci, co the operator's
 * channel counts (8, 16 or 32; 32 -> 16 is not built). */
typedef struct tem_wino_layer {
  int64_t src_off, dst_off;
  int32_t ci, co, flip;
} tem_wino_layer;

/* Once per network and step, after the optimizer update (as tem_flip_transpose): the Winograd-domain copies of the
 * listed kernels.  `layers_dev` is device memory.  (The reference leaves the choice of convolution algorithm to
 * TF/cuDNN, which picks Winograd forms for 3x3 kernels the same way.) */
int tem_winograd_weights(const float *theta, float *u, const tem_wino_layer *layers_dev, int32_t nlayers,
                         tem_stream_t stream);

/* Kernel gradient of a k 3, s 1 layer (C_in -> C_out: 16 -> 16, 8 -> 16, 8 -> 8) in the Winograd domain of
 * tem_winograd_weights (2.25x fewer matrix flops than tem_conv_bwd_weight; same operator, fp32 rounding differs).  Same
 * slab contract as tem_conv_bwd_weight: the launch writes exactly tem_conv_bwd_weight_winograd_nslab(a) slabs (query
 * with a->nslab = the caller's upper bound; negative: TEM_E*, e.g. TEM_EUNSUPPORTED for other geometries; `name`, if
 * non-NULL, receives the kernel's name), one per workgroup, already transformed back to [27][ci][co]; pass that value
 * as a->nslab.  Deterministic (fixed summation orders). */
int tem_conv_bwd_weight_winograd_nslab(const tem_bww_args *a, char *name, int32_t name_len);
int tem_conv_bwd_weight_winograd(const tem_bww_args *a, tem_stream_t stream);

/* g(view) = saved(view) > 0 ? g : slope * g, in place: LeakyReLU gradient gated on the saved output where no
 * convolution epilogue can carry it (gradient entering the frozen prior network, discriminator.py:62-66). */
int tem_leaky_gate_view(const tem_view *g, const tem_view *saved, float slope, tem_stream_t stream);

/* ---- bf16 mixed precision (BASELINE config 5; the reference is fp32 only, cgan.py:13-14) -------------------------
 * Activations, gate / add views and the kernel copy are bfloat16: the `float *` fields of tem_view / tem_conv_args
 * carry bf16 pointers, strides stay in ELEMENTS.  `w` is the layer's kernel packed [tap][co][ci] in bf16
 * (tem_pack_weights_bf16); w_layout == TEM_W_FLIP_CO_CI reverses the taps (input-gradient of a stride-1 layer over
 * the un-transposed copy).  Accumulation, bias, LeakyReLU / gate / dropout run in fp32; outputs are rounded to
 * bf16 (nearest even) on store.  Same operator definition and epilogue order as tem_conv. */
int tem_conv_bf16(const tem_conv_args *a, tem_stream_t stream);

/* TEM_OK and the kernel's name (as rocprofv3 prints it) if tem_conv_bf16 accepts these arguments. */
int tem_conv_bf16_describe(const tem_conv_args *a, char *name, int32_t name_len);

/* bf16 form of tem_conv_transpose (k4 s2 only): `w` = bf16 kernel [tap][co][ci]. */
int tem_conv_transpose_bf16(const tem_conv_args *a, tem_stream_t stream);
int tem_conv_transpose_bf16_describe(const tem_conv_args *a, char *name, int32_t name_len);

/* bf16 form of tem_conv_bwd_weight: in0 / in1 / dout are bf16 views, the partial slabs are float32 (the split-K
 * finish, the gradient vector and Adam stay fp32).  The launch writes exactly the number of slabs that
 * tem_conv_bwd_weight_bf16_nslab(a) returns for a->nslab = the caller's upper bound (negative: TEM_E*; `name`, if
 * non-NULL, receives the kernel's name); pass that value back as a->nslab. */
int tem_conv_bwd_weight_bf16(const tem_bww_args *a, tem_stream_t stream);
int tem_conv_bwd_weight_bf16_nslab(const tem_bww_args *a, char *name, int32_t name_len);

/* dst[i] = bf16(src[i]) (round to nearest even) */
int tem_cast_f32_to_bf16(const float *src, void *dst, int64_t n, tem_stream_t stream);

/* The per-step bf16 kernel copies of one network: theta_h = bf16(theta), same layout; theta_ht = every kernel of
 * the table with its last two axes transposed ([tap][A][B] -> [tap][B][A]).  tem_conv_bf16 contracts over the LAST
 * axis of the copy it is given: forward of a Conv layer reads theta_ht, its stride-1 input-gradient theta_h with
 * TEM_W_FLIP_CO_CI; a ConvTranspose layer (and the input-gradient of a stride-2 Conv through
 * tem_conv_transpose_bf16) reads theta_h, the ConvTranspose input-gradient theta_ht. */
int tem_pack_weights_bf16(const float *theta, void *theta_h, void *theta_ht, const tem_wlayer *layers_dev,
                          int32_t nlayers, int64_t total, tem_stream_t stream);

/* bf16 forms of tem_focal_logits / tem_focal_match / tem_copy_view / tem_add_view / tem_channel_sum: bf16 views in
 * and out, fp32 arithmetic, double loss sums. */
int tem_focal_logits_bf16(const tem_view *z, int32_t target, float gamma, double *losses, uint32_t slot_mask,
                          float loss_scale, const tem_view *dz, float grad_scale, tem_stream_t stream);
int tem_focal_match_bf16(const tem_view *a, const tem_view *b, float gamma, double *losses, uint32_t slot_mask,
                         float loss_scale, const tem_view *db, float grad_scale, tem_stream_t stream);
int tem_copy_view_bf16(const tem_view *src, const tem_view *dst, tem_stream_t stream);
int tem_add_view_bf16(const tem_view *src, const tem_view *dst, tem_stream_t stream);
int tem_channel_sum_bf16(const tem_view *g, float *out, int32_t accumulate, tem_stream_t stream);

/* InstanceNormalization (models/utils.py:10-38; defined there, every call site commented out at
 * models/utils.py:75-76,81-82,124-125,131): per sample and channel, over the spatial axes,
 *   mean, variance = tf.nn.moments(x);  y = scale[c] * (x - mean) * rsqrt(variance + eps) + offset[c].
 * mean / rstd ([N*C] each) are written for the backward pass. */
int tem_instance_norm(const tem_view *x, const float *scale, const float *offset, float eps,
                      const tem_view *y, float *mean, float *rstd, tem_stream_t stream);

/* Gradient of tem_instance_norm: dx (may alias dy), dscale[C], doffset[C] (either may be NULL).
 * workspace: 2*N*C doubles. */
int tem_instance_norm_bwd(const tem_view *x, const tem_view *dy, const float *scale, const float *mean,
                          const float *rstd, const tem_view *dx, float *dscale, float *doffset,
                          double *workspace, tem_stream_t stream);

/* Library identification: returns TEM_ABI_VERSION; *arch (if non-NULL) receives a
 * static string naming the compiled offload target ("gfx950"). */
int tem_abi_version(const char **arch);

#ifdef __cplusplus
}
#endif
#endif /* TEM_HIP_H */
