/* dmmfods_hip.h -- C ABI of libdmmfods_hip.so: the MI355X (gfx950) implementation of the DMMFODS
 * Dense_U_Net_lidar training hot path.
 *
 * The reference (p-mc-grath/DMMFODS) is pure Python on torch.nn and has NO FFI of its own; its drop-in
 * boundary is the Python surface of  dmmfods/graphs/models/Dense_U_Net_lidar.py  and
 * dmmfods/agents/Dense_U_Net_lidar_Agent.py.  Each entry point below names the reference interface it
 * stands behind (paths relative to the reference tree):
 *
 *   dmm_plan_create / dmm_plan_destroy ... Dense_U_Net_lidar.__init__   graphs/models/Dense_U_Net_lidar.py:29-208
 *   dmm_plan_tensor_*  .................... nn.Module.state_dict() key/shape layout   (same file, :71-192)
 *   dmm_plan_forward ..................... Dense_U_Net_lidar.forward    graphs/models/Dense_U_Net_lidar.py:210-267
 *   dmm_plan_loss_backward ............... BCEWithLogitsLoss(reduction='none') + metrics + backward(ones)
 *                                          agents/Dense_U_Net_lidar_Agent.py:247-264, utils/...helper.py:311-401
 *   dmm_plan_set_loss / dmm_loss_forward . FocalLoss / ClassWiseFocalLoss  graphs/losses/FocalLoss.py:9-91
 *   dmm_adam_step ........................ torch.optim.Adam.step        agents/Dense_U_Net_lidar_Agent.py:57-61,265
 *   dmm_conv_forward / dmm_conv_wgrad .... single-kernel entry points for unit tests (torch.nn.functional.conv2d,
 *                                          conv_transpose2d as dispatched by the modules built at :72-131)
 *
 * Conventions: every function returns 0 on success and a negative dmm_status otherwise; dmm_last_error()
 * gives the message (thread-local).  The caller owns all buffers (device pointers are plain void*); the
 * library owns only the opaque plan.  A plan is bound to one device and is not thread-safe.  All launches go
 * to the hipStream_t passed as `stream` (a void* here so that the header needs no HIP include).
 */
#ifndef DMMFODS_HIP_H
#define DMMFODS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  DMM_OK = 0,
  DMM_ERR_INVALID = -1,      /* bad argument / unsupported configuration (Python raises AttributeError/ValueError) */
  DMM_ERR_SHAPE = -2,        /* spatial size not a multiple of 32 (reference: ValueError from ConvTranspose2d) */
  DMM_ERR_HIP = -3,          /* a HIP call failed */
  DMM_ERR_STATE = -4,        /* plan not bound / wrong call order */
  DMM_ERR_NO_DEVICE = -5
} dmm_status;

enum { DMM_F32 = 0, DMM_F16 = 1, DMM_BF16 = 2 };

/* tensor kinds in the state_dict table */
enum { DMM_T_CONV = 0, DMM_T_CONVT = 1, DMM_T_BN_WEIGHT = 2, DMM_T_BN_BIAS = 3, DMM_T_BN_MEAN = 4, DMM_T_BN_VAR = 5,
       DMM_T_BN_TRACKED = 6 };

/* Mirrors config.model.* of the reference (utils/Dense_U_Net_lidar_helper.py:110-123) plus the run shape. */
typedef struct {
  int32_t growth_rate;
  int32_t num_blocks;
  int32_t block_config[8];
  int32_t num_init_features;
  int32_t bn_size;
  int32_t num_classes;
  int32_t concat_before_block_num;
  int32_t stream_1_in_channels;
  int32_t stream_2_in_channels;
  int32_t batch, height, width; /* per-GPU minibatch and input size (H, W multiples of 32) */
  int32_t dtype;                /* DMM_F32 (parity), DMM_F16 or DMM_BF16 (storage/MFMA type; fp32 accumulate) */
  float loss_scale;             /* multiplies d(loss)/d(logit); gradients are un-scaled before they are returned */
  float bn_momentum, bn_eps;    /* 0.1, 1e-5 */
  float iou_threshold;          /* config.agent.iou_threshold, 0.7, applied to raw logits (reference quirk) */
  int32_t use_mfma;             /* 1 = MFMA kernels; 0 = scalar check kernels (bring-up / debugging) */
} dmm_model_desc;

typedef struct dmm_plan dmm_plan;

const char* dmm_last_error(void);
int dmm_version(void);
/* Switches for tests and A/B timing: "graph" (1 = replay captured launch lists, 2 = the forward list only; see dmm_plan_num_graph_replays), "overlap_wgrad" (1 = weight-gradient GEMMs on a second stream beside the
 * data-gradient chain, 0 = one stream; read at every call), and the kernel families "thin_logits" (gather-once kernel for the
 * heat-map head's last convolution), "conv3" (LDS halo-tile kernels of the multi-tap convolutions), "wg3" (the growth convolution's
 * weight gradient), "wgp" (weight gradients of the parity-phase convolutions), "wg5" (of the 5x5 head / 7x7 stem convolutions),
 * "cvp" (the ConvTranspose kernels), "bw1" (fused backward of the 1x1 bottleneck convolutions), "pig" (persistent forward of the
 * 1x1 convolutions): 1 = on, 0 = generic kernels.
 * A plan chooses the family of each of its launches ONCE, in dmm_plan_bind, from the switches of that moment, and keeps it
 * (labels reported by dmm_plan_profile_op name the kernels that really run): toggle a switch BEFORE binding a plan.  The
 * single-kernel entry points below read the switches at every call.  "grad_bucket_mb": size of the data-parallel gradient
 * buckets of plans created afterwards.  Returns DMM_ERR_INVALID for an unknown name.  Results are identical up to the fp32
 * summation order. */
int dmm_set_option(const char* name, int value);

/* Plan construction needs no GPU: it derives the layer table, the state_dict layout and the workspace size. */
int dmm_plan_create(const dmm_model_desc* desc, dmm_plan** out);
/* Teardown (nothing upstream: the reference never frees a model explicitly, agents/Dense_U_Net_lidar_Agent.py:442-450).  The call
 * first SYNCHRONISES every helper stream the plan has launched on, so when it returns nothing the library enqueued outside the
 * caller's own stream still touches the workspace or the arenas; what was enqueued on the `stream` arguments of earlier calls is
 * the caller's to order before it frees those buffers.  The plan owns no stream: the helper streams belong to a per-device,
 * process-lifetime pool and the plan's events go back to it.  Returns DMM_OK, or DMM_ERR_HIP naming the first HIP call that
 * failed - the plan is gone either way and the handle must not be used again.  dmm_plan_destroy(NULL) is DMM_OK.
 * DMM_TRACE_DESTROY=1 in the environment writes one line per teardown step to stderr. */
int dmm_plan_destroy(dmm_plan* plan);

/* state_dict layout, in the reference's registration order */
int dmm_plan_num_tensors(const dmm_plan* plan);
/* shape has 4 entries (unused = 0), *arena_offset is in elements of the param arena (trainable tensors) or of the
 * buffer arena (running_mean / running_var); num_batches_tracked has no arena slot (offset -1). */
int dmm_plan_tensor_info(const dmm_plan* plan, int index, const char** name, int32_t* kind, int32_t* ndim,
                         int64_t shape[4], int64_t* arena_offset);
int64_t dmm_plan_num_params(const dmm_plan* plan);        /* elements of the param / grad arena */
int64_t dmm_plan_num_buffer_elems(const dmm_plan* plan);  /* elements of the running-stat arena */
size_t dmm_plan_workspace_bytes(const dmm_plan* plan);
double dmm_plan_forward_flops(const dmm_plan* plan);      /* 2*MACs of all convolutions, whole batch */

/* Bind device memory: workspace (>= workspace_bytes, 256-byte aligned), fp32 param arena, fp32 grad arena,
 * fp32 running-stat arena.  May be called again after a re-allocation. */
int dmm_plan_bind(dmm_plan* plan, void* workspace, size_t workspace_bytes, float* params, float* grads, float* buffers);

/* stream_1 (B,s1,H,W) and stream_2 (B,s2,H,W; may be NULL when s2 == 0) are fp32 NCHW device tensors;
 * logits_out is (B,num_classes,H,W) fp32 NCHW.  training != 0: batch statistics + running-stat update. */
int dmm_plan_forward(dmm_plan* plan, const float* stream_1, const float* stream_2, float* logits_out, int training,
                     void* stream);

/* After a training-mode forward: per-pixel BCE against target (B,num_classes,H,W fp32), metric counts, and the
 * backward pass of the SUM of all loss elements.  Gradients land in the bound grad arena (fully overwritten).
 * metrics_out (device, doubles): [NC loss sums | NC equal-counts | B x (NC intersections, NC unions)]. */
int dmm_plan_loss_backward(dmm_plan* plan, const float* logits, const float* target, double* metrics_out, void* stream);

/* Loss epilogue of dmm_plan_loss_backward / dmm_plan_loss_metrics.  DMM_LOSS_BCE (default): BCEWithLogitsLoss(reduction=
 * 'none'), agents/Dense_U_Net_lidar_Agent.py:54.  DMM_LOSS_FOCAL: alpha[c]*(1 - exp(-bce))**gamma[c]*bce per class c
 * (graphs/losses/FocalLoss.py:41-50 with equal entries, ClassWiseFocalLoss :78-91 otherwise); nclass = num_classes. */
enum { DMM_LOSS_BCE = 0, DMM_LOSS_FOCAL = 1 };
int dmm_plan_set_loss(dmm_plan* plan, int kind, const float* alpha, const float* gamma, int nclass);

/* The same loss kernel without a plan, on any (batch, nclass <= 8, height, width) fp32 NCHW tensors: unreduced loss
 * (loss_out, nullable) and d(sum of loss)/d(input) (dinput_out, nullable).  from_prob != 0: `input` holds probabilities
 * (FocalLoss(logits=False): F.binary_cross_entropy, graphs/losses/FocalLoss.py:43-44). */
int dmm_loss_forward(int kind, int from_prob, const float* alpha, const float* gamma, const float* input, const float* target,
                     float* loss_out, float* dinput_out, int batch, int nclass, int height, int width, void* stream);

/* Backward from an externally computed d(loss)/d(logit) (B,num_classes,H,W fp32), e.g. from torch autograd of any
 * loss on the returned logits (reference: loss.backward(...), agents/Dense_U_Net_lidar_Agent.py:264). */
int dmm_plan_backward(dmm_plan* plan, const float* dlogits, void* stream);

/* Data-parallel training (new here; the reference's torch.distributed import, graphs/models/Dense_U_Net_lidar.py:7, is unused):
 * the gradient arena is cut into buckets of whole tensors (about dmm_set_option("grad_bucket_mb", 25) each, set before
 * dmm_plan_create) listed in the order in which backward finishes them (head, decoder, block 4 ... stems).
 * dmm_plan_grad_bucket_wait makes `stream` wait until bucket `index` of the most recently enqueued backward is final, so an
 * all-reduce enqueued on that stream runs beside the rest of backward. */
int dmm_plan_num_grad_buckets(const dmm_plan* plan);
int dmm_plan_grad_bucket(const dmm_plan* plan, int index, int64_t* offset, int64_t* count);
int dmm_plan_grad_bucket_wait(dmm_plan* plan, int index, void* stream);

/* Per-launch timing with HIP events recorded on the launch stream (used by bench.py for the roofline block).
 * which: 0 = training forward, 1 = loss + backward.  profile_begin(plan, n) arms recording for the next n passes of
 * each; profile_collect sums per-op milliseconds over the recorded passes (synchronise the stream first).
 * profile_filter(plan, "igemm.bnbwd.n128/") restricts the event pairs to the launches whose label starts with the prefix
 * (NULL or "" = every launch).  Unfiltered profiling serialises everything on one stream so that each pair brackets one
 * kernel alone; a filtered profile runs exactly as in production (weight gradients on the side stream) and costs two
 * event records per selected launch. */
int dmm_plan_profile_begin(dmm_plan* plan, int max_passes);
int dmm_plan_profile_filter(dmm_plan* plan, const char* label_prefix);
int dmm_plan_profile_num_ops(const dmm_plan* plan, int which);
int dmm_plan_profile_op(const dmm_plan* plan, int which, int index, const char** label, double* flops, double* bytes);
int dmm_plan_profile_collect(dmm_plan* plan, int which, double* ms_sum, int n, int* passes);

/* Loss + metrics only (validation). */
int dmm_plan_loss_metrics(dmm_plan* plan, const float* logits, const float* target, double* metrics_out, void* stream);

/* Launch-list replay.  dmm_plan_forward (training) and dmm_plan_loss_backward run their ~350 / ~750 launches eagerly the first time;
 * the second time, the part of the list that touches no caller pointer (everything between the stem convolution and the logits
 * kernel; everything behind the loss kernel; as one chain on one stream) is captured once into a hipGraph
 * and replayed by ONE call from then on, whatever tensors the caller passes.  OFF by default (dmm_set_option("graph", 1) /
 * DMM_GRAPH=1 turn it on): measured on MI355X it cuts the host's enqueue time of a step from 14-25 ms to 0.5 ms but not the GPU's
 * time, and the replayed chain is single-stream, i.e. slower than the eager two-stream schedule (capi.cpp, launch_list).  Any
 * dmm_set_option drops the captured graphs.  Not replayed: profiled passes, the eval forward,
 * dmm_plan_backward, and the backward of a plan whose gradient buckets are waited for (dmm_plan_grad_bucket_wait: data-parallel
 * overlap needs the bucket events at their place inside the list).  which: 0 training forward, 1 loss + backward. */
long long dmm_plan_num_graph_replays(const dmm_plan* plan, int which);

/* Flat fused Adam over n fp32 elements (amsgrad unsupported).  step is 1-based. */
int dmm_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream);

/* ---- single-kernel entry points (unit tests) ---- */
typedef struct {
  int32_t dtype, use_mfma;
  int32_t B, H, W;         /* input spatial size */
  int32_t Cin, Cout;       /* real channels; Cin is a multiple of 8 */
  int32_t R, S, stride, pad;
  int32_t transposed;      /* 1: ConvTranspose2d(k=3, s=2, p=1, output_padding=1), weight (Cin, Cout, 3, 3) */
  int32_t mode;            /* 0 plain, 1 nearest-upsample x2 source, 2 avg-pool 2x2 then 1x1 */
  int32_t bn_relu;         /* apply relu(x*scale+shift) to the input first */
} dmm_conv_desc;

/* x: T NHWC (B,H,W,Cin); w: fp32 master weights; y: T NHWC output; stats: 2*Cout doubles (sum, sumsq; zeroed by the
 * callee) or NULL; scratch: >= dmm_conv_scratch_bytes. */
size_t dmm_conv_scratch_bytes(const dmm_conv_desc* d);
int dmm_conv_forward(const dmm_conv_desc* d, const void* x, const float* w, const float* scale, const float* shift, void* y,
                     double* stats, void* scratch, void* stream);
/* dw: fp32, master layout, overwritten.  dy: T NHWC gradient of the conv output. */
int dmm_conv_wgrad(const dmm_conv_desc* d, const void* x, const void* dy, const float* scale, const float* shift, float* dw,
                   void* scratch, void* stream);
/* BN+ReLU-fused data gradient: gx (T, NHWC like x) = scale * relu'(x*scale+shift) * conv_dgrad(dy); red: 2*Cin doubles
 * (sum dz, sum dz*xhat with xhat = (x - mean)*invstd; zeroed by the callee).  `shift` points at 3*Cin floats:
 * shift, mean, invstd. */
int dmm_conv_dgrad(const dmm_conv_desc* d, const void* x, const void* dy, const float* w, const float* scale, const float* shift,
                   void* gx, double* red, void* scratch, void* stream);


/* The same two gradients as the plan's launches see them (autograd of torch.nn.functional.conv2d behind the reference's
 * BatchNorm2d + ReLU, graphs/models/Dense_U_Net_lidar.py:85-92 via torchvision _DenseLayer):
 *   - the incoming gradient may carry the deferred BatchNorm-backward correction of the layer behind it:
 *     dy_eff = dy + q[c] + r[c] * yfwd[.., c]  (yfwd: T NHWC forward output of this convolution; q, r: Cout floats; all NULL = none);
 *   - transposed_form != 0: the weight gradient with the taps on the gradient side, the form the plan uses for thin outputs
 *     (the dense layers' 3x3 growth convolution: wg3.hip in 16-bit storage);
 *   - accumulate != 0: gx += instead of gx =. */
int dmm_conv_wgrad_ex(const dmm_conv_desc* d, const void* x, const void* dy, const float* scale, const float* shift, const void* yfwd,
                      const float* q, const float* r, int transposed_form, float* dw, void* scratch, void* stream);
int dmm_conv_dgrad_ex(const dmm_conv_desc* d, const void* x, const void* dy, const float* w, const float* scale, const float* shift,
                      const void* yfwd, const float* q, const float* r, void* gx, int accumulate, double* red, void* scratch,
                      void* stream);
/* The head's last convolution (5x5, 64 channels onto <= 4 classes behind BatchNorm2d + ReLU; reference
 * graphs/models/Dense_U_Net_lidar.py:128-131), 16-bit storage: its weight gradient AND the BatchNorm-backward reductions of the norm
 * in front of it - red[0..64) = sum dz, red[64..128) = sum dz*xhat over all pixels, dz = relu'(x*scale+shift) * conv_dgrad(dy) with
 * the weights rounded to the storage type, as dmm_conv_dgrad reports them - from ONE pass over x and dy (wg5.hip, PA = 3: x enters as
 * the two factors of its activation; wg5_fin64_kernel).  w: fp32 master weights (Cout, 64, 5, 5); `shift` points at 3*64 floats:
 * shift, mean, invstd; dw: fp32, master layout, overwritten; red: 128 doubles, zeroed by the callee.  DMM_ERR_INVALID for any other
 * shape.  (Nothing upstream: autograd runs conv2d's and batch_norm's backward as separate ATen calls.) */
int dmm_conv5_wgrad_stats(const dmm_conv_desc* d, const void* x, const void* dy, const float* w, const float* scale, const float* shift,
                          float* dw, double* red, void* scratch, void* stream);
/* Data gradient AND weight gradient of a 1x1 bottleneck convolution in one pass (bw1.hip; 16-bit storage, Cout == 128,
 * Cin % 32 == 0; DMM_ERR_INVALID otherwise): the two results of dmm_conv_dgrad_ex and dmm_conv_wgrad_ex on the same operands. */
int dmm_conv1x1_backward_fused(const dmm_conv_desc* d, const void* x, const void* dy, const float* w, const float* scale,
                               const float* shift, const void* yfwd, const float* q, const float* r, void* gx, int accumulate,
                               float* dw, double* red, void* scratch, void* stream);

/* Which kernel family ran the calling thread's most recent single-kernel launch (dmm_conv_forward / _wgrad(_ex) / _dgrad(_ex) /
 * dmm_conv1x1_backward_fused; for multi-launch entry points: the last launch).  The single-kernel entry points dispatch like a
 * plan does and fall back to the generic implicit-GEMM kernels when no specialised family accepts the shape - silently, so a
 * per-kernel parity test asserts the family it names: dmm_impl_name(dmm_last_impl()) is one of
 * "generic", "thin", "conv3", "cvp", "halo", "wg3", "wg5", "wgp", "pig", "bw1"  ("auto": nothing launched yet).
 * (Nothing upstream: the reference has no kernel families; torch dispatches inside ATen.) */
int dmm_last_impl(void);
const char* dmm_impl_name(int impl);
/* Bit (1 << family) for every such launch of the calling thread since the mask was last reset (entry points that launch several
 * kernels - the parity phases of a ConvTranspose - leave more than one bit); reset != 0 clears it after reading. */
unsigned dmm_impl_mask(int reset);

#ifdef __cplusplus
}
#endif
#endif /* DMMFODS_HIP_H */
