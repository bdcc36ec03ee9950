// Argument blocks and launchers of the helper kernels (see pointwise.hip) and of the two GEMM kernels.
#pragma once
#include "common.h"

namespace dmm {

struct ConvertArgs {
  const float* src1;  // (B, C1, H, W) fp32
  const float* src2;  // (B, C2, H, W) fp32 or null
  int C1, C2;         // C1 + C2 <= 8
  void* dst;          // T NHWC, 8 channels per pixel, zero padded
  int B, H, W;
  double* stat_sum;   // 8 doubles each (nullable)
  double* stat_sq;
  float scale;        // multiplies every value (1 for inputs; loss_scale for an external d(loss)/d(logit))
};

struct BnFinalizeArgs {
  const double* sum;
  const double* sq;
  int stat_stride;  // doubles between the STAT_REPS replicas of sum / sq (0: a single copy)
  double count;           // positions the sums were taken over
  double count_unbiased;  // positions PyTorch's BatchNorm sees (differs for the nearest-upsampled head input)
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  float* scale;
  float* shift;
  float* mean;
  float* invstd;
  int C;
  int training;
  float momentum, eps;
};

struct BnBwdFinalizeArgs {
  const double* red1;  // sum dz
  const double* red2;  // sum dz*xhat
  int stat_stride;     // doubles between the STAT_REPS replicas of red1 / red2 (0: a single copy)
  const float* mean;
  const float* invstd;
  const float* scale;
  float* dgamma;
  float* dbeta;
  double* qd;  // deferred correction accumulators of the normalised tensor, fp64 (nullable)
  double* rd;
  float* q;    // float hi / lo split of the accumulators, rewritten after every update
  float* r;
  float* ql;
  float* rl;
  double count;      // positions of the normalised tensor
  float grad_scale;  // 1 / loss_scale
  int C;
};

struct MaxpoolArgs {
  const void* y0;  // (B, H0, W0, ld0) conv0 output
  int ld0, H0, W0, B, C;
  const float* scale;
  const float* shift;
  void* out;  // (B, Hp, Wp, ldo), pre-offset to the destination channels
  int ldo, Hp, Wp;
  unsigned char* argmax;  // (B, Hp, Wp, C)
  double* stat_sum;
  double* stat_sq;
  int stat_stride;  // doubles between the STAT_REPS replicas of sum / sq (0: a single copy)
};

struct MaxpoolBwdArgs {
  const void* y0;
  int ld0, H0, W0, B, C;
  const float* scale;
  const float* shift;
  const void* gpool;  // raw gradient of the pooled tensor (pre-offset), pixel stride ldg
  const void* xpool;  // pooled forward tensor (same layout) for the deferred correction
  const float* q;
  const float* r;
  const float* ql;
  const float* rl;
  const float* mean;    // norm0 batch statistics of y0
  const float* invstd;
  int ldg, Hp, Wp;
  const unsigned char* argmax;
  void* gy0;  // (B, H0, W0, ld0): s * dz0
  double* red1;
  double* red2;
  int stat_stride;  // doubles between the STAT_REPS replicas of red1 / red2 (0: a single copy)
};

struct BceArgs {
  const float* logits;  // (B, NC, H, W) fp32
  const float* target;
  void* dlogits;  // T NHWC8: (sigmoid(x) - t) * loss_scale   (nullable)
  double* out;    // [NC loss | NC equal | B x (NC inter, NC union)]
  int B, NC, H, W;
  float thr, loss_scale;
  // loss epilogue: 0 = BCE-with-logits (A:54); 1 = focal loss alpha[c]*(1-exp(-bce))^gamma[c]*bce on top of it
  // (reference graphs/losses/FocalLoss.py:41-50, class-wise :78-91).  from_prob: the input holds probabilities
  // (binary_cross_entropy with torch's log clamp at -100) instead of logits.
  int kind, from_prob;
  float alpha[8], gamma[8];
  float* loss_out;  // fp32 NCHW unreduced loss (nullable)
  float* dx_out;    // fp32 NCHW d(sum loss)/d(input), unscaled (nullable)
  int metrics;      // 0: skip the loss sums / metric counts (out may be null)
};

struct ApplyCorrArgs {
  void* g;        // T gradient, pixel stride ldg: g += (q + ql) + (r + rl) * y
  const void* y;  // T forward tensor, pixel stride ldy
  const float* q;
  const float* r;
  const float* ql;
  const float* rl;
  size_t npix;
  int C, ldg, ldy;
};

struct AdamArgs {
  float* p;
  const float* g;
  float* m;
  float* v;
  size_t n;
  float beta1, beta2, eps, weight_decay, step_size, bc2_sqrt, grad_scale;
};

struct PackSeg {
  int Creal, Cpad, ntaps, nchunks, koff;
  unsigned tapw[MAX_TAPS];  // up to four master tap indices per packed tap, 0xff = none
};

struct PackDesc {
  const float* w;  // master weights
  void* dst;       // T packed [chunk][Npad][BK]
  float* gw;       // master gradient (nullable)
  const float* dpack;  // fp32 packed gradient (nullable)
  int N, Npad, nseg;
  int shared_master;  // several descriptors add into the same master elements
  long long sn, sk, st;  // master index = n*sn + k*sk + tap*st
  PackSeg seg[2];
  int rs;             // taps of the master tensor (R * S)
  int tiled;          // 0: generic pack / unpack kernels; 1: tile kernels, the master is contiguous along (k, tap) for a fixed n (sk == rs);
                      // 2: tile kernels, contiguous along (n, tap) for a fixed k (sn == rs)
};
// One workgroup of the tile kernels (pack_tiles_kernel / unpack_tiles_kernel): 32 rows n x 32 channels x all taps of a master tensor,
// for the `nsib` consecutive descriptors that share it (the parity phases of a ConvTranspose: together they cover every tap once).
struct PackTile { int desc, nsib, n0, cg; };   // nsib: count | 0x100 when no master tap is written twice by the group

// Fused backward of a dense layer's 1x1 bottleneck convolution (bw1.hip): the data-gradient launch `c` (EPI_BNBWD) plus the
// packed weight gradient of the same convolution.
struct Bw1Args {
  ConvArgs c;
  float* dpack;       // fp32 packed weight gradient [chunk = c / 32][dNpad][32]
  int dNpad, wC;      // its row pitch (128) and the real number of input channels
  // Weight-gradient partials (nullable).  Every workgroup ends with a 64 KB slice of the weight gradient in registers.  Added to
  // dpack with fp32 atomics that is 33 MB per launch at the atomics' 1.3 TB/s - 0.9 ms of the step on the data-gradient chain.
  // With `part` each workgroup STORES its slice to slot (row range * nct + channel slice) (plain stores: 6 TB/s) and a second
  // launch (launch_bw1_reduce, on the weight-gradient stream) adds the slots up into dpack.
  float* part;
  int part_slots;     // capacity of `part` in 64 KB slots
  int nct, ntiles, tiles_per_wg, xcd_group, nsplit;  // (filled by the launcher)
};
struct Bw1Geom { int nct, ntiles, tiles_per_wg, nsplit, xcd_group, nwg; };
Bw1Geom bw1_geometry(const ConvArgs& a);   // how launch_bw1 splits the launch (device-dependent: compute units)
constexpr int B1_SLOT_FLOATS = 4 * 128 * 32;  // one workgroup's slice: 4 chunks of 32 input channels x 128 bottleneck channels
bool bw1_eligible(const WgradArgs& w, const ConvArgs& d, int dtype);
hipError_t launch_bw1(const Bw1Args& g, int dtype, hipStream_t st);
hipError_t launch_bw1_reduce(const Bw1Args& g, hipStream_t st);  // dpack = sum of the slots (no-op without `part`)
hipError_t launch_igemm(const ConvArgs& a, int dtype, int epi, bool mfma, hipStream_t st, int impl = IMPL_AUTO);
hipError_t launch_wgrad(WgradArgs a, int dtype, bool mfma, hipStream_t st, int impl = IMPL_AUTO);
int igemm_pick(const ConvArgs& a, int dtype, int epi, bool mfma);   // the family (enum Impl) that would run the launch now
int wgrad_pick(const WgradArgs& a, int dtype, bool mfma);
void thin_set_enabled(bool on);  // thin.hip
void conv3_set_enabled(bool on);  // conv3.hip
void wg3_set_enabled(bool on);    // wg3.hip
void wgp_set_enabled(bool on);    // wgp.hip
void wg5_set_enabled(bool on);    // wg5.hip
void cvp_set_enabled(bool on);    // cvp.hip
void bw1_set_enabled(bool on);    // bw1.hip
void pig_set_enabled(bool on);    // pig.hip
bool cvp_handles(const ConvArgs& a, int dtype, int epi);
bool conv3_handles(const ConvArgs& a, int dtype, int epi);
bool wg3_handles(const WgradArgs& a, int dtype);
bool wgp_handles(const WgradArgs& a, int dtype);
bool wg5_handles(const WgradArgs& a, int dtype);
hipError_t launch_wg5_rawfin(const RawFinArgs& a, hipStream_t st);
hipError_t launch_wg5_fin64(const Fin64Args& a, hipStream_t st);
hipError_t launch_convert_input(const ConvertArgs& a, int dtype, hipStream_t st);
hipError_t launch_bn_finalize(const BnFinalizeArgs& a, hipStream_t st);
hipError_t launch_bn_bwd_finalize(const BnBwdFinalizeArgs& a, hipStream_t st);
hipError_t launch_maxpool_fwd(const MaxpoolArgs& a, int dtype, hipStream_t st);
hipError_t launch_maxpool_bwd(const MaxpoolBwdArgs& a, int dtype, hipStream_t st);
hipError_t launch_bce_metrics(const BceArgs& a, int dtype, hipStream_t st);
hipError_t launch_adam(const AdamArgs& a, hipStream_t st);
hipError_t launch_apply_corr(const ApplyCorrArgs& a, int dtype, hipStream_t st);
hipError_t launch_pack(const PackDesc* descs_dev, const int* prefix_dev, int ndesc, int total_rows, int dtype, hipStream_t st,
                       const PackDesc* tile_descs = nullptr, const PackTile* tiles_dev = nullptr, int nt1 = 0, int nt9 = 0);
hipError_t launch_unpack(const PackDesc* descs_dev, const int* prefix_dev, int ndesc, int total_rows, int dtype, float grad_scale,
                         hipStream_t st, const PackDesc* tile_descs = nullptr, const PackTile* tiles_dev = nullptr, int nt1 = 0, int nt9 = 0);

#if defined(__HIPCC__)
// ---- the finalize steps, per channel: bodies of bn_finalize_kernel / bn_bwd_finalize_kernel ----
// COH: the sums were added by OTHER workgroups of the running launch (float atomics execute at the memory side and leave nothing in
// any L2; the loads bypass this CU's L1: agent-scope relaxed atomic loads = global_load ... sc1).  Only the round-3 experiment that ran
// the steps as the tail of the producing launch used it (profiles/r03/ablations.txt: 0.8 ms SLOWER per step than launches of their own).
template <bool COH>
__device__ __forceinline__ double stat_load(const double* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}

template <bool COH>
__device__ __forceinline__ void bn_finalize_channel(const BnFinalizeArgs& a, int c) {
  float mean, var;
  if (a.training) {
    double su = stat_load<COH>(a.sum + c), sq = stat_load<COH>(a.sq + c);
    if (a.stat_stride)
      for (int k = 1; k < STAT_REPS; ++k) { su += stat_load<COH>(a.sum + c + (size_t)k * a.stat_stride); sq += stat_load<COH>(a.sq + c + (size_t)k * a.stat_stride); }
    const double m = su / a.count;
    double v = sq / a.count - m * m;
    if (v < 0) v = 0;
    mean = (float)m;
    var = (float)v;
    const double unb = a.count_unbiased > 1 ? v * (a.count_unbiased / (a.count_unbiased - 1.0)) : v;
    a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * mean;
    a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unb;
  } else {
    mean = a.running_mean[c];
    var = a.running_var[c];
  }
  const float invstd = 1.0f / sqrtf(var + a.eps);
  const float s = a.gamma[c] * invstd;
  a.scale[c] = s;
  a.shift[c] = a.beta[c] - mean * s;
  a.mean[c] = mean;
  a.invstd[c] = invstd;
}

template <bool COH>
__device__ __forceinline__ void bn_bwd_finalize_channel(const BnBwdFinalizeArgs& a, int c) {
  double S1 = stat_load<COH>(a.red1 + c), S2 = stat_load<COH>(a.red2 + c);
  if (a.stat_stride)
    for (int k = 1; k < STAT_REPS; ++k) { S1 += stat_load<COH>(a.red1 + c + (size_t)k * a.stat_stride); S2 += stat_load<COH>(a.red2 + c + (size_t)k * a.stat_stride); }
  const double mu = a.mean[c], is = a.invstd[c];
  const double dotp = S2;  // sum dz * xhat, reduced in centred form by the producing kernel
  a.dgamma[c] = (float)(dotp * a.grad_scale);
  a.dbeta[c] = (float)(S1 * a.grad_scale);
  if (a.qd != nullptr) {
    const double s = a.scale[c];
    const double c1 = S1 / a.count, c2 = dotp / a.count;
    // contribution of this consumer to d/dx:  s*dz (stored by the dgrad epilogue)  - s*c1 - s*c2*(x-mu)*is.
    // Accumulated in fp64 over all consumers of the channel (they cancel heavily inside dense blocks) and handed to
    // the gathers as a two-float split: a rounding error here would be a coherent per-channel gradient offset.
    const double q = a.qd[c] + (-s * c1 + s * c2 * mu * is);
    const double r = a.rd[c] + (-s * c2 * is);
    a.qd[c] = q;
    a.rd[c] = r;
    const float qh = (float)q, rh = (float)r;
    a.q[c] = qh; a.ql[c] = (float)(q - (double)qh);
    a.r[c] = rh; a.rl[c] = (float)(r - (double)rh);
  }
}

#endif

}  // namespace dmm
