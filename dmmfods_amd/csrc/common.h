// Shared definitions for the gfx950 (MI355X) kernels of the Dense_U_Net_lidar hot path.
// Layout convention everywhere: activations are NHWC ("pixel-major"): element (b,y,x,c) of a tensor with
// pixel stride ld lives at ((b*H + y)*W + x)*ld + c.  A "slot" is 16 bytes of channels (8 x f16 or 4 x f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

namespace dmm {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16;  // storage/MFMA type of the "mixed bf16" configuration (BASELINE configs[4]); fp32 accumulate like f16
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum DType { DT_F32 = 0, DT_F16 = 1, DT_BF16 = 2 };
static inline size_t dtype_size(int dt) { return dt == DT_F32 ? 4 : 2; }

template <typename T> struct TT;
template <> struct TT<float> {
  static constexpr int SLOT = 4;
  typedef f32x4 vec;
};
template <> struct TT<f16> {
  static constexpr int SLOT = 8;
  typedef f16x8 vec;
};
template <> struct TT<bf16> {
  static constexpr int SLOT = 8;
  typedef bf16x8 vec;
};

constexpr int MAX_TAPS = 52;
#ifndef DMM_STAT_REPS
#define DMM_STAT_REPS 8
#endif
constexpr int STAT_REPS = DMM_STAT_REPS;  // replicas of the BatchNorm reduction accumulators (low 3 bits of the workgroup id = XCD)
constexpr int DESIGN_CUS = 256; // compute units of the MI355X: launch geometries that size plan workspace are computed for it, not queried
constexpr int BM = 128;        // rows (pixels) per workgroup tile
constexpr int NTHREADS = 256;  // 4 waves of 64
constexpr int ROWB = 64;       // LDS bytes per tile row = one K-chunk; the four 16-byte slots of a row are XOR-swizzled
                               // with (row >> 2) & 3, which makes ds_read_b128 conflict-free for its 16-lane groups
                               // {0-3,12-15,20-27}, ... (each group then covers 4 bank quarters x 4 distinct slots)

enum GatherMode { G_PLAIN = 0, G_UP2 = 1, G_POOL2 = 2 };
enum Epilogue { EPI_STORE = 0, EPI_BNBWD = 1, EPI_LOGITS = 2 };

// One K-segment of the gathered ("A") operand of an implicit-GEMM convolution.
struct Seg {
  const void* src;     // T*, NHWC, pixel stride ld
  const void* src2;    // companion tensor for the effective-gradient prologue (nullable)
  const float* scale;  // BN+ReLU prologue: v = max(v*scale[c] + shift[c], 0)   (nullable)
  const float* shift;
  const float* q;      // effective gradient: v = src + (q+ql)[c] + (r+rl)[c]*src2  (nullable); the per-channel
  const float* r;      // constants are fp64 values split into float hi (q, r) and lo (ql, rl) parts
  const float* ql;
  const float* rl;
  int ld, ld2;
  int Hs, Ws;   // source spatial size
  int C;        // real channels (multiple of SLOT)
  int Cpad;     // channels per tap in the K enumeration (multiple of SLOT, >= C)
  int ntaps;
  int nchunks;  // ceil(ntaps*Cpad / BK)
  int mode;     // GatherMode
  int istride;  // source pixel = row pixel * istride + tap offset
  short taps[MAX_TAPS];  // (dy & 0xff) | ((dx & 0xff) << 8)
};

struct ConvArgs {
  Seg seg[2];
  int nseg;
  int B, Ho, Wo, M;  // row grid (b, y, x), M = B*Ho*Wo
  const void* wpack; // T* packed weights [chunk][Npad][BK]
  int N, Npad;
  // output pixel of row (b,y,x) = (b, y*ostride + py, x*ostride + px) in a tensor Hout x Wout, pixel stride ldo
  void* out;
  int ldo, coff, Hout, Wout, ostride, py, px;
  // EPI_STORE: per-channel sum / sum of squares of the stored values (nullable), indexed like out channels
  double* stat_sum;
  double* stat_sq;
  // EPI_LOGITS: fp32 NCHW logits (B, N, Hout, Wout)
  float* logits;
  // EPI_BNBWD: acc = d(relu(bn(x))).  x and the gradient buffer `out` share the pixel mapping above.
  const void* bx;
  int ldbx;
  const float* bscale;
  const float* bshift;
  const float* bmean;    // batch mean / inverse std of x: the second reduction is sum dz * (x - mean) * invstd
  const float* binvstd;
  double* red1;    // sum dz
  double* red2;    // sum dz*xhat
  int accumulate;  // out += s*dz  instead of  out = s*dz
  // The per-channel fp64 accumulators (stat_sum/stat_sq or red1/red2) exist in STAT_REPS replicas `stat_stride` doubles
  // apart (0 = one copy): a workgroup adds into replica blockIdx % 8, i.e. the one its XCD owns, so the L2 atomics of the
  // 8 XCDs never meet on a line and each address sees 1/8 of the traffic.  The finalize kernels add the replicas up.
  int stat_stride;
  int pool2;       // each row is a 2x2-average-pooled pixel: distribute 0.25*acc to the 4 source pixels
  // conv3.hip, data gradient of the dense 3x3 convolution: the gathered operand AFTER its prologue (the effective output gradient of
  // the 32 growth channels), interior pixels of every tile, as a compact [pixel][32] tensor (nullable).  wg3.hip reads it instead
  // of gathering 64 bytes per pixel from two [pixel][ld] tensors - the L2 fetches 128-byte lines, so those gathers moved 4x the bytes.
  void* eff_out;
  // EPI_BNBWD, second pass of a two-pass BatchNorm backward (conv3.hip; plan.cpp emit_conv_bwd): the per-channel constants of the
  // deferred correction are already known, so the kernel stores the FINAL gradient s*dz + q[c] + r[c]*x instead of s*dz (nullable;
  // red1 / red2 are null in that pass: the reductions were the first pass, which stored nothing).
  const float* eq;
  const float* er;
  // conv3.hip, forward: several output-parity phases of ONE convolution in one launch (nphase = 0: one phase, the fields above).
  // The phases share both gathered operands and the output tensor and differ in their taps, packed weights and output parity; the
  // four phases of a tile are neighbours in the tile order, i.e. they run at the same time on one XCD and the half-resolution
  // input is fetched from HBM once (four launches swept it four times).
  int nphase;
  const void* ph_wpack[4];
  short ph_taps0[4][4];    // taps of seg[0] per phase
  short ph_taps1[4][9];    // taps of seg[1] per phase
  signed char ph_py[4], ph_px[4];
  // cvp.hip, forward: the four parity phases of a ConvTranspose in one launch (nphase = 4; ph_wpack / ph_taps0 / ph_py / ph_px as above,
  // ph_ntaps = taps of each phase: 1, 2, 2, 4 in some order).  Separate launches of 600 (1200) workgroups each ran two (three) rounds on
  // the chip's 512 slots, the last one mostly empty; one launch of all phases, longest first, fills them.
  signed char ph_ntaps[4];
};

// Weight-gradient GEMM:  dP[chunk][n][k] += sum_m dYeff[m][n] * A[m][k], same A gather as the forward conv.
struct WgradArgs {
  Seg seg[2];
  int nseg;
  int B, Ho, Wo, M;
  Seg dy;        // the pixel-aligned operand P: a one-tap Seg (tap = (py,px), istride = ostride for ConvTranspose phases)
  int N, Npad;
  float* dpack;  // fp32 packed gradient [chunk][Npad][BK]
  int rows_per_split;  // multiple of BM
  int kgroups;         // number of K groups (each WG_CHUNKS chunks)
  // wg3.hip: per-workgroup slots for the partial results (nullable: fp32 atomics into dpack).  One buffer serves every launch of
  // the family - they run in order on one stream and each is followed by its reduction.
  float* part;
  int part_slots;      // capacity of `part` in slots of W3_SLOT_FLOATS
  // wgp.hip: several output-parity phases of ONE convolution in one launch (nphase = 0: one phase, the fields above).  The phases
  // share the gathered operand (seg[0] without its taps) and differ in their taps, the parity of the gradient rows and the packed
  // gradient they add into.  Phase p of a tile range runs beside the other phases of that range on the same XCD, so the operand
  // is fetched from HBM once and served to the others by that XCD's L2 (four launches swept it four times).
  int nphase;
  short ph_xtaps[4][4];   // taps of seg[0] per phase (ntaps each)
  short ph_ytap[4];       // parity tap of dy per phase
  float* ph_dpack[4];
  // wg5.hip, PY = 2 (round 5): the thin operand enters as the two factors of its activation - batch mean / inverse std of its
  // BatchNorm (scale / shift: seg[0]) - and the factor correlations go to sbuf [5][64][32] instead of dpack (nullable: off).
  // PA = 3 (25 taps): the same for the 64-channel operand `dy` (t_mean / t_invstd: 64 channels; sbuf [2][4][64][32]; the thin operand's real channels number <= 4)
  const float* t_mean;
  const float* t_invstd;
  float* sbuf;
  // wgpw.hip: the taps of each phase of a multi-phase launch when they differ (the ConvTranspose's 1, 2, 2, 4); all 0 = seg[0].ntaps each
  signed char ph_ntaps[4];
};
// wg5_rawfin_kernel (wg5.hip): see there
struct RawFinArgs {
  const float* sbuf;     // [5][64][32] factor correlations
  float* dpack;          // packed gradient of the raw-input segment [3][Npad][32]
  int Npad;
  const float* w;        // master weights [64][Kin][9]
  int Kin, koff, nreal;  // input channels of the convolution, first raw-input channel, real raw-input channels (<= 8)
  int tapw[9];           // master tap of packed tap t
  const float* gamma;    // of the raw-input channels (8 readable)
  const float* beta;
  double* red1;          // BatchNorm-backward reductions of the raw-input channels, replica 0
  double* red2;
};
constexpr int W5_SBUF_FLOATS = 5 * 64 * 32;
// wg5_fin64_kernel (wg5.hip): from the factor correlations of the 64-channel operand (wg5_kernel, PA = 3) to the packed weight gradient
// of the head's 5x5 convolution and the BatchNorm-backward reductions of the norm in front of it
struct Fin64Args {
  const float* sbuf;     // (doubles) [2][4][64][32]: variant 0 = S2 (m x), 1 = S1 (m); column 32 chunk + k = 4 tap + n
  float* dpack;          // packed gradient [7][Npad][32]
  int Npad;
  const float* w;        // master weights [nreal][Kin][25] (the forward convolution's own)
  int Kin, nreal;        // input channels of the convolution (64), its real output channels (classes, <= 4)
  int dtype;             // storage type of the packed weights (DT_F16 / DT_BF16)
  unsigned char tapw[28];  // master tap of packed tap t
  const float* scale;    // norm1's forward constants, 64 channels: relu(bn(x)) = scale (m x) + shift m
  const float* shift;
  const float* mean;     // its batch statistics: sum dz xhat = (sum dz x - mean sum dz) invstd
  const float* invstd;
  double* red1;          // BatchNorm-backward reductions of the 64 channels, replica 0
  double* red2;
};
constexpr int W5_SBUF64_FLOATS = 2 * 2 * 4 * 64 * 32;   // [2][4][64][32] doubles
constexpr int W3_SLOT_FLOATS = 9 * 128 * 32;  // one workgroup's partial result of the dense 3x3 weight gradient (147 KB)
constexpr int W3_MAX_SLOTS = 256;             // = workgroups of a launch at most (device-independent: plans are sized without a GPU)

// Kernel family of a convolution / weight-gradient launch.  A plan decides it ONCE per launch when it is built (igemm_pick /
// wgrad_pick walk the dispatch without launching and honour the dmm_set_option switches of that moment) and the executor
// dispatches from the recorded value, so a plan's labels, its profile classes and the kernels it runs cannot drift apart when an
// option is toggled afterwards.  IMPL_AUTO (the single-kernel test entry points): decide at the call.
enum Impl { IMPL_AUTO = 0, IMPL_GENERIC = 1, IMPL_THIN, IMPL_CONV3, IMPL_CVP, IMPL_HALO, IMPL_WG3, IMPL_WG5, IMPL_WGP, IMPL_PIG, IMPL_BW1, IMPL_HF, IMPL_CF, IMPL_WGPW /* wgp in its wave-specialised form: noted beside IMPL_WGP */, IMPL_CVW /* likewise cvp forward (cvw.hip) */, IMPL_COUNT };
struct LaunchCtl {
  bool dry = false;      // walk the eligibility tests, launch nothing
  int impl = IMPL_AUTO;  // the one family allowed to take the launch (IMPL_AUTO: every enabled family, in dispatch order)
  unsigned deny = 0;     // 1 << family for families a plan under construction must not pick (PlanSwitches, plan.h)
};
extern thread_local LaunchCtl g_ctl;  // (defined in pointwise.hip)
// The family that took the calling thread's most recent convolution / weight-gradient / fused-backward launch (dmm_last_impl):
// a per-kernel test asserts the family it names really ran - IMPL_AUTO falls back to the generic kernels silently.
extern thread_local int g_last_impl;
extern thread_local unsigned g_impl_mask;  // 1 << family for every launch since the mask was last reset (dmm_impl_mask)
inline void note_impl(int impl) { g_last_impl = impl; g_impl_mask |= 1u << impl; }
inline bool family_on(bool enabled, int family) { return g_ctl.impl == IMPL_AUTO ? (enabled && !((g_ctl.deny >> family) & 1u)) : g_ctl.impl == family; }

// Lab knobs: launch-geometry and ablation switches whose experiments are recorded (profiles/*/ablations.txt).  The
// shipped library fixes them at their defaults - no getenv, nothing an embedding process can trip over; a lab build
// (-DDMM_LAB=1: tools/build_variant.sh, loaded through DMM_LIB_PATH) reads them from the environment again, once per process.
// The run-time switches that remain are listed in plan.h (PlanSwitches) and capi.cpp; each has a test.
#ifndef DMM_LAB
#define DMM_LAB 0
#endif
#if DMM_LAB
inline int lab_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
inline bool lab_flag(const char* name) { return getenv(name) != nullptr; }
inline const char* lab_str(const char* name) { return getenv(name); }
#else
inline constexpr int lab_int(const char*, int dflt) { return dflt; }
inline constexpr bool lab_flag(const char*) { return false; }
inline constexpr const char* lab_str(const char*) { return nullptr; }
#endif

#if defined(__HIPCC__)
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(f16 v) { return (float)v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ f16 from_f32<f16>(float v) { return (f16)v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }  // v_cvt_pk_bf16_f32: round to nearest even

// one 32x32x16 matrix-core step on 16-bit fragments (8 k-values per lane), fp32 accumulate
__device__ __forceinline__ f32x16 mma16(const f16x8& a, const f16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its own L2.  Neighbouring tiles share data
// (conv halos, the operand that several K groups / N tiles re-read), so give every XCD one CONTIGUOUS range of the logical
// tile order: logical = start(xcd) + blockIdx / 8.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int x = bid & 7, i = bid >> 3;
  const int lo = nblocks >> 3, rem = nblocks & 7;
  return x * lo + (x < rem ? x : rem) + i;
}

__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }
#endif

}  // namespace dmm
