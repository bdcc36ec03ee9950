// Bandwidth-bound helper kernels of the Dense_U_Net_lidar training step (gfx950).
//   convert_input   : fp32 NCHW streams -> T NHWC8 (+ per-channel sum / sum^2 for the head BatchNorm)
//   bn_finalize     : batch statistics -> (scale, shift, mean, invstd), running-stat update (momentum .1)
//   bn_bwd_finalize : (sum dz, sum dz*x) -> dgamma, dbeta and the deferred per-channel correction (q, r)
//   maxpool_fwd/bwd : 3x3 s2 p1 max pooling fused with the BN+ReLU in front of it (first-max argmax saved)
//   bce_metrics     : BCE-with-logits (sum reduction), d(loss)/d(logit), IoU / accuracy counts
//   adam            : flat fused Adam (torch.optim.Adam semantics, amsgrad off)
//   pack / unpack   : OIHW fp32 master weights <-> K-chunked compute layout [chunk][Npad][BK]
#include <type_traits>

#include "common.h"
#include "gather.h"
#include "pointwise.h"

namespace dmm {

thread_local LaunchCtl g_ctl;  // see common.h
thread_local int g_last_impl = IMPL_AUTO;
thread_local unsigned g_impl_mask = 0;

// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void convert_input_kernel(ConvertArgs a) {
  // one thread per pixel; reads are coalesced along x within each NCHW plane
  __shared__ double red[16];
  if (threadIdx.x < 16) red[threadIdx.x] = 0.0;
  __syncthreads();
  const size_t plane = (size_t)a.H * a.W;
  const size_t npix = (size_t)a.B * plane;
  double s1[8], s2[8];  // fp64 from the first add (see igemm.hip)
#pragma unroll
  for (int c = 0; c < 8; ++c) { s1[c] = 0.0; s2[c] = 0.0; }
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
    const size_t b = p / plane, rem = p - b * plane;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float x = 0.f;
      if (c < a.C1) x = a.src1[(b * a.C1 + c) * plane + rem];
      else if (c < a.C1 + a.C2) x = a.src2[(b * a.C2 + (c - a.C1)) * plane + rem];
      v[c] = x * a.scale;
    }
    T* d = (T*)a.dst + p * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const T t = from_f32<T>(v[c]);
      d[c] = t;
      const double f = (double)to_f32(t);
      s1[c] += f;
      s2[c] += f * f;
    }
  }
  if (a.stat_sum == nullptr) return;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    double x = s1[c], y = s2[c];
    for (int o = 32; o > 0; o >>= 1) { x += __shfl_down(x, o); y += __shfl_down(y, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&red[c], x); atomicAdd(&red[8 + c], y); }
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    atomic_add_f64(a.stat_sum + threadIdx.x, red[threadIdx.x]);
    atomic_add_f64(a.stat_sq + threadIdx.x, red[8 + threadIdx.x]);
  }
}

hipError_t launch_convert_input(const ConvertArgs& a, int dtype, hipStream_t st) {
  const size_t npix = (size_t)a.B * a.H * a.W;
  int grid = (int)((npix + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (dtype == DT_F16) hipLaunchKernelGGL(convert_input_kernel<f16>, dim3(grid), dim3(256), 0, st, a);
  else if (dtype == DT_BF16) hipLaunchKernelGGL(convert_input_kernel<bf16>, dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(convert_input_kernel<float>, dim3(grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(BnFinalizeArgs a) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < a.C) bn_finalize_channel<false>(a, c);
}

hipError_t launch_bn_finalize(const BnFinalizeArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((a.C + 127) / 128), dim3(128), 0, st, a);
  return hipGetLastError();
}

__global__ void bn_bwd_finalize_kernel(BnBwdFinalizeArgs a) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < a.C) bn_bwd_finalize_channel<false>(a, c);
}

hipError_t launch_bn_bwd_finalize(const BnBwdFinalizeArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((a.C + 127) / 128), dim3(128), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Block-wide per-channel reduction helper: NCV slot columns x (256/NCV) row lanes.
template <int SLOT>
__device__ __forceinline__ void block_channel_reduce(double* red, int C, int cbase, const double (&s1)[SLOT],
                                                     const double (&s2)[SLOT], double* d1, double* d2, int stat_stride = 0) {
  // red: 2*C doubles, zeroed and synchronised by the caller
#pragma unroll
  for (int i = 0; i < SLOT; ++i) {
    atomicAdd(&red[cbase + i], s1[i]);
    atomicAdd(&red[C + cbase + i], s2[i]);
  }
  __syncthreads();
  // thousands of workgroups adding to the same C addresses serialise in the L2 atomic unit: one replica per XCD (summed by the
  // finalize kernels), as in igemm.hip
  const size_t rep = (size_t)(blockIdx.x & (STAT_REPS - 1)) * stat_stride;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    atomic_add_f64(d1 + rep + c, red[c]);
    atomic_add_f64(d2 + rep + c, red[C + c]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(MaxpoolArgs a) {
  constexpr int SLOT = TT<T>::SLOT;
  typedef typename TT<T>::vec V;
  extern __shared__ double red[];
  for (int i = threadIdx.x; i < 2 * a.C; i += blockDim.x) red[i] = 0.0;
  __syncthreads();
  const int ncv = a.C / SLOT;
  const int cv = threadIdx.x % ncv, rl = threadIdx.x / ncv, rpb = blockDim.x / ncv;
  const int c = cv * SLOT;
  float sc[SLOT], sh[SLOT];
  double s1[SLOT], s2[SLOT];
  load_f32s<SLOT>(a.scale + c, sc);
  load_f32s<SLOT>(a.shift + c, sh);
#pragma unroll
  for (int i = 0; i < SLOT; ++i) { s1[i] = 0.0; s2[i] = 0.0; }
  const int npix = a.B * a.Hp * a.Wp;
  const T* y0 = (const T*)a.y0;
  T* out = (T*)a.out;
  if (rl < rpb) {
    for (int p = blockIdx.x * rpb + rl; p < npix; p += gridDim.x * rpb) {
      int b, oy, ox;
      row_to_byx(p, a.Hp, a.Wp, b, oy, ox);
      float best[SLOT];
      int arg[SLOT];
#pragma unroll
      for (int i = 0; i < SLOT; ++i) { best[i] = -INFINITY; arg[i] = 0; }
      // all nine window loads are issued before the first compare (out-of-image taps re-read the centre pixel and are
      // skipped by the predicate), so one memory latency is paid per output instead of nine
      V win[9];
      bool ok[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int iy = 2 * oy - 1 + k / 3, ix = 2 * ox - 1 + k % 3;
        ok[k] = iy >= 0 && ix >= 0 && iy < a.H0 && ix < a.W0;
        const int cy = ok[k] ? iy : 2 * oy, cx = ok[k] ? ix : 2 * ox;
        win[k] = *(const V*)(y0 + ((size_t)(b * a.H0 + cy) * a.W0 + cx) * a.ld0 + c);
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        float f[SLOT];
        vec_to_f32<T>(win[k], f);
#pragma unroll
        for (int i = 0; i < SLOT; ++i) {
          const float v = fmaxf(fmaf(f[i], sc[i], sh[i]), 0.f);
          if (ok[k] && v > best[i]) { best[i] = v; arg[i] = k; }
        }
      }
      const V ov = f32_to_vec<T>(best);
      *(V*)(out + (size_t)p * a.ldo + c) = ov;
      float r[SLOT];
      vec_to_f32<T>(ov, r);
      unsigned long long packed = 0;  // the slot's argmax bytes leave in one store
#pragma unroll
      for (int i = 0; i < SLOT; ++i) {
        packed |= (unsigned long long)(unsigned)arg[i] << (8 * i);
        s1[i] += (double)r[i];
        s2[i] += (double)r[i] * (double)r[i];
      }
      if constexpr (SLOT == 8) *(unsigned long long*)(a.argmax + (size_t)p * a.C + c) = packed;
      else *(unsigned*)(a.argmax + (size_t)p * a.C + c) = (unsigned)packed;
    }
  }
  block_channel_reduce<SLOT>(red, a.C, c, s1, s2, a.stat_sum, a.stat_sq, a.stat_stride);
}

hipError_t launch_maxpool_fwd(const MaxpoolArgs& a, int dtype, hipStream_t st) {
  const int slot = dtype == DT_F32 ? 4 : 8;
  const int rpb = 256 / (a.C / slot);
  const int npix = a.B * a.Hp * a.Wp;
  int grid = (npix + rpb - 1) / rpb;
  if (grid > 2048) grid = 2048;  // 8 workgroups per CU: every workgroup ends with 2 C fp64 atomics
  const size_t smem = 2 * a.C * sizeof(double);
  if (dtype == DT_F16) hipLaunchKernelGGL(maxpool_fwd_kernel<f16>, dim3(grid), dim3(256), smem, st, a);
  else if (dtype == DT_BF16) hipLaunchKernelGGL(maxpool_fwd_kernel<bf16>, dim3(grid), dim3(256), smem, st, a);
  else hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid), dim3(256), smem, st, a);
  return hipGetLastError();
}

// grad wrt conv0 output: dz0[y,x,c] = relu'(.) * sum over pooling windows whose saved argmax is (y,x);
// stores s*dz0 into gy0 and reduces sum dz0, sum dz0*y0 for norm0's backward.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(MaxpoolBwdArgs a) {
  constexpr int SLOT = TT<T>::SLOT;
  typedef typename TT<T>::vec V;
  extern __shared__ double red[];
  for (int i = threadIdx.x; i < 2 * a.C; i += blockDim.x) red[i] = 0.0;
  __syncthreads();
  const int ncv = a.C / SLOT;
  const int cv = threadIdx.x % ncv, rl = threadIdx.x / ncv, rpb = blockDim.x / ncv;
  const int c = cv * SLOT;
  float sc[SLOT], sh[SLOT], q[SLOT], rr[SLOT], qlo[SLOT], rlo[SLOT];
  double s1[SLOT], s2[SLOT];
  load_f32s<SLOT>(a.scale + c, sc);
  load_f32s<SLOT>(a.shift + c, sh);
  load_f32s<SLOT>(a.q + c, q);
  load_f32s<SLOT>(a.r + c, rr);
  load_f32s<SLOT>(a.ql + c, qlo);
  load_f32s<SLOT>(a.rl + c, rlo);
  float mu[SLOT], istd[SLOT];
  load_f32s<SLOT>(a.mean + c, mu);
  load_f32s<SLOT>(a.invstd + c, istd);
#pragma unroll
  for (int i = 0; i < SLOT; ++i) { s1[i] = 0.0; s2[i] = 0.0; }
  // per-thread partial sums in fp32, moved to the fp64 accumulators every 8 pixels (round 4: two conversions and two fp64 adds per ELEMENT
  // were a third of this kernel's vector time, and it is one of the two launches the end of the step waits for - the stem's leaves)
  float f1[SLOT], f2[SLOT];
#pragma unroll
  for (int i = 0; i < SLOT; ++i) { f1[i] = 0.f; f2[i] = 0.f; }
  int pending = 0;
  const int npix = a.B * a.H0 * a.W0;
  const T* y0 = (const T*)a.y0;
  const T* gp = (const T*)a.gpool;
  const T* xp = (const T*)a.xpool;
  T* gy0 = (T*)a.gy0;
  if (rl < rpb) {
    // (b, y, x) of the thread's pixel walk forward by the decomposed stride: two divisions per pixel were a sixth of the loop's instructions
    const int stride = gridDim.x * rpb;
    int sb, sy, sx;
    row_to_byx(stride, a.H0, a.W0, sb, sy, sx);
    int b, y, x;
    row_to_byx(blockIdx.x * rpb + rl, a.H0, a.W0, b, y, x);
    for (int p = blockIdx.x * rpb + rl; p < npix; p += stride) {
      float g[SLOT];
#pragma unroll
      for (int i = 0; i < SLOT; ++i) g[i] = 0.f;
      // windows (oy, ox) with 2*oy-1 <= y <= 2*oy+1: up to two per axis.  All four candidates are loaded before the first use
      // (clamped to a valid window, dropped by the predicate): one memory latency per pixel instead of one per window.
      const int oy0 = y >> 1, oy1 = (y + 1) >> 1;  // oy0 <= oy1, may coincide
      const int ox0 = x >> 1, ox1 = (x + 1) >> 1;
      V wg[4], wx_[4];
      unsigned long long wam[4];
      bool wok[4];
      int wk[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int wy = (w >> 1) ? oy1 : oy0, wx = (w & 1) ? ox1 : ox0;
        wok[w] = ((w >> 1) == 0 || oy1 != oy0) && ((w & 1) == 0 || ox1 != ox0) && wy < a.Hp && wx < a.Wp;
        const int cy = min(wy, a.Hp - 1), cx = min(wx, a.Wp - 1);
        wk[w] = (y - (2 * wy - 1)) * 3 + (x - (2 * wx - 1));
        const size_t op = (size_t)(b * a.Hp + cy) * a.Wp + cx;
        wg[w] = *(const V*)(gp + op * a.ldg + c);
        wx_[w] = *(const V*)(xp + op * a.ldg + c);
        if constexpr (SLOT == 8) wam[w] = *(const unsigned long long*)(a.argmax + op * a.C + c);
        else wam[w] = *(const unsigned*)(a.argmax + op * a.C + c);
      }
      const V y0v = *(const V*)(y0 + (size_t)p * a.ld0 + c);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        if constexpr (std::is_same<T, f16>::value) {
          // f16: the effective gradient g + q + r x of a window on the packed halves (v_fma_mix_f32: two instructions per element; the lo parts
          // of q, r are far below the storage rounding, as in gather.h eff_grad_slot)
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 gi = __builtin_bit_cast(u32x4, wg[w]), xi = __builtin_bit_cast(u32x4, wx_[w]);
          const float one = 1.f;
#pragma unroll
          for (int pr = 0; pr < 4; ++pr) {
            float t0, t1, v0, v1;
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(t0) : "v"(xi[pr]), "v"(rr[2 * pr]), "v"(q[2 * pr]));
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t1) : "v"(xi[pr]), "v"(rr[2 * pr + 1]), "v"(q[2 * pr + 1]));
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(v0) : "v"(gi[pr]), "v"(one), "v"(t0));
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(v1) : "v"(gi[pr]), "v"(one), "v"(t1));
            if (wok[w] && (int)((wam[w] >> (16 * pr)) & 0xff) == wk[w]) g[2 * pr] += v0;
            if (wok[w] && (int)((wam[w] >> (16 * pr + 8)) & 0xff) == wk[w]) g[2 * pr + 1] += v1;
          }
        } else {
        float gf[SLOT], xf[SLOT];
        vec_to_f32<T>(wg[w], gf);
        vec_to_f32<T>(wx_[w], xf);
#pragma unroll
        for (int i = 0; i < SLOT; ++i)
          if (wok[w] && (int)((wam[w] >> (8 * i)) & 0xff) == wk[w]) g[i] += (gf[i] + fmaf(rr[i], xf[i], q[i])) + fmaf(rlo[i], xf[i], qlo[i]);
        }
      }
      float yf[SLOT], o[SLOT];
      vec_to_f32<T>(y0v, yf);
#pragma unroll
      for (int i = 0; i < SLOT; ++i) {
        const float dz = (fmaf(yf[i], sc[i], sh[i]) > 0.f) ? g[i] : 0.f;
        f1[i] += dz;
        f2[i] = fmaf(dz, (yf[i] - mu[i]) * istd[i], f2[i]);
        o[i] = sc[i] * dz;
      }
      *(V*)(gy0 + (size_t)p * a.ld0 + c) = f32_to_vec<T>(o);
      if (++pending == 8) {
        pending = 0;
#pragma unroll
        for (int i = 0; i < SLOT; ++i) { s1[i] += (double)f1[i]; s2[i] += (double)f2[i]; f1[i] = 0.f; f2[i] = 0.f; }
      }
      x += sx; if (x >= a.W0) { x -= a.W0; ++y; }
      y += sy; if (y >= a.H0) { y -= a.H0; ++b; }
      b += sb;
    }
#pragma unroll
    for (int i = 0; i < SLOT; ++i) { s1[i] += (double)f1[i]; s2[i] += (double)f2[i]; }
  }
  block_channel_reduce<SLOT>(red, a.C, c, s1, s2, a.red1, a.red2, a.stat_stride);
}

hipError_t launch_maxpool_bwd(const MaxpoolBwdArgs& a, int dtype, hipStream_t st) {
  const int slot = dtype == DT_F32 ? 4 : 8;
  const int rpb = 256 / (a.C / slot);
  const int npix = a.B * a.H0 * a.W0;
  int grid = (npix + rpb - 1) / rpb;
  static const int cap = lab_int("DMM_MPB_GRID", 2048);
  if (grid > cap) grid = cap;  // 8 workgroups per CU: every workgroup ends with 2 C fp64 atomics
  const size_t smem = 2 * a.C * sizeof(double);
  if (dtype == DT_F16) hipLaunchKernelGGL(maxpool_bwd_kernel<f16>, dim3(grid), dim3(256), smem, st, a);
  else if (dtype == DT_BF16) hipLaunchKernelGGL(maxpool_bwd_kernel<bf16>, dim3(grid), dim3(256), smem, st, a);
  else hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid), dim3(256), smem, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// out[] (doubles): [0..NC) loss sums, [NC..2NC) equal counts, then per image b: inter[NC], union[NC]
// Per-element loss and its derivative.  kind 0: BCE; kind 1: focal  F = alpha*(1-pt)^gamma*bce with pt = exp(-bce), so
// dF/dx = alpha*(1-pt)^(gamma-1) * bce' * (gamma*pt*bce + (1-pt))   (d(1-pt)/dx = pt*bce').
__device__ __forceinline__ void loss_elem(const BceArgs& a, int n, float x, float t, float& loss, float& dx) {
  float bce, dbce;
  if (!a.from_prob) {
    const float e = expf(-fabsf(x));
    bce = fmaxf(x, 0.f) - x * t + log1pf(e);
    const float sig = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    dbce = sig - t;
  } else {  // torch.binary_cross_entropy: logs clamped at -100
    const float lp = fmaxf(logf(x), -100.f), lq = fmaxf(logf(1.f - x), -100.f);
    bce = -(t * lp + (1.f - t) * lq);
    dbce = -(t * (lp > -100.f ? 1.f / x : 0.f) - (1.f - t) * (lq > -100.f ? 1.f / (1.f - x) : 0.f));
  }
  if (a.kind == 0) { loss = bce; dx = dbce; return; }
  const float al = a.alpha[n], ga = a.gamma[n];
  const float omp = -expm1f(-bce);  // 1 - pt, without cancellation for small bce
  const float pt = 1.f - omp;
  if (!(omp > 0.f)) { loss = 0.f; dx = 0.f; return; }
  const float pw1 = powf(omp, ga - 1.f);
  loss = al * pw1 * omp * bce;
  dx = al * pw1 * dbce * fmaf(ga * pt, bce, omp);
}

template <typename T>
__global__ __launch_bounds__(256) void bce_metrics_kernel(BceArgs a) {
  __shared__ float red[4 * 8];
  if (threadIdx.x < 32) red[threadIdx.x] = 0.f;
  __syncthreads();
  const int b = blockIdx.y;
  const size_t plane = (size_t)a.H * a.W;
  float ls[8], eq[8], in_[8], un[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) { ls[n] = eq[n] = in_[n] = un[n] = 0.f; }
  // Four consecutive pixels per thread and step: 16-byte loads of every class plane (2 x NC of them in flight per step) and four
  // 16-byte stores of the gradient slots.  (Round 5: one pixel per step - 4-byte loads, one memory latency per pixel and thread - ran at
  // 1.7 TB/s: 0.23 ms on the data-gradient chain for 0.09 ms of bytes at C2.)  plane % 4 == 0 (the launcher checks); the tail loop below
  // serves planes that are not.
  const bool vec4 = (plane & 3) == 0 && ((((uintptr_t)a.logits) | ((uintptr_t)a.target)) & 15) == 0 && a.loss_out == nullptr && a.dx_out == nullptr;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x, gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec4) {
    for (size_t p = gid * 4; p < plane; p += nthreads * 4) {
      f32x4 xv[8], tv[8];
#pragma unroll
      for (int n = 0; n < 8; ++n)
        if (n < a.NC) {
          const size_t idx = ((size_t)b * a.NC + n) * plane + p;
          xv[n] = *(const f32x4*)(a.logits + idx);
          tv[n] = *(const f32x4*)(a.target + idx);
        }
      float g[4][8];
#pragma unroll
      for (int n = 0; n < 8; ++n) {
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j][n] = 0.f;
        if (n < a.NC) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float x = xv[n][j], t = tv[n][j];
            float l, d;
            loss_elem(a, n, x, t, l, d);
            ls[n] += l;
            g[j][n] = d * a.loss_scale;
            const bool pp = x >= a.thr, gg = t >= a.thr;
            eq[n] += (pp == gg) ? 1.f : 0.f;
            in_[n] += (pp && gg) ? 1.f : 0.f;
            un[n] += (pp || gg) ? 1.f : 0.f;
          }
        }
      }
      if (a.dlogits != nullptr) {
        T* d = (T*)a.dlogits + ((size_t)b * plane + p) * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < 8; ++n) d[j * 8 + n] = from_f32<T>(g[j][n]);
      }
    }
  } else
  for (size_t p = gid; p < plane; p += nthreads) {
    float g[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      g[n] = 0.f;
      if (n < a.NC) {
        const size_t idx = ((size_t)b * a.NC + n) * plane + p;
        const float x = a.logits[idx], t = a.target[idx];
        float l, d;
        loss_elem(a, n, x, t, l, d);
        ls[n] += l;
        g[n] = d * a.loss_scale;
        if (a.loss_out != nullptr) a.loss_out[idx] = l;
        if (a.dx_out != nullptr) a.dx_out[idx] = d;
        const bool pp = x >= a.thr, gg = t >= a.thr;
        eq[n] += (pp == gg) ? 1.f : 0.f;
        in_[n] += (pp && gg) ? 1.f : 0.f;
        un[n] += (pp || gg) ? 1.f : 0.f;
      }
    }
    if (a.dlogits != nullptr) {
      T* d = (T*)a.dlogits + ((size_t)b * plane + p) * 8;
#pragma unroll
      for (int n = 0; n < 8; ++n) d[n] = from_f32<T>(g[n]);
    }
  }
  if (!a.metrics) return;
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    if (n >= a.NC) break;
    float v0 = ls[n], v1 = eq[n], v2 = in_[n], v3 = un[n];
    for (int o = 32; o > 0; o >>= 1) {
      v0 += __shfl_down(v0, o); v1 += __shfl_down(v1, o); v2 += __shfl_down(v2, o); v3 += __shfl_down(v3, o);
    }
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&red[n], v0); atomicAdd(&red[8 + n], v1); atomicAdd(&red[16 + n], v2); atomicAdd(&red[24 + n], v3);
    }
  }
  __syncthreads();
  if (threadIdx.x < a.NC) {
    const int n = threadIdx.x;
    atomic_add_f64(a.out + n, (double)red[n]);
    atomic_add_f64(a.out + a.NC + n, (double)red[8 + n]);
    atomic_add_f64(a.out + 2 * a.NC + (size_t)b * 2 * a.NC + n, (double)red[16 + n]);
    atomic_add_f64(a.out + 2 * a.NC + (size_t)b * 2 * a.NC + a.NC + n, (double)red[24 + n]);
  }
}

hipError_t launch_bce_metrics(const BceArgs& a, int dtype, hipStream_t st) {
  const size_t plane = (size_t)a.H * a.W;
  // <= 4096 pixels per thread keeps the per-thread float counters exact.  Every workgroup ends with 4 x NC fp64 atomics, 2 x NC of them
  // on addresses ALL workgroups share (the loss sums and the accuracy counts): 2400 workgroups at C2 queued 2400 additions on each -
  // about as long as the kernel's bytes take.  So: ~512 workgroups (two per CU; 96 bytes of loads in flight per thread and step), more
  // only where a thread would otherwise walk more than 4096 pixels.
  static const int bce_wgs = lab_int("DMM_BCE_WGS", 512);
  int gx = std::max(1, bce_wgs / std::max(1, a.B));
  gx = std::min(gx, (int)((plane + 256 * 4 - 1) / (256 * 4)));                 // no more than one step of four pixels per thread
  gx = std::max(gx, (int)((plane + 256 * 4096 - 1) / ((size_t)256 * 4096)));
  dim3 grid(gx, a.B);
  if (dtype == DT_F16) hipLaunchKernelGGL(bce_metrics_kernel<f16>, grid, dim3(256), 0, st, a);
  else if (dtype == DT_BF16) hipLaunchKernelGGL(bce_metrics_kernel<bf16>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(bce_metrics_kernel<float>, grid, dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
    float g = a.g[i] * a.grad_scale;
    float p = a.p[i];
    if (a.weight_decay != 0.f) g = fmaf(a.weight_decay, p, g);
    const float m = a.beta1 * a.m[i] + (1.f - a.beta1) * g;
    const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
    a.m[i] = m;
    a.v[i] = v;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    a.p[i] = p - a.step_size * (m / denom);
  }
}

hipError_t launch_adam(const AdamArgs& a, hipStream_t st) {
  int grid = (int)((a.n + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Materialise the deferred BatchNorm-backward correction of a gradient tensor that will be gathered many times
// (several taps / output tiles / K groups): afterwards its consumers read a plain tensor.
// A thread owns ONE channel slot (its eight q / r / ql / rl constants stay in registers for the whole launch: fetched per element they were
// eight times the bytes of the tensors themselves, from cache) and walks pixels four at a time (eight 16-byte loads in flight).
template <typename T>
__global__ __launch_bounds__(256) void apply_corr_kernel(ApplyCorrArgs a, unsigned rows_per_step) {
  constexpr int SLOT = TT<T>::SLOT;
  typedef typename TT<T>::vec V;
  const unsigned ncs = (unsigned)(a.C / SLOT);
  const unsigned gtid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned cs = gtid % ncs, prow = gtid / ncs;
  if (prow >= rows_per_step) return;  // (the grid is rounded up to whole blocks)
  const int c = (int)cs * SLOT;
  float q[SLOT], r[SLOT], ql[SLOT], rl[SLOT];
  load_f32s<SLOT>(a.q + c, q); load_f32s<SLOT>(a.r + c, r); load_f32s<SLOT>(a.ql + c, ql); load_f32s<SLOT>(a.rl + c, rl);
  T* gbase = (T*)a.g + c;
  const T* ybase = (const T*)a.y + c;
  constexpr int U = 4;
  size_t p = prow;
  for (; p + (size_t)(U - 1) * rows_per_step < a.npix; p += (size_t)U * rows_per_step) {
    V gv[U], yv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t pp = p + (size_t)u * rows_per_step;
      gv[u] = *(const V*)(gbase + pp * a.ldg);
      yv[u] = *(const V*)(ybase + pp * a.ldy);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float f[SLOT], y[SLOT];
      vec_to_f32<T>(gv[u], f);
      vec_to_f32<T>(yv[u], y);
#pragma unroll
      for (int k = 0; k < SLOT; ++k) f[k] = (f[k] + fmaf(r[k], y[k], q[k])) + fmaf(rl[k], y[k], ql[k]);
      *(V*)(gbase + (p + (size_t)u * rows_per_step) * a.ldg) = f32_to_vec<T>(f);
    }
  }
  for (; p < a.npix; p += rows_per_step) {
    float f[SLOT], y[SLOT];
    vec_to_f32<T>(*(const V*)(gbase + p * a.ldg), f);
    vec_to_f32<T>(*(const V*)(ybase + p * a.ldy), y);
#pragma unroll
    for (int k = 0; k < SLOT; ++k) f[k] = (f[k] + fmaf(r[k], y[k], q[k])) + fmaf(rl[k], y[k], ql[k]);
    *(V*)(gbase + p * a.ldg) = f32_to_vec<T>(f);
  }
}

hipError_t launch_apply_corr(const ApplyCorrArgs& a, int dtype, hipStream_t st) {
  const unsigned ncs = (unsigned)(a.C / (dtype == DT_F32 ? 4 : 8));
  if (ncs == 0 || a.npix == 0) return hipSuccess;
  // pixel rows walked side by side: up to ~4 M threads, a whole number of channel-slot groups
  size_t rows = std::min<size_t>(a.npix, std::max<size_t>(1, (size_t)(16384 * 256) / ncs));
  const size_t threads = rows * ncs;
  const int grid = (int)((threads + 255) / 256);
  if (dtype == DT_F16) hipLaunchKernelGGL(apply_corr_kernel<f16>, dim3(grid), dim3(256), 0, st, a, (unsigned)rows);
  else if (dtype == DT_BF16) hipLaunchKernelGGL(apply_corr_kernel<bf16>, dim3(grid), dim3(256), 0, st, a, (unsigned)rows);
  else hipLaunchKernelGGL(apply_corr_kernel<float>, dim3(grid), dim3(256), 0, st, a, (unsigned)rows);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// pack: one thread per 16-byte slot of a (chunk, n) row (adjacent lanes write adjacent slots: full-line stores).
// unpack: the same enumeration run backwards.
__device__ __forceinline__ bool pack_locate(const PackDesc& d, int chunk, int kk, int BK, int& seg, int& tap, int& ch) {
  int s = 0, lc = chunk;
  while (s < d.nseg && lc >= d.seg[s].nchunks) { lc -= d.seg[s].nchunks; ++s; }
  if (s >= d.nseg) return false;
  const int e = lc * BK + kk;
  tap = e / d.seg[s].Cpad;
  ch = e - tap * d.seg[s].Cpad;
  seg = s;
  return tap < d.seg[s].ntaps && ch < d.seg[s].Creal;
}

template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const PackDesc* descs, const int* row_prefix, int ndesc, int total_rows) {
  constexpr int SLOT = TT<T>::SLOT, BK = 4 * SLOT;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = gid >> 2, sl = gid & 3;
  if (row >= total_rows) return;
  int lo = 0, hi = ndesc - 1;  // last desc with row_prefix[desc] <= row
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (row_prefix[mid] <= row) lo = mid; else hi = mid - 1;
  }
  const PackDesc& d = descs[lo];
  const int lr = row - row_prefix[lo];
  const int chunk = lr / d.Npad, n = lr - chunk * d.Npad;
  T* dst = (T*)d.dst + ((size_t)chunk * d.Npad + n) * BK + sl * SLOT;
  float out[SLOT];
#pragma unroll
  for (int e = 0; e < SLOT; ++e) {
    const int kk = sl * SLOT + e;
    int s, tap, ch;
    float v = 0.f;
    if (n < d.N && pack_locate(d, chunk, kk, BK, s, tap, ch)) {
      const unsigned tw = d.seg[s].tapw[tap];
      const size_t base = (size_t)n * d.sn + (size_t)(d.seg[s].koff + ch) * d.sk;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned mt = (tw >> (8 * u)) & 0xff;
        if (mt != 0xff) v += d.w[base + (size_t)mt * d.st];
      }
    }
    out[e] = v;
  }
  *(typename TT<T>::vec*)dst = f32_to_vec<T>(out);
}

template <typename T>
__global__ __launch_bounds__(256) void unpack_kernel(const PackDesc* descs, const int* row_prefix, int ndesc, int total_rows,
                                                     float grad_scale) {
  constexpr int SLOT = TT<T>::SLOT, BK = 4 * SLOT;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = gid >> 2, sl = gid & 3;
  if (row >= total_rows) return;
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (row_prefix[mid] <= row) lo = mid; else hi = mid - 1;
  }
  const PackDesc& d = descs[lo];
  if (d.dpack == nullptr || d.gw == nullptr) return;
  const int lr = row - row_prefix[lo];
  const int chunk = lr / d.Npad, n = lr - chunk * d.Npad;
  if (n >= d.N) return;
  const float* src = d.dpack + ((size_t)chunk * d.Npad + n) * BK + sl * SLOT;
  float in[SLOT];
  load_f32s<SLOT>(src, in);
#pragma unroll
  for (int e = 0; e < SLOT; ++e) {
    const int kk = sl * SLOT + e;
    int s, tap, ch;
    if (!pack_locate(d, chunk, kk, BK, s, tap, ch)) continue;
    const float v = in[e] * grad_scale;
    const unsigned tw = d.seg[s].tapw[tap];
    const size_t base = (size_t)n * d.sn + (size_t)(d.seg[s].koff + ch) * d.sk;
    const bool merged = ((tw >> 8) & 0xff) != 0xff;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned mt = (tw >> (8 * u)) & 0xff;
      if (mt == 0xff) continue;
      if (merged || d.shared_master) atomic_add_f32(d.gw + base + (size_t)mt * d.st, v);
      else d.gw[base + (size_t)mt * d.st] = v;
    }
  }
}

// ---- tile kernels (round 4) ----
// The generic kernels above gather one master element per load - at stride 36 bytes for 3x3 weights - and find their descriptor by a
// binary search of dependent loads per THREAD: 0.28 + 0.21 ms at C2, 0.98 + 0.52 ms at C5 for what is a 0.5 / 3 GB copy (10 % of the HBM
// roofline).  Here a workgroup owns 32 rows x 32 channels x all taps of a master tensor: it reads them as 32 contiguous runs of 32 RS
// floats (whichever index the master is contiguous along, PackDesc::tiled), turns the tile in LDS and writes whole 64-byte row pieces
// of the packed layout, 2 KB contiguous per (tap, channel group) - for every descriptor that shares the master (the parity phases of
// a ConvTranspose), which is read once for all of them.  unpack: the same walk backwards; the phases' contributions meet in LDS, so
// the master gradient is written once with plain stores (no atomics, nothing to zero).
constexpr int PT_N = 32;  // rows and channels of a tile
template <typename T, int RS>
__global__ __launch_bounds__(256) void pack_tiles_kernel(const PackDesc* __restrict__ descs, const PackTile* __restrict__ tiles) {
  constexpr int RUN = PT_N * RS, ROW = RUN + 1;
  __shared__ float tile[PT_N * ROW];
  const PackTile t = tiles[blockIdx.x];
  const PackDesc& d = descs[t.desc];
  const int tid = threadIdx.x;
  const bool rows_n = d.tiled == 1;                    // LDS rows are n (else channels)
  const int nvalid = min(PT_N, d.N - t.n0), cvalid = min(PT_N, d.seg[0].Creal - PT_N * t.cg);
  const int outer_valid = rows_n ? nvalid : cvalid, inner_valid = (rows_n ? cvalid : nvalid) * RS;
  const long long ostride = rows_n ? d.sn : d.sk;
  const float* base = d.w + (long long)t.n0 * d.sn + (long long)(PT_N * t.cg) * d.sk;
  if ((((size_t)base | (size_t)(ostride * 4)) & 15) == 0) {   // 16-byte loads (every tensor of the arena starts on a multiple of 4 floats)
    for (int idx = tid; idx < PT_N * RUN / 4; idx += 256) {
      const int o = idx / (RUN / 4), j = (idx - o * (RUN / 4)) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (o < outer_valid && j + 3 < inner_valid) v = *(const f32x4*)(base + o * ostride + j);
      else if (o < outer_valid)
        for (int q = 0; q < 4; ++q) if (j + q < inner_valid) v[q] = base[o * ostride + j + q];
      float* tp = tile + o * ROW + j;   // (ROW is odd: scalar LDS stores)
      tp[0] = v[0]; tp[1] = v[1]; tp[2] = v[2]; tp[3] = v[3];
    }
  } else {
    for (int idx = tid; idx < PT_N * RUN; idx += 256) {
      const int o = idx / RUN, j = idx - o * RUN;
      tile[o * ROW + j] = (o < outer_valid && j < inner_valid) ? base[o * ostride + j] : 0.f;
    }
  }
  __syncthreads();
  const int nsib = t.nsib & 0xff;
  for (int sb = 0; sb < nsib; ++sb) {
    const PackDesc& ds = descs[t.desc + sb];
    const PackSeg& sg = ds.seg[0];
    const int cpt = sg.Cpad / 32;
    for (int u = tid; u < sg.ntaps * (PT_N * 4); u += 256) {
      const int tap = u / (PT_N * 4), rem = u - tap * (PT_N * 4), nn = rem >> 2, sl = rem & 3;
      const unsigned tw = sg.tapw[tap];
      float out[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = sl * 8 + e;
        const int at = rows_n ? nn * ROW + c * RS : c * ROW + nn * RS;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned mt = (tw >> (8 * q)) & 0xff;
          if (mt != 0xff) v += tile[at + mt];
        }
        out[e] = v;
      }
      T* dst = (T*)ds.dst + ((size_t)(tap * cpt + t.cg) * ds.Npad + t.n0 + nn) * 32 + sl * 8;
      *(typename TT<T>::vec*)dst = f32_to_vec<T>(out);
    }
  }
}

template <int RS>
__global__ __launch_bounds__(256) void unpack_tiles_kernel(const PackDesc* __restrict__ descs, const PackTile* __restrict__ tiles, float grad_scale) {
  constexpr int RUN = PT_N * RS, ROW = RUN + 1;
  __shared__ float tile[PT_N * ROW];
  const PackTile t = tiles[blockIdx.x];
  const PackDesc& d = descs[t.desc];
  const int tid = threadIdx.x;
  const bool rows_n = d.tiled == 1;
  const bool exact = (t.nsib & 0x100) != 0;   // every master tap is written by exactly one packed tap of the group: plain LDS stores
  const int nsib = t.nsib & 0xff;
  if (!exact) {
    for (int idx = tid; idx < PT_N * ROW; idx += 256) tile[idx] = 0.f;
    __syncthreads();
  }
  for (int sb = 0; sb < nsib; ++sb) {
    const PackDesc& ds = descs[t.desc + sb];
    const PackSeg& sg = ds.seg[0];
    const int cpt = sg.Cpad / 32;
    for (int u = tid; u < sg.ntaps * (PT_N * 4); u += 256) {
      const int tap = u / (PT_N * 4), rem = u - tap * (PT_N * 4), nn = rem >> 2, sl = rem & 3;
      const unsigned tw = sg.tapw[tap];
      float in[8];
      load_f32s<8>(ds.dpack + ((size_t)(tap * cpt + t.cg) * ds.Npad + t.n0 + nn) * 32 + sl * 8, in);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = sl * 8 + e;
        const int at = rows_n ? nn * ROW + c * RS : c * ROW + nn * RS;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned mt = (tw >> (8 * q)) & 0xff;
          if (mt == 0xff) continue;
          if (exact) tile[at + mt] = in[e];
          else atomicAdd(&tile[at + mt], in[e]);   // (LDS; merged taps and phases meet here)
        }
      }
    }
  }
  __syncthreads();
  const int nvalid = min(PT_N, d.N - t.n0), cvalid = min(PT_N, d.seg[0].Creal - PT_N * t.cg);
  const int outer_valid = rows_n ? nvalid : cvalid, inner_valid = (rows_n ? cvalid : nvalid) * RS;
  const long long ostride = rows_n ? d.sn : d.sk;
  float* base = d.gw + (long long)t.n0 * d.sn + (long long)(PT_N * t.cg) * d.sk;
  if ((((size_t)base | (size_t)(ostride * 4)) & 15) == 0) {
    for (int idx = tid; idx < PT_N * RUN / 4; idx += 256) {
      const int o = idx / (RUN / 4), j = (idx - o * (RUN / 4)) * 4;
      const float* tp = tile + o * ROW + j;
      if (o < outer_valid && j + 3 < inner_valid) {
        const f32x4 v = {tp[0] * grad_scale, tp[1] * grad_scale, tp[2] * grad_scale, tp[3] * grad_scale};
        *(f32x4*)(base + o * ostride + j) = v;
      } else if (o < outer_valid) {
        for (int q = 0; q < 4; ++q) if (j + q < inner_valid) base[o * ostride + j + q] = tp[q] * grad_scale;
      }
    }
  } else {
    for (int idx = tid; idx < PT_N * RUN; idx += 256) {
      const int o = idx / RUN, j = idx - o * RUN;
      if (o < outer_valid && j < inner_valid) base[o * ostride + j] = tile[o * ROW + j] * grad_scale;
    }
  }
}

// A tile range holds the tiles of the 1x1 masters first (nt1), then those of the 3x3 masters (nt9): one launch per instantiation.
hipError_t launch_pack(const PackDesc* descs_dev, const int* prefix_dev, int ndesc, int total_rows, int dtype, hipStream_t st,
                       const PackDesc* tile_descs, const PackTile* tiles_dev, int nt1, int nt9) {
  if (dtype != DT_F32 && tiles_dev != nullptr) {
    if (dtype == DT_F16) {
      if (nt1 > 0) hipLaunchKernelGGL((pack_tiles_kernel<f16, 1>), dim3(nt1), dim3(256), 0, st, tile_descs, tiles_dev);
      if (nt9 > 0) hipLaunchKernelGGL((pack_tiles_kernel<f16, 9>), dim3(nt9), dim3(256), 0, st, tile_descs, tiles_dev + nt1);
    } else {
      if (nt1 > 0) hipLaunchKernelGGL((pack_tiles_kernel<bf16, 1>), dim3(nt1), dim3(256), 0, st, tile_descs, tiles_dev);
      if (nt9 > 0) hipLaunchKernelGGL((pack_tiles_kernel<bf16, 9>), dim3(nt9), dim3(256), 0, st, tile_descs, tiles_dev + nt1);
    }
  }
  if (total_rows <= 0) return hipSuccess;
  dim3 grid((total_rows * 4 + 255) / 256), block(256);
  if (dtype == DT_F16) hipLaunchKernelGGL(pack_kernel<f16>, grid, block, 0, st, descs_dev, prefix_dev, ndesc, total_rows);
  else if (dtype == DT_BF16) hipLaunchKernelGGL(pack_kernel<bf16>, grid, block, 0, st, descs_dev, prefix_dev, ndesc, total_rows);
  else hipLaunchKernelGGL(pack_kernel<float>, grid, block, 0, st, descs_dev, prefix_dev, ndesc, total_rows);
  return hipGetLastError();
}

hipError_t launch_unpack(const PackDesc* descs_dev, const int* prefix_dev, int ndesc, int total_rows, int dtype, float grad_scale,
                         hipStream_t st, const PackDesc* tile_descs, const PackTile* tiles_dev, int nt1, int nt9) {
  if (dtype != DT_F32 && tiles_dev != nullptr) {
    if (nt1 > 0) hipLaunchKernelGGL(unpack_tiles_kernel<1>, dim3(nt1), dim3(256), 0, st, tile_descs, tiles_dev, grad_scale);
    if (nt9 > 0) hipLaunchKernelGGL(unpack_tiles_kernel<9>, dim3(nt9), dim3(256), 0, st, tile_descs, tiles_dev + nt1, grad_scale);
  }
  if (total_rows <= 0) return hipSuccess;
  dim3 grid((total_rows * 4 + 255) / 256), block(256);
  if (dtype != DT_F32)  // the packed gradient is fp32 for every storage type: only the chunk geometry (BK = 32) matters
    hipLaunchKernelGGL(unpack_kernel<f16>, grid, block, 0, st, descs_dev, prefix_dev, ndesc, total_rows, grad_scale);
  else
    hipLaunchKernelGGL(unpack_kernel<float>, grid, block, 0, st, descs_dev, prefix_dev, ndesc, total_rows, grad_scale);
  return hipGetLastError();
}

}  // namespace dmm
