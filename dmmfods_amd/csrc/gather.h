// Device-side operand gather shared by the implicit-GEMM forward/dgrad kernel and the wgrad kernel.
// A K-dimension "chunk" is 64 bytes (BK = 4 slots) of the enumeration  k = tap*Cpad + c  over a segment.
#pragma once
#include "common.h"

namespace dmm {

template <typename T>
__device__ __forceinline__ void vec_to_f32(const typename TT<T>::vec& v, float (&f)[TT<T>::SLOT]) {
#pragma unroll
  for (int i = 0; i < TT<T>::SLOT; ++i) f[i] = to_f32(v[i]);
}
template <typename T>
__device__ __forceinline__ typename TT<T>::vec f32_to_vec(const float (&f)[TT<T>::SLOT]) {
  typename TT<T>::vec v;
#pragma unroll
  for (int i = 0; i < TT<T>::SLOT; ++i) v[i] = from_f32<T>(f[i]);
  return v;
}
template <int S>
__device__ __forceinline__ void load_f32s(const float* p, float (&f)[S]) {
#pragma unroll
  for (int i = 0; i < S; i += 4) {
    f32x4 t = *(const f32x4*)(p + i);
    f[i] = t[0]; f[i + 1] = t[1]; f[i + 2] = t[2]; f[i + 3] = t[3];
  }
}

// Position of a thread in the K enumeration: segment, tap, channel; `left` = chunks left in this segment.
struct KWalk {
  int s, tap, c, left;
};

template <int SLOT>
__device__ __forceinline__ void kw_enter(KWalk& w, const Seg* seg, int s, int j) {
  w.s = s;
  w.left = seg[s].nchunks;
  w.tap = 0;
  w.c = j * SLOT;
  const int cp = seg[s].Cpad;
  while (w.c >= cp) { w.c -= cp; ++w.tap; }
}
template <int SLOT>
__device__ __forceinline__ void kw_next(KWalk& w, const Seg* seg, int nseg, int j) {
  if (--w.left == 0) {
    if (w.s + 1 < nseg) kw_enter<SLOT>(w, seg, w.s + 1, j);
    else w.tap = 1 << 20;  // past the end: gathers return zero
  } else {
    w.c += 4 * SLOT;
    const int cp = seg[w.s].Cpad;
    while (w.c >= cp) { w.c -= cp; ++w.tap; }
  }
}

// Prologue constants of one slot column (8 or 4 channels), loaded once by kernels whose threads keep a fixed channel
// position (wgrad): BN+ReLU uses k0 = scale, k1 = shift; the effective gradient uses k0..k3 = q, r, q_lo, r_lo.
template <int S>
struct SlotK {
  typedef float fv __attribute__((ext_vector_type(S)));  // vector-typed so that the struct lives in registers
  fv k0, k1, k2, k3;
};
template <int S>
__device__ __forceinline__ typename SlotK<S>::fv load_fv(const float* p) {
  typename SlotK<S>::fv v;
#pragma unroll
  for (int i = 0; i < S; i += 4) {
    const f32x4 t = *(const f32x4*)(p + i);
    v[i] = t[0]; v[i + 1] = t[1]; v[i + 2] = t[2]; v[i + 3] = t[3];
  }
  return v;
}
template <int S>
__device__ __forceinline__ SlotK<S> load_slot_consts(const Seg& sg, int c) {
  SlotK<S> k;
  k.k0 = 0.f; k.k1 = 0.f; k.k2 = 0.f; k.k3 = 0.f;
  if (c >= 0 && c < sg.C) {
    if (sg.scale != nullptr) { k.k0 = load_fv<S>(sg.scale + c); k.k1 = load_fv<S>(sg.shift + c); }
    else if (sg.q != nullptr) {
      k.k0 = load_fv<S>(sg.q + c); k.k1 = load_fv<S>(sg.r + c); k.k2 = load_fv<S>(sg.ql + c); k.k3 = load_fv<S>(sg.rl + c);
    }
  }
  return k;
}

// The same constants read from an LDS image built once per workgroup (stage_consts): [k0 | k1 | k2 | k3] x Cst floats.
template <int S>
__device__ __forceinline__ SlotK<S> lds_slot_consts(const float* lk, int cst, int narr, int c) {
  SlotK<S> k;
  k.k0 = 0.f; k.k1 = 0.f; k.k2 = 0.f; k.k3 = 0.f;
  if (narr >= 2 && c >= 0 && c < cst) {
    k.k0 = load_fv<S>(lk + c);
    k.k1 = load_fv<S>(lk + cst + c);
    if (narr == 4) { k.k2 = load_fv<S>(lk + 2 * cst + c); k.k3 = load_fv<S>(lk + 3 * cst + c); }
  }
  return k;
}
template <int S, int NARR>
__device__ __forceinline__ SlotK<S> lds_slot_consts_n(const float* lk, int cst, int c) {
  SlotK<S> k;
  k.k0 = 0.f; k.k1 = 0.f; k.k2 = 0.f; k.k3 = 0.f;
  if constexpr (NARR >= 2) {
    const int cc = c < cst ? c : 0;  // tails read channel 0's constants; their slots are zeroed by state anyway
    k.k0 = load_fv<S>(lk + cc);
    k.k1 = load_fv<S>(lk + cst + cc);
    if constexpr (NARR == 4) { k.k2 = load_fv<S>(lk + 2 * cst + cc); k.k3 = load_fv<S>(lk + 3 * cst + cc); }
  }
  return k;
}
// Cooperative fill of that image; returns the number of floats used.  Call before a __syncthreads().
__device__ __forceinline__ int stage_consts(const Seg& sg, float* lk, int tid, int nthreads) {
  const int cst = sg.C;
  if (sg.scale != nullptr) {
    for (int i = tid; i < cst; i += nthreads) { lk[i] = sg.scale[i]; lk[cst + i] = sg.shift[i]; }
    return 2 * cst;
  }
  if (sg.q != nullptr) {
    for (int i = tid; i < cst; i += nthreads) {
      lk[i] = sg.q[i]; lk[cst + i] = sg.r[i]; lk[2 * cst + i] = sg.ql[i]; lk[3 * cst + i] = sg.rl[i];
    }
    return 4 * cst;
  }
  return 0;
}
__host__ __device__ inline int seg_const_floats(const Seg& sg) { return sg.scale ? 2 * sg.C : (sg.q ? 4 * sg.C : 0); }

// One 16-byte slot of the gathered operand for row pixel (b,y,x) at K position (tap, c) of segment sg,
// with the segment's prologue applied.  Out-of-image taps, channels >= C, taps >= ntaps and invalid rows
// give zeros (zero padding applies AFTER BN+ReLU, as in conv(relu(bn(x)))).
template <typename T, bool PRE = false>
__device__ __forceinline__ typename TT<T>::vec gather_slot(const Seg& sg, int b, int y, int x, bool rowvalid,
                                                           int tap, int c, const SlotK<TT<T>::SLOT>& pre = SlotK<TT<T>::SLOT>()) {
  constexpr int S = TT<T>::SLOT;
  typedef typename TT<T>::vec V;
  V zero;
#pragma unroll
  for (int i = 0; i < S; ++i) zero[i] = (T)0;
  if (!rowvalid || tap >= sg.ntaps || c >= sg.C) return zero;
  const T* base = (const T*)sg.src;
  if (sg.mode == G_POOL2) {
    float acc[S], sc[S], sh[S];
#pragma unroll
    for (int i = 0; i < S; ++i) acc[i] = 0.f;
    const bool bn = sg.scale != nullptr;
    if (bn) {
      if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < S; ++i) { sc[i] = pre.k0[i]; sh[i] = pre.k1[i]; }
      } else { load_f32s<S>(sg.scale + c, sc); load_f32s<S>(sg.shift + c, sh); }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const size_t pix = ((size_t)(b * sg.Hs + 2 * y + (a >> 1)) * sg.Ws + (2 * x + (a & 1)));
      V v = *(const V*)(base + pix * sg.ld + c);
      float f[S];
      vec_to_f32<T>(v, f);
#pragma unroll
      for (int i = 0; i < S; ++i) acc[i] += bn ? fmaxf(fmaf(f[i], sc[i], sh[i]), 0.f) : f[i];
    }
#pragma unroll
    for (int i = 0; i < S; ++i) acc[i] *= 0.25f;
    return f32_to_vec<T>(acc);
  }
  const int t = sg.taps[tap];
  const int dy = (int)(signed char)(t & 0xff), dx = (int)(signed char)((t >> 8) & 0xff);
  int sy = y * sg.istride + dy, sx = x * sg.istride + dx;
  if (sg.mode == G_UP2) {
    if (sy < 0 || sx < 0 || sy >= 2 * sg.Hs || sx >= 2 * sg.Ws) return zero;
    sy >>= 1;
    sx >>= 1;
  } else {
    if (sy < 0 || sx < 0 || sy >= sg.Hs || sx >= sg.Ws) return zero;
  }
  const size_t pix = ((size_t)(b * sg.Hs + sy) * sg.Ws + sx);
  V v = *(const V*)(base + pix * sg.ld + c);
  if (sg.scale != nullptr) {
    float f[S], sc[S], sh[S];
    vec_to_f32<T>(v, f);
    if constexpr (PRE) {
#pragma unroll
      for (int i = 0; i < S; ++i) { sc[i] = pre.k0[i]; sh[i] = pre.k1[i]; }
    } else { load_f32s<S>(sg.scale + c, sc); load_f32s<S>(sg.shift + c, sh); }
#pragma unroll
    for (int i = 0; i < S; ++i) f[i] = fmaxf(fmaf(f[i], sc[i], sh[i]), 0.f);
    return f32_to_vec<T>(f);
  }
  if (sg.q != nullptr) {
    V v2 = *(const V*)((const T*)sg.src2 + pix * sg.ld2 + c);
    float f[S], f2[S], q[S], r[S], ql[S], rl[S];
    vec_to_f32<T>(v, f);
    vec_to_f32<T>(v2, f2);
    if constexpr (PRE) {
#pragma unroll
      for (int i = 0; i < S; ++i) { q[i] = pre.k0[i]; r[i] = pre.k1[i]; ql[i] = pre.k2[i]; rl[i] = pre.k3[i]; }
    } else {
      load_f32s<S>(sg.q + c, q);
      load_f32s<S>(sg.r + c, r);
      load_f32s<S>(sg.ql + c, ql);
      load_f32s<S>(sg.rl + c, rl);
    }
#pragma unroll
    for (int i = 0; i < S; ++i) f[i] = (f[i] + fmaf(r[i], f2[i], q[i])) + fmaf(rl[i], f2[i], ql[i]);
    return f32_to_vec<T>(f);
  }
  return v;
}

// ---- split gather (issue-early / write-late): `gather_issue` only computes addresses and issues the global loads, so
// they stay in flight while the MFMAs of the current K-step run; `gather_finish` applies the prologue one step later,
// just before the slot is written to LDS.  state: 0 = zero slot, 1 = raw data pending, 2 = already final (pool mode).
template <typename T>
struct RawSlot {
  typename TT<T>::vec v, v2;
  int state;
};

template <typename T>
__device__ __forceinline__ RawSlot<T> gather_issue(const Seg& sg, int b, int y, int x, bool rowvalid, int tap, int c,
                                                    const SlotK<TT<T>::SLOT>& kpool) {
  constexpr int S = TT<T>::SLOT;
  typedef typename TT<T>::vec V;
  RawSlot<T> r;
#pragma unroll
  for (int i = 0; i < S; ++i) { r.v[i] = (T)0; r.v2[i] = (T)0; }
  r.state = 0;
  if (!rowvalid || tap >= sg.ntaps || c >= sg.C) return r;
  if (sg.mode == G_POOL2) {  // four loads + averaging: rare (transitions), done synchronously
    r.v = gather_slot<T, true>(sg, b, y, x, rowvalid, tap, c, kpool);
    r.state = 2;
    return r;
  }
  const int t = sg.taps[tap];
  const int dy = (int)(signed char)(t & 0xff), dx = (int)(signed char)((t >> 8) & 0xff);
  int sy = y * sg.istride + dy, sx = x * sg.istride + dx;
  if (sg.mode == G_UP2) {
    if (sy < 0 || sx < 0 || sy >= 2 * sg.Hs || sx >= 2 * sg.Ws) return r;
    sy >>= 1;
    sx >>= 1;
  } else {
    if (sy < 0 || sx < 0 || sy >= sg.Hs || sx >= sg.Ws) return r;
  }
  const size_t pix = ((size_t)(b * sg.Hs + sy) * sg.Ws + sx);
  r.v = *(const V*)((const T*)sg.src + pix * sg.ld + c);
  if (sg.q != nullptr) r.v2 = *(const V*)((const T*)sg.src2 + pix * sg.ld2 + c);
  r.state = 1;
  return r;
}

// PRO: -1 = decide at run time from the segment; 0 = none; 1 = BN+ReLU; 2 = effective gradient (compile-time variants carry
// only their own straight-line code)
template <typename T, int PRO = -1>
__device__ __forceinline__ typename TT<T>::vec gather_finish(const Seg& sg, const RawSlot<T>& r, const SlotK<TT<T>::SLOT>& k) {
  constexpr int S = TT<T>::SLOT;
  if constexpr (PRO >= 0) {
    typename TT<T>::vec out = r.v;
    if constexpr (PRO == 1) {
      float f[S];
      vec_to_f32<T>(r.v, f);
#pragma unroll
      for (int i = 0; i < S; ++i) f[i] = fmaxf(fmaf(f[i], k.k0[i], k.k1[i]), 0.f);
      out = f32_to_vec<T>(f);
    } else if constexpr (PRO == 2) {
      float f[S], f2[S];
      vec_to_f32<T>(r.v, f);
      vec_to_f32<T>(r.v2, f2);
#pragma unroll
      for (int i = 0; i < S; ++i) f[i] = (f[i] + fmaf(k.k1[i], f2[i], k.k0[i])) + fmaf(k.k3[i], f2[i], k.k2[i]);
      out = f32_to_vec<T>(f);
    }
    typename TT<T>::vec zero;
#pragma unroll
  for (int e = 0; e < TT<T>::SLOT; ++e) zero[e] = (T)0;
  // state 0: r.v holds zeros; 2: already final; 3: loaded from a clamped address to keep the load count constant -> zeros
  return r.state == 1 ? out : (r.state == 3 ? zero : r.v);
  }
  if (r.state != 1) return r.v;  // zero or already final
  if (sg.scale != nullptr) {
    float f[S];
    vec_to_f32<T>(r.v, f);
#pragma unroll
    for (int i = 0; i < S; ++i) f[i] = fmaxf(fmaf(f[i], k.k0[i], k.k1[i]), 0.f);
    return f32_to_vec<T>(f);
  }
  if (sg.q != nullptr) {
    float f[S], f2[S];
    vec_to_f32<T>(r.v, f);
    vec_to_f32<T>(r.v2, f2);
#pragma unroll
    for (int i = 0; i < S; ++i) f[i] = (f[i] + fmaf(k.k1[i], f2[i], k.k0[i])) + fmaf(k.k3[i], f2[i], k.k2[i]);
    return f32_to_vec<T>(f);
  }
  return r.v;
}

// BN+ReLU on one 16-byte slot of f16: v_fma_mixlo/mixhi_f16 take the f16 half as an fp32 operand and round the fp32 fma
// once to f16 (identical to (f16)fmaf((float)x, s, t)); ReLU commutes with that rounding and runs packed.
__device__ __forceinline__ f16x8 bn_relu_slot(const f16x8& x, const SlotK<8>& k) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 xi = __builtin_bit_cast(u32x4, x);
  u32x4 o;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(xi[p]), "v"(k.k0[2 * p]), "v"(k.k1[2 * p]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "+v"(d) : "v"(xi[p]), "v"(k.k0[2 * p + 1]), "v"(k.k1[2 * p + 1]));
    // ReLU on the packed pair; written as the instruction itself because fmax() on a value the compiler cannot see through
    // comes with a canonicalising v_pk_max_f16 x, x in front of it
    asm("v_pk_max_f16 %0, %1, 0" : "=v"(d) : "v"(d));
    o[p] = d;
  }
  return __builtin_bit_cast(f16x8, o);
}
// bf16 has no mixed-precision fma: unpack by shift / mask (a bf16 is the upper half of an fp32), fma + max in fp32, one
// v_cvt_pk_bf16_f32 per pair (7 instructions per 2 elements; 3 for f16)
__device__ __forceinline__ bf16x8 bn_relu_slot(const bf16x8& x, const SlotK<8>& k) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 xi = __builtin_bit_cast(u32x4, x);
  u32x4 o;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float lo = __builtin_bit_cast(float, xi[p] << 16), hi = __builtin_bit_cast(float, xi[p] & 0xffff0000u);
    bf16x2 pk;
    pk[0] = (bf16)fmaxf(fmaf(lo, k.k0[2 * p], k.k1[2 * p]), 0.f);
    pk[1] = (bf16)fmaxf(fmaf(hi, k.k0[2 * p + 1], k.k1[2 * p + 1]), 0.f);
    o[p] = __builtin_bit_cast(unsigned, pk);
  }
  return __builtin_bit_cast(bf16x8, o);
}
__device__ __forceinline__ f32x4 bn_relu_slot(const f32x4& x, const SlotK<4>& k) {
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = fmaxf(fmaf(x[i], k.k0[i], k.k1[i]), 0.f);
  return o;
}

// Effective gradient g + q[c] + r[c]*x on one slot.  fp32 tensors use the hi/lo split constants (exact to ~1e-14, the
// parity configuration); for f16 tensors the lo parts are far below the storage rounding, so the slot costs two mixed
// instructions per element: t = fma(f32(x), r, q) and f16(f32(g) + t), rounded once.
__device__ __forceinline__ f16x8 eff_grad_slot(const f16x8& g, const f16x8& x, const SlotK<8>& k) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 gi = __builtin_bit_cast(u32x4, g), xi = __builtin_bit_cast(u32x4, x);
  const float one = 1.f;
  u32x4 o;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float t0, t1;
    unsigned d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(t0) : "v"(xi[p]), "v"(k.k1[2 * p]), "v"(k.k0[2 * p]));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=v"(t1) : "v"(xi[p]), "v"(k.k1[2 * p + 1]), "v"(k.k0[2 * p + 1]));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(gi[p]), "v"(one), "v"(t0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(gi[p]), "v"(one), "v"(t1));
    o[p] = d;
  }
  return __builtin_bit_cast(f16x8, o);
}
__device__ __forceinline__ bf16x8 eff_grad_slot(const bf16x8& g, const bf16x8& x, const SlotK<8>& k) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 gi = __builtin_bit_cast(u32x4, g), xi = __builtin_bit_cast(u32x4, x);
  u32x4 o;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float xl = __builtin_bit_cast(float, xi[p] << 16), xh = __builtin_bit_cast(float, xi[p] & 0xffff0000u);
    const float gl = __builtin_bit_cast(float, gi[p] << 16), gh = __builtin_bit_cast(float, gi[p] & 0xffff0000u);
    bf16x2 pk;
    pk[0] = (bf16)(gl + fmaf(xl, k.k1[2 * p], k.k0[2 * p]));
    pk[1] = (bf16)(gh + fmaf(xh, k.k1[2 * p + 1], k.k0[2 * p + 1]));
    o[p] = __builtin_bit_cast(unsigned, pk);
  }
  return __builtin_bit_cast(bf16x8, o);
}
__device__ __forceinline__ f32x4 eff_grad_slot(const f32x4& g, const f32x4& x, const SlotK<4>& k) {
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (g[i] + fmaf(k.k1[i], x[i], k.k0[i])) + fmaf(k.k3[i], x[i], k.k2[i]);
  return o;
}

// BatchNorm + ReLU backward on one 16-byte slot of f16 (the epilogue of the fused data-gradient kernels): dy = the data gradient of
// relu(bn(x)) (fp32, from the accumulators), x = the raw input of the norm, gold = the gradient already stored for x (ACC).
//   dz = dy where fma(x, sc, sh) > 0 (the forward's own arithmetic), else 0;   s1 += dz;   s2 += dz * x  (UNCENTRED: the caller turns
//   the finished fp64 total into the centred sum  (S2 - mean S1) * invstd  once per channel);   returns f16(gold + sc * dz).
// The mixed-precision fmas read the f16 halves in place and round the result to f16 themselves: 6 instructions per element (mask 3,
// sums 2, result 1) where the generic form (convert x, convert gold, centre, scale, convert back) needs 10-11.
template <bool ACC>
__device__ __forceinline__ f16x8 bnbwd_slot(const float (&dy)[8], const f16x8& x, const f16x8& gold, const float (&sc)[8],
                                            const float (&sh)[8], float (&s1)[8], float (&s2)[8]) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 xi = __builtin_bit_cast(u32x4, x), gi = __builtin_bit_cast(u32x4, gold);
  u32x4 o;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float t0, t1;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(t0) : "v"(xi[p]), "v"(sc[2 * p]), "v"(sh[2 * p]));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t1) : "v"(xi[p]), "v"(sc[2 * p + 1]), "v"(sh[2 * p + 1]));
    const float d0 = t0 > 0.f ? dy[2 * p] : 0.f, d1 = t1 > 0.f ? dy[2 * p + 1] : 0.f;
    s1[2 * p] += d0;
    s1[2 * p + 1] += d1;
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(s2[2 * p]) : "v"(xi[p]), "v"(d0));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(s2[2 * p + 1]) : "v"(xi[p]), "v"(d1));
    unsigned d;
    if constexpr (ACC) {
      asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[0,0,1]" : "=v"(d) : "v"(sc[2 * p]), "v"(d0), "v"(gi[p]));
      asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(d) : "v"(sc[2 * p + 1]), "v"(d1), "v"(gi[p]));
    } else {
      asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(d) : "v"(sc[2 * p]), "v"(d0));
      asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(d) : "v"(sc[2 * p + 1]), "v"(d1));
    }
    o[p] = d;
  }
  return __builtin_bit_cast(f16x8, o);
}

// narr: 0 = no prologue, 2 = BN+ReLU, 4 = effective gradient (run-time value used by PRO < 0 only)
template <typename T, int PRO>
__device__ __forceinline__ typename TT<T>::vec finish_slot(int narr, const RawSlot<T>& r, const SlotK<TT<T>::SLOT>& k) {
  typename TT<T>::vec out = r.v;
  if (PRO == 1 || (PRO < 0 && narr == 2)) out = bn_relu_slot(r.v, k);
  if (PRO == 2 || (PRO < 0 && narr == 4)) out = eff_grad_slot(r.v, r.v2, k);
  typename TT<T>::vec zero;
#pragma unroll
  for (int e = 0; e < TT<T>::SLOT; ++e) zero[e] = (T)0;
  // state 0: r.v holds zeros; 2: already final; 3: loaded from a clamped address to keep the load count constant -> zeros
  return r.state == 1 ? out : (r.state == 3 ? zero : r.v);
}

// Decompose a row index into (b, y, x) of the row grid.
// Sums of per-lane partials over the four 16-lane rows of a wave with the gfx950 lane swaps (v_permlane32_swap / v_permlane16_swap:
// VALU, no LDS round trip).  A swap exchanges halves (rows) of TWO registers, so one swap + one add folds two values at once:
// 16 values cost 12 swaps + 12 adds (two __shfl_xor steps per value: 32 ds_bpermute + 32 adds) and leave every lane with ONE
// finished wave total per group of four values instead of all of them in every lane.
//   in:  v[0 .. 4 NQ)   this lane's partials
//   out: w[m], m < NQ = the wave total (over lanes l, l^16, l^32, l^48) of value 4 m + fold_pick(lane)
// (The builtins: hipcc pads the 2 wait states the swaps need behind a VALU write of either operand itself - cdna_hip_programming.md T21.)
#ifndef FOLD_VARIANT
#define FOLD_VARIANT 0  // experiment builds: bit 0 the lane swaps as ds_bpermute shuffles (same sums, same order), bit 1 the DPP row rotations as shuffles
#endif
__device__ __forceinline__ float fold_swap32(float a, float b) {  // lanes 0-31: a(l) + a(l+32); lanes 32-63: b(l-32) + b(l)
  if constexpr (FOLD_VARIANT & 1) {
    const float ax = __shfl_xor(a, 32, 64), bx = __shfl_xor(b, 32, 64);
    return (__lane_id() & 32) ? bx + b : a + ax;
  }
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float fold_swap16(float a, float b) {  // rows 0,2: a(row) + a(row+1); rows 1,3: b(row-1) + b(row)
  if constexpr (FOLD_VARIANT & 1) {
    const float ax = __shfl_xor(a, 16, 64), bx = __shfl_xor(b, 16, 64);
    return (__lane_id() & 16) ? bx + b : a + ax;
  }
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ int fold_pick(int lane) {  // rows 0..3 of the wave end up with values 0, 2, 1, 3 of each group of four
  const int g = lane >> 4;
  return ((g & 1) << 1) | (g >> 1);
}
template <int NQ>
__device__ __forceinline__ void fold_rows(const float (&v)[4 * NQ], float (&w)[NQ]) {
#pragma unroll
  for (int m = 0; m < NQ; ++m) {
    const float u0 = fold_swap32(v[4 * m], v[4 * m + 1]);      // rows 0,1: value 4m (rows r, r+2 added); rows 2,3: value 4m+1
    const float u1 = fold_swap32(v[4 * m + 2], v[4 * m + 3]);  // rows 0,1: value 4m+2;                  rows 2,3: value 4m+3
    w[m] = fold_swap16(u0, u1);                                // rows 0..3: totals of 4m, 4m+2, 4m+1, 4m+3
  }
}

// Epilogue reductions of the convolution kernels: every lane holds partials s1[SLOT] | s2[SLOT] of its slot column cv = lane % NCV
// (NCV = 4, 8, 16 or 32 columns; the lanes l, l + NCV, ... of a wave share a column).  Adds the wave totals to the workgroup's fp64
// accumulators red[fold_slot(0, ch)] (s1) and red[fold_slot(1, ch)] (s2) in LDS.  Lanes inside a 16-lane row are folded by DPP row rotations
// (one v_add_f32 each), the rows by fold_rows: every lane is left with a quarter of the column's totals and issues SLOT/2 LDS atomics
// (a __shfl_xor per step and value, then 2 SLOT atomics on 1/4 .. 1/16 of the lanes before round 3).
// Layout of the accumulators: VALUE-major - red[v * NCV + cv], v = which * SLOT + e - so that the lanes of one LDS atomic (16 / 8 / 4
// columns x the four values the rows hold) touch 64 / 32 / 16 CONSECUTIVE doubles: every bank the minimum number of times.  (Channel-major,
// as in round 2, the 64 lanes met 4-deep on a quarter of the banks: SQ_LDS_BANK_CONFLICT per LDS-active cycle 0.49 -> 0.76 on bw1 when all
// lanes started to issue these atomics.)  fold_slot: where channel ch of the tile (ch = cv * SLOT + e) keeps sum `which`.
template <int NCV, int SLOT>
__device__ __forceinline__ int fold_slot(int which, int ch) { return (which * SLOT + (ch % SLOT)) * NCV + ch / SLOT; }

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  if constexpr (FOLD_VARIANT & 2) return v + __shfl_xor(v, CTRL == 0x128 ? 8 : 4, 64);  // (row_ror:8 / :4 on values already equal across the rotated halves)
  return v + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
template <int NCV, int SLOT, int BN>
__device__ __forceinline__ void fold_to_lds(const float (&s1)[SLOT], const float (&s2)[SLOT], double* red, int cv, bool colvalid, int lane) {
  static_assert((NCV == 4 || NCV == 8 || NCV == 16 || NCV == 32) && SLOT % 4 == 0, "slot columns per wave");
  float v[2 * SLOT];
#pragma unroll
  for (int e = 0; e < SLOT; ++e) { v[e] = s1[e]; v[SLOT + e] = s2[e]; }
  if constexpr (NCV == 32) {  // two lanes per column: one swap folds a pair of values
#pragma unroll
    for (int k = 0; k < SLOT; ++k) {
      const float t = fold_swap32(v[2 * k], v[2 * k + 1]);  // lanes 0-31: value 2k, lanes 32-63: value 2k+1
      const int vi = 2 * k + (lane >> 5);
      if (colvalid) atomicAdd(&red[vi * NCV + cv], (double)t);
    }
  } else {
    if constexpr (NCV <= 8) {
#pragma unroll
      for (int e = 0; e < 2 * SLOT; ++e) v[e] = dpp_add<0x128>(v[e]);  // row_ror:8
    }
    if constexpr (NCV <= 4) {
#pragma unroll
      for (int e = 0; e < 2 * SLOT; ++e) v[e] = dpp_add<0x124>(v[e]);  // row_ror:4
    }
    float w[SLOT / 2];
    fold_rows<SLOT / 2>(v, w);
    if (colvalid && (lane & 15) < NCV) {
      const int pick = fold_pick(lane);
#pragma unroll
      for (int m = 0; m < SLOT / 2; ++m) {
        const int vi = 4 * m + pick;
        atomicAdd(&red[vi * NCV + cv], (double)w[m]);
      }
    }
  }
}

__device__ __forceinline__ void row_to_byx(int m, int Ho, int Wo, int& b, int& y, int& x) {
  x = m % Wo;
  const int t = m / Wo;
  y = t % Ho;
  b = t / Ho;
}

}  // namespace dmm
