// Weight gradient of the heat-map head's last convolution `refine1` (5x5, 64 -> 3 channels at full resolution; reference
// M:128-131) on LDS tiles, gfx950, 16-bit storage types.
//
//   dW[n][c][tap] = sum over pixels p of  A[p][c] * dY[p + t_tap][n]          (transposed form of wgrad.hip: taps on the dY side)
//     A  = relu(bn(x))   64 channels, normalised ONCE per pixel
//     dY = dL/dlogits    3 channels stored as one 16-byte slot of 8
// The result has 25 x 8 = 200 (tap, n) columns per input channel: seven 32-column chunks of the packed gradient, four taps each.
// The generic kernel runs this as a GEMM with K = 200 gathered from dY tap by tap and re-reads (and re-normalises) A for every
// chunk group: 1.2 TB/s on a layer whose whole traffic is one pass over A (1.26 GB at C2).  Here a PERSISTENT workgroup walks
// 8 x 16 pixel tiles: A tile (128 px x 64 ch) and the 12 x 20 pixel dY halo go to LDS once, the (tap, n) columns of a chunk are
// simply per-lane addresses into the halo image (a lane's four consecutive columns belong to one tap), both operands come from
// ds_read_b64_tr_b16, and the 7 x 64 x 32 result lives in the accumulators of the four waves until the end (one round of atomics).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "gather.h"

namespace dmm {

constexpr int W5_TH = 8, W5_TW = 16;
#ifndef W5_F_PER_CU
#define W5_F_PER_CU 3   // workgroups per CU of the factor forms (PA = 3: 166 registers; PA = 0, PY = 2: 154): three waves per SIMD (lab: -DW5_F_PER_CU=2)
#endif
constexpr int W5_CA = 64;
constexpr int W5_A_BYTES = BM * W5_CA * 2;                    // 16 KB, 128-byte rows, 64-byte granule XOR-ed with (row >> 1) & 1
// TR = tap radius, STR = source stride of the thin operand: (5x5, stride 1) = the head's last convolution; (7x7, stride 2) = the
// stem convolution conv0 (reference M:47-52 / torchvision features.conv0), whose thin operand is the raw input.
// TCOL = columns per tap: 8 (the thin operand's eight channels), 16 (PY == 2: two variants of each channel, see below) or 4 (PA == 3: the
//        first four channels of the thin operand only - its real channels, the classes, must number <= 4 - so that the 25 taps are
//        four 32-column chunks instead of seven; the column order is private to wg5_kernel<PA = 3> and wg5_fin64_kernel)
template <int TR, int STR, int TCOL = 8>
struct W5Geo {
  static constexpr int NT = (2 * TR + 1) * (2 * TR + 1);        // taps
  static constexpr int NCH = (NT * TCOL + 31) / 32;             // 32-column chunks of the (tap, thin channel) columns: 7 / 13 / 5
  static constexpr int NQ = (NCH + 1) / 2;                      // chunks per wave pair
  static constexpr int HH = STR * (W5_TH - 1) + 2 * TR + 1, HW = STR * (W5_TW - 1) + 2 * TR + 1;  // halo: 12 x 20 / 21 x 37
  static constexpr int YP = 2 * TCOL;                           // bytes of a halo pixel
  static constexpr int Y_BYTES = HH * HW * YP + 64;             // halo image + a zero line for the column groups past the last tap
  static constexpr int LDS = W5_A_BYTES + Y_BYTES;
  static constexpr int LDS2 = 2 * W5_A_BYTES + Y_BYTES + 2 * W5_CA * 4;  // PA == 3: two images of the 64-channel operand + its constants
  static constexpr int NYL = (HH * HW + NTHREADS - 1) / NTHREADS;  // halo pixels per thread
};

struct Wg5Args {
  WgradArgs w;
  int tiles_y, tiles_x, ntiles, tiles_per_wg;
};

typedef unsigned w5_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ w5_u32x2 w5_tr16(const unsigned char* p) {
  typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
  h4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(p));
  return __builtin_bit_cast(w5_u32x2, r);
}
template <typename T>
__device__ __forceinline__ typename TT<T>::vec w5_frag(const w5_u32x2& lo, const w5_u32x2& hi) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(typename TT<T>::vec, v);
}

// The two factors of an activation on one 16-byte slot: relu(bn(x)) = scale (m x) + shift m with m = [f16(fma(x, scale, shift)) > 0]
// (the forward's own value).  mx = m x is the STORED x or zero - exact, no rounding of its own - and mm = m; `okm` = all ones / zero
// (the slot's pixel lies inside / outside the image).  k0 scale, k1 shift.
// f16: mixed-precision fmas on the packed halves and integer mask arithmetic, 4 instructions per element:
//   d = f16(fma(x, k0, k1))  (2 per pair)    t = min_u16(max_i16(d, 0), 1) & okm: 1 where d > 0 - as integers negative halves and -0
//   order below +0 - (3)    mm = t * 0x3C00 = 1.0h, mall = t * 0xFFFF (2)    mx = x & mall (1)
__device__ __forceinline__ void factor_slot(const f16x8& x, const SlotK<8>& k, unsigned okm, f16x8& mx, f16x8& mm) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 xi = __builtin_bit_cast(u32x4, x);
  const unsigned zero2 = 0u, one2 = 0x00010001u, h1 = 0x3C003C00u, all2 = 0xFFFFFFFFu;
  u32x4 om, ox;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned d, t;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(xi[p]), "v"(k.k0[2 * p]), "v"(k.k1[2 * p]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(xi[p]), "v"(k.k0[2 * p + 1]), "v"(k.k1[2 * p + 1]));
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(t) : "v"(d), "v"(zero2));
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(t), "v"(one2));
    t &= okm;
    unsigned m1, ma;
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(m1) : "v"(t), "v"(h1));
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(ma) : "v"(t), "v"(all2));
    om[p] = m1;
    ox[p] = xi[p] & ma;
  }
  mx = __builtin_bit_cast(f16x8, ox);
  mm = __builtin_bit_cast(f16x8, om);
}
__device__ __forceinline__ void factor_slot(const bf16x8& x, const SlotK<8>& k, unsigned okm, bf16x8& mx, bf16x8& mm) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 xi = __builtin_bit_cast(u32x4, x);
  u32x4 om, ox;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float lo = __builtin_bit_cast(float, xi[p] << 16), hi = __builtin_bit_cast(float, xi[p] & 0xffff0000u);
    // (the forward's own rounded value: bn_relu_slot)
    const bool on0 = okm != 0u && (float)(bf16)fmaf(lo, k.k0[2 * p], k.k1[2 * p]) > 0.f;
    const bool on1 = okm != 0u && (float)(bf16)fmaf(hi, k.k0[2 * p + 1], k.k1[2 * p + 1]) > 0.f;
    om[p] = (on0 ? 0x3F80u : 0u) | (on1 ? 0x3F800000u : 0u);
    ox[p] = xi[p] & ((on0 ? 0xFFFFu : 0u) | (on1 ? 0xFFFF0000u : 0u));
  }
  mx = __builtin_bit_cast(bf16x8, ox);
  mm = __builtin_bit_cast(bf16x8, om);
}

// PA = prologue of the 64-channel operand: 1 BN+ReLU (an activation), 0 none / 2 effective gradient (an output gradient),
//      3 (round 5, second half) the two FACTORS of the activation on the 64-channel side, relu(bn(x)) = scale (m x) + shift m with
//      m = [bn(x) > 0]: two LDS images [m x | m] - the stored x itself or zero: no rounding -, two accumulator sets, results to `sbuf`
//      [2][4][64][32] doubles (variant 0 = S2 = corr(m x, dY), 1 = S1 = corr(m, dY)).  The head's 5x5 convolution `refine1` (reference
//      M:128-131): its weight gradient dW = scale S2 + shift S1 AND the BatchNorm-backward reductions of norm1, sum dz = sum_{tap,n}
//      W[n][c][tap] S1[c][tap][n] and sum dz xhat = (sum W S2 - mean sum W S1) invstd (wg5_fin64_kernel) - the reductions-only FIRST
//      pass of the two-pass data gradient (0.68 ms, one pass over 1.26 GB at C2) is not run any more;
// PY = prologue of the thin operand: 1 BN+ReLU (the raw-input channels of the head's first convolution sit behind norm0), 0 none,
//      2 (round 5) the two FACTORS of that activation: relu(bn(x)) = gamma * (m * xhat) + beta * m with m = [bn(x) > 0] and xhat the
//      normalised input.  The halo pixel holds [m * xhat (8 channels) | m (8 channels)], the kernel correlates both with the output
//      gradient - S2[c][tap][n] = sum_p gY[p][c] (m xhat)[p + tap][n], S1 likewise with m - and writes them to `sbuf` instead of the
//      packed gradient.  From S1, S2 follow BOTH the weight gradient dW = gamma S2 + beta S1 AND the BatchNorm-backward reductions of the
//      raw-input channels, sum dz = sum_{c,tap} W[c][n][tap] S1[c][tap][n] and sum dz xhat likewise with S2 (wg5_rawfin_kernel) - the
//      full-resolution data gradient towards the raw input (0.69 ms, one more pass over the 1.26 GB gradient at C2) existed only for
//      those two sums per channel.
template <typename T, int TR, int STR, int PA, int PY>
__global__ __launch_bounds__(NTHREADS, (TR == 3 ? 1 : ((PA == 3 || (PA == 0 && PY == 2)) ? W5_F_PER_CU : 2))) void wg5_kernel(const Wg5Args g) {  // (the stem form: 7 accumulator tiles + two operand sets)
  static_assert(sizeof(T) == 2, "16-bit storage");
  typedef typename TT<T>::vec V;
  constexpr int TCOL = PY == 2 ? 16 : (PA == 3 ? 4 : 8);
  typedef W5Geo<TR, STR, TCOL> G5;
  constexpr int W5_HH = G5::HH, W5_HW = G5::HW, W5_NCH = G5::NCH, NYL = G5::NYL, YP = G5::YP;
  constexpr int SLOT = 8;
  constexpr int NA = BM * (W5_CA / SLOT) / NTHREADS;  // 4 A slots per thread
  const WgradArgs& a = g.w;
  const Seg& sy = a.seg[0];  // dY, 25 taps
  const Seg& sa = a.dy;      // A, pixel aligned

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NIMG = PA == 3 ? 2 : 1;
  unsigned char* As = smem;
  unsigned char* Ys = smem + NIMG * W5_A_BYTES;
  constexpr int ZERO = W5_HH * W5_HW * YP;  // offset of the zero line in the dY image

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t_beg = blockIdx.x * g.tiles_per_wg, t_end = min(g.ntiles, t_beg + g.tiles_per_wg);
  if (t_beg >= t_end) return;

  // ---- fixed channel positions: prologue constants once ----
  const int ca = tid & 7, pa0 = tid >> 3;  // A: slot column, pixels pa0 + 32 i
  SlotK<SLOT> ka;
  ka.k0 = 0.f; ka.k1 = 0.f; ka.k2 = 0.f; ka.k3 = 0.f;
  if (PA == 1) { ka.k0 = load_fv<SLOT>(sa.scale + ca * SLOT); ka.k1 = load_fv<SLOT>(sa.shift + ca * SLOT); }
  if (PA == 2) { ka.k0 = load_fv<SLOT>(sa.q + ca * SLOT); ka.k1 = load_fv<SLOT>(sa.r + ca * SLOT); }
  // PA == 3: [scale | shift] x 64 in LDS, read where a tile is stored; held in registers across the contraction the constants spill
  // the two accumulator sets
  float* kcl = (float*)(smem + NIMG * W5_A_BYTES + G5::Y_BYTES);
  if (PA == 3 && tid < W5_CA) { kcl[tid] = sa.scale[tid]; kcl[W5_CA + tid] = sa.shift[tid]; }
  if (PA == 3) __syncthreads();
  const T* asrc = (const T*)sa.src + ca * SLOT;
  const T* asrc2 = (const T*)sa.src2 + ca * SLOT;
  const T* ysrc = (const T*)sy.src;
  SlotK<SLOT> kyk;  // the thin operand's eight channels: the same constants for every thread
  kyk.k0 = 0.f; kyk.k1 = 0.f; kyk.k2 = 0.f; kyk.k3 = 0.f;
  if (PY == 1) { kyk.k0 = load_fv<SLOT>(sy.scale); kyk.k1 = load_fv<SLOT>(sy.shift); }
  if (PY == 2) { kyk.k0 = load_fv<SLOT>(sy.scale); kyk.k1 = load_fv<SLOT>(sy.shift); kyk.k2 = load_fv<SLOT>(a.t_mean); kyk.k3 = load_fv<SLOT>(a.t_invstd); }
  int alds[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int p = pa0 + 32 * i;
    alds[i] = p * 128 + ((ca * 16) ^ (((p >> 1) & 1) << 6));
  }
  if (tid < 4) *(V*)(Ys + ZERO + tid * 16) = V{};  // the zero line (never rewritten)
  (void)NA;

  // Two register sets: the loads of tile t + 2 are issued while tile t is contracted (a tile's contraction is short next to a
  // memory latency).  Every load is issued unconditionally (clamped tile / pixel), so the number of loads behind a given one is a
  // compile-time constant and the compiler's counted s_waitcnt in store() leaves the other set in flight.
  struct Regs {
    V ra[NA], ra2[PA == 2 ? NA : 1], ry[NYL];
    unsigned oka, oky;
  };
  Regs R0, R1;
  const int tiles_img = g.tiles_y * g.tiles_x;
  auto issue = [&](Regs& R, int tile_) {
    const int tile = min(tile_, t_end - 1);  // past the end: the last tile again (never stored)
    const int b = tile / tiles_img, tr = tile - b * tiles_img;
    const int y0 = (tr / g.tiles_x) * W5_TH, x0 = (tr % g.tiles_x) * W5_TW;
    R.oka = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {  // branch-free: clamped address, zeroed at the write if outside
      const int p = pa0 + 32 * i;
      const int y = y0 + (p >> 4), x = x0 + (p & 15);
      if (y < a.Ho && x < a.Wo) R.oka |= 1u << i;
      const size_t pix = (size_t)(b * sa.Hs + min(y, sa.Hs - 1)) * sa.Ws + min(x, sa.Ws - 1);
      R.ra[i] = *(const V*)(asrc + pix * sa.ld);
      if constexpr (PA == 2) R.ra2[i] = *(const V*)(asrc2 + pix * sa.ld2);
    }
    R.oky = 0;
#pragma unroll
    for (int i = 0; i < NYL; ++i) {
      const int hp = tid + NTHREADS * i;
      const int hy = hp / W5_HW, hx = hp - hy * W5_HW;
      const int y = y0 * STR - TR + hy, x = x0 * STR - TR + hx;
      if (hp < W5_HH * W5_HW && (unsigned)y < (unsigned)sy.Hs && (unsigned)x < (unsigned)sy.Ws) R.oky |= 1u << i;
      const size_t pix = (size_t)(b * sy.Hs + min(max(y, 0), sy.Hs - 1)) * sy.Ws + min(max(x, 0), sy.Ws - 1);
      R.ry[i] = *(const V*)(ysrc + pix * sy.ld);
    }
  };
  auto store = [&](const Regs& R) {
    V z;
#pragma unroll
    for (int e = 0; e < SLOT; ++e) z[e] = (T)0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if constexpr (PA == 3) {
        SlotK<SLOT> kx;
        kx.k0 = load_fv<SLOT>(kcl + ca * SLOT); kx.k1 = load_fv<SLOT>(kcl + W5_CA + ca * SLOT); kx.k2 = 0.f; kx.k3 = 0.f;
        V mx, mm;
        factor_slot(R.ra[i], kx, ((R.oka >> i) & 1) ? 0xFFFFFFFFu : 0u, mx, mm);
        *(V*)(As + alds[i]) = mx;
        *(V*)(As + W5_A_BYTES + alds[i]) = mm;
      } else {
      V v = R.ra[i];
      if constexpr (PA == 1) v = bn_relu_slot(R.ra[i], ka);
      if constexpr (PA == 2) v = eff_grad_slot(R.ra[i], R.ra2[i], ka);
      *(V*)(As + alds[i]) = ((R.oka >> i) & 1) ? v : z;
      }
    }
#pragma unroll
    for (int i = 0; i < NYL; ++i) {
      const int hp = tid + NTHREADS * i;
      if (hp < W5_HH * W5_HW) {
        if constexpr (PY == 2) {
          float xf[SLOT], mx[SLOT], mm[SLOT];
          vec_to_f32<T>(R.ry[i], xf);
#pragma unroll
          for (int e = 0; e < SLOT; ++e) {
            const bool on = fmaf(xf[e], kyk.k0[e], kyk.k1[e]) > 0.f;    // the forward's own test (scale / shift of the same tables)
            mm[e] = on ? 1.f : 0.f;
            mx[e] = on ? (xf[e] - kyk.k2[e]) * kyk.k3[e] : 0.f;
          }
          const bool ok = (R.oky >> i) & 1;
          *(V*)(Ys + hp * YP) = ok ? f32_to_vec<T>(mx) : z;
          *(V*)(Ys + hp * YP + 16) = ok ? f32_to_vec<T>(mm) : z;
        } else if constexpr (TCOL == 4) {   // the first four channels of the slot only
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          const u32x4 yv = __builtin_bit_cast(u32x4, R.ry[i]);
          const bool ok = (R.oky >> i) & 1;
          const u32x2 y2 = {ok ? yv[0] : 0u, ok ? yv[1] : 0u};
          *(u32x2*)(Ys + hp * YP) = y2;
        } else {
          V v = R.ry[i];
          if constexpr (PY == 1) v = bn_relu_slot(R.ry[i], kyk);
          *(V*)(Ys + hp * 16) = ((R.oky >> i) & 1) ? v : z;  // zero padding applies AFTER the prologue
        }
      }
    }
  };

  // wave w: 64-channel operand rows 32 (w & 1) .., chunks (w >> 1), (w >> 1) + 2, ...
  constexpr int NQ = G5::NQ;
  const int cw = wave & 1, q0 = wave >> 1;
  f32x16 acc[NQ], acc1[PA == 3 ? NQ : 1];
#pragma unroll
  for (int m = 0; m < NQ; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[m][i] = 0.f; if (PA == 3) acc1[m][i] = 0.f; }

  // transposed-read lane geometry (see wgrad.hip): group tg = lane >> 4 covers columns 16 (tg & 1) .., rows 8 (tg >> 1) + tq (+4)
  const int tg = lane >> 4, ti = lane & 15, tq = ti >> 2, tp = ti & 3;
  const int arow = 8 * (tg >> 1) + tq;
  const int acol = ((32 * cw + 16 * (tg & 1) + 4 * tp) * 2) ^ (((arow >> 1) & 1) << 6);
  // a lane's four consecutive (tap, n) columns 16 (tg & 1) + 4 tp .. belong to tap 4 q + 2 (tg & 1) + (tp >> 1), n = 4 (tp & 1) ..
  int boff[NQ], bstep[NQ], bsec[NQ];
#pragma unroll
  for (int m = 0; m < NQ; ++m) {
    const int q = q0 + 2 * m;
    // TCOL = 8: tap 4 q + 2 (tg & 1) + (tp >> 1), channels 4 (tp & 1) ..; TCOL = 16: tap 2 q + (tg & 1), variant tp >> 1, channels 4 (tp & 1) ..
    // TCOL = 4: tap 8 q + 4 (tg & 1) + tp, channels 0 .. 3
    const int tap = TCOL == 8 ? 4 * q + 2 * (tg & 1) + (tp >> 1) : (TCOL == 4 ? 8 * q + 4 * (tg & 1) + tp : 2 * q + (tg & 1));
    const int sub = TCOL == 8 ? (tp & 1) * 8 : (TCOL == 4 ? 0 : (tp >> 1) * 16 + (tp & 1) * 8);
    const bool live = q < W5_NCH && tap < sy.ntaps;
    const int tw = sy.taps[live ? tap : 0];
    const int dy = (int)(signed char)(tw & 0xff), dx = (int)(signed char)((tw >> 8) & 0xff);
    boff[m] = live ? ((TR + dy) * W5_HW + STR * arow + TR + dx) * YP + sub : ZERO + (tp & 1) * 8;
    bstep[m] = live ? STR * W5_HW * YP : 0;  // one tile row further in the halo image
    bsec[m] = live ? STR * 4 * YP : 0;       // the fragment's second half: 4 pixels further
  }

  auto contract = [&]() {
#pragma unroll 2
    for (int ms = 0; ms < W5_TH; ++ms) {  // one tile row = 16 pixels of the contraction per step
      const unsigned char* ap = As + (16 * ms + arow) * 128 + acol;
      const V af = w5_frag<T>(w5_tr16(ap), w5_tr16(ap + 4 * 128));
      V af1 = af;
      if constexpr (PA == 3) af1 = w5_frag<T>(w5_tr16(ap + W5_A_BYTES), w5_tr16(ap + W5_A_BYTES + 4 * 128));
#pragma unroll
      for (int m = 0; m < NQ; ++m) {
        if (q0 + 2 * m < W5_NCH) {  // (wave-uniform)
          const unsigned char* yp = Ys + boff[m] + ms * bstep[m];
          const V bf = w5_frag<T>(w5_tr16(yp), w5_tr16(yp + bsec[m]));
          acc[m] = mma16(af, bf, acc[m]);
          if constexpr (PA == 3) acc1[m] = mma16(af1, bf, acc1[m]);
        }
      }
    }
  };
  issue(R0, t_beg);
  issue(R1, t_beg + 1);
  for (int tile = t_beg; tile < t_end; tile += 2) {
    store(R0);        // waits for this tile's loads (R1's stay in flight)
    __syncthreads();  // images complete
    issue(R0, tile + 2);
    contract();
    __syncthreads();  // all waves done with the images
    if (tile + 1 >= t_end) break;
    store(R1);
    __syncthreads();
    issue(R1, tile + 3);
    contract();
    __syncthreads();
  }

  // ---- add the partial result to the packed gradient: dP[chunk][c][k % 32] ----
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int m = 0; m < NQ; ++m) {
    const int q = q0 + 2 * m;
    if (q >= W5_NCH) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = 32 * cw + (i & 3) + 8 * (i >> 2) + 4 * h;
      if constexpr (PA == 3) {
        // fp64 across workgroups: these sums become BatchNorm-backward reductions of a norm ON the data-gradient chain - with fp32 atomics
        // (4e-7 of run-to-run noise in a sum) the whole encoder's gradients moved by 3e-3 between two runs (DenseNet-121 amplifies a
        // per-channel offset of the head's gradient ~1e4-fold); a workgroup's own fp32 accumulation is deterministic and averages out
        double* dst = (double*)a.sbuf + ((size_t)q * a.Npad + c) * 32 + r;
        atomic_add_f64(dst, (double)acc[m][i]);
        atomic_add_f64(dst + (size_t)W5_NCH * a.Npad * 32, (double)acc1[m][i]);
      } else {
        atomic_add_f32((PY == 2 ? a.sbuf : a.dpack) + ((size_t)q * a.Npad + c) * 32 + r, acc[m][i]);
      }
    }
  }
}

static bool g_wg5 = !lab_flag("DMM_NO_WG5");
void wg5_set_enabled(bool on) { g_wg5 = on; }


template <typename T, int TR, int STR, int PA, int PY = 0>
static hipError_t launch_wg5_t(const Wg5Args& g, int nwg, hipStream_t st) {
  auto kern = wg5_kernel<T, TR, STR, PA, PY>;
  constexpr int lds = PA == 3 ? W5Geo<TR, STR, 4>::LDS2 : W5Geo<TR, STR, PY == 2 ? 16 : 8>::LDS;
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NTHREADS), lds, st, g);
  return hipGetLastError();
}

// Takes a weight gradient whose tapped operand (seg[0]) is ONE 8-channel slot and whose pixel-aligned operand (dy) has 64 channels,
// 16-bit storage:  (a) 25 taps at stride 1, the 64 channels an activation (BN+ReLU): the head's 5x5 convolution, transposed form;
// (b) 49 taps at stride 2 over the raw input, the 64 channels an output gradient (plain or with the deferred correction): the
// stem's 7x7 convolution, normal form;  (c) 9 taps at stride 1 over the BN+ReLU-normalised raw input, the 64 channels an output
// gradient: the raw-input segment of the head's first convolution (reference M:126-127), normal form, all four output parities of
// the plan's phase decomposition in one pass over the full-resolution gradient.  Returns hipErrorNotSupported otherwise.
hipError_t launch_wg5(const WgradArgs& a, int dtype, hipStream_t st) {
  if (!family_on(g_wg5, IMPL_WG5) || dtype == DT_F32 || a.nseg != 1) return hipErrorNotSupported;
  const Seg& q = a.seg[0];
  const Seg& p = a.dy;
  const bool stem = q.ntaps == 49, raw3 = q.ntaps == 9;
  const int tr = stem ? 3 : (raw3 ? 1 : 2), str = stem ? 2 : 1;
  if (q.mode != G_PLAIN || q.istride != str || (q.ntaps != 25 && q.ntaps != 49 && q.ntaps != 9) || q.C != 8 || q.Cpad != 8 || q.Hs != str * a.Ho ||
      q.Ws != str * a.Wo || (q.scale != nullptr) != raw3 || q.q != nullptr)
    return hipErrorNotSupported;
  if (q.nchunks != (q.ntaps * 8 + 31) / 32) return hipErrorNotSupported;
  const bool factors = a.sbuf != nullptr;   // PY = 2: the activation's two factors, results to sbuf
  const bool head5 = !stem && !raw3;
  if (factors && !((raw3 || head5) && a.t_mean != nullptr && a.t_invstd != nullptr)) return hipErrorNotSupported;
  if (p.mode != G_PLAIN || p.istride != 1 || p.ntaps != 1 || p.taps[0] != 0 || p.C != W5_CA || p.Hs != a.Ho || p.Ws != a.Wo) return hipErrorNotSupported;
  if ((stem || raw3) ? p.scale != nullptr : (p.scale == nullptr || p.q != nullptr)) return hipErrorNotSupported;
  if (a.N != W5_CA || a.Npad != W5_CA) return hipErrorNotSupported;
  bool seen[49];
  for (int t = 0; t < 49; ++t) seen[t] = false;
  for (int t = 0; t < q.ntaps; ++t) {
    const int dy = (int)(signed char)(q.taps[t] & 0xff), dx = (int)(signed char)((q.taps[t] >> 8) & 0xff);
    if (dy < -tr || dy > tr || dx < -tr || dx > tr || seen[(dy + tr) * (2 * tr + 1) + dx + tr]) return hipErrorNotSupported;
    seen[(dy + tr) * (2 * tr + 1) + dx + tr] = true;
  }
  if (g_ctl.dry) return hipSuccess;
  Wg5Args g;
  g.w = a;
  g.tiles_y = (a.Ho + W5_TH - 1) / W5_TH;
  g.tiles_x = (a.Wo + W5_TW - 1) / W5_TW;
  g.ntiles = a.B * g.tiles_y * g.tiles_x;
  static const int cus = [] { hipDeviceProp_t pr; int dev = 0; hipGetDevice(&dev);
                              return (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
  static const int per_cu = lab_int("DMM_WG5_PER_CU", 2);
  // every workgroup ends with 57 (stem: 106) KB of atomics; a tile costs ~1 us: two workgroups per CU unless the picture is small
  const bool three = (head5 && factors) || (raw3 && factors && !p.q);
  int nwg = std::max(1, std::min((stem ? 1 : (three ? W5_F_PER_CU : per_cu)) * cus, g.ntiles / 8));  // (the stem form holds one workgroup per CU)
  g.tiles_per_wg = (g.ntiles + nwg - 1) / nwg;
  nwg = (g.ntiles + g.tiles_per_wg - 1) / g.tiles_per_wg;
  const bool f16t = dtype == DT_F16;
  if (raw3 && factors) {
    if (p.q) return f16t ? launch_wg5_t<f16, 1, 1, 2, 2>(g, nwg, st) : launch_wg5_t<bf16, 1, 1, 2, 2>(g, nwg, st);
    return f16t ? launch_wg5_t<f16, 1, 1, 0, 2>(g, nwg, st) : launch_wg5_t<bf16, 1, 1, 0, 2>(g, nwg, st);
  }
  if (raw3) {
    if (p.q) return f16t ? launch_wg5_t<f16, 1, 1, 2, 1>(g, nwg, st) : launch_wg5_t<bf16, 1, 1, 2, 1>(g, nwg, st);
    return f16t ? launch_wg5_t<f16, 1, 1, 0, 1>(g, nwg, st) : launch_wg5_t<bf16, 1, 1, 0, 1>(g, nwg, st);
  }
  if (head5 && factors) return f16t ? launch_wg5_t<f16, 2, 1, 3>(g, nwg, st) : launch_wg5_t<bf16, 2, 1, 3>(g, nwg, st);
  if (!stem) return f16t ? launch_wg5_t<f16, 2, 1, 1>(g, nwg, st) : launch_wg5_t<bf16, 2, 1, 1>(g, nwg, st);
  if (p.q) return f16t ? launch_wg5_t<f16, 3, 2, 2>(g, nwg, st) : launch_wg5_t<bf16, 3, 2, 2>(g, nwg, st);
  return f16t ? launch_wg5_t<f16, 3, 2, 0>(g, nwg, st) : launch_wg5_t<bf16, 3, 2, 0>(g, nwg, st);
}

// From the factor correlations S (wg5_kernel, PY = 2; layout [chunk][64 output channels c][32 columns], column 32 chunk + k = 16 tap +
// 8 variant + n, variant 0 = S2 (m xhat), 1 = S1 (m)) to
//   * the packed weight gradient of the raw-input segment, dP[(8 tap + n) / 32][c][(8 tap + n) % 32] += gamma[n] S2 + beta[n] S1, and
//   * the BatchNorm-backward reductions of the raw-input channels (fp64): red1[n] += sum_{c,tap} W[c][n][tap] S1, red2[n] likewise with S2
//     (the forward convolution's own master weights: W[c][koff + n][tapw[tap]] of a [64][Kin][9] tensor).
// One workgroup: 64 x 9 x 8 = 4608 (c, tap, n) triples, 18 per thread.
__global__ __launch_bounds__(256) void wg5_rawfin_kernel(const RawFinArgs a) {
  __shared__ double r1[8], r2[8];
  const int tid = threadIdx.x;
  if (tid < 8) { r1[tid] = 0.0; r2[tid] = 0.0; }
  __syncthreads();
  for (int i = tid; i < 64 * 9 * 8; i += 256) {
    const int n = i & 7, tap = (i >> 3) % 9, c = i / 72;
    const int j = 16 * tap + n;                                            // S2 column; S1: + 8
    const float s2 = a.sbuf[((size_t)(j >> 5) * 64 + c) * 32 + (j & 31)];
    const float s1 = a.sbuf[((size_t)((j + 8) >> 5) * 64 + c) * 32 + ((j + 8) & 31)];
    const int jp = 8 * tap + n;
    a.dpack[((size_t)(jp >> 5) * a.Npad + c) * 32 + (jp & 31)] += a.gamma[n] * s2 + a.beta[n] * s1;
    if (n < a.nreal) {
      const double w = (double)a.w[((size_t)c * a.Kin + a.koff + n) * 9 + a.tapw[tap]];
      atomicAdd(&r1[n], w * (double)s1);
      atomicAdd(&r2[n], w * (double)s2);
    }
  }
  __syncthreads();
  if (tid < a.nreal) { a.red1[tid] += r1[tid]; a.red2[tid] += r2[tid]; }   // (replica 0; this launch is the only writer of these channels)
}

hipError_t launch_wg5_rawfin(const RawFinArgs& a, hipStream_t st) {
  if (g_ctl.dry) return hipSuccess;
  hipLaunchKernelGGL(wg5_rawfin_kernel, dim3(1), dim3(256), 0, st, a);
  return hipGetLastError();
}

// From the factor correlations of the 64-channel operand (wg5_kernel, PA = 3; sbuf [2][4][64 c][32] DOUBLES, column 32 chunk + k =
// 4 tap + n, variant 0 = S2 = corr(m x, dY), 1 = S1 = corr(m, dY)) to
//   * the packed weight gradient of the 5x5 convolution, dP[(8 tap + n) / 32][c][(8 tap + n) % 32] += scale[c] S2 + shift[c] S1
//     (relu(bn(x)) = scale (m x) + shift m), and
//   * norm1's BatchNorm-backward reductions (fp64): red1[c] += sum_{tap,n} W[n][c][tap] S1[c][tap][n] = sum dz,
//     red2[c] += (sum W S2 - mean[c] sum W S1) invstd[c] = sum dz xhat - what the reductions-only pass of the 5x5 data gradient
//     computed from dz = m * (sum_{tap,n} W dY).
// Eight workgroups of 8 channels: 32 threads share a channel's 100 columns (a one-workgroup form took 31 us: 50 dependent rounds).
__global__ __launch_bounds__(256) void wg5_fin64_kernel(const Fin64Args a) {
  const int tid = threadIdx.x, c = blockIdx.x * 8 + (tid >> 5), part = tid & 31;
  const double sc = (double)a.scale[c], sh = (double)a.shift[c];
  const double* sb = (const double*)a.sbuf;
  double r1 = 0.0, r2 = 0.0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int j = part + 32 * u;   // = chunk u, column `part`
    if (j >= 100) break;
    const size_t at = ((size_t)u * 64 + c) * 32 + part;
    const double s2 = sb[at], s1 = sb[(size_t)4 * 64 * 32 + at];
    const int n = j & 3, tap = j >> 2, jp = 8 * tap + n;
    a.dpack[((size_t)(jp >> 5) * a.Npad + c) * 32 + (jp & 31)] += (float)(sc * s2 + sh * s1);
    if (n < a.nreal) {
      // the weight the data gradient's STORING pass multiplies with: the master weight rounded to the storage type (pack).  With the
      // fp32 master the two sums were those of a slightly different dz than the one stored (6e-5 of a sum - 3e-3 on conv0's gradient)
      const float wm = a.w[((size_t)n * a.Kin + c) * 25 + a.tapw[tap]];
      const double w = a.dtype == DT_F16 ? (double)(float)(f16)wm : (double)(float)(bf16)wm;
      r1 += w * s1;
      r2 += w * s2;
    }
  }
#pragma unroll
  for (int s = 1; s < 32; s <<= 1) { r1 += __shfl_xor(r1, s); r2 += __shfl_xor(r2, s); }
  if (part == 0) {   // (replica 0; this launch is the only writer of these channels)
    a.red1[c] += r1;
    a.red2[c] += (r2 - (double)a.mean[c] * r1) * (double)a.invstd[c];
  }
}

hipError_t launch_wg5_fin64(const Fin64Args& a, hipStream_t st) {
  if (g_ctl.dry) return hipSuccess;
  hipLaunchKernelGGL(wg5_fin64_kernel, dim3(8), dim3(256), 0, st, a);
  return hipGetLastError();
}

bool wg5_handles(const WgradArgs& a, int dtype) {
  const LaunchCtl keep = g_ctl;
  g_ctl.dry = true;
  const hipError_t e = launch_wg5(a, dtype, nullptr);
  g_ctl = keep;
  return e == hipSuccess;
}

}  // namespace dmm
