// Multi-tap convolutions on an LDS halo tile, gfx950, 16-bit storage types:
//   * the dense layers' 3x3 growth convolution, forward and data gradient (torchvision _DenseLayer.conv2, 128 -> 32 channels;
//     reference call sites M:85-92, M:169-176);
//   * the heat-map head's first convolution `refine0` (reference M:126-127): forward, in the per-output-parity form of plan.cpp
//     (2x2 merged taps over the half-resolution decoder output, 128 channels, plus 3x3 stride-2 taps over the raw input,
//     8 channels; 64 output channels), and its data gradient towards the raw input.
//
// The generic implicit-GEMM kernel (igemm.hip) gathers, bounds-tests and normalises the input once per TAP: 9x (4x) the
// BatchNorm+ReLU arithmetic and address work for 8 MFMAs per tap - 58 vector instructions per MFMA on the growth convolution.
// Here a workgroup owns an 8 x 16 pixel tile of the row grid:
//   1. the input halo (tile + tap extent) of every segment is loaded ONCE (16-byte slots, all loads in flight together,
//      branch-free), the prologue (BN+ReLU, or the deferred BatchNorm-backward correction g + q + r*x for data gradients) is
//      applied ONCE per element and the result is written to an LDS image [halo row][halo pixel][channel]; pixels outside the
//      picture are zero AFTER the prologue, as in conv(relu(bn(x)));
//   2. the K loop walks the packed-weight chunks in groups: a tap only shifts the LDS address of the A fragment, the weights
//      stream through a 3-deep ring filled by LDS-DMA two groups ahead (global_load_lds_dwordx4 from inline asm, swizzle on the
//      source address; counted vmcnt + raw s_barrier, so the DMA stays in flight across the barrier);
//   3. epilogues as in igemm.hip: store + BatchNorm statistics (forward; the statistics are reduced straight from the
//      accumulator layout), or fused BN/ReLU backward (data gradient).
// LDS image of the wide segment: pixel pitch = C*2 + 16 bytes (an odd number of 16-byte slots), row pitch a multiple of 256
// bytes: the 16 lanes of a ds_read_b128 group (pixels x = 0-3, 12-15 of one tile row and 4-11 of the next) hit 16 different slots.
// Two workgroups share a CU (<= 80 KB of LDS each): one fills its halo while the other runs its MFMAs.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gather.h"

namespace dmm {

constexpr int C3_TH = 8, C3_TW = 16;
#ifndef C3_DBG
#define C3_DBG 0  // timing experiments only (experiment builds: -DC3_DBG=n; the shipped library carries no ablation switch): 1 halo
                  // loads hit one line, 2 no MFMA, 4 no weight DMA, 8 no epilogue, 16 no global statistics atomics, 32 no statistics
#endif
constexpr int C3_WRING = 3;  // weight ring slots
#ifndef C3_RAW_EPI_BAR
#define C3_RAW_EPI_BAR 0  // experiment (round 3): LDS-only raw barriers in the forward epilogue so that the next tile's halo loads are not drained: +-0
#endif

struct Conv3Args {
  ConvArgs c;
  int tiles_y, tiles_x, ntiles;
  int dymin0, dxmin0;  // smallest tap offsets of segment 0 (its halo starts there)
  int dymin1, dxmin1;  // ... of the raw-input segment (stride-2 source)
  signed char ph_dymin0[4], ph_dxmin0[4], ph_dymin1[4], ph_dxmin1[4];  // ... per phase of a multi-phase launch (ConvArgs::nphase)
};

// CS = 16-byte channel slots per pixel of the wide segment (C = 8 CS, a multiple of 32; 0: no wide segment); SPAN = its tap
// extent - 1 (2: 3x3, 1: 2x2).  TSPAN / TSTR: a thin segment of 8 channels (ONE slot per pixel) with (TSPAN+1)^2 taps at source
// stride TSTR (TSPAN = -1: none): the head's raw-input taps (3x3, stride 2) or the logits gradient under the 5x5 conv (stride 1).
// NT = 32-column output tiles; GC = chunks per ring slot.
template <typename T, int CS, int SPAN, int TSPAN, int TSTR, int NT, int GC>
struct Conv3Cfg {
  static constexpr bool SEG0 = CS > 0, SEG1 = TSPAN >= 0;
  static constexpr int BN = 32 * NT;
  static constexpr int NTAPS0 = SEG0 ? (SPAN + 1) * (SPAN + 1) : 0, CG = CS / 4;
  static constexpr int NTAPS1 = SEG1 ? (TSPAN + 1) * (TSPAN + 1) : 0, NCH1 = (NTAPS1 + 3) / 4;
  static constexpr int NCH0 = NTAPS0 * CG, NCH = NCH0 + NCH1;
  static constexpr int NGRP = (NCH + GC - 1) / GC;
  static constexpr int HH = C3_TH + SPAN, HW = C3_TW + SPAN;
  static constexpr int PP = CS * 16 + 16;
  static constexpr int RP = (HW * PP + 255) / 256 * 256;
  static constexpr int HALO0 = SEG0 ? HH * RP : 0;
  static constexpr int HH1 = TSTR * (C3_TH - 1) + TSPAN + 1, HW1 = TSTR * (C3_TW - 1) + TSPAN + 1;
  static constexpr int RP1 = HW1 * 16;
  static constexpr int HALO1 = SEG1 ? HH1 * RP1 : 0;
  static constexpr int WSLOT = GC * BN * 64;
  static constexpr int MAIN0 = HALO0 + HALO1 + C3_WRING * WSLOT;
  static constexpr int STAGE = BM * (BN + 8) * 2;  // accumulators staged as T (16-bit) for both epilogues
  static constexpr int MAIN = MAIN0 > STAGE ? MAIN0 : STAGE;
  static constexpr int EXTRA = BM * 4 + 2 * BN * 8 + 4 * 2 * BN * 4;  // rowpix + fp64 reduction scratch + per-wave partials
  static constexpr int bytes = MAIN + EXTRA;
};

// PRO = 0 none, 1 BN+ReLU, 2 effective gradient (16-bit form: q, r only).
template <typename T, int CS, int SPAN, int TSPAN, int TSTR, int NT, int GC, int EPI, int PRO>
__global__ __launch_bounds__(NTHREADS, 2) void conv3_kernel(const Conv3Args g) {
  static_assert(sizeof(T) == 2 && CS % 4 == 0, "16-bit storage, whole 64-byte chunks");
  typedef typename TT<T>::vec V;
  typedef Conv3Cfg<T, CS, SPAN, TSPAN, TSTR, NT, GC> SM;
  constexpr bool SEG0 = SM::SEG0, SEG1 = SM::SEG1;
  constexpr int SLOT = 8, BN = SM::BN, CG = SM::CG, NCH0 = SM::NCH0, NCH = SM::NCH, NGRP = SM::NGRP;
  constexpr int PP = SM::PP, RP = SM::RP, RP1 = SM::RP1, HH = SM::HH, HW = SM::HW;
  constexpr int CSD = SEG0 ? CS : 1;                   // (divisor that stays legal without a wide segment)
  constexpr int NSL = SEG0 ? HH * HW * CS : 0;         // halo slots of the wide segment per tile
  constexpr int NI = SEG0 ? (NSL + NTHREADS - 1) / NTHREADS : 1;  // per thread
  constexpr int PSTEP = NTHREADS / CSD;                // halo pixels between a thread's consecutive slots
  constexpr int NSL1 = SM::HH1 * SM::HW1;
  constexpr int NI1 = SEG1 ? (NSL1 + NTHREADS - 1) / NTHREADS : 0;
  constexpr int PPC = BN / 16;                         // 1-KiB weight pieces per chunk
  constexpr int NPW = (GC * PPC) / 4;                  // ... per wave and ring slot
  static_assert((GC * BN) % 64 == 0 && NTHREADS % CSD == 0, "weight pieces must split evenly over the waves");
  const ConvArgs& a = g.c;
  const Seg& sg = a.seg[0];
  const Seg& sg1 = a.seg[(SEG0 && SEG1) ? 1 : 0];  // the thin segment

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* halo = smem;
  unsigned char* halo1 = smem + SM::HALO0;
  unsigned char* wring = smem + SM::HALO0 + SM::HALO1;
  int* rowpix = (int*)(smem + SM::MAIN);
  double* red = (double*)(smem + SM::MAIN + BM * 4);
  float* wpart = (float*)(smem + SM::MAIN + BM * 4 + 2 * BN * 8);  // [wave][2][BN]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // A workgroup walks tiles lt, lt + gridDim.x, ... of the launch (gridDim.x is a multiple of 8 whenever there is more than one
  // round, so a workgroup stays on "its" XCD's contiguous range of the tile order): the halo loads of the NEXT tile are issued right
  // behind the last MFMA of the current one and fly under its epilogue (round 3 ablation: load wait, K loop and epilogue of a tile
  // simply added up - 0.5 + 0.85 + 0.6 ms on the head's forward phases - overlapped only across the 2-3 workgroups of a CU).
  // Multi-phase launches (forward, the head's 3x3 over the upsampled map): the walk is over (tile, phase) pairs, phase fastest, so
  // the four phases of a tile are neighbours in the XCD's range - four workgroups of one XCD work on them at the same time and the
  // halo comes from HBM once.
  constexpr bool MULTI = EPI == EPI_STORE && SEG0 && SEG1;
  const int nph = (MULTI && a.nphase > 0) ? a.nphase : 1;
  const int nitems = g.ntiles * nph;
  int lt = blockIdx.x;
  int b, y0, x0, ph = 0;
  int dymin0 = g.dymin0, dxmin0 = g.dxmin0, dymin1 = g.dymin1, dxmin1 = g.dxmin1;
  auto decode = [&](int l) {
    int tile = xcd_remap(l, nitems);
    if constexpr (MULTI) {
      if (nph > 1) {
        ph = tile % nph; tile /= nph;
        dymin0 = g.ph_dymin0[ph]; dxmin0 = g.ph_dxmin0[ph]; dymin1 = g.ph_dymin1[ph]; dxmin1 = g.ph_dxmin1[ph];
      }
    }
    const int tx_i = tile % g.tiles_x; tile /= g.tiles_x;
    const int ty_i = tile % g.tiles_y;
    b = tile / g.tiles_y;
    y0 = ty_i * C3_TH; x0 = tx_i * C3_TW;
  };
  decode(lt);

  // ---- the first two weight groups start streaming now ----
  // LDS-DMA issued from inline asm: hipcc does not count it, so it does not drain it (vmcnt(0)) in front of the next ds_read
  // as it does for the builtin (it cannot prove that the LDS ranges differ); the waits are the counted ones in the K loop.
  const T* wp = (const T*)a.wpack;   // (per phase in a multi-phase launch: set at the top of the tile loop)
  int woff[NPW];   // per-lane source offset (elements) inside a group's block (the group's chunks are contiguous: Npad == BN)
#pragma unroll
  for (int q = 0; q < NPW; ++q) {
    const int row = 16 * (wave * NPW + q) + (lane >> 2);
    woff[q] = row * 32 + (((lane & 3) ^ ((lane >> 4) & 3)) << 3);   // swizzle on the source side
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned wdst0 = __builtin_amdgcn_readfirstlane(lds0 + SM::HALO0 + SM::HALO1 + wave * NPW * 1024);
  auto issue_w = [&](int grp) {
    if (C3_DBG & 4) return;
    const unsigned dst = wdst0 + (grp % C3_WRING) * SM::WSLOT;
    const T* src = wp + (size_t)grp * (GC * BN * 32);
#pragma unroll
    for (int q = 0; q < NPW; ++q) {
      // a piece of a chunk past the end of the pack (last, partial group) re-reads the chunk before it: never used
      const int gc = (wave * NPW + q) / PPC;
      const T* s = (grp * GC + gc < NCH) ? src : src - (size_t)(grp * GC + gc - (NCH - 1)) * (BN * 32);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(s + woff[q]), "s"(dst + q * 1024) : "memory");
    }
  };
  // ---- halo: every slot loaded once, all loads in flight together ----
  const int cs = tid % CSD, hp0 = tid / CSD;
  SlotK<SLOT> kk, kk1;
  kk.k0 = 0.f; kk.k1 = 0.f; kk.k2 = 0.f; kk.k3 = 0.f;
  kk1 = kk;
  if (SEG0 && PRO == 1) { kk.k0 = load_fv<SLOT>(sg.scale + cs * SLOT); kk.k1 = load_fv<SLOT>(sg.shift + cs * SLOT); }
  if (SEG0 && PRO == 2) { kk.k0 = load_fv<SLOT>(sg.q + cs * SLOT); kk.k1 = load_fv<SLOT>(sg.r + cs * SLOT); }
  if (SEG1 && PRO == 1) { kk1.k0 = load_fv<SLOT>(sg1.scale); kk1.k1 = load_fv<SLOT>(sg1.shift); }
  V raw[NI], raw2[PRO == 2 ? NI : 1], rawb[SEG1 ? NI1 : 1];
  bool ok[NI], okb[SEG1 ? NI1 : 1];
  const T* src = (const T*)sg.src + cs * SLOT;
  const T* src2 = (const T*)sg.src2 + cs * SLOT;
  auto load_halo = [&]() {  // of tile (b, y0, x0)
    if constexpr (SEG0)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int hp = hp0 + PSTEP * i;
      const int hy = hp / HW, hx = hp - hy * HW;
      const int sy = y0 + dymin0 + hy, sx = x0 + dxmin0 + hx;
      ok[i] = hp < HH * HW && (unsigned)sy < (unsigned)sg.Hs && (unsigned)sx < (unsigned)sg.Ws;
      // branch-free: a slot outside the picture loads a clamped (valid) address and is zeroed when the image is written; a
      // conditional load would make the compiler wait for each load before the next branch
      const int cy = min(max(sy, 0), sg.Hs - 1), cx = min(max(sx, 0), sg.Ws - 1);
      const size_t pix = (C3_DBG & 1) ? 0 : (size_t)(b * sg.Hs + cy) * sg.Ws + cx;
      raw[i] = *(const V*)(src + pix * sg.ld);
      if constexpr (PRO == 2) raw2[i] = *(const V*)(src2 + pix * sg.ld2);
    }
    if constexpr (SEG1) {
#pragma unroll
      for (int i = 0; i < NI1; ++i) {
        const int hp = tid + NTHREADS * i;
        const int hy = hp / SM::HW1, hx = hp - hy * SM::HW1;
        const int sy = TSTR * y0 + dymin1 + hy, sx = TSTR * x0 + dxmin1 + hx;
        okb[i] = hp < NSL1 && (unsigned)sy < (unsigned)sg1.Hs && (unsigned)sx < (unsigned)sg1.Ws;
        const int cy = min(max(sy, 0), sg1.Hs - 1), cx = min(max(sx, 0), sg1.Ws - 1);
        const size_t pix = (C3_DBG & 1) ? 0 : (size_t)(b * sg1.Hs + cy) * sg1.Ws + cx;
        rawb[i] = *(const V*)((const T*)sg1.src + pix * sg1.ld);
      }
    }
  };
  load_halo();
  constexpr int NCV = BN / SLOT;       // slot columns of the output tile
  constexpr int RPP = NTHREADS / NCV;  // rows per pass
  constexpr int NIT = BM / RPP;        // rows per thread
  const int cv = tid % NCV, rr = tid / NCV;
  const int n = cv * SLOT;
  const bool colvalid = n < a.N;
  int ppre[EPI == EPI_BNBWD ? NIT : 1];
  V xpre[EPI == EPI_BNBWD ? NIT : 1];
  V z;
#pragma unroll
  for (int e = 0; e < SLOT; ++e) z[e] = (T)0;
  // this lane's pixel: tile row 2*wave + (r >> 4), column r & 15
  const int ty = 2 * wave + (r >> 4), tx = r & 15;
  int abase = (ty - g.dymin0) * RP + (tx - g.dxmin0) * PP + h * 16;
  const int bsw = (r >> 2) & 3;
  // tap offsets: scalar loads from the kernel arguments, all before the loop - no compiler-counted memory operation may sit
  // between the DMA issue and the counted wait, or hipcc's wait for it drains the DMA as well
  int toffs[SEG0 ? SM::NTAPS0 : 1];
  if constexpr (SEG0)
#pragma unroll
  for (int tap = 0; tap < SM::NTAPS0; ++tap) {
    const int tw = sg.taps[tap];
    toffs[tap] = (int)(signed char)(tw & 0xff) * RP + (int)(signed char)((tw >> 8) & 0xff) * PP;
  }
  // thin segment: one 16-byte slot per tap; k-step (chunk c, half s) of lane half h reads tap j = 4c + 2s + h (j >= taps: zeros)
  constexpr int NT1 = SM::NTAPS1;
  int off1[SEG1 ? 2 * SM::NCH1 : 1];
  if constexpr (SEG1) {
    const int abase1 = (TSTR * ty - g.dymin1) * RP1 + (TSTR * tx - g.dxmin1) * 16;
#pragma unroll
    for (int c = 0; c < SM::NCH1; ++c)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int j0 = 4 * c + 2 * s;  // tap of lane half 0; half 1 reads j0 + 1
        const int t0 = sg1.taps[j0 < NT1 ? j0 : 0], t1 = sg1.taps[j0 + 1 < NT1 ? j0 + 1 : 0];
        const int o0 = (int)(signed char)(t0 & 0xff) * RP1 + (int)(signed char)((t0 >> 8) & 0xff) * 16;
        const int o1 = (int)(signed char)(t1 & 0xff) * RP1 + (int)(signed char)((t1 >> 8) & 0xff) * 16;
        off1[2 * c + s] = (j0 + h < NT1) ? abase1 + (h ? o1 : o0) : -1;
      }
  }

  int opy = a.py, opx = a.px;   // output parity of the item being computed
  while (true) {   // ---- one tile (b, y0, x0); its halo is in the registers ----
  if constexpr (MULTI) {
    if (nph > 1) {   // this item's phase: weights, tap offsets, output parity (scalar loads from the kernel arguments)
      wp = (const T*)a.ph_wpack[ph];
      opy = a.ph_py[ph]; opx = a.ph_px[ph];
      abase = (ty - dymin0) * RP + (tx - dxmin0) * PP + h * 16;
#pragma unroll
      for (int tap = 0; tap < SM::NTAPS0; ++tap) {
        const int tw = a.ph_taps0[ph][tap];
        toffs[tap] = (int)(signed char)(tw & 0xff) * RP + (int)(signed char)((tw >> 8) & 0xff) * PP;
      }
      const int abase1 = (TSTR * ty - dymin1) * RP1 + (TSTR * tx - dxmin1) * 16;
#pragma unroll
      for (int c = 0; c < SM::NCH1; ++c)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int j0 = 4 * c + 2 * s2;
          const int t0 = a.ph_taps1[ph][j0 < NT1 ? j0 : 0], t1 = a.ph_taps1[ph][j0 + 1 < NT1 ? j0 + 1 : 0];
          const int o0 = (int)(signed char)(t0 & 0xff) * RP1 + (int)(signed char)((t0 >> 8) & 0xff) * 16;
          const int o1 = (int)(signed char)(t1 & 0xff) * RP1 + (int)(signed char)((t1 >> 8) & 0xff) * 16;
          off1[2 * c + s2] = (j0 + h < NT1) ? abase1 + (h ? o1 : o0) : -1;
        }
    }
  }
  issue_w(0);      // the first two weight groups start streaming now (the ring is free: the previous tile's staging has been read)
  if (NGRP > 1) issue_w(1);
  if (tid < BM) {
    const int y = y0 + tid / C3_TW, x = x0 + tid % C3_TW;
    rowpix[tid] = (y < a.Ho && x < a.Wo) ? (b * a.Hout + y * a.ostride + opy) * a.Wout + x * a.ostride + opx : -1;
  }
  // (forward: the BatchNorm sums of ALL tiles of the walk gather in `red`, zeroed in front of the first; data gradients: one tile per workgroup)
  if (tid < 2 * BN && lt == (int)blockIdx.x) red[tid] = 0.0;

  // ---- epilogue operands of the fused BN/ReLU backward: x at every output position, issued now, used after the K loop ----
  if constexpr (EPI == EPI_BNBWD) {
    const T* bx = (const T*)a.bx;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {  // (rowpix is not visible yet: recompute this thread's rows)
      const int row = rr + RPP * i;
      const int y = y0 + row / C3_TW, x = x0 + row % C3_TW;
      ppre[i] = (colvalid && y < a.Ho && x < a.Wo) ? (b * a.Hout + y * a.ostride + a.py) * a.Wout + x * a.ostride + a.px : -1;
#pragma unroll
      for (int e = 0; e < SLOT; ++e) xpre[i][e] = (T)0;
      if (ppre[i] >= 0) xpre[i] = *(const V*)(bx + (size_t)ppre[i] * a.ldbx + n);
    }
  }

  // ---- prologue once per element, then the LDS images ----
  if constexpr (SEG0)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int hp = hp0 + PSTEP * i;
    if (hp < HH * HW) {
      const int hy = hp / HW, hx = hp - hy * HW;
      V v = raw[i];
      if constexpr (PRO == 1) v = bn_relu_slot(raw[i], kk);
      if constexpr (PRO == 2) v = eff_grad_slot(raw[i], raw2[i], kk);
      *(V*)(halo + hy * RP + hx * PP + cs * 16) = ok[i] ? v : z;  // zero padding applies AFTER the prologue
    }
  }
  if constexpr (SEG1) {
#pragma unroll
    for (int i = 0; i < NI1; ++i) {
      const int hp = tid + NTHREADS * i;
      if (hp < NSL1) {
        V v = rawb[i];
        if constexpr (PRO == 1) v = bn_relu_slot(rawb[i], kk1);
        *(V*)(halo1 + hp * 16) = okb[i] ? v : z;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the first two weight groups (issued first) and every load after them
  __syncthreads();                                   // images + first two ring slots complete
  if constexpr (EPI == EPI_BNBWD && SEG0 && !SEG1 && CS == 4 && SPAN == 2) {
    // the dense 3x3 convolution's data gradient: hand the effective output gradient of this tile (prologue applied, 16-bit) to the
    // weight-gradient kernel as a compact [pixel][32] tensor; the stores fly under the K loop
    if (a.eff_out != nullptr) {
      T* eo = (T*)a.eff_out;
#pragma unroll
      for (int i = 0; i < BM * CS / NTHREADS; ++i) {
        const int sidx = tid + NTHREADS * i, px = sidx / CS, c4 = sidx % CS;
        const int py_ = px / C3_TW, px_ = px % C3_TW;
        const int y = y0 + py_, x = x0 + px_;
        const V v = *(const V*)(halo + (py_ - g.dymin0) * RP + (px_ - g.dxmin0) * PP + c4 * 16);
        if (y < a.Ho && x < a.Wo) *(V*)(eo + ((size_t)(b * a.Ho + y) * a.Wo + x) * (CS * SLOT) + c4 * SLOT) = v;
      }
    }
  }

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

#pragma unroll
  for (int grp = 0; grp < NGRP; ++grp) {
    if (grp + 2 < NGRP) issue_w(grp + 2);  // ring slot (grp + 2) % 3 was last read during grp - 1: all waves are past its barrier
    const unsigned char* Bp = wring + (grp % C3_WRING) * SM::WSLOT;
    if (!(C3_DBG & 2)) {
#pragma unroll
      for (int gc = 0; gc < GC; ++gc) {
        const int ck = grp * GC + gc;
        if (ck < NCH) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            V av = z;
            if constexpr (SEG0) {
              if (ck < NCH0) av = *(const V*)(halo + abase + toffs[ck / (SEG0 ? CG : 1)] + (ck % (SEG0 ? CG : 1)) * 64 + s * 32);
            }
            if constexpr (SEG1) if (ck >= NCH0) {
              const int o = off1[2 * (ck - NCH0) + s];
              const V ld = *(const V*)(halo1 + (o >= 0 ? o : 0));
              av = o >= 0 ? ld : z;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const V bv = *(const V*)(Bp + (gc * BN + 32 * t + r) * 64 + (((2 * s + h) ^ bsw) << 4));
              acc[t] = mma16(av, bv, acc[t]);
            }
          }
        }
      }
    }
    // The next group's weights (issued one iteration ago) must have landed; this iteration's DMA stays in flight.  lgkmcnt(0): every
    // fragment read of THIS group must have RETURNED before the barrier, because the first thing behind it is the DMA that refills
    // the slot they read (WAR).  The data dependence of the MFMAs does not give that: hipcc sinks a group's last MFMAs - and the
    // wait for their operands - below the barrier, and with the stem's small, L2-resident weights the refill can land (~250 cycles)
    // before a queued ds_read has executed (seen as 0.1 % wrong outputs of the 7x7 stem convolution at 4 x 1280 x 1920, run to run
    // different, never at parity-test sizes).  Wait and barrier are ONE asm statement so that nothing is scheduled between them.
    if (grp + 2 < NGRP) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  // all waves are past the last barrier: the images and the ring are dead.  The next tile's halo is requested NOW, so that the loads
  // fly under this tile's epilogue; rowpix / ppre of this tile stay valid (b, y0, x0 are only read again at the top of the loop).
  // (forward variants only: the data-gradient variants carry their epilogue operands in registers across the K loop, and holding
  // the next halo as well costs them a workgroup per CU - measured 1.87 -> 2.6 ms on the dense 3x3 data gradients)
  constexpr bool PERSIST = EPI == EPI_STORE;
  const int lnext = lt + gridDim.x;
  const bool more = PERSIST && lnext < nitems;   // (workgroup-uniform)
  if (more) { decode(lnext); load_halo(); }
  if (!((C3_DBG & 8) && acc[0][0] != 123.f)) {
  // reuse the images for staging
  T* Cs = (T*)smem;
  constexpr int CPITCH = BN + 8;
  float ps1[NT], ps2[NT];  // forward: this lane's column sums over its 16 rows, of the values as stored (rounded to T)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    ps1[t] = 0.f; ps2[t] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h;
      const T v = from_f32<T>(acc[t][i]);
      Cs[row * CPITCH + 32 * t + r] = v;
      if (EPI == EPI_STORE && rowpix[row] >= 0) { const float f = to_f32(v); ps1[t] += f; ps2[t] = fmaf(f, f, ps2[t]); }
    }
  }
  if constexpr (EPI == EPI_STORE) {
    // BatchNorm statistics straight from the accumulator layout: lane (r, h) holds column 32 t + r; fold the two lane halves,
    // one LDS word per column and wave, fp64 from there on (per-lane partials cover 16 rows: fp32 is exact enough, see igemm.hip)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      wpart[(wave * 2 + h) * BN + 32 * t + r] = fold_swap32(ps1[t], ps2[t]);  // lane half 0: the sum, half 1: the sum of squares
    }
  }
  // (forward variants: a raw barrier behind an LDS-only wait - __syncthreads() is a fence, s_waitcnt vmcnt(0) on gfx9, and drained the
  // next tile's halo loads here, a few hundred cycles after they had been requested)
  if constexpr (PERSIST && C3_RAW_EPI_BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else __syncthreads();

  if constexpr (EPI == EPI_STORE) {
    T* out = (T*)a.out;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int row = rr + RPP * i;
      const int pix = rowpix[row];
      if (pix < 0 || !colvalid) continue;
      *(V*)(out + (size_t)pix * a.ldo + n) = *(const V*)(Cs + row * CPITCH + cv * SLOT);
    }
    // (round 5: gathered over the walk in LDS - thread tid owns red[tid] -, one round of global atomics behind the loop)
    if (!(a.stat_sum == nullptr || (C3_DBG & 32)))
    if (!(C3_DBG & 16) && tid < 2 * BN) {
      const int col = tid % BN, which = tid / BN;
      if (col < a.N) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += (double)wpart[(w * 2 + which) * BN + col];
        red[tid] += s;
      }
    }
  } else {  // EPI_BNBWD: acc = d(relu(bn(x))); mask, reduce, scatter s*dz (see igemm.hip)
    float s1[SLOT], s2[SLOT];
#pragma unroll
    for (int i = 0; i < SLOT; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
    T* gout = (T*)a.out;
    float sc[SLOT], sh[SLOT], mu[SLOT], is[SLOT];
    if (colvalid) {
      load_f32s<SLOT>(a.bscale + n, sc); load_f32s<SLOT>(a.bshift + n, sh);
      load_f32s<SLOT>(a.bmean + n, mu); load_f32s<SLOT>(a.binvstd + n, is);
    }
    // second pass of a two-pass BatchNorm backward (thin-only variants: the logits gradient under the 5x5 head convolution): the
    // correction constants are final, store s*dz + q + r*x - what apply_corr made of s*dz in a pass of its own (read g, read x, write g)
    constexpr bool TWO_PASS = !SEG0 && SEG1;
    const bool fin = TWO_PASS && a.eq != nullptr;   // (workgroup-uniform)
    float qv[TWO_PASS ? SLOT : 1], rv[TWO_PASS ? SLOT : 1];
    if constexpr (TWO_PASS) {
#pragma unroll
      for (int e = 0; e < SLOT; ++e) { qv[e] = 0.f; rv[e] = 0.f; }
      if (fin && colvalid) { load_f32s<SLOT>(a.eq + n, qv); load_f32s<SLOT>(a.er + n, rv); }
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      if (ppre[i] < 0) continue;
      const int row = rr + RPP * i;
      float av[SLOT], xf[SLOT], gf[SLOT];
      vec_to_f32<T>(*(const V*)(Cs + row * CPITCH + cv * SLOT), av);
      vec_to_f32<T>(xpre[i], xf);
      if (a.accumulate && gout != nullptr) vec_to_f32<T>(*(const V*)(gout + (size_t)ppre[i] * a.ldo + n), gf);
#pragma unroll
      for (int e = 0; e < SLOT; ++e) {
        const float dz = (fmaf(xf[e], sc[e], sh[e]) > 0.f) ? av[e] : 0.f;
        s1[e] += dz;
        s2[e] = fmaf(dz, (xf[e] - mu[e]) * is[e], s2[e]);
        gf[e] = ((a.accumulate && gout != nullptr) ? gf[e] : 0.f) + sc[e] * dz;
        if constexpr (TWO_PASS) gf[e] += fmaf(rv[e], xf[e], qv[e]);   // (zeros unless `fin`)
      }
      if (gout != nullptr) *(V*)(gout + (size_t)ppre[i] * a.ldo + n) = f32_to_vec<T>(gf);
    }
    if (a.red1 != nullptr) {  // (workgroup-uniform; null in the second pass of a two-pass BatchNorm backward)
      // per-channel reductions: lanes -> LDS (fp64) -> one fp64 atomic per channel and workgroup (see igemm.hip)
      fold_to_lds<NCV, SLOT, BN>(s1, s2, red, cv, colvalid, lane);
      __syncthreads();
      if (!(C3_DBG & (16 | 32)) && tid < BN && tid < a.N) {
        const size_t rep = (size_t)(blockIdx.x & (STAT_REPS - 1)) * a.stat_stride;
        atomic_add_f64(a.red1 + rep + tid, red[fold_slot<NCV, SLOT>(0, tid)]);
        atomic_add_f64(a.red2 + rep + tid, red[fold_slot<NCV, SLOT>(1, tid)]);
      }
    }
  }
  }  // (epilogue)
  if (!more) break;
  lt = lnext;
  // staging / reduction scratch read: the next tile may overwrite the images, the ring, rowpix and red
  if constexpr (PERSIST && C3_RAW_EPI_BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else __syncthreads();
  }  // (tile loop)
  if constexpr (EPI == EPI_STORE) {
    if (!(a.stat_sum == nullptr || (C3_DBG & (16 | 32))) && tid < 2 * BN) {   // (red[tid] was written by this very thread)
      const int col = tid % BN, which = tid / BN;
      if (col < a.N) {
        const size_t rep = (size_t)(blockIdx.x & (STAT_REPS - 1)) * a.stat_stride;
        atomic_add_f64((which ? a.stat_sq : a.stat_sum) + rep + col, red[tid]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
static bool g_conv3 = !lab_flag("DMM_NO_CONV3");
void conv3_set_enabled(bool on) { g_conv3 = on; }


template <typename T, int CS, int SPAN, int TSPAN, int TSTR, int NT, int GC, int EPI, int PRO>
static hipError_t launch_c3(const Conv3Args& g, hipStream_t st) {
  typedef Conv3Cfg<T, CS, SPAN, TSPAN, TSTR, NT, GC> SM;
  static_assert(SM::bytes <= 80 * 1024, "two workgroups per CU");
  if (g_ctl.dry) return hipSuccess;
  auto kern = conv3_kernel<T, CS, SPAN, TSPAN, TSTR, NT, GC, EPI, PRO>;
  static bool attr_done = false;
  if (SM::bytes > 48 * 1024 && !attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SM::bytes);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  // persistent: the workgroups a CU holds at a time (LDS, <= 4) x CUs x 2 rounds' worth of slots, in whole groups of 8 (one per XCD);
  // a launch with fewer tiles than that runs one tile per workgroup, as before
  static const int cus = [] { hipDeviceProp_t pr; int dev = 0; hipGetDevice(&dev);
                              return (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
  static const int rounds = lab_int("DMM_C3_ROUNDS", 1);
  const int per_cu = std::min(4, (160 * 1024) / SM::bytes);
  int nwg = g.ntiles * (g.c.nphase > 0 ? g.c.nphase : 1);
  if (EPI == EPI_STORE && rounds > 0 && nwg > per_cu * cus * rounds) nwg = per_cu * cus * rounds / 8 * 8;
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NTHREADS), SM::bytes, st, g);
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_c3_type(const Conv3Args& g, int epi, int pro, int cs, int span, int tspan, int tstr, int nt, hipStream_t st) {
  if (epi == EPI_STORE && pro == 1) {
    if (tspan < 0 && span == 2 && nt == 1 && cs == 16) return launch_c3<T, 16, 2, -1, 1, 1, 4, EPI_STORE, 1>(g, st);  // dense conv2
    if (tspan < 0 && span == 2 && nt == 1 && cs == 8) return launch_c3<T, 8, 2, -1, 1, 1, 2, EPI_STORE, 1>(g, st);
    if (tspan == 2 && tstr == 2 && span == 1 && nt == 2 && cs == 16) return launch_c3<T, 16, 1, 2, 2, 2, 2, EPI_STORE, 1>(g, st);  // refine0 phase
  }
  if (epi == EPI_BNBWD && tspan < 0 && span == 2) {
    if (cs == 4 && nt == 4 && pro == 2) return launch_c3<T, 4, 2, -1, 1, 4, 1, EPI_BNBWD, 2>(g, st);              // dense conv2 dgrad
    if (cs == 4 && nt == 4 && pro == 0) return launch_c3<T, 4, 2, -1, 1, 4, 1, EPI_BNBWD, 0>(g, st);
    if (cs == 4 && nt == 2 && pro == 2) return launch_c3<T, 4, 2, -1, 1, 2, 1, EPI_BNBWD, 2>(g, st);
    if (cs == 4 && nt == 2 && pro == 0) return launch_c3<T, 4, 2, -1, 1, 2, 1, EPI_BNBWD, 0>(g, st);
    if (cs == 8 && nt == 1 && pro == 0) return launch_c3<T, 8, 2, -1, 1, 1, 2, EPI_BNBWD, 0>(g, st);              // refine0 -> raw input
    if (cs == 8 && nt == 1 && pro == 2) return launch_c3<T, 8, 2, -1, 1, 1, 2, EPI_BNBWD, 2>(g, st);
  }
  if (epi == EPI_STORE && cs == 0 && tspan == 6 && tstr == 2 && pro == 0 && nt == 2) return launch_c3<T, 0, 0, 6, 2, 2, 2, EPI_STORE, 0>(g, st);  // stem conv0 (7x7 stride 2)
  if (epi == EPI_BNBWD && cs == 0 && tspan == 4 && tstr == 1 && pro == 0) {                                          // refine1 dgrad (5x5)
    if (nt == 2) return launch_c3<T, 0, 0, 4, 1, 2, 2, EPI_BNBWD, 0>(g, st);
    if (nt == 1) return launch_c3<T, 0, 0, 4, 1, 1, 2, EPI_BNBWD, 0>(g, st);
  }
  return hipErrorNotSupported;
}

static bool tap_box(const Seg& sg, int& dymin, int& dxmin, int& span) {
  int dymax = -128, dxmax = -128;
  dymin = 127; dxmin = 127;
  for (int t = 0; t < sg.ntaps; ++t) {
    const int dy = (int)(signed char)(sg.taps[t] & 0xff), dx = (int)(signed char)((sg.taps[t] >> 8) & 0xff);
    dymin = dy < dymin ? dy : dymin; dymax = dy > dymax ? dy : dymax;
    dxmin = dx < dxmin ? dx : dxmin; dxmax = dx > dxmax ? dx : dxmax;
  }
  span = dymax - dymin;
  if (span < 0 || span > 6 || dxmax - dxmin != span || sg.ntaps != (span + 1) * (span + 1)) return false;
  bool seen[49] = {false};
  for (int t = 0; t < sg.ntaps; ++t) {  // every offset of the box exactly once, in any order
    const int dy = (int)(signed char)(sg.taps[t] & 0xff) - dymin, dx = (int)(signed char)((sg.taps[t] >> 8) & 0xff) - dxmin;
    if (seen[dy * (span + 1) + dx]) return false;
    seen[dy * (span + 1) + dx] = true;
  }
  return true;
}

// Returns hipErrorNotSupported when the layer is not one of the shapes built (16-bit storage, unit-stride multi-tap segment of
// C % 32 == 0 channels on the row grid, optionally the 8-channel stride-2 raw-input segment of the head).
hipError_t launch_conv3(const ConvArgs& a, int dtype, int epi, hipStream_t st) {
  if (!family_on(g_conv3, IMPL_CONV3) || dtype == DT_F32 || a.nseg < 1 || a.nseg > 2 || a.pool2 || (epi != EPI_STORE && epi != EPI_BNBWD)) return hipErrorNotSupported;
  const Seg& sg = a.seg[0];
  if (a.Npad % 32 || a.Npad > 128) return hipErrorNotSupported;
  Conv3Args g;
  g.c = a;
  g.dymin0 = g.dxmin0 = g.dymin1 = g.dxmin1 = 0;
  int cs = 0, span = 0, tspan = -1, tstr = 1;
  auto thin_ok = [&](const Seg& s1, int stride) {
    return s1.mode == G_PLAIN && s1.istride == stride && s1.C == 8 && s1.Cpad == 8 && s1.Hs == stride * a.Ho && s1.Ws == stride * a.Wo &&
           s1.q == nullptr && tap_box(s1, g.dymin1, g.dxmin1, tspan);
  };
  if (a.nseg == 1 && sg.C == 8 && sg.istride == 2) {  // thin segment only, stride 2: the stem's 7x7 convolution over the raw input
    tstr = 2;
    if (epi != EPI_STORE || !thin_ok(sg, 2) || tspan != 6 || sg.scale != nullptr) return hipErrorNotSupported;
  } else if (a.nseg == 1 && sg.C == 8) {   // thin segment only: the logits gradient under the 5x5 head convolution
    if (!thin_ok(sg, 1) || tspan != 4 || sg.scale != nullptr) return hipErrorNotSupported;
  } else {
    if (sg.mode != G_PLAIN || sg.istride != 1 || sg.C % 32 || sg.Cpad != sg.C || sg.Hs != a.Ho || sg.Ws != a.Wo) return hipErrorNotSupported;
    if (!tap_box(sg, g.dymin0, g.dxmin0, span) || span < 1 || span > 2) return hipErrorNotSupported;
    cs = sg.C / 8;
    if (a.nseg == 2) {
      tstr = 2;
      if (!thin_ok(a.seg[1], 2) || tspan != 2 || (a.seg[1].scale != nullptr) != (sg.scale != nullptr)) return hipErrorNotSupported;
    }
  }
  if (a.nphase < 0 || a.nphase > 4) return hipErrorNotSupported;
  if (a.nphase > 0) {  // several output-parity phases in one launch: the forward of the two-segment head convolution only
    if (epi != EPI_STORE || a.nseg != 2 || sg.ntaps != 4 || a.seg[1].ntaps != 9) return hipErrorNotSupported;
    for (int ph = 0; ph < a.nphase; ++ph) {
      Seg t0 = sg, t1 = a.seg[1];
      for (int t = 0; t < 4; ++t) t0.taps[t] = a.ph_taps0[ph][t];
      for (int t = 0; t < 9; ++t) t1.taps[t] = a.ph_taps1[ph][t];
      int y0, x0, y1, x1, sp0, sp1;
      if (!tap_box(t0, y0, x0, sp0) || sp0 != span || !tap_box(t1, y1, x1, sp1) || sp1 != tspan || a.ph_wpack[ph] == nullptr ||
          a.ph_py[ph] < 0 || a.ph_py[ph] >= a.ostride || a.ph_px[ph] < 0 || a.ph_px[ph] >= a.ostride)
        return hipErrorNotSupported;
      g.ph_dymin0[ph] = (signed char)y0; g.ph_dxmin0[ph] = (signed char)x0; g.ph_dymin1[ph] = (signed char)y1; g.ph_dxmin1[ph] = (signed char)x1;
    }
  }
  if (a.nseg == 1 && (a.ostride != 1 || a.Hout != a.Ho || a.Wout != a.Wo)) return hipErrorNotSupported;
  if (cs == 0 && tspan == 6 && a.Npad != 64) return hipErrorNotSupported;
  const int pro = sg.scale ? 1 : (sg.q ? 2 : 0);
  if (epi == EPI_BNBWD && a.accumulate && a.out == nullptr) return hipErrorNotSupported;
  // the final-gradient epilogue (eq / er) exists in ONE variant: the thin-only 5x5 data gradient, not accumulating, with an output
  if ((a.eq != nullptr) != (a.er != nullptr)) return hipErrorNotSupported;
  if (a.eq != nullptr && !(epi == EPI_BNBWD && cs == 0 && tspan == 4 && tstr == 1 && pro == 0 && a.out != nullptr && !a.accumulate &&
                           a.red1 == nullptr && a.red2 == nullptr))
    return hipErrorNotSupported;
  g.tiles_y = (a.Ho + C3_TH - 1) / C3_TH;
  g.tiles_x = (a.Wo + C3_TW - 1) / C3_TW;
  g.ntiles = a.B * g.tiles_y * g.tiles_x;
  const int nt = a.Npad / 32;
  static const bool trace = lab_flag("DMM_C3_TRACE");
  if (trace && !g_ctl.dry) fprintf(stderr, "conv3: epi %d pro %d cs %d span %d tspan %d tstr %d nt %d M %d\n", epi, pro, cs, span, tspan, tstr, nt, a.M);
  return dtype == DT_F16 ? launch_c3_type<f16>(g, epi, pro, cs, span, tspan, tstr, nt, st)
                         : launch_c3_type<bf16>(g, epi, pro, cs, span, tspan, tstr, nt, st);
}

// Does launch_conv3 take this launch?  (the plan labels its launches by the kernel family that runs them)
bool conv3_handles(const ConvArgs& a, int dtype, int epi) {
  const LaunchCtl keep = g_ctl;
  g_ctl.dry = true;
  const hipError_t e = launch_conv3(a, dtype, epi, nullptr);
  g_ctl = keep;
  return e == hipSuccess;
}

}  // namespace dmm
