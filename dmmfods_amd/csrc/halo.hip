// Halo-tile convolution for gfx950: multi-tap (3x3, 5x5, phase-decomposed) convolutions whose whole weight set fits in LDS.
//
// The generic implicit-GEMM kernel (igemm.hip) re-gathers and re-normalises the input once per tap.  Here a persistent
// workgroup (one per CU, 4 waves) keeps ALL packed weights of the layer in LDS for its lifetime and walks over 8x16-pixel
// output tiles.  For each tile the input halo (tile + tap extent) of every segment is loaded ONCE, the prologue (BN+ReLU or
// the deferred-gradient correction) is applied ONCE per element, and the result is written to an LDS image.  The K loop is
// then only: ds_read_b128 (A fragment at pixel base + precomputed per-slot tap/channel offset), ds_read_b128 (B fragment),
// v_mfma_f32_32x32x16 -- no global loads, no barriers, no address arithmetic beyond one add.  The next tile's halo loads are
// issued before the current tile's MFMAs and land in registers meanwhile (issue-early / write-late).
// Epilogues are those of igemm.hip (store + BN statistics, fused BN/ReLU backward, fp32 NCHW logits); per-channel reductions
// are kept in LDS (fp64) across all tiles of the workgroup and flushed with one set of global atomics at the end.
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "gather.h"

namespace dmm {

constexpr int TH = 8, TW = 16;  // 128 output pixels per tile (BM)
constexpr int HALO_MAX_NS = 16; // halo slots per thread (register-prefetched)

struct HaloSeg {
  int dymin, dxmin, HH, HW, pitch, lds_off, ncs;  // halo extent, LDS row pitch (bytes), image offset, 16-byte slots per pixel
};
struct HaloArgs {
  ConvArgs c;
  HaloSeg hs[2];
  int tiles_y, tiles_x, ntiles;
  int nsl0, nsl_total;        // halo slots of segment 0 / all segments (per tile)
  int total_chunks;
  int w_off, stage_off, rowpix_off, red_off, slot_off, zero_off, k_off, lds_bytes;
};

template <typename T> struct MmaH;
template <> struct MmaH<f16> {
  static __device__ __forceinline__ void run(f32x16& acc, const f16x8& a, const f16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
};
template <> struct MmaH<float> {
  static __device__ __forceinline__ void run(f32x16& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
  }
};

template <typename T, int BN, int EPI, bool EFF, int NS>
__global__ __launch_bounds__(NTHREADS) void halo_kernel(const HaloArgs h) {
  constexpr int SLOT = TT<T>::SLOT;
  constexpr int BK = 4 * SLOT;
  constexpr int E = (int)sizeof(T);
  typedef typename TT<T>::vec V;
  constexpr int NT = BN / 32;
  constexpr int SCOLS = (EPI == EPI_BNBWD && BN > 64) ? 64 : BN;  // staged columns per epilogue pass
  constexpr int SPITCH_T = SCOLS + SLOT, SPITCH_F = SCOLS + 4;
  const ConvArgs& a = h.c;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Wl = smem + h.w_off;
  int* rowpix = (int*)(smem + h.rowpix_off);
  double* red = (double*)(smem + h.red_off);
  int* slot_tab = (int*)(smem + h.slot_off);
  float* lk0 = (float*)(smem + h.k_off);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int nslots = h.total_chunks * 4;

  // ------------------------------------------------------------------ one-time setup
  {  // all packed weights -> LDS, 64-byte rows with the slot swizzle of igemm.hip
    const T* wp = (const T*)a.wpack;
    const int nws = h.total_chunks * a.Npad * 4;
    for (int ws = tid; ws < nws; ws += NTHREADS) {
      const int row = ws >> 2, q = ws & 3;
      *(V*)(Wl + row * 64 + ((q ^ ((row >> 2) & 3)) << 4)) = *(const V*)(wp + (size_t)row * BK + q * SLOT);
    }
  }
  for (int gs = tid; gs < nslots; gs += NTHREADS) {  // per 16-byte K slot: segment, tap and channel -> LDS offset
    int lc = gs >> 2;
    const bool s1 = a.nseg > 1 && lc >= a.seg[0].nchunks;
    if (s1) lc -= a.seg[0].nchunks;
    auto entry = [&](const Seg& sg, const HaloSeg& g, int flag) {
      const int e = lc * BK + (gs & 3) * SLOT;
      const int tap = e / sg.Cpad, c = e - tap * sg.Cpad;
      int v = h.zero_off | (1 << 30);  // absolute: a slot of zeros
      if (lc < sg.nchunks && tap < sg.ntaps && c < sg.C) {
        const int t = sg.taps[tap];
        const int dy = (int)(signed char)(t & 0xff), dx = (int)(signed char)((t >> 8) & 0xff);
        v = (g.lds_off + ((dy - g.dymin) * g.HW + (dx - g.dxmin)) * g.pitch + c * E) | flag;
      }
      return v;
    };
    slot_tab[gs] = s1 ? entry(a.seg[1], h.hs[1], (int)0x80000000) : entry(a.seg[0], h.hs[0], 0);
  }
  if (tid < 4) ((int*)(smem + h.zero_off))[tid] = 0;
  if (tid < 2 * BN) red[tid] = 0.0;
  const int nk0 = stage_consts(a.seg[0], lk0, tid, NTHREADS);
  float* lk1 = lk0 + nk0;
  if (a.nseg > 1) stage_consts(a.seg[1], lk1, tid, NTHREADS);

  // this thread's halo slots (tile independent): segment, halo pixel (hy,hx), channel, LDS destination
  int hmeta[NS], hdst[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    int sid = tid + i * NTHREADS;
    hmeta[i] = -1;
    hdst[i] = 0;
    if (sid < h.nsl_total) {
      const int s = sid >= h.nsl0 ? 1 : 0;
      if (s) sid -= h.nsl0;
      const int ncs = s ? h.hs[1].ncs : h.hs[0].ncs, HWs = s ? h.hs[1].HW : h.hs[0].HW;
      const int pitch = s ? h.hs[1].pitch : h.hs[0].pitch, loff = s ? h.hs[1].lds_off : h.hs[0].lds_off;
      const int hp = sid / ncs, cs = sid - hp * ncs;
      const int hy = hp / HWs, hx = hp - hy * HWs;
      hmeta[i] = (s << 30) | (hy << 20) | (hx << 10) | cs;
      hdst[i] = loff + hp * pitch + cs * 16;
    }
  }
  // this lane's A-fragment pixel base in each segment's halo image
  int pixbase[2];
  {
    const int p = 32 * wave + r, ty = p / TW, tx = p - ty * TW;
    pixbase[0] = ((ty * a.seg[0].istride) * h.hs[0].HW + tx * a.seg[0].istride) * h.hs[0].pitch;
    pixbase[1] = ((ty * a.seg[1].istride) * h.hs[1].HW + tx * a.seg[1].istride) * h.hs[1].pitch;
  }
  const int tiles_per_img = h.tiles_y * h.tiles_x;

  RawSlot<T> raw[NS];
  // the segment of a slot varies per thread: both segments get their own statically-addressed code path (indexing the
  // kernel-argument struct with a per-thread value would push it to scratch)
  auto issue_one = [&](const Seg& sg, const HaloSeg& g, RawSlot<T>& rs, int m, int b, int ty0, int tx0) {
    const int hy = (m >> 20) & 0x3ff, hx = (m >> 10) & 0x3ff, c = (m & 0x3ff) * SLOT;
    const int sy = ty0 * sg.istride + g.dymin + hy, sx = tx0 * sg.istride + g.dxmin + hx;
    if (sy < 0 || sx < 0 || sy >= sg.Hs || sx >= sg.Ws || c >= sg.C) return;
    const size_t pix = (size_t)(b * sg.Hs + sy) * sg.Ws + sx;
    rs.v = *(const V*)((const T*)sg.src + pix * sg.ld + c);
    if (EFF && sg.q != nullptr) rs.v2 = *(const V*)((const T*)sg.src2 + pix * sg.ld2 + c);
    rs.state = 1;
  };
  auto issue_halo = [&](int tile) {
    const int b = tile / tiles_per_img, tr = tile - b * tiles_per_img;
    const int ty0 = (tr / h.tiles_x) * TH, tx0 = (tr % h.tiles_x) * TW;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      RawSlot<T>& rs = raw[i];
#pragma unroll
      for (int e = 0; e < SLOT; ++e) { rs.v[e] = (T)0; if (EFF) rs.v2[e] = (T)0; }
      rs.state = 0;
      const int m = hmeta[i];
      if (m < 0) continue;
      if ((m >> 30) & 1) issue_one(a.seg[1], h.hs[1], rs, m, b, ty0, tx0);
      else issue_one(a.seg[0], h.hs[0], rs, m, b, ty0, tx0);
    }
  };
  auto store_one = [&](const Seg& sg, const float* lk, const RawSlot<T>& rs, int m, int dst) {
    const int c = (m & 0x3ff) * SLOT;
    const int narr = sg.scale ? 2 : (sg.q ? 4 : 0);
    const SlotK<SLOT> kk = lds_slot_consts<SLOT>(lk, sg.C, rs.state == 1 ? narr : 0, c);
    *(V*)(smem + dst) = gather_finish<T>(sg, rs, kk);
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int m = hmeta[i];
      if (m < 0) continue;
      if ((m >> 30) & 1) store_one(a.seg[1], lk1, raw[i], m, hdst[i]);
      else store_one(a.seg[0], lk0, raw[i], m, hdst[i]);
    }
  };

  __syncthreads();
  int tile = blockIdx.x;
  if (tile < h.ntiles) { issue_halo(tile); store_halo(); }
  __syncthreads();

  for (; tile < h.ntiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    if (next < h.ntiles) issue_halo(next);  // lands in registers while this tile computes

    // ---------------------------------------------------------------- K loop: LDS reads + MFMA only
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // One wave per SIMD and no global loads in this loop: latency has to be hidden by the wave itself.  A step is one
    // (chunk, k-half); fragments are prefetched PD steps ahead and their slot offsets 2*PD steps ahead, in register rings
    // with compile-time indices.
    {
      constexpr int PD = NT == 1 ? 8 : 4;
      const int nsteps = 2 * h.total_chunks;
      const int zslot = h.zero_off | (1 << 30);
      auto ld_so = [&](int step) { return step < nsteps ? slot_tab[2 * step + hh] : zslot; };
      auto ld_frag = [&](int step, int so, V& av, V (&bv)[NT]) {
        const int kc = step >> 1, s = step & 1;
        const int base = (so & (1 << 30)) ? 0 : pixbase[(so >> 31) & 1];
        av = *(const V*)(smem + base + (so & 0x3fffffff));
        const int sw = ((2 * s + hh) ^ ((r >> 2) & 3)) << 4;
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = *(const V*)(Wl + (kc * a.Npad + 32 * t + r) * 64 + sw);
      };
      int so_r[PD];
      V a_r[PD], b_r[PD][NT];
      {
        int so0[PD];
#pragma unroll
        for (int u = 0; u < PD; ++u) so0[u] = ld_so(u);
#pragma unroll
        for (int u = 0; u < PD; ++u) so_r[u] = ld_so(PD + u);
#pragma unroll
        for (int u = 0; u < PD; ++u) ld_frag(u < nsteps ? u : 0, so0[u], a_r[u], b_r[u]);
      }
      for (int base = 0; base < nsteps; base += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
          const int step = base + u;
          if (step < nsteps) {
#pragma unroll
            for (int t = 0; t < NT; ++t) MmaH<T>::run(acc[t], a_r[u], b_r[u][t]);
          }
          if (step + PD < nsteps) ld_frag(step + PD, so_r[u], a_r[u], b_r[u]);
          so_r[u] = ld_so(step + 2 * PD);
        }
      }
    }

    // ---------------------------------------------------------------- epilogue
    const int b = tile / tiles_per_img, tr = tile - b * tiles_per_img;
    const int ty0 = (tr / h.tiles_x) * TH, tx0 = (tr % h.tiles_x) * TW;
    if (tid < BM) {
      const int y = ty0 + tid / TW, x = tx0 + tid % TW;
      rowpix[tid] = (y < a.Ho && x < a.Wo) ? (b * a.Hout + y * a.ostride + a.py) * a.Wout + x * a.ostride + a.px : -1;
    }
    for (int cg = 0; cg < BN; cg += SCOLS) {
      __syncthreads();  // staging buffer free (and rowpix visible)
      if (EPI == EPI_STORE) {
        T* Cs = (T*)(smem + h.stage_off);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (32 * t < cg || 32 * t >= cg + SCOLS) continue;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * hh;
            Cs[row * SPITCH_T + 32 * t - cg + r] = from_f32<T>(acc[t][i]);
          }
        }
      } else {
        float* Cs = (float*)(smem + h.stage_off);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (32 * t < cg || 32 * t >= cg + SCOLS) continue;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * hh;
            Cs[row * SPITCH_F + 32 * t - cg + r] = acc[t][i];
          }
        }
      }
      __syncthreads();
      if (EPI == EPI_LOGITS) {
        const float* Cs = (const float*)(smem + h.stage_off);
        const size_t plane = (size_t)a.Hout * a.Wout;
        for (int idx = tid; idx < BM * a.N; idx += NTHREADS) {
          const int n = idx / BM, row = idx - n * BM;
          const int pix = rowpix[row];
          if (pix < 0) continue;
          const int bimg = pix / (int)plane;
          a.logits[((size_t)bimg * a.N + n) * plane + (pix - bimg * (int)plane)] = Cs[row * SPITCH_F + n];
        }
        continue;
      }
      constexpr int NCV = SCOLS / SLOT, RPP = NTHREADS / NCV;
      const int cv = tid % NCV, rr = tid / NCV;
      const int n = cg + cv * SLOT;
      const bool colvalid = n < a.N;
      float s1[SLOT], s2[SLOT];
#pragma unroll
      for (int i = 0; i < SLOT; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
      if (EPI == EPI_STORE) {
        const T* Cs = (const T*)(smem + h.stage_off);
        T* out = (T*)a.out;
        for (int row = rr; row < BM; row += RPP) {
          const int pix = rowpix[row];
          if (pix < 0 || !colvalid) continue;
          const V v = *(const V*)(Cs + row * SPITCH_T + cv * SLOT);
          *(V*)(out + (size_t)pix * a.ldo + n) = v;
          float f[SLOT];
          vec_to_f32<T>(v, f);
#pragma unroll
          for (int i = 0; i < SLOT; ++i) { s1[i] += f[i]; s2[i] = fmaf(f[i], f[i], s2[i]); }
        }
      } else {  // EPI_BNBWD
        const float* Cs = (const float*)(smem + h.stage_off);
        const T* bx = (const T*)a.bx;
        T* g = (T*)a.out;
        float sc[SLOT], sh[SLOT], mu[SLOT], is[SLOT];
        if (colvalid) {
          load_f32s<SLOT>(a.bscale + n, sc); load_f32s<SLOT>(a.bshift + n, sh);
          load_f32s<SLOT>(a.bmean + n, mu); load_f32s<SLOT>(a.binvstd + n, is);
        }
        for (int row = rr; row < BM; row += RPP) {
          const int pix = rowpix[row];
          if (pix < 0 || !colvalid) continue;
          float av[SLOT], xf[SLOT], gf[SLOT];
#pragma unroll
          for (int i = 0; i < SLOT; i += 4) {
            const f32x4 t4 = *(const f32x4*)(Cs + row * SPITCH_F + cv * SLOT + i);
            av[i] = t4[0]; av[i + 1] = t4[1]; av[i + 2] = t4[2]; av[i + 3] = t4[3];
          }
          vec_to_f32<T>(*(const V*)(bx + (size_t)pix * a.ldbx + n), xf);
          if (a.accumulate) vec_to_f32<T>(*(const V*)(g + (size_t)pix * a.ldo + n), gf);
#pragma unroll
          for (int i = 0; i < SLOT; ++i) {
            const float dz = (fmaf(xf[i], sc[i], sh[i]) > 0.f) ? av[i] : 0.f;
            s1[i] += dz;
            s2[i] = fmaf(dz, (xf[i] - mu[i]) * is[i], s2[i]);
            gf[i] = (a.accumulate ? gf[i] : 0.f) + sc[i] * dz;
          }
          *(V*)(g + (size_t)pix * a.ldo + n) = f32_to_vec<T>(gf);
        }
      }
      if (colvalid) {
#pragma unroll
        for (int i = 0; i < SLOT; ++i) {
          atomicAdd(&red[n + i], (double)s1[i]);
          atomicAdd(&red[BN + n + i], (double)s2[i]);
        }
      }
    }
    __syncthreads();                       // all waves done with this tile's halo image and the staging buffer
    if (next < h.ntiles) store_halo();     // prologue + LDS write of the prefetched halo
    __syncthreads();
  }

  // ---- flush the per-channel reductions of all tiles of this workgroup
  __syncthreads();
  if (EPI != EPI_LOGITS && tid < BN && tid < a.N) {
    double* d1 = (EPI == EPI_STORE) ? a.stat_sum : a.red1;
    double* d2 = (EPI == EPI_STORE) ? a.stat_sq : a.red2;
    if (d1 != nullptr) {
      atomic_add_f64(d1 + tid, red[tid]);
      atomic_add_f64(d2 + tid, red[BN + tid]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
static int halo_cus() {
  static int n = 0;
  if (!n) {
    hipDeviceProp_t p;
    int dev = 0;
    hipGetDevice(&dev);
    n = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return n;
}

// Fills `h`; returns false when the layer does not fit the halo kernel (caller falls back to igemm).
static bool halo_plan(const ConvArgs& a, int dtype, int epi, HaloArgs& h) {
  const int E = (int)dtype_size(dtype), SLOT = 16 / E;
  if (a.nseg < 1 || a.nseg > 2 || a.Npad > 128 || a.Npad % 32) return false;
  bool multitap = false;
  int total = 0;
  for (int s = 0; s < a.nseg; ++s) {
    const Seg& sg = a.seg[s];
    if (sg.mode != G_PLAIN || sg.C % SLOT || sg.Cpad != sg.C || sg.ntaps < 1) return false;
    if (sg.ntaps > 1) multitap = true;
    total += sg.nchunks;
  }
  if (!multitap) return false;
  memset((void*)&h, 0, sizeof(h));
  h.c = a;
  h.total_chunks = total;
  int off = 0;
  auto take = [&](int bytes) { off = (off + 15) / 16 * 16; const int o = off; off += bytes; return o; };
  h.w_off = take(total * a.Npad * 64);
  h.nsl_total = 0;
  for (int s = 0; s < 2; ++s) {
    HaloSeg& g = h.hs[s];
    g.HW = 1; g.HH = 1; g.pitch = 16; g.ncs = 1;
    if (s >= a.nseg) continue;
    const Seg& sg = a.seg[s];
    int dymin = 127, dymax = -128, dxmin = 127, dxmax = -128;
    for (int t = 0; t < sg.ntaps; ++t) {
      const int dy = (int)(signed char)(sg.taps[t] & 0xff), dx = (int)(signed char)((sg.taps[t] >> 8) & 0xff);
      dymin = dy < dymin ? dy : dymin; dymax = dy > dymax ? dy : dymax;
      dxmin = dx < dxmin ? dx : dxmin; dxmax = dx > dxmax ? dx : dxmax;
    }
    g.dymin = dymin; g.dxmin = dxmin;
    g.HH = (TH - 1) * sg.istride + (dymax - dymin) + 1;
    g.HW = (TW - 1) * sg.istride + (dxmax - dxmin) + 1;
    if (g.HH > 1000 || g.HW > 1000) return false;
    g.ncs = sg.C / SLOT;
    g.pitch = sg.C * E + 16;  // odd number of 16-byte slots per pixel: conflict-free ds_read_b128 across pixels
    g.lds_off = take(g.HH * g.HW * g.pitch);
    const int nsl = g.HH * g.HW * g.ncs;
    if (s == 0) h.nsl0 = nsl;
    h.nsl_total += nsl;
  }
  if (h.nsl_total > HALO_MAX_NS * NTHREADS) return false;
  const int scols = (epi == EPI_BNBWD && a.Npad > 64) ? 64 : a.Npad;
  h.stage_off = take(epi == EPI_STORE ? BM * (scols + SLOT) * E : BM * (scols + 4) * 4);
  h.rowpix_off = take(BM * 4);
  h.red_off = take(2 * a.Npad * 8);
  h.slot_off = take(total * 4 * 4);
  h.zero_off = take(16);
  int kfl = 0;
  for (int s = 0; s < a.nseg; ++s) kfl += seg_const_floats(a.seg[s]);
  h.k_off = take(kfl * 4 + 16);
  h.lds_bytes = (off + 15) / 16 * 16;
  if (h.lds_bytes > 160 * 1024) return false;
  if ((1 << 30) <= off) return false;
  h.tiles_y = (a.Ho + TH - 1) / TH;
  h.tiles_x = (a.Wo + TW - 1) / TW;
  h.ntiles = a.B * h.tiles_y * h.tiles_x;
  return h.ntiles > 0;
}

template <typename T, int BN, int EPI, bool EFF>
static hipError_t launch_halo_ns(const HaloArgs& h, hipStream_t st) {
  const int ns = (h.nsl_total + NTHREADS - 1) / NTHREADS;
  int grid = halo_cus();
  if (grid > h.ntiles) grid = h.ntiles;
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), h.lds_bytes, st, h);
    return hipGetLastError();
  };
  if (ns <= 4) return go(halo_kernel<T, BN, EPI, EFF, 4>);
  if (ns <= 8) return go(halo_kernel<T, BN, EPI, EFF, 8>);
  if (ns <= 12) return go(halo_kernel<T, BN, EPI, EFF, 12>);
  return go(halo_kernel<T, BN, EPI, EFF, HALO_MAX_NS>);
}

// Variants built: forward (EPI_STORE) for N <= 64, logits for N <= 32, data gradients for N <= 128.
template <typename T>
static hipError_t launch_halo_type(const HaloArgs& h, int epi, hipStream_t st) {
  const int bn = h.c.Npad;
  if (epi == EPI_STORE) {
    if (bn == 64) return launch_halo_ns<T, 64, EPI_STORE, false>(h, st);
    if (bn == 32) return launch_halo_ns<T, 32, EPI_STORE, false>(h, st);
  } else if (epi == EPI_LOGITS) {
    if (bn == 32) return launch_halo_ns<T, 32, EPI_LOGITS, false>(h, st);
  } else {
    if (bn == 128) return launch_halo_ns<T, 128, EPI_BNBWD, true>(h, st);
    if (bn == 64) return launch_halo_ns<T, 64, EPI_BNBWD, true>(h, st);
    if (bn == 32) return launch_halo_ns<T, 32, EPI_BNBWD, true>(h, st);
  }
  return hipErrorNotSupported;
}

#if defined(HALO_F32_PART)
hipError_t launch_halo_f32(const HaloArgs& h, int epi, hipStream_t st) { return launch_halo_type<float>(h, epi, st); }
#else
hipError_t launch_halo_f32(const HaloArgs& h, int epi, hipStream_t st);  // halo32.o (same source, -DHALO_F32_PART)

// Returns hipErrorNotSupported when the layer is not eligible.
hipError_t launch_halo(const ConvArgs& a, int dtype, int epi, hipStream_t st) {
  HaloArgs h;
  static const bool no_halo = lab_flag("DMM_NO_HALO");
  if (!family_on(!no_halo, IMPL_HALO) || dtype == DT_BF16) return hipErrorNotSupported;  // bf16 layers take the generic / thin kernels
  if (a.pool2 || !halo_plan(a, dtype, epi, h)) return hipErrorNotSupported;
  if (epi != EPI_BNBWD && a.seg[0].q != nullptr) return hipErrorNotSupported;
  if (epi == EPI_STORE && a.Npad > 64) return hipErrorNotSupported;
  if (epi == EPI_LOGITS && a.Npad > 32) return hipErrorNotSupported;
  // Measured on MI355X (C2, b4): with one wave per SIMD the tile phases of this kernel do not overlap yet, so it only beats
  // the generic kernel (2-3 workgroups per CU) where the tap count is large: the 5x5 logits conv (7.9 -> 4.3 ms).  The
  // other eligible layers stay on igemm unless DMM_HALO_ALL is set.
  static const bool all = lab_flag("DMM_HALO_ALL");
  if (!all && epi != EPI_LOGITS) return hipErrorNotSupported;
  if (g_ctl.dry) return (epi == EPI_STORE && (h.c.Npad == 64 || h.c.Npad == 32)) || (epi == EPI_LOGITS && h.c.Npad == 32) ||
                        (epi == EPI_BNBWD && (h.c.Npad == 128 || h.c.Npad == 64 || h.c.Npad == 32)) ? hipSuccess : hipErrorNotSupported;
  return dtype == DT_F16 ? launch_halo_type<f16>(h, epi, st) : launch_halo_f32(h, epi, st);
}
#endif

}  // namespace dmm
