// Persistent 1x1 convolution, forward (gfx950, 16-bit storage): the dense layers' bottleneck convolution conv1 (torchvision
// _DenseLayer: norm1 -> relu -> 1x1 K -> 128; reference call sites M:85-92, M:169-176), the decoder's conv_reduce (M:111-112) and the
// mid-fusion concat_module (M:187-192):  out[m][n] = sum_k relu(bn(x))[m][k] * W[n][k]  + per-channel sum / sum of squares of the
// stored values (the BatchNorm batch statistics of the NEXT norm).
//
// Same tile, K pipeline and arithmetic as igemm.hip's lean path (128 rows x 128 columns per 256-thread workgroup, K in 64-byte
// chunks, two chunks per stage, A gathered through registers with the BN+ReLU prologue, W by LDS-DMA, XOR-swizzled 64-byte LDS
// rows) - what changes is the life of a workgroup.  igemm.hip launches one workgroup per tile: kernel arguments, the prologue
// constants of all K channels (8 KB for K = 992), the first operand loads, the K loop, accumulator staging, stores and the
// statistics atomics run strictly one after the other, overlapped only across the 2 workgroups a CU holds (ablation, round 2:
// 46 % of a block-1 launch is the epilogue; round 3, conv3.hip: every phase of such a kernel adds up).  Here a workgroup stays
// on its CU and walks row tiles of ONE 128-column tile:
//   * prologue constants, weight-DMA offsets and column geometry once per workgroup;
//   * the first K stage of the NEXT tile is requested right behind the last MFMA of the current one, so those loads fly under the
//     epilogue (staging, stores) instead of in front of an idle matrix core;
//   * the BatchNorm sums of all tiles of the walk are accumulated in LDS (fp64, one owner thread per column): one global atomic
//     per column and WORKGROUP instead of per tile (4800 tiles -> 512 workgroups on block 1).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "gather.h"
#include "pointwise.h"

namespace dmm {

constexpr int PG_BN = 128, PG_KS = 2, PG_NT = PG_BN / 32;
constexpr int PG_A_BYTES = PG_KS * BM * ROWB, PG_B_BYTES = PG_KS * PG_BN * ROWB;  // 16 KB each per buffer
constexpr int PG_MAIN = 2 * (PG_A_BYTES + PG_B_BYTES);                             // 64 KB: two operand buffers
constexpr int PG_PITCH = PG_BN + 8;                                                // staging row pitch (elements)
static_assert(BM * PG_PITCH * 2 <= PG_MAIN, "the staged tile aliases the operand buffers");
constexpr int PG_EXTRA = 4 * 2 * PG_BN * 4 + 2 * PG_BN * 8;                        // per-wave partials + per-workgroup fp64 sums

struct PigArgs {
  ConvArgs c;
  int mtiles, ntiles, walkers;  // walkers = workgroups per column tile; workgroup (w, ntile) takes row tiles w, w + walkers, ...
};

template <typename T>
__global__ __launch_bounds__(NTHREADS, 2) void pig_kernel(const PigArgs g) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  typedef typename TT<T>::vec V;
  constexpr int SLOT = 8, BK = 32, BN = PG_BN, KS = PG_KS, NT = PG_NT;
  constexpr int NB = KS * BN / 64;          // 1-KiB LDS-DMA pieces of the weight image per wave and stage
  constexpr int RST = NTHREADS / KS / 4;    // 32 row groups; a thread owns rows rg + 32 i of ONE slot column j of ONE chunk u
  constexpr int NR = BM / RST;              // 4
  const ConvArgs& a = g.c;
  const Seg& sg = a.seg[0];

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wpart = (float*)(smem + PG_MAIN);                       // [wave][2][BN]
  double* wsum = (double*)(smem + PG_MAIN + 4 * 2 * BN * 4);     // [2][BN]: sums of the whole walk
  float* lk = (float*)(smem + PG_MAIN + PG_EXTRA);               // [scale | shift] x C

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ntile = blockIdx.x % g.ntiles, walker = blockIdx.x / g.ntiles;
  const int n0 = ntile * BN;
  if (walker >= g.mtiles) return;

  const int C = sg.C, ld = sg.ld;
  const int total = sg.nchunks;
  const int nstages = ((total + 2 * KS - 1) / (2 * KS)) * 2;  // even: the loop body holds two stages (a dead stage has zero A)
  stage_consts(sg, lk, tid, NTHREADS);
  if (tid < 2 * BN) wsum[tid] = 0.0;

  const int u = tid / (NTHREADS / KS);  // chunk of the stage this thread gathers
  const int j = tid & 3;                // slot column
  const int rg = (tid >> 2) & (RST - 1);

  // W: per-lane source offsets of this wave's LDS-DMA pieces (fixed over K and over the walk)
  const T* wp = (const T*)a.wpack;
  constexpr int PPC = BN / 16;
  const int ub = (wave * NB) / PPC;
  int boff[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int nn = ((wave * NB + q) % PPC) * 16 + (lane >> 2);
    const int sl = (lane & 3) ^ ((nn >> 2) & 3);
    boff[q] = (n0 + nn) * BK + sl * SLOT;
  }
  auto issue_b = [&](int buf, int stage) {  // (a dead chunk re-reads chunk 0: its A slots are zero)
    const int gch = stage * KS + ub;
    const T* bsrc = wp + (size_t)(gch < total ? gch : 0) * a.Npad * BK;
    unsigned char* Bs = smem + buf * (PG_A_BYTES + PG_B_BYTES) + PG_A_BYTES + wave * NB * 1024;
#pragma unroll
    for (int q = 0; q < NB; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc + boff[q]),
                                       (__attribute__((address_space(3))) void*)(Bs + q * 1024), 16, 0, 0);
  };

  struct ARing {
    RawSlot<T> raw[NR];
    int c;
  };
  // Two register sets: the loads of stage s + 2 are requested while stage s is multiplied, and the barriers of the K loop are raw
  // s_barrier instructions behind a COUNTED wait - __syncthreads() is a workgroup-scope fence, on gfx9 an s_waitcnt vmcnt(0), which
  // drained every prefetch twice per stage: the loads of a stage then had a fraction of one MFMA block to land, and the K-deep launches
  // (blocks 3-4, the decoder's conv_reduce with 1.5-2 K channels) ran at one HBM latency per stage.
  ARing RA, RB;
  unsigned roff[NR];   // byte offsets of the cursor tile's rows (the launcher keeps the operand below 4 GiB)
  bool rv[NR];
  auto rows_of = [&](int mtile) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int m = mtile * BM + rg + RST * i;
      rv[i] = m < a.M;
      roff[i] = (unsigned)(rv[i] ? m : 0) * (unsigned)ld * (unsigned)sizeof(T);
    }
  };
  // the issue cursor walks (tile, stage) pairs in the order they are consumed and runs ahead of the tile being multiplied
  int cur_tile = walker, cur_stage = 0;
  const unsigned char* abase = (const unsigned char*)sg.src;
  auto issue_a = [&](ARing& R) {  // branch-free: clamped addresses, dropped by `state` (see igemm.hip); ALWAYS NR loads (counted waits)
    const int gch = cur_stage * KS + u;
    const int c = gch * BK + j * SLOT;
    const bool cv = gch < total && c < C && cur_tile < g.mtiles;
    const int cc = cv ? c : 0;
    R.c = cc;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      R.raw[i].v = *(const V*)(abase + roff[i] + (unsigned)cc * (unsigned)sizeof(T));
      R.raw[i].state = (cv && rv[i]) ? 1 : 3;
    }
    if (++cur_stage == nstages) {
      cur_stage = 0;
      cur_tile += g.walkers;
      rows_of(cur_tile < g.mtiles ? cur_tile : walker);
    }
  };
  auto store_a = [&](const ARing& R, int buf) {
    unsigned char* As = smem + buf * (PG_A_BYTES + PG_B_BYTES);
    const SlotK<SLOT> kk = lds_slot_consts_n<SLOT, 2>(lk, C, R.c);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int row = u * BM + rg + RST * i;
      *(V*)(As + row * ROWB + ((j ^ ((rg >> 2) & 3)) << 4)) = finish_slot<T, 1>(2, R.raw[i], kk);
    }
  };
  // raw barriers: this wave's LDS traffic has returned (lgkmcnt) and all but its N youngest vector-memory operations have completed
  auto bar_all = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto bar_keep = [&]() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NR) : "memory"); };   // the newest A request stays in flight
  auto bar_lds = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };                        // LDS only
  f32x16 acc[NT];
  auto mma = [&](int buf) {
    const unsigned char* As = smem + buf * (PG_A_BYTES + PG_B_BYTES);
    const unsigned char* Bs = As + PG_A_BYTES;
#pragma unroll
    for (int uu = 0; uu < KS; ++uu)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int sw = ((2 * s + h) ^ ((r >> 2) & 3)) << 4;
        const V av = *(const V*)(As + (uu * BM + 32 * wave + r) * ROWB + sw);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const V bv = *(const V*)(Bs + (uu * BN + 32 * t + r) * ROWB + sw);
          acc[t] = mma16(av, bv, acc[t]);
        }
      }
  };

  // epilogue geometry: thread = (slot column cv, row phase rr), rows rr + 16 i
  constexpr int NCV = BN / SLOT, RPP = NTHREADS / NCV, NIT = BM / RPP;
  const int cv = tid % NCV, rr = tid / NCV;
  const int n = n0 + cv * SLOT;
  const bool colvalid = n < a.N;
  T* out = (T*)a.out;

  int mtile = walker;
  rows_of(mtile);
  __syncthreads();  // constants staged
  issue_a(RA);
  issue_a(RB);
  while (true) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    issue_b(0, 0);
    for (int it = 0; it < nstages; it += 2) {
      store_a(RA, 0);
      if (it == 0) bar_all(); else bar_keep();  // the weight DMA of this stage has landed (it == 0: it was the newest request)
      issue_b(1, it + 1);
      // bar_keep's vmcnt(NR) assumes the weight DMA is OLDER in issue order than the NR loads of issue_a: nothing else orders an
      // LDS-DMA builtin against plain loads it does not alias, so pin the order (no instruction crosses a sched_barrier(0))
      __builtin_amdgcn_sched_barrier(0);
      issue_a(RA);                              // stage it + 2 (or the next tile's: the cursor runs on)
      mma(0);
      store_a(RB, 1);
      bar_keep();
      if (it + 2 < nstages) issue_b(0, it + 2);  // (the staging below reuses the image: no DMA may be left in flight)
      __builtin_amdgcn_sched_barrier(0);
      issue_a(RB);
      mma(1);
    }
    const int m0 = mtile * BM;
    const int next = mtile + g.walkers;
    const bool more = next < g.mtiles;   // (workgroup-uniform)
    if (nstages > 0) { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NR) : "memory"); }  // all waves done with the operand images (no DMA left: only the two A requests of the next tile fly on): stage the tile over them
    {
      T* Cs = (T*)smem;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float ps1 = 0.f, ps2 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h;
          const T v = from_f32<T>(acc[t][i]);
          Cs[row * PG_PITCH + 32 * t + r] = v;
          if (m0 + row < a.M) { const float f = to_f32(v); ps1 += f; ps2 = fmaf(f, f, ps2); }  // sums of the values AS STORED
        }
        wpart[(wave * 2 + h) * BN + 32 * t + r] = fold_swap32(ps1, ps2);  // lane half 0: the sum, half 1: the sum of squares
      }
    }
    bar_lds();  // (LDS only: the next tile's two A requests keep flying under the epilogue)
    {
      const T* Cs = (const T*)smem;
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int row = rr + RPP * i;
        if (m0 + row < a.M && colvalid) *(V*)(out + (size_t)(m0 + row) * a.ldo + n) = *(const V*)(Cs + row * PG_PITCH + cv * SLOT);
      }
      if (tid < 2 * BN) {  // four wave partials per column -> fp64; thread tid owns wsum[tid] for the whole walk
        const int col = tid % BN, which = tid / BN;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += (double)wpart[(w * 2 + which) * BN + col];
        wsum[tid] += s;
      }
    }
    if (!more) break;
    mtile = next;
    bar_lds();  // staging read: the next tile's weight DMA may overwrite it
  }
  if (a.stat_sum != nullptr && tid < 2 * BN) {
    const int col = tid % BN, which = tid / BN;
    if (n0 + col < a.N) {
      const size_t rep = (size_t)(blockIdx.x & (STAT_REPS - 1)) * a.stat_stride;
      atomic_add_f64((which ? a.stat_sq : a.stat_sum) + rep + n0 + col, wsum[tid]);
    }
  }
}

static bool g_pig = !lab_flag("DMM_NO_PIG");
void pig_set_enabled(bool on) { g_pig = on; }

// Takes a forward launch (EPI_STORE) of a plain 1x1 convolution behind BN+ReLU in a 16-bit storage type whose padded output width
// is a multiple of 128 and whose rows are the pixels of the output tensor themselves.  Returns hipErrorNotSupported otherwise.
hipError_t launch_pig(const ConvArgs& a, int dtype, int epi, hipStream_t st) {
  if (!family_on(g_pig, IMPL_PIG) || dtype == DT_F32 || epi != EPI_STORE || a.nseg != 1 || a.pool2) return hipErrorNotSupported;
  const Seg& s = a.seg[0];
  if (s.mode != G_PLAIN || s.istride != 1 || s.ntaps != 1 || s.taps[0] != 0 || s.Hs != a.Ho || s.Ws != a.Wo || s.scale == nullptr || s.q != nullptr)
    return hipErrorNotSupported;
  if (s.C % 8 || s.Cpad != s.C || a.Npad % PG_BN || a.out == nullptr) return hipErrorNotSupported;
  if (2.0 * (double)a.M * s.ld >= 4294967296.0) return hipErrorNotSupported;  // 32-bit byte offsets into the operand
  if (a.ostride != 1 || a.Hout != a.Ho || a.Wout != a.Wo || a.py != 0 || a.px != 0) return hipErrorNotSupported;
  const int lds = PG_MAIN + PG_EXTRA + 2 * s.C * 4 + 16;
  if (lds > 160 * 1024) return hipErrorNotSupported;
  const int fit = lds <= 80 * 1024 ? 2 : 1;  // workgroups per CU (K > 1278 channels: the prologue constants push it past 80 KB)
  if (g_ctl.dry) return hipSuccess;
  PigArgs g;
  g.c = a;
  g.mtiles = (a.M + BM - 1) / BM;
  g.ntiles = a.Npad / PG_BN;
  static const int cus = [] { hipDeviceProp_t pr; int dev = 0; hipGetDevice(&dev);
                              return (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
  static const int per_cu = lab_int("DMM_PIG_PER_CU", 2);
  g.walkers = std::max(1, std::min(g.mtiles, (std::min(per_cu, fit) * cus) / g.ntiles));
  const int nwg = g.walkers * g.ntiles;
  auto kern = dtype == DT_F16 ? pig_kernel<f16> : pig_kernel<bf16>;
  static bool attr_done[2] = {false, false};
  if (!attr_done[dtype == DT_F16 ? 0 : 1]) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done[dtype == DT_F16 ? 0 : 1] = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NTHREADS), lds, st, g);
  return hipGetLastError();
}

}  // namespace dmm
