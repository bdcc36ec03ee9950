// Thin-output convolution for gfx950: few output channels (N <= 4), many taps - the 5x5 heat-map head
// (dec_out_to_heat_maps.refine1, reference M:233-237: 64 -> num_classes).
//
// As an implicit GEMM this layer wastes the matrix core on padding (N = 3 of a 32-wide tile) and gathers every input
// pixel 25 times.  Here every input pixel is gathered ONCE: a 1x1 GEMM produces, per input pixel, the products with all
// (tap, class) weight columns (25*3 = 75 of 96 MFMA columns), and the convolution sum over taps becomes a shifted
// reduction of those partial products in LDS:
//     logits[y][x][cls] = sum_tap  P[y + dy_tap][x + dx_tap][tap*N + cls],   P[p][tap*N + cls] = sum_c relu(bn(x[p][c])) w[cls][c][tap]
// A workgroup walks down a strip of 128 input columns (124 output columns for a 5x5), one input row per step:
//   gather + BN + ReLU (registers -> LDS, issued one row ahead) -> 12 MFMA per wave with the weight fragments held in
//   registers -> P[96][128] fp32 to LDS -> per output column, add the row's taps into a ring of 2R+2 output rows
//   -> the output row completed by the previous step is written out (fp32 NCHW) and its ring slot cleared.
// 16-bit storage only; fp32 tensors stay on the generic kernels (the parity configuration).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gather.h"

namespace dmm {

constexpr int TH_PX = 128;        // input pixels per row step (4 waves x 32)
constexpr int TH_COLS = 96;       // MFMA columns (tap*N + cls)
constexpr int TH_PP = TH_PX + 4;  // floats per P column (pitch 132 words: b128 writes of 32 columns spread over the banks)
constexpr int TH_RING = 6;        // output rows in flight: 2R+1 receive taps, one is being written out
constexpr int TH_MAXTAPS = 32;

struct ThinArgs {
  const void* src;   // T (f16 or bf16) NHWC
  int ld, B, H, W;
  const float* scale;
  const float* shift;
  const void* wpack;  // T forward pack [chunk = tap*2 + c/32][Npad][32]
  int Npad, N, ntaps, R;
  float* logits;  // fp32 NCHW (B, N, H, W)
  int rows_per_wg, nys, nxs;
  signed char dy[TH_MAXTAPS], dx[TH_MAXTAPS];
};

struct ThinSmem {
  static constexpr int A_BYTES = 2 * TH_PX * ROWB;            // two 32-channel chunks
  static constexpr int P_BYTES = TH_COLS * TH_PP * 4;
  static constexpr int O_BYTES = TH_RING * 4 * TH_PX * 4;     // ring[slot][cls < 4][xl]
  static constexpr int bytes = A_BYTES + P_BYTES + O_BYTES;
};

// KR, KN: tap radius and class count at compile time (full row-major (2KR+1)^2 tap table), so that the reduction is
// straight-line code with immediate LDS offsets.
template <typename T, int KR, int KN>
__global__ __launch_bounds__(NTHREADS, 2) void thin_logits_kernel(const ThinArgs a) {
  typedef typename TT<T>::vec V8;
  const T* const src = (const T*)a.src;
  const T* const wpack = (const T*)a.wpack;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;
  float* Ps = (float*)(smem + ThinSmem::A_BYTES);
  float* Os = (float*)(smem + ThinSmem::A_BYTES + ThinSmem::P_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int xs = bid % a.nxs; bid /= a.nxs;
  const int ys = bid % a.nys;
  const int b = bid / a.nys;
  constexpr int R = KR;
  constexpr int KW5 = 2 * KR + 1;
  const int wout = TH_PX - 2 * R;  // output columns per strip
  const int x0 = xs * wout, y0 = ys * a.rows_per_wg;
  constexpr int ncols = KW5 * KW5 * KN;

  // ---- gather role: slot column j (8 channels), pixels pg + 32 i ----
  const int j = tid & 7, pg = tid >> 3;
  SlotK<8> kk;
  kk.k0 = load_fv<8>(a.scale + 8 * j);
  kk.k1 = load_fv<8>(a.shift + 8 * j);
  kk.k2 = 0.f; kk.k3 = 0.f;
  int gx[4];
  bool gxok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    gx[i] = x0 - R + pg + 32 * i;
    gxok[i] = gx[i] >= 0 && gx[i] < a.W;
  }
  const int aoff = ((j >> 2) * TH_PX) * ROWB + (((j & 3) ^ ((pg >> 2) & 3)) << 4);  // + px * ROWB; (px >> 2) & 3 == (pg >> 2) & 3

  // ---- MFMA role: weight fragments of this lane's column, kept in registers for the whole strip ----
  const int r = lane & 31, h = lane >> 5;
  V8 bfrag[2][2][3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int col = 32 * t + r;
    const int tap = col / KN, cls = col - tap * KN;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bfrag[u][s][t][e] = (T)0;
        if (col < ncols) bfrag[u][s][t] = *(const V8*)(wpack + ((size_t)(tap * 2 + u) * a.Npad + cls) * 32 + (2 * s + h) * 8);
      }
  }

  // ---- reduce role: output column xl, (dy, class) groups [half * GH, ...) - every ring element has ONE owner thread ----
  const int xl = tid & 127, half = tid >> 7;
  constexpr int NG = KW5 * KN, GH = (NG + 1) / 2;

  for (int i = tid; i < TH_RING * 4 * TH_PX; i += NTHREADS) Os[i] = 0.f;

  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  V8 raw[4];
  bool rawok[4];
  auto issue = [&](int iy) {
    const bool rowok = iy >= 0 && iy < a.H;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rawok[i] = rowok && gxok[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) raw[i][e] = (T)0;
      if (rawok[i]) raw[i] = *(const V8*)(src + ((size_t)(b * a.H + iy) * a.W + gx[i]) * a.ld + 8 * j);
    }
  };

  const int nsteps = a.rows_per_wg + 2 * R;
  const int yend = min(a.H, y0 + a.rows_per_wg);
  issue(y0 - R);
  int s_lo = (((y0 - 2 * R - 1) % TH_RING) + TH_RING) % TH_RING;  // ring slot of output row iy - R - 1 (the row written out)
  __syncthreads();  // ring cleared
  for (int it = 0; it <= nsteps; ++it) {
    const int iy = y0 - R + it;  // input row of this step (it == nsteps: drain, only writes the last output row)
    if (it < nsteps) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const V8 v = bn_relu_slot(raw[i], kk);
        V8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (T)0;
        *(V8*)(As + aoff + (pg + 32 * i) * ROWB) = rawok[i] ? v : z;  // zero padding applies AFTER BN+ReLU
      }
    }
    __syncthreads();  // A image complete; previous step's reduction has finished reading P
    if (it < nsteps) {
      if (it + 1 < nsteps) issue(iy + 1);
      f32x16 acc[3];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const V8 av = *(const V8*)(As + (u * TH_PX + 32 * wave + r) * ROWB + (((2 * s + h) ^ ((r >> 2) & 3)) << 4));
#pragma unroll
          for (int t = 0; t < 3; ++t) acc[t] = mma16(av, bfrag[u][s][t], acc[t]);
        }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int col = 32 * t + r;
        if (col < ncols) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v4 = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
            *(f32x4*)(Ps + col * TH_PP + 32 * wave + 8 * g + 4 * h) = v4;
          }
        }
      }
    }
    __syncthreads();  // P complete
    // write out the row completed by the previous step and clear its slot (no tap of this step lands there)
    {
      const int yd = iy - 1 - R;
      for (int idx = tid; idx < KN * TH_PX; idx += NTHREADS) {
        const int cls = idx >> 7, x = idx & 127;
        float* o = Os + (s_lo * 4 + cls) * TH_PX + x;
        const float v = *o;
        *o = 0.f;
        const int xg = x0 + x;
        if (yd >= y0 && yd < yend && x < wout && xg < a.W) a.logits[(((size_t)b * KN + cls) * a.H + yd) * a.W + xg] = v;
      }
    }
    if (it < nsteps && xl < wout) {
      // group g = (dyi, cls): output row iy - (dyi - R) = (iy - R - 1) + (2R + 1 - dyi), i.e. ring slot s_lo + 2R + 1 - dyi
      auto reduce = [&](auto G0) {
        constexpr int g0 = decltype(G0)::value;
#pragma unroll
        for (int g = g0; g < g0 + GH && g < NG; ++g) {
          const int dyi = g / KN, cls = g % KN;
          float sum = 0.f;
#pragma unroll
          for (int dxi = 0; dxi < KW5; ++dxi) sum += Ps[((dyi * KW5 + dxi) * KN + cls) * TH_PP + xl + dxi];
          int slot = s_lo + 2 * R + 1 - dyi;
          if (slot >= TH_RING) slot -= TH_RING;
          Os[(slot * 4 + cls) * TH_PX + xl] += sum;
        }
      };
      if (half == 0) reduce(std::integral_constant<int, 0>());
      else reduce(std::integral_constant<int, GH>());
    }
    if (++s_lo == TH_RING) s_lo = 0;
  }
}

static bool g_no_thin = false;
void thin_set_enabled(bool on) { g_no_thin = !on; }

// Returns hipErrorNotSupported when the layer is not eligible.
hipError_t launch_thin_logits(const ConvArgs& c, int dtype, int epi, hipStream_t st) {
  if (!family_on(!g_no_thin, IMPL_THIN) || dtype == DT_F32 || epi != EPI_LOGITS || c.nseg != 1) return hipErrorNotSupported;
  const Seg& sg = c.seg[0];
  if (sg.mode != G_PLAIN || sg.istride != 1 || sg.Hs != c.Ho || sg.Ws != c.Wo || c.ostride != 1 || c.py != 0 || c.px != 0 ||
      c.Hout != c.Ho || c.Wout != c.Wo)
    return hipErrorNotSupported;
  if (sg.C != 64 || sg.Cpad != 64 || sg.scale == nullptr || sg.q != nullptr) return hipErrorNotSupported;
  if (c.N < 1 || c.N > 4 || sg.ntaps > TH_MAXTAPS || sg.ntaps * c.N > TH_COLS) return hipErrorNotSupported;
  ThinArgs a;
  a.src = sg.src; a.ld = sg.ld; a.B = c.B; a.H = c.Ho; a.W = c.Wo;
  a.scale = sg.scale; a.shift = sg.shift;
  a.wpack = c.wpack; a.Npad = c.Npad; a.N = c.N; a.ntaps = sg.ntaps;
  a.logits = c.logits;
  // the kernel wants the full row-major (2R+1)^2 tap table
  int R = 0;
  for (int t = 0; t < sg.ntaps; ++t) R = std::max(R, abs((int)(signed char)(sg.taps[t] & 0xff)));
  const int kw = 2 * R + 1;
  if (sg.ntaps != kw * kw) return hipErrorNotSupported;
  for (int t = 0; t < sg.ntaps; ++t) {
    const int dy = (int)(signed char)(sg.taps[t] & 0xff), dx = (int)(signed char)((sg.taps[t] >> 8) & 0xff);
    if (dy != t / kw - R || dx != t % kw - R) return hipErrorNotSupported;
    a.dy[t] = (signed char)dy; a.dx[t] = (signed char)dx;
  }
  void (*kern)(const ThinArgs) = nullptr;
  const bool bf = dtype == DT_BF16;
  if (R == 2 && c.N == 3) kern = bf ? thin_logits_kernel<bf16, 2, 3> : thin_logits_kernel<f16, 2, 3>;
  else if (R == 2 && c.N == 1) kern = bf ? thin_logits_kernel<bf16, 2, 1> : thin_logits_kernel<f16, 2, 1>;
  else if (R == 2 && c.N == 2) kern = bf ? thin_logits_kernel<bf16, 2, 2> : thin_logits_kernel<f16, 2, 2>;
  if (kern == nullptr) return hipErrorNotSupported;
  if (g_ctl.dry) return hipSuccess;
  a.R = R;
  const int wout = TH_PX - 2 * R;
  a.nxs = (a.W + wout - 1) / wout;
  // Rows per workgroup: the launch runs in ROUNDS of the chip's 512 slots (two 79 KB workgroups per CU) and a strip of r output rows
  // reads r + 2R input rows - pick the split whose rounds x (r + 2R) is smallest (C2, 4 x 1280 x 1920: 64 rows were 1280 workgroups =
  // 2.5 rounds of 68 rows, the third round half empty; 160 rows are ONE round of 164: 0.57 -> 0.47 ms), the finer split on a tie.
  {
    constexpr int slots = 2 * DESIGN_CUS;
    long best = -1;
    a.rows_per_wg = 8; a.nys = (a.H + 7) / 8;
    for (int nys = 1; nys <= std::max(1, a.H / 8); ++nys) {
      const int rows = (a.H + nys - 1) / nys;
      const long wgs = (long)a.B * ((a.H + rows - 1) / rows) * a.nxs;
      const long cost = ((wgs + slots - 1) / slots) * (rows + 2 * R);
      if (best < 0 || cost <= best) { best = cost; a.rows_per_wg = rows; a.nys = (a.H + rows - 1) / rows; }
    }
  }
  static const void* attr = nullptr;
  if (attr != (const void*)kern) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ThinSmem::bytes);
    if (e != hipSuccess) return e;
    attr = (const void*)kern;
  }
  hipLaunchKernelGGL(kern, dim3(a.B * a.nys * a.nxs), dim3(NTHREADS), ThinSmem::bytes, st, a);
  return hipGetLastError();
}

}  // namespace dmm
