// Checks gather.h fold_to_lds against a plain reduction for every (NCV, SLOT) the kernels instantiate.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I dmmfods_amd/csrc tools/probes/fold_probe.hip -o build_var/fold_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "common.h"
#include "gather.h"
using namespace dmm;

template <int NCV, int SLOT>
__global__ void k(const float* in, double* out, int ncolvalid) {
  constexpr int BN = NCV * SLOT;
  __shared__ double red[2 * BN];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2 * BN; i += blockDim.x) red[i] = 0.0;
  __syncthreads();
  float s1[SLOT], s2[SLOT];
  for (int e = 0; e < SLOT; ++e) { s1[e] = in[(tid * 2 + 0) * SLOT + e]; s2[e] = in[(tid * 2 + 1) * SLOT + e]; }
  const int cv = tid % NCV;
  fold_to_lds<NCV, SLOT, BN>(s1, s2, red, cv, cv < ncolvalid, lane);
  __syncthreads();
  for (int i = tid; i < BN; i += blockDim.x) { out[i] = red[fold_slot<NCV, SLOT>(0, i)]; out[BN + i] = red[fold_slot<NCV, SLOT>(1, i)]; }
}

template <int NCV, int SLOT>
int run(int ncolvalid) {
  constexpr int BN = NCV * SLOT, NT = 256;
  std::vector<float> h(NT * 2 * SLOT);
  for (auto& v : h) v = (float)(rand() % 2001 - 1000) / 64.f;
  float* din; double* dout;
  hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 2 * BN * 8);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k<NCV, SLOT>), dim3(1), dim3(NT), 0, 0, din, dout, ncolvalid);
  std::vector<double> o(2 * BN), ref(2 * BN, 0.0);
  hipMemcpy(o.data(), dout, 2 * BN * 8, hipMemcpyDeviceToHost);
  for (int t = 0; t < NT; ++t) {
    const int cv = t % NCV;
    if (cv >= ncolvalid) continue;
    for (int e = 0; e < SLOT; ++e) { ref[cv * SLOT + e] += h[(t * 2 + 0) * SLOT + e]; ref[BN + cv * SLOT + e] += h[(t * 2 + 1) * SLOT + e]; }
  }
  int bad = 0;
  for (int i = 0; i < 2 * BN; ++i) if (fabs(o[i] - ref[i]) > 1e-3) { if (bad < 4) printf("  [%d] got %f want %f\n", i, o[i], ref[i]); ++bad; }
  printf("NCV %2d SLOT %d valid cols %2d: %s (%d bad)\n", NCV, SLOT, ncolvalid, bad ? "FAIL" : "ok", bad);
  hipFree(din); hipFree(dout);
  return bad;
}

int main() {
  int bad = 0;
  bad += run<4, 8>(4); bad += run<8, 8>(8); bad += run<16, 8>(16); bad += run<16, 8>(10);
  bad += run<8, 4>(8); bad += run<16, 4>(16); bad += run<16, 4>(10); bad += run<32, 4>(32); bad += run<32, 4>(20); bad += run<8, 4>(5); bad += run<4, 8>(3);
  return bad ? 1 : 0;
}
