// Stand-alone timing of lnfwd0_kernel (mineral_amd/csrc/ln_fwd0.hip.h) at the benchmark shape: 32,768 rows, two towers.
// Build variants with -DLNF0_ABL=<bits> (see the header) to see where the time goes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops tools/lnfwd0_probe.hip -o tools/bin/lnfwd0_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
#include "../mineral_amd/csrc/gemm.hip.h"
#include "../mineral_amd/csrc/split_bf16.hip.h"
#include "../mineral_amd/csrc/ln_wgrad0.hip.h"
#include "../mineral_amd/csrc/ln_fwd0.hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  using namespace mrl;
  const int M = argc > 1 ? atoi(argv[1]) : 32768, N = 512, IN = 64, towers = 2;
  LnFwd0Args a;
  memset(&a, 0, sizeof(a));
  std::vector<float> h((size_t)M * IN);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f - 0.5f;
  float *x, *mean, *std_;
  CK(hipMalloc(&x, (size_t)M * IN * 4)); CK(hipMemcpy(x, h.data(), (size_t)M * IN * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&mean, IN * 4)); CK(hipMalloc(&std_, IN * 4));
  CK(hipMemcpy(mean, h.data(), IN * 4, hipMemcpyHostToDevice));
  std::vector<float> one(N, 1.0f);
  CK(hipMemcpy(std_, one.data(), IN * 4, hipMemcpyHostToDevice));
  for (int k = 0; k < towers; ++k) {
    float *W, *b, *xh, *rs;
    CK(hipMalloc(&W, (size_t)N * IN * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&xh, (size_t)M * N * 4)); CK(hipMalloc(&rs, (size_t)M * 4));
    CK(hipMemcpy(W, h.data(), (size_t)N * IN * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, h.data(), N * 4, hipMemcpyHostToDevice));
    a.W[k] = W; a.bias[k] = b; a.xhat[k] = xh; a.rstd[k] = rs;
  }
  a.x = x; a.in = IN; a.mean = mean; a.std = std_; a.clamp_lo = -5.f; a.clamp_hi = 5.f; a.M = M; a.Mpad = M; a.eps = 1e-5f;
  a.rows_per_block = std::max(LNF0_ROWS, ((M + 127) / 128 + LNF0_ROWS - 1) / LNF0_ROWS * LNF0_ROWS);
  const int nblk = (M + a.rows_per_block - 1) / a.rows_per_block;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(lnfwd0_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 30;
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(lnfwd0_kernel, dim3(nblk, towers), dim3(LNF0_THREADS), LNF0_LDS, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(lnfwd0_kernel, dim3(nblk, towers), dim3(LNF0_THREADS), LNF0_LDS, 0, a);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / reps, bytes = (double)towers * M * (N + IN) * 4;
#ifdef LNF0_TRACE
  long long tr[2][64];
  CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_lnf0_trace), sizeof(tr)));
  const char* names[2] = {"row wave 0", "matrix wave 4"};
  for (int r = 0; r < 2; ++r) {
    printf("%s (cycles since its first stamp): start, prologue done, first barrier passed | per step: work A done, work B done, barrier passed\n  ", names[r]);
    for (int k = 0; k < 3 + 3 * 8 + 1; ++k) printf("%lld%s", tr[r][k] ? tr[r][k] - tr[r][0] : -1, (k == 2 || (k > 2 && (k - 2) % 3 == 0)) ? " | " : " ");
    printf("\n");
  }
#endif
  printf("LNF0_ABL=%d rows %d blocks %d x %d (%d rows each): %.1f us per launch, %.2f TB/s\n", LNF0_ABL, M, nblk, towers, a.rows_per_block, us, bytes / us * 1e-6);
  return 0;
}
