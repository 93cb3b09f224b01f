// Calibration: what fp32 MFMA rate does this GPU sustain with nothing but v_mfma_f32_32x32x2_f32 in flight?
// Runs register-only MFMA loops with 1..4 waves per SIMD and prints TFLOP/s and the shader clock seen by the waves.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_probe.hip -o tools/bin/mfma_peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, unsigned long long* clk) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-6f, b = 1.0f + blockIdx.x * 1e-9f;
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {   // fixed trip count
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int NACC>
static void run(int blocks_per_cu, int iters) {
  const int blocks = 256 * blocks_per_cu;
  float* out;
  unsigned long long* clk;
  if (hipMalloc(&out, blocks * 256 * 4) != hipSuccess || hipMalloc(&clk, 16) != hipSuccess) { printf("alloc failed\n"); return; }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double flops = (double)blocks * 4 * iters * 8 * NACC * 4096.0;
  printf("%d accumulators, %d wave(s)/SIMD, %7d iters: %8.3f ms  %7.1f TFLOP/s   cycle counter / 100 MHz clock = %.1f MHz-equivalent, block 0 ran %.1f us\n",
         NACC, blocks_per_cu, iters, ms, flops / ms * 1e-9, (double)h[0] / (double)h[1] * 100.0, h[1] / 100.0);
  (void)hipFree(out);
  (void)hipFree(clk);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  run<4>(1, iters);
  run<4>(2, iters);
  run<4>(3, iters);
  run<4>(4, iters);
  run<8>(1, iters);
  run<8>(2, iters);
  run<4>(2, iters * 8);
  return 0;
}
