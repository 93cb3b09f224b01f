// Calibration: the six-term split product on v_mfma_f32_32x32x16_bf16 (6 MFMAs per 32x32 tile and 16 k) against the same
// work on v_mfma_f32_16x16x32_bf16 with two terms per instruction (3 MFMAs per 16x16 tile and 16 k), register-only loops on
// pseudo-random operands (the clock the chip holds depends on the data and on the MFMA shape).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o tools/bin/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ inline bf16x8 rnd_frag(unsigned seed) {
  u32x4 v;
  for (int i = 0; i < 4; ++i) {
    seed = seed * 1664525u + 1013904223u;
    // two bf16 values in [-2, 2) with random mantissas: sign | exponent 0x3f / 0x3e | 7 mantissa bits
    const unsigned lo = (seed & 0x80ffu) | 0x3f00u, hi = ((seed >> 16) & 0x80ffu) | 0x3e80u;
    v[i] = lo | (hi << 16);
  }
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256, 2) void loop32(float* out, int iters, unsigned long long* clk) {
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bf16x8 a[2][3], b[2][3];
  for (int i = 0; i < 2; ++i) for (int p = 0; p < 3; ++p) { a[i][p] = rnd_frag(threadIdx.x * 97 + i * 7 + p + blockIdx.x); b[i][p] = rnd_frag(threadIdx.x * 131 + i * 11 + p + 5); }
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][TA[t]], b[j][TB[t]], acc[i][j], 0, 0, 0);
    if ((it & 63) == 63)      // keep the sums bounded: alternate the sign of one operand set
      for (int i = 0; i < 2; ++i) for (int p = 0; p < 3; ++p) { u32x4 v = __builtin_bit_cast(u32x4, a[i][p]); for (int q = 0; q < 4; ++q) v[q] ^= 0x80008000u; a[i][p] = __builtin_bit_cast(bf16x8, v); }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

__global__ __launch_bounds__(256, 2) void loop16(float* out, int iters, unsigned long long* clk) {
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  bf16x8 a[4][3], b[4][2];
  for (int i = 0; i < 4; ++i) { for (int p = 0; p < 3; ++p) a[i][p] = rnd_frag(threadIdx.x * 97 + i * 7 + p + blockIdx.x); for (int p = 0; p < 2; ++p) b[i][p] = rnd_frag(threadIdx.x * 131 + i * 11 + p + 5); }
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  constexpr int TB[3] = {0, 1, 1};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][t], b[j][TB[t]], acc[i][j], 0, 0, 0);
    if ((it & 63) == 63)
      for (int i = 0; i < 4; ++i) for (int p = 0; p < 3; ++p) { u32x4 v = __builtin_bit_cast(u32x4, a[i][p]); for (int q = 0; q < 4; ++q) v[q] ^= 0x80008000u; a[i][p] = __builtin_bit_cast(bf16x8, v); }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <typename K>
static void run(const char* name, K kernel, int blocks_per_cu, int iters) {
  const int blocks = 256 * blocks_per_cu;
  float* out;
  unsigned long long* clk;
  if (hipMalloc(&out, blocks * 256 * 4) != hipSuccess || hipMalloc(&clk, 16) != hipSuccess) { printf("alloc failed\n"); return; }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 3;
  unsigned long long h[2];
  (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  // bf16 flops per wave and iteration: 24 MFMAs of 32x32x16 = 24 * 32768 (the 16x16x32 loop does the same work in 48)
  const double flops = (double)blocks * 4 * iters * 24 * 32768.0;
  printf("%-28s %d wave(s)/SIMD: %8.3f ms  %7.1f bf16 TFLOP/s = %6.1f fp32-equivalent  (wave clock %.0f MHz)\n", name, blocks_per_cu, ms, flops / ms * 1e-9,
         flops / ms * 1e-9 / 6.0, (double)h[0] / (double)h[1] * 100.0);
  (void)hipFree(out);
  (void)hipFree(clk);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  for (int rep = 0; rep < 2; ++rep) {
    run("32x32x16, six terms", loop32, 1, iters);
    run("16x16x32, two terms per MFMA", loop16, 1, iters);
    run("32x32x16, six terms", loop32, 2, iters);
    run("16x16x32, two terms per MFMA", loop16, 2, iters);
  }
  return 0;
}
