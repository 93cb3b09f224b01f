// Does VALU / transcendental / LDS work of a wave hide under its own MFMAs?  One wave per SIMD runs a loop of
// {1 MFMA 32x32x2 f32 + N independent VALU ops}; cycles per iteration are printed for several N.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap_probe.hip -o tools/bin/mfma_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int KIND>
__global__ __launch_bounds__(256) void loop_kernel(float* out, int iters, unsigned long long* clk) {
  __shared__ float lds[4096];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-6f, b = 1.0f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
  lds[threadIdx.x] = a;
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (KIND == 0) v[j % 16] = v[j % 16] * 1.0001f + 0.5f;                       // v_fma_f32
        else if (KIND == 1) v[j % 16] = __builtin_amdgcn_exp2f(v[j % 16]);           // v_exp_f32
        else v[j % 16] += lds[(threadIdx.x + j * 64 + it) & 4095];                   // ds_read_b32 + add
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}

template <int NV, int KIND>
static void run(const char* what, int iters, int waves_per_simd = 1) {
  float* out;
  unsigned long long* clk;
  if (hipMalloc(&out, 4 * 256 * 256 * 4) != hipSuccess || hipMalloc(&clk, 8) != hipSuccess) return;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((loop_kernel<NV, KIND>), dim3(256 * waves_per_simd), dim3(256), 0, 0, out, iters, clk);
  (void)hipDeviceSynchronize();
  unsigned long long h = 0;
  (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  printf("%d wave(s)/SIMD, 1 MFMA + %2d x %-12s: %6.1f cycles per MFMA per wave -> MFMA pipe busy %.0f%%\n", waves_per_simd, NV, what, (double)h / iters / 4.0, 100.0 * 64.0 * waves_per_simd / ((double)h / iters / 4.0));
  (void)hipFree(out);
  (void)hipFree(clk);
}

int main() {
  const int iters = 2000;
  run<0, 0>("v_fma_f32", iters);
  run<4, 0>("v_fma_f32", iters);
  run<8, 0>("v_fma_f32", iters);
  run<12, 0>("v_fma_f32", iters);
  run<16, 0>("v_fma_f32", iters);
  run<24, 0>("v_fma_f32", iters);
  run<2, 1>("v_exp_f32", iters);
  run<4, 1>("v_exp_f32", iters);
  run<8, 1>("v_exp_f32", iters);
  run<2, 2>("ds_read+add", iters);
  run<4, 2>("ds_read+add", iters);
  run<8, 2>("ds_read+add", iters);
  for (int w = 2; w <= 3; ++w) {
    run<0, 0>("v_fma_f32", iters, w);
    run<8, 0>("v_fma_f32", iters, w);
    run<16, 0>("v_fma_f32", iters, w);
    run<24, 0>("v_fma_f32", iters, w);
    run<4, 1>("v_exp_f32", iters, w);
    run<4, 2>("ds_read+add", iters, w);
  }
  return 0;
}
