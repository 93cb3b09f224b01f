// Stand-alone timing of the slab reduction's access pattern: `nslabs` partial copies of a tensor, `count` floats each,
// summed in a fixed order by blocks of CH column chunks (of 4 floats) x QL slab lanes, up to 16 loads of a lane in flight.
// Every timed launch follows a 1 GB fill that empties the caches (the slabs of the training step were written by other
// kernels long before they are read back).  Usage: reduce_probe [count] [nslabs]
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops tools/reduce_probe.hip -o tools/bin/reduce_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH, int QL>
__global__ __launch_bounds__(256) void reduce_kernel(const float* __restrict__ srcb, float* __restrict__ dst, int64_t count, int64_t stride, int nslabs) {
  static_assert(CH * QL == 256, "one thread per (chunk, slab lane)");
  __shared__ f32x4 red[QL][CH];
  const int lb = blockIdx.x, nb = gridDim.x;
  const int cx = threadIdx.x & (CH - 1), q = threadIdx.x / CH;
  const int64_t n4 = count >> 2;
  for (int64_t c0 = (int64_t)lb * CH; c0 < n4; c0 += (int64_t)nb * CH) {
    const int64_t i = min(c0 + cx, n4 - 1);
    const float* src = srcb + 4 * i;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int k = q;
    for (; k + 15 * QL < nslabs; k += 16 * QL) {
      f32x4 v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const f32x4*>(src + (int64_t)(k + j * QL) * stride);
#pragma unroll
      for (int j = 0; j < 16; j += 4) { s0 += v[j]; s1 += v[j + 1]; s2 += v[j + 2]; s3 += v[j + 3]; }
    }
    for (; k + 3 * QL < nslabs; k += 4 * QL) {
      f32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(src + (int64_t)(k + j * QL) * stride);
      s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
    }
    for (; k < nslabs; k += QL) s0 += *reinterpret_cast<const f32x4*>(src + (int64_t)k * stride);
    red[q][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && c0 + cx < n4) {
      f32x4 t = red[0][cx];
#pragma unroll
      for (int w = 1; w < QL; ++w) t += red[w][cx];
      *reinterpret_cast<f32x4*>(dst + 4 * (c0 + cx)) = t;
    }
    __syncthreads();
  }
}

__global__ void fill_kernel(float* p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

int main(int argc, char** argv) {
  const int64_t count = argc > 1 ? atoll(argv[1]) : 131072;
  const int nslabs = argc > 2 ? atoi(argv[2]) : 32;
  float *src, *dst, *junk;
  const int64_t junk_n = 256ll << 20;
  CK(hipMalloc(&src, (size_t)nslabs * count * 4));
  CK(hipMalloc(&dst, count * 4));
  CK(hipMalloc(&junk, junk_n * 4));
  CK(hipMemset(src, 0, (size_t)nslabs * count * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int cold = 0; cold < 2; ++cold)
    for (int shape = 0; shape < 3; ++shape)
      for (int nb : {256, 512, 1024, 2048}) {
        auto run = [&] {
          if (shape == 0) hipLaunchKernelGGL((reduce_kernel<16, 16>), dim3(nb), dim3(256), 0, 0, src, dst, count, count, nslabs);
          else if (shape == 1) hipLaunchKernelGGL((reduce_kernel<32, 8>), dim3(nb), dim3(256), 0, 0, src, dst, count, count, nslabs);
          else hipLaunchKernelGGL((reduce_kernel<64, 4>), dim3(nb), dim3(256), 0, 0, src, dst, count, count, nslabs);
        };
        double tot = 0;
        const int reps = 6;
        for (int i = 0; i < reps + 1; ++i) {
          if (cold) hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, junk, junk_n, (float)i);
          CK(hipEventRecord(e0, 0));
          run();
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (i) tot += ms;
        }
        const double us = tot / reps * 1e3;
        printf("%s count %lld slabs %d chunks x slab lanes %2d x %2d blocks %4d: %6.1f us, %.2f TB/s\n", cold ? "cold" : "hot ", (long long)count, nslabs,
               shape == 0 ? 16 : shape == 1 ? 32 : 64, shape == 0 ? 16 : shape == 1 ? 8 : 4, nb, us, (double)count * nslabs * 4 / (us * 1e-6) / 1e12);
      }
  return 0;
}
