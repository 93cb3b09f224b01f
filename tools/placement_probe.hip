// Where does the workgroup dispatcher put the blocks of a grid that is smaller than the chip's capacity?
// Launches a bounded busy kernel with a given static LDS footprint and records, per block, the XCC / SE / CU it ran
// on and its start and end time; prints how many blocks each CU received and how many ran at the same time.
//   hipcc --offload-arch=gfx950 -O2 tools/placement_probe.hip -o gpurun_out/placement_probe && ./placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>

struct Rec { unsigned hw_id, xcc_id; unsigned long long t0, t1; };

template <int LDS_BYTES>
__global__ __launch_bounds__(256) void busy_kernel(Rec* out, int iters, float* sink) {
  __shared__ float pad[LDS_BYTES / 4];
  const unsigned long long t0 = wall_clock64();
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  pad[threadIdx.x] = a;
  __syncthreads();
  for (int i = 0; i < iters; ++i) a = a * b + pad[(threadIdx.x + i) & 255];   // fixed trip count: always exits
  if (a == 123.456f) sink[0] = a;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    Rec r;
    r.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_ID, all 32 bits
    r.xcc_id = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // XCC_ID[3:0]
    r.t0 = t0;
    r.t1 = wall_clock64();
    out[(blockIdx.x + gridDim.x * blockIdx.z) * 4 + (threadIdx.x >> 6)] = r;
  }
}

template <int LDS_BYTES>
static void probe(int blocks, int towers, int iters, unsigned dyn) {
  Rec* d;
  float* sink;
  const int n = blocks * towers * 4;   // one record per wave
  hipMalloc(&d, n * sizeof(Rec));
  hipMalloc(&sink, 4);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(busy_kernel<LDS_BYTES>, dim3(blocks, 1, towers), dim3(256), dyn, 0, d, iters, sink);
    hipDeviceSynchronize();
  }
  std::vector<Rec> h(n);
  hipMemcpy(h.data(), d, n * sizeof(Rec), hipMemcpyDeviceToHost);
  std::map<unsigned, int> per_cu;
  unsigned long long tmin = ~0ull, tmax = 0;
  for (auto& r : h) {
    unsigned cu = (r.hw_id >> 8) & 0xf, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7;
    per_cu[(r.xcc_id << 16) | (se << 8) | (sh << 4) | cu]++;
    tmin = std::min(tmin, r.t0);
    tmax = std::max(tmax, r.t1);
  }
  std::map<unsigned, int> per_simd;
  for (auto& r : h) {
    unsigned cu = (r.hw_id >> 8) & 0xf, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7, simd = (r.hw_id >> 4) & 3;
    per_simd[(r.xcc_id << 16) | (se << 8) | (sh << 4) | cu | (simd << 24)]++;
  }
  std::map<int, int> shist;
  for (auto& kv : per_simd) shist[kv.second]++;
  printf("waves-per-SIMD histogram:");
  for (auto& kv : shist) printf("  %d waves x %d SIMDs", kv.first, kv.second);
  printf("\n   first 6 blocks (simd.slot of the 4 waves):");
  for (int i = 0; i < 6; ++i) {
    printf("  [");
    for (int w = 0; w < 4; ++w) printf(" %u.%u", (h[i * 4 + w].hw_id >> 4) & 3, h[i * 4 + w].hw_id & 0xf);
    printf(" ]");
  }
  printf("\n");
  std::map<int, int> hist;
  for (auto& kv : per_cu) hist[kv.second]++;
  double mean_dur = 0;
  for (auto& r : h) mean_dur += (double)(r.t1 - r.t0);
  mean_dur /= n;
  printf("static LDS %6d B + dynamic %6u B, %5d waves: %zu distinct CUs; waves-per-CU histogram:", LDS_BYTES, dyn, n, per_cu.size());
  for (auto& kv : hist) printf("  %d waves x %d CUs", kv.first, kv.second);
  printf(";  span %.1f us, mean block %.1f us (100 MHz ticks)\n", (tmax - tmin) / 100.0, mean_dur / 100.0);
  // first blocks: where they landed
  printf("   first 24 blocks (xcc/se/cu):");
  for (int i = 0; i < 24 && i < n; ++i)
    printf(" %u/%u/%u", h[i].xcc_id, (h[i].hw_id >> 13) & 7, (h[i].hw_id >> 8) & 0xf);
  printf("\n");
  hipFree(d);
  hipFree(sink);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  probe<33792>(256, 2, iters, 0);
  probe<33792>(256, 2, iters, 48128);
  probe<33792>(512, 1, iters, 0);
  probe<25600>(256, 2, iters, 0);
  probe<61440>(256, 2, iters, 0);
  probe<37376>(1024, 2, iters, 0);
  probe<37376>(384, 2, iters, 0);
  return 0;
}
