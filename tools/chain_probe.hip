// Stand-alone timing of the small-batch kernels (chain.hip.h) on the DFlexAntPPO network, 2,048 rows, two towers.
// Build variants with -DCH_ABL=<bits> (see chain.hip.h) to price parts of a kernel; `cold` runs rewrite the weights between
// launches the way the optimizer does.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../mineral_amd/csrc/chain.hip.h"
using namespace mrl;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u + seed;
  x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
  p[i] = ((x & 0xffff) / 32768.0f - 1.0f) * scale;
}
__global__ void touch_kernel(float* p, size_t n, float d) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] += d;
}

// issue rate of v_mfma_f32_16x16x4_f32: NACC independent accumulators per wave, `n` rounds; lane 0 of every wave reports shader
// cycles (s_memtime) and 100 MHz ticks (s_memrealtime) spent in the loop
template <int NACC>
__global__ void mfma_rate_kernel(int n, float* out, unsigned long long* clk) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.0f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (s == 12345.678f) out[0] = s;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    clk[2 * w] = t1 - t0;
    clk[2 * w + 1] = r1 - r0;
  }
}

static float* dalloc(size_t n, unsigned seed, float scale) {
  float* p;
  CK(hipMalloc(&p, n * 4));
  hipLaunchKernelGGL(fill_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, p, n, seed, scale);
  return p;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 2048;
  const int iters = 200;
  const int in0 = 64, nl = 3, units[3] = {512, 256, 128};
  const int Mpad = (M + 127) / 128 * 128;
  ChainFwdP f; memset(&f, 0, sizeof(f));
  ChainBwdP b; memset(&b, 0, sizeof(b));
  WgradSmallP w; memset(&w, 0, sizeof(w));
  ReblockP rb; memset(&rb, 0, sizeof(rb));
  float* x = dalloc((size_t)M * in0, 1, 1.0f);
  f.x = x; f.M = M; f.in_dim = in0; f.n_layers = nl; f.ld = 512; f.rows_written = Mpad; f.ln_eps = 1e-5f;
  b.M = M; b.n_layers = nl; b.ld = 512;
  std::vector<std::pair<float*, size_t>> params;   // natural-layout weights, to be touched in cold runs
  int piece = 0, task = 0;
  w.M = M; w.nsplit = 10; w.ksplit = 224; w.nsplit = (M + w.ksplit - 1) / w.ksplit; w.lo = -5; w.hi = 5;
  if (argc > 2) { w.nsplit = atoi(argv[2]); w.ksplit = ((M + w.nsplit - 1) / w.nsplit + 31) / 32 * 32; w.nsplit = (M + w.ksplit - 1) / w.ksplit; }
  for (int t = 0; t < 2; ++t) {
    float* xh[3]; float* dz[3]; float* g[3]; float* be[3]; float* hact[3];
    std::vector<float*> cps;
    float* seeds = dalloc((size_t)M * 32, 77 + t, 0.01f);
    b.seeds[t] = seeds;
    for (int l = 0; l < nl; ++l) {
      const int u = units[l], in = l ? units[l - 1] : in0, kn = l + 1 < nl ? units[l + 1] : 32;
      f.units[l] = b.units[l] = u;
      float* W = dalloc((size_t)u * in, 10 * t + l, 0.05f);
      float* Wn = dalloc((size_t)kn * u, 20 * t + l, 0.05f);
      params.push_back({W, (size_t)u * in});
      float* Wf = dalloc((size_t)u * ((in + 31) & ~31), 0, 0.f);
      float* Wb = dalloc((size_t)kn * u, 0, 0.f);
      g[l] = dalloc(u, 30 + l, 1.0f); be[l] = dalloc(u, 40 + l, 0.1f);
      xh[l] = dalloc((size_t)Mpad * u, 50 + l, 1.0f); dz[l] = dalloc((size_t)Mpad * u, 60 + l, 0.01f);
      f.W[t][l] = Wf; f.b[t][l] = dalloc(u, 70 + l, 0.1f); f.g[t][l] = g[l]; f.be[t][l] = be[l];
      f.xhat[t][l] = xh[l]; f.rstd[t][l] = dalloc(Mpad, 80 + l, 1.0f);
      hact[l] = dalloc((size_t)Mpad * u, 90 + l, 1.0f); f.hact[t][l] = hact[l];
      b.Wn[t][l] = Wb; b.xhat[t][l] = xh[l]; b.rstd[t][l] = f.rstd[t][l]; b.g[t][l] = g[l]; b.be[t][l] = be[l]; b.dz[t][l] = dz[l];
      b.dy[t][l] = dalloc((size_t)Mpad * u, 61 + l, 0.01f);
      cps.push_back(dalloc((size_t)w.nsplit * 3 * u, 0, 0.f));
      ReblockDesc& d0 = rb.d[rb.nd++];
      d0.kind = 0; d0.src = W; d0.dst = Wf; d0.N = u; d0.K = in; d0.rows = u; d0.piece0 = piece; piece += u * ((in + 31) & ~31) / 4;
      ReblockDesc& d1 = rb.d[rb.nd++];
      d1.kind = 1; d1.src = Wn; d1.dst = Wb; d1.N = u; d1.K = kn; d1.rows = kn; d1.piece0 = piece; piece += kn * u / 4;
    }
    for (int l = nl; l >= 0; --l) {
      WgDesc& d = w.d[w.nd++];
      const bool head = l == nl;
      d.u = head ? 32 : units[l]; d.in = head ? units[nl - 1] : (l ? units[l - 1] : in0);
      d.ldz = d.u; d.slab_rows = d.u;
      d.dz = head ? b.seeds[t] : dz[l];
      d.xin = l == 0 ? x : hact[l - 1];
      d.pro = PRO_NONE;
      d.pro0 = l == 0 ? nullptr : g[l - 1]; d.pro1 = l == 0 ? nullptr : be[l - 1];
      d.slab = dalloc((size_t)w.nsplit * d.u * d.in, 0, 0.f);
      if (!head) { d.colpart = cps[l]; d.dy = b.dy[t][l]; d.xout = xh[l]; }
      d.tiles_n = (d.in + 63) / 64; d.tiles = ((d.u + 63) / 64) * d.tiles_n; d.task0 = task; task += d.tiles * w.nsplit;
    }
  }
  rb.npieces = piece; w.ntasks = task;
  // block-tile weight gradients: 9 row ranges for the 128-column tiles, 5 for the 64-column ones
  WgTileP wt; memset(&wt, 0, sizeof(wt));
  wt.M = M;
  int ttask = 0;
  const int sfull = argc > 4 ? atoi(argv[4]) : 10;
  for (int i = 0; i < w.nd; ++i) {
    const WgDesc& o = w.d[i];
    WgTileDesc& d = wt.d[wt.nd++];
    d.dz = o.dz; d.xin = o.xin; d.dy = o.dy; d.xout = o.xout; d.u = o.u; d.in = o.in;
    d.bm = (d.u <= 32 && d.in > 64) ? 32 : 128;
    d.bn = d.in <= 64 ? 64 : 128;
    d.nsplit = d.bm == 32 ? (sfull + 3) / 4 : (d.bn == 128 ? sfull : (sfull + 1) / 2);
    d.slab = dalloc((size_t)d.nsplit * d.u * d.in, 0, 0.f);
    d.colpart = dalloc((size_t)d.nsplit * 3 * d.u, 0, 0.f);
    d.tiles_n = (d.in + d.bn - 1) / d.bn; d.tiles = ((d.u + d.bm - 1) / d.bm) * d.tiles_n; d.task0 = ttask; ttask += d.tiles * d.nsplit;
  }
  wt.ntasks = ttask;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  auto run_wt = [&] { hipLaunchKernelGGL(wgrad_tile_kernel, dim3(ttask), dim3(512), WT_LDS_F * 4, 0, wt); };
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fwd_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_bwd_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  const size_t lds = (CH_RED + 2 * CH_ROWS * 512) * 4;
  const int wg_lds = argc > 3 ? atoi(argv[3]) : 84 * 1024;
  auto run_rb = [&] { hipLaunchKernelGGL(reblock_kernel, dim3((piece + 255) / 256), dim3(256), 0, 0, rb); };
  auto run_f = [&] { hipLaunchKernelGGL(chain_fwd_kernel<0>, dim3(Mpad / 16, 2), dim3(512), lds, 0, f); };
  auto run_b = [&] { hipLaunchKernelGGL(chain_bwd_kernel<0>, dim3((M + 15) / 16, 2), dim3(512), lds, 0, b); };
  auto run_w = [&] { hipLaunchKernelGGL(wgrad_small_kernel, dim3((task + 3) / 4), dim3(512), wg_lds, 0, w); };
  auto touch = [&] { for (auto& p : params) hipLaunchKernelGGL(touch_kernel, dim3((p.second + 255) / 256), dim3(256), 0, 0, p.first, p.second, 1e-6f); };
  run_rb(); CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_it = [&](const char* name, auto fn, bool cold) {
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) {
      if (cold) { touch(); run_rb(); }
      CK(hipEventRecord(e0, 0)); fn(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-22s %s  median %7.1f us  min %7.1f us\n", name, cold ? "cold" : "warm", ts[iters / 2], ts[0]);
  };
  printf("CH_ABL=%d M=%d tasks=%d (nsplit %d, ksplit %d) wg_lds %d\n", CH_ABL, M, task, w.nsplit, w.ksplit, wg_lds);
  {
    unsigned long long* clk; CK(hipMalloc(&clk, 4096 * 16));
    float* outp = dalloc(16, 0, 0.f);
    std::vector<unsigned long long> h(4096 * 2);
    auto rate = [&](const char* name, auto kern, int threads, int nacc, int n) {
      std::vector<float> ts;
      for (int i = 0; i < 50; ++i) {
        CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, n, outp, clk); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms * 1e3f);
      }
      std::sort(ts.begin(), ts.end());
      CK(hipMemcpy(h.data(), clk, 256 * (threads / 64) * 16, hipMemcpyDeviceToHost));
      std::vector<double> cyc, mhz;
      for (int w = 0; w < 256 * (threads / 64); ++w) { cyc.push_back((double)h[2 * w]); mhz.push_back(100.0 * h[2 * w] / (double)h[2 * w + 1]); }
      std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
      printf("%-28s %d MFMAs per wave: %7.1f us; %.1f cycles per MFMA and wave (median), clock %.0f MHz (median)\n", name, nacc * n, ts[25],
             cyc[cyc.size() / 2] / (nacc * n), mhz[mhz.size() / 2]);
    };
    rate("1 wave/SIMD, 16 accumulators", mfma_rate_kernel<16>, 256, 16, 24);
    rate("1 wave/SIMD, 1 accumulator", mfma_rate_kernel<1>, 256, 1, 384);
    rate("2 waves/SIMD, 2 accumulators", mfma_rate_kernel<2>, 512, 2, 192);
    rate("2 waves/SIMD, 1 accumulator", mfma_rate_kernel<1>, 512, 1, 384);
    rate("1 wave/SIMD, 16 acc, long", mfma_rate_kernel<16>, 256, 16, 2400);
  }
  time_it("reblock", run_rb, false);
  time_it("chain_fwd", run_f, false);
  time_it("chain_fwd", run_f, true);
  time_it("chain_bwd", run_b, false);
  time_it("chain_bwd", run_b, true);
  time_it("wgrad_small", run_w, false);
  printf("wgrad_tile: %d blocks\n", ttask);
  time_it("wgrad_tile", run_wt, false);
#ifdef CH_TRACE
  auto dump_trace = [&](const char* name, auto fn) {
    for (int i = 0; i < 5; ++i) fn();
    CK(hipDeviceSynchronize());
    unsigned long long h[8][64];
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(ch_trace_buf), sizeof(h)));
    printf("%s: cycles since kernel entry of block (8,0), per wave; per layer: main loop end | stats barrier passed | stores issued | layer barrier passed\n", name);
    for (int w = 0; w < 8; ++w) {
      printf("  wave %d:", w);
      for (int sl = 1; sl <= 12; ++sl) printf(" %6lld%s", (long long)(h[w][sl] - h[w][0]), sl % 4 == 0 ? " |" : "");
      printf("\n");
    }
  };
  dump_trace("chain_fwd", run_f);
  dump_trace("chain_bwd", run_b);
  {
    for (int i = 0; i < 5; ++i) run_w();
    CK(hipDeviceSynchronize());
    unsigned long long h[8][64];
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(ch_trace_buf), sizeof(h)));
    printf("wgrad_small block 16: cycles since task start, per wave: [before | after] each 32-row group ... then after the slab stores\n");
    for (int wv = 0; wv < 8; ++wv) {
      printf("  wave %d:", wv);
      for (int sl = 1; sl <= 2 * ((w.ksplit + 31) / 32); ++sl) printf(" %6lld", (long long)(h[wv][sl] - h[wv][0]));
      printf("  | stores %6lld\n", (long long)(h[wv][40] - h[wv][0]));
    }
  }
  {
    for (int i = 0; i < 5; ++i) run_wt();
    CK(hipDeviceSynchronize());
    unsigned long long h[8][64];
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(ch_trace_buf), sizeof(h)));
    printf("wgrad_tile block 16: cycles since task start, per wave: first barrier | [stage computed, stage barrier] ... | slab stores issued\n");
    for (int wv = 0; wv < 8; ++wv) {
      printf("  wave %d: %6lld |", wv, (long long)(h[wv][1] - h[wv][0]));
      for (int sl = 2; sl <= 17; ++sl) printf(" %6lld", (long long)(h[wv][sl] - h[wv][0]));
      printf("  | stores %6lld; %.0f MHz\n", (long long)(h[wv][40] - h[wv][0]), 100.0 * (double)(h[wv][40] - h[wv][0]) / (double)(h[wv][51] - h[wv][50]));
    }
  }
  {
    CK(hipDeviceSynchronize());
    run_wt();
    CK(hipDeviceSynchronize());
    static unsigned long long hb[1024][2];
    CK(hipMemcpyFromSymbol(hb, HIP_SYMBOL(ch_block_clk), sizeof(hb)));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < ttask; ++i) { t0 = std::min(t0, hb[i][0]); t1 = std::max(t1, hb[i][1]); }
    printf("wgrad_tile blocks: first entry to last exit %.2f us; per block [start offset us, duration us]:\n", (t1 - t0) / 100.0);
    for (int i = 0; i < ttask; ++i) printf("%s%5.2f,%5.2f", i % 8 ? "  " : "\n  ", (hb[i][0] - t0) / 100.0, (hb[i][1] - hb[i][0]) / 100.0);
    printf("\n");
  }
#endif
  // back-to-back sequence as in a step
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) { run_rb(); run_f(); run_b(); run_w(); }
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("sequence reblock+fwd+bwd+wgrad: %.1f us per step\n", ms * 1e3f / iters);
  return 0;
}
