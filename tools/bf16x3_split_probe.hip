// Probe: an fp32-accurate weight-gradient GEMM on the bf16 matrix cores of gfx950.
//
//   C[z][m][n] = sum_k A[z][k][m] * B[z][k][n]        (A = dZ [rows][out], B = X [rows][in]; both row-major, k = batch row)
//
// Every fp32 operand is split exactly into three bf16 terms x = h + m + l (8 significant bits each, rounded to nearest), and the
// product is accumulated in fp32 from the six terms hh, hm, mh, mm, hl, lh (the dropped ones are below 2^-23 of |a*b|): six
// v_mfma_f32_32x32x16_bf16 (32 cycles each, 16 k) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each, 2 k): 192 against
// 512 matrix-core cycles per 16 k.  The probe measures what is left of that 2.7x once the split (VALU) and the staging are
// paid, and the error against an fp64 sum next to that of a plain fp32 sum.
//
//   hipcc --offload-arch=gfx950 -O3 -Xclang -target-feature -Xclang -packed-fp32-ops tools/bf16x3_split_probe.hip -o tools/bin/bf16x3_split_probe
//   tools/bin/bf16x3_split_probe [M N K splits]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

#ifndef PROBE_SCHED
#define PROBE_SCHED 1     // interleave the split and the LDS traffic with the MFMAs of the same wave
#endif
#ifndef PROBE_VARIANT
#define PROBE_VARIANT 0   // timing diagnostics (wrong results): 2 = no split / LDS writes in the loop, 3 = no global loads either,
                          // 4 = no fragment reads either, 5 = no barrier either (bare MFMA loop)
#endif
constexpr int BM = 128, BN = 128, BK = 16;
constexpr int ROWB = 320;                 // bytes per k-row of a plane: 128 bf16 + 64 B, so four consecutive rows tile the 64 banks
constexpr int PLANE = BK * ROWB;          // 5,120 B
constexpr int OPER = 3 * PLANE;           // three planes per operand
constexpr int STAGE = 2 * OPER;           // A and B of one k-step: 30,720 B; two stages in flight

// x (4 floats) -> three packed bf16 quadruples (exact: x = h + m + l), every term rounded to nearest even
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float lo_f(unsigned pk) { return __uint_as_float(__builtin_amdgcn_perm(pk, 0u, 0x05040c0cu)); }
__device__ __forceinline__ float hi_f(unsigned pk) { return __uint_as_float(pk & 0xffff0000u); }
__device__ __forceinline__ void split4(const f32x4 x, u32x2& h, u32x2& m, u32x2& l) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float x0 = x[2 * e], x1 = x[2 * e + 1];
    const unsigned hp = pack2(x0, x1);
    const float r0 = x0 - lo_f(hp), r1 = x1 - hi_f(hp);
    const unsigned mp = pack2(r0, r1);
    const float s0 = r0 - lo_f(mp), s1 = r1 - hi_f(mp);
    h[e] = hp;
    m[e] = mp;
    l[e] = pack2(s0, s1);
  }
}

__device__ __forceinline__ bf16x8 tr_frag(const char* base) {
  // two transposed 4-row blocks: k .. k+3 and k+4 .. k+7 of this lane's column
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + 4 * ROWB));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// grid: x = (M/128)*(N/128) tiles * splits, z = towers.  slab[z][split][M][N]
__global__ __launch_bounds__(256, 2) void wgrad_bf16x3(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ slab, int M, int N,
                                                         int K, int splits, int same_rows) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = N / BN, tiles = (M / BM) * tiles_n;
  // workgroups go round-robin over the 8 XCDs: give each XCD a contiguous range of logical ids, so that the tiles of one
  // k-split (which share their A and B rows) run on one XCD and meet in its L2
  const int logical = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int tile = logical % tiles, split = logical / tiles;
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int ks = K / splits, k_begin = same_rows ? 0 : split * ks, nsteps = ks / BK;   // same_rows: diagnostic, every split reads split 0's rows
  const float* Az = A + (size_t)blockIdx.z * K * M;
  const float* Bz = B + (size_t)blockIdx.z * K * N;

  // staging map: chunk c (4 columns) of rows (t >> 5) and (t >> 5) + 8
  const int c = t & 31, kr = t >> 5;
  f32x4 ra[2][2], rb[2][2];                // [register set = step parity][row]
  auto gload = [&](int st, auto SET) {
    constexpr int S = decltype(SET)::value;
    const int k = k_begin + st * BK + kr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const size_t kk = (size_t)min(k + 8 * i, K - 1);   // steps past the end are loaded (clamped) and staged, never multiplied
      ra[S][i] = *reinterpret_cast<const f32x4*>(Az + kk * M + m0 + 4 * c);
      rb[S][i] = *reinterpret_cast<const f32x4*>(Bz + kk * N + n0 + 4 * c);
    }
  };
  auto stage = [&](char* buf, auto SET) {
    constexpr int S = decltype(SET)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x2 h, m, l;
      char* p = buf + (kr + 8 * i) * ROWB + c * 8;
      split4(ra[S][i], h, m, l);
      *reinterpret_cast<u32x2*>(p) = h;
      *reinterpret_cast<u32x2*>(p + PLANE) = m;
      *reinterpret_cast<u32x2*>(p + 2 * PLANE) = l;
      split4(rb[S][i], h, m, l);
      *reinterpret_cast<u32x2*>(p + OPER) = h;
      *reinterpret_cast<u32x2*>(p + OPER + PLANE) = m;
      *reinterpret_cast<u32x2*>(p + OPER + 2 * PLANE) = l;
    }
  };

  // fragment map of v_mfma_f32_32x32x16_bf16 through transposed reads: 16-lane group g takes columns 16 (g & 1) .. +15 and
  // k rows 8 (g >> 1) .. +7; lane 4q + p of the group supplies the address of row q, columns 4p .. 4p+3
  const int g = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
  const int frag_off = (8 * (g >> 1) + q) * ROWB + (16 * (g & 1) + 4 * p4) * 2;
  const int fa = frag_off + (wm * 64) * 2;
  const int fb = OPER + frag_off + (wn * 64) * 2;
  bf16x8 fra[2][2][3], frb[2][2][3];       // [fragment set = step parity][tile][plane]
  auto frags = [&](auto SET) {
    constexpr int S = decltype(SET)::value;
    const char* cur = lds + S * STAGE;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fra[S][i][pl] = tr_frag(cur + fa + pl * PLANE + i * 64);
        frb[S][i][pl] = tr_frag(cur + fb + pl * PLANE + i * 64);
      }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto mfmas = [&](auto SET) {
    constexpr int S = decltype(SET)::value;
    // six terms per accumulator, smallest first; consecutive MFMAs go to different accumulators
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[S][i][TA[term]], frb[S][j][TB[term]], acc[i][j], 0, 0, 0);
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // Step st (parity P): the fragments of step st are read from stage P while the MFMAs of step st-1 run from the other
  // fragment set; between the MFMAs the operands of step st+1 (register set P^1) are split and written to stage P^1, whose
  // fragments were read a step ago; then the loads of step st+3 refill that register set.  One barrier per step.
  auto steady = [&](int st, auto PAR) {
    constexpr int P = decltype(PAR)::value;
    using Q = std::integral_constant<int, P ^ 1>;
#if PROBE_VARIANT < 4
    frags(PAR);
#endif
    mfmas(Q{});
#if PROBE_VARIANT == 0
    stage(lds + (P ^ 1) * STAGE, Q{});
#endif
#if PROBE_VARIANT < 3
    gload(st + 3, Q{});
#endif
#if PROBE_SCHED
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // four VALU
      if (i % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // one LDS write
    }
    __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);     // the global loads
#endif
#if PROBE_VARIANT != 5
    __syncthreads();
#endif
  };

  gload(0, I0{});
  gload(1, I1{});
  stage(lds, I0{});
  gload(2, I0{});
  __syncthreads();
  frags(I0{});
#if PROBE_VARIANT >= 4
  frags(I1{});
#endif
  stage(lds + STAGE, I1{});
  gload(3, I1{});
  __syncthreads();
  int st = 1;
  for (; st + 1 < nsteps; st += 2) {                   // nsteps is even: pairs (1,2) .. (nsteps-3, nsteps-2)
    steady(st, I1{});
    steady(st + 1, I0{});
  }
  steady(st, I1{});                                    // st = nsteps - 1
  mfmas(I1{});

  float* out = slab + ((size_t)blockIdx.z * splits + split) * M * N;
  const int col = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        out[(size_t)row * N + n0 + wn * 64 + j * 32 + col] = acc[i][j][r];
      }
}

static float lcg(uint64_t& s) {
  s = s * 6364136223846793005ull + 1442695040888963407ull;
  return (float)((s >> 40) & 0xffffff) / 8388608.0f - 1.0f;   // [-1, 1)
}

int main(int argc, char** argv) {
  int M = 256, N = 512, K = 32768, splits = 64;
  const int Z = 2;
  int same_rows = 0;
  if (argc >= 5) { M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); splits = atoi(argv[4]); }
  if (argc >= 6) same_rows = atoi(argv[5]);
  if (M % BM || N % BN || splits < 1 || K % (splits * BK * 2) || M < BM || N < BN || ((M / BM) * (N / BN) * splits) % 8) {
    fprintf(stderr, "need M %% 128 == 0, N %% 128 == 0, K %% (splits * 32) == 0, tiles * splits %% 8 == 0\n");
    return 2;
  }
  printf("wgrad C[%d][%d] over K=%d, %d towers, %d splits (%d k-steps per block), grid %d blocks\n", M, N, K, Z, splits, K / splits / BK,
         (M / BM) * (N / BN) * splits * Z);
  fflush(stdout);
  std::vector<float> hA((size_t)Z * K * M), hB((size_t)Z * K * N);
  uint64_t s = 12345;
  // magnitudes spread over several binades, as gradients and activations are
  for (auto& v : hA) { const float u = lcg(s); v = u * u * u * 1e-3f; }
  for (auto& v : hB) { const float u = lcg(s); v = u * (1.0f + 3.0f * u * u); }
  float *dA, *dB, *dS;
  const size_t slab_elems = (size_t)Z * splits * M * N;
  CK(hipMalloc(&dA, hA.size() * 4));
  CK(hipMalloc(&dB, hB.size() * 4));
  CK(hipMalloc(&dS, slab_elems * 4));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  const size_t smem = 2 * STAGE;
  CK(hipFuncSetAttribute((const void*)wgrad_bf16x3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const dim3 grid((M / BM) * (N / BN) * splits, 1, Z);
  hipLaunchKernelGGL(wgrad_bf16x3, grid, dim3(256), smem, 0, dA, dB, dS, M, N, K, splits, same_rows);
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  printf("first launch done\n");
  fflush(stdout);

  // accuracy on sampled entries: fp64 sum as the truth, a sequential fp32 sum for scale
  std::vector<float> hS(slab_elems);
  CK(hipMemcpy(hS.data(), dS, slab_elems * 4, hipMemcpyDeviceToHost));
  double worst_split = 0, worst_f32 = 0;
  uint64_t r = 999;
  for (int it = 0; it < 256; ++it) {
    r = r * 6364136223846793005ull + 1442695040888963407ull;
    const int z = (r >> 33) % Z, m = (r >> 40) % M, n = (r >> 20) % N;
    double ref = 0, mag = 0, got = 0;
    float f32 = 0.f;
    for (int k = 0; k < K; ++k) {
      const float a = hA[((size_t)z * K + k) * M + m], b = hB[((size_t)z * K + k) * N + n];
      ref += (double)a * (double)b;
      mag += fabs((double)a * (double)b);
      f32 = fmaf(a, b, f32);
    }
    for (int sp = 0; sp < splits; ++sp) got += (double)hS[(((size_t)z * splits + sp) * M + m) * N + n];
    worst_split = fmax(worst_split, fabs(got - ref) / mag);
    worst_f32 = fmax(worst_f32, fabs((double)f32 - ref) / mag);
  }
  if (!same_rows) printf("max |err| / sum|a*b| over 256 sampled entries: split-bf16 %.3e   sequential fp32 fma %.3e\n", worst_split, worst_f32);
  fflush(stdout);

  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 50;
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(wgrad_bf16x3, grid, dim3(256), smem, 0, dA, dB, dS, M, N, K, splits, same_rows);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(wgrad_bf16x3, grid, dim3(256), smem, 0, dA, dB, dS, M, N, K, splits, same_rows);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, flop = 2.0 * Z * M * (double)N * K;
  printf("%.1f us per launch = %.1f TFLOP/s of fp32-equivalent work (fp32 matrix-core peak 157.3, six-term bf16 bound %.0f)\n", us, flop / us * 1e-6,
         2500.0 / 6.0);
  return 0;
}
