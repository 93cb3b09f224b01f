// Forward of a 512-wide LayerNorm FIRST layer on at most 64 inputs, on batches above the chain kernels' row count:
//   z = norm(x) W^T + b,  xhat = (z - mean(z)) * rstd,  rstd = 1 / sqrt(var(z) + eps)          (xhat and rstd are what is stored)
//
// The layer writes 2 KB per batch row and reads 256 B: it is a streaming kernel with a small GEMM inside.  gemm_kernel<64,512>
// gives every 64-row tile a block of its own, which stages the whole weight matrix again (196 KB of planes from L2 per block)
// and runs load -> 4 MFMA steps -> LayerNorm -> 128 KB of stores in sequence with one other block to overlap with: 3.0 TB/s.
// Here the weights never move: a block of 12 waves walks a range of batch rows in steps of 32.
//   matrix waves 4 .. 11  wave m keeps the three bf16 planes of W rows 64 m .. +63 (all 64 k) in 96 registers for the whole
//                         range; per step it reads the planes of the step's 32 x 64 input tile from LDS, runs the 48
//                         v_mfma_f32_32x32x16_bf16 of the six-term product in the transposed orientation (a lane holds one
//                         batch row and 4-column runs of z) and writes z [32][512] to the LDS stage of the step
//   row waves 0 .. 3      normalise, clamp and split the NEXT step's input tile into its LDS planes; then the LayerNorm of
//                         the PREVIOUS step: wave w owns 8 rows, a lane two 16-byte chunks per row, bias add, two-pass
//                         statistics on DPP adds (a wave holds whole rows), xhat out in 1 KB pieces per wave and store
// One barrier per step; two z stages and two input stages (158 KB of LDS, one block per CU).
#pragma once

#ifndef LNF0_ABL
#define LNF0_ABL 0   // timing-only builds (tools/lnfwd0_probe.hip): bit 0 no MFMAs, 1 no LayerNorm arithmetic, 2 no xhat stores, 3 no z stage writes
#endif

namespace mrl {

#ifdef LNF0_TRACE
__device__ long long g_lnf0_trace[2][64];   // [role][slot] cycle stamps of block 0 (probe builds only)
#define LNF0_STAMP(role, slot) do { if (blockIdx.x == 7 && blockIdx.y == 0 && lane == 0 && (wave == 0 || wave == 4) && (slot) < 64) g_lnf0_trace[role][slot] = __builtin_readcyclecounter(); } while (0)
#else
#define LNF0_STAMP(role, slot) do { } while (0)
#endif

struct LnFwd0Args {
  const float* x;           // [rows][in]
  int in;                   // inputs per row: a multiple of 4, at most 64
  const float* mean;        // input normalisation (nullptr: the input is used as it is)
  const float* std;
  float clamp_lo, clamp_hi;
  const float* W[2];        // [512][in]
  const float* bias[2];     // [512]
  float* xhat[2];           // [rows_pad][512]
  float* rstd[2];           // [rows_pad]
  int M, Mpad;              // rows with data; rows to write (the activation tensors are kept finite up to a whole 128-row tile)
  int rows_per_block;       // a multiple of 32
  int tw0;
  float eps;
};

constexpr int LNF0_N = 512, LNF0_K = 64, LNF0_ROWS = 32;
constexpr int LNF0_ZP = 520;                           // row pitch of a z stage in floats
constexpr int LNF0_XP = 144;                           // row pitch of an input plane in bytes (64 bf16 + 16)
constexpr int LNF0_ZSTAGE = LNF0_ROWS * LNF0_ZP * 4;
constexpr int LNF0_XSTAGE = 3 * LNF0_ROWS * LNF0_XP;
constexpr int LNF0_LDS = 2 * LNF0_ZSTAGE + 2 * LNF0_XSTAGE;
constexpr int LNF0_THREADS = 64 * 12;

__global__ __launch_bounds__(LNF0_THREADS, 1) void lnfwd0_kernel(const LnFwd0Args a) {
  using namespace wsplit;
  constexpr int N = LNF0_N, K = LNF0_K, R = LNF0_ROWS, ZP = LNF0_ZP, XP = LNF0_XP;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  auto zst = [&](int i) __attribute__((always_inline)) { return lds + i * LNF0_ZSTAGE; };
  auto xst = [&](int i) __attribute__((always_inline)) { return lds + 2 * LNF0_ZSTAGE + i * LNF0_XSTAGE; };
  const int t = threadIdx.x, lane = t & 63, tw = blockIdx.y + a.tw0;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r0 = blockIdx.x * a.rows_per_block, r1 = min(a.Mpad, r0 + a.rows_per_block);
  const int nsteps = (r1 - r0 + R - 1) / R;
  if (nsteps <= 0) return;

  if (wave < 4) {
    // ================= row waves =================
    const float* __restrict__ xin = a.x;
    float* __restrict__ xo = a.xhat[tw];
    float* __restrict__ rso = a.rstd[tw];
    f32x4 b4[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) b4[c] = *reinterpret_cast<const f32x4*>(a.bias[tw] + 4 * lane + 256 * c);
    // input staging: chunk cb (4 inputs) of rows rb and rb + 16 of the step
    const int cb = t & 15, rb = t >> 4;
    const bool in_b = 4 * cb < a.in;
    const bool norm = a.mean != nullptr;
    f32x4 pv0 = {0.f, 0.f, 0.f, 0.f}, pv1 = {1.f, 1.f, 1.f, 1.f};
    if (norm && in_b) {
      pv0 = *reinterpret_cast<const f32x4*>(a.mean + 4 * cb);
      pv1 = *reinterpret_cast<const f32x4*>(a.std + 4 * cb);
    }
    f32x4 ob[2];
    auto load_in = [&](int st) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int64_t row = min(r0 + st * R + rb + 16 * i, a.M - 1);      // rows past the data repeat the last row: finite padding
        ob[i] = *reinterpret_cast<const f32x4*>(xin + row * a.in + (in_b ? 4 * cb : 0));   // chunks past the width are zeroed when staged
      }
    };
    auto stage_in = [&](int st) __attribute__((always_inline)) {
      char* buf = xst(st & 1);
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x4 v = in_b ? ob[i] : zero;
        if (norm && in_b) {                               // mineral/common/normalizers.py:65-68
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf((v[e] - pv0[e]) / pv1[e], a.clamp_lo), a.clamp_hi);
        }
        u32x2 h, m, l;
        split4(v, h, m, l);
        char* q = buf + (rb + 16 * i) * XP + cb * 8;
        *reinterpret_cast<u32x2*>(q) = h;
        *reinterpret_cast<u32x2*>(q + R * XP) = m;
        *reinterpret_cast<u32x2*>(q + 2 * R * XP) = l;
      }
    };
    // LayerNorm of the z stage of step st: wave w owns rows 8 w .. +7.  The eight rows advance together, phase by phase,
    // in straight-line code: one row's chain (LDS read, two wave totals, store) is some 800 cycles of latency with
    // nothing else on the SIMD to hide it, eight interleaved chains cost little more than one.  Every row of the block
    // exists in the padded activation tensor (the block's row count is a multiple of the step), so nothing is guarded.
    auto layernorm = [&](int st) __attribute__((always_inline)) {
      const float* zb = reinterpret_cast<const float*>(zst(st & 1));
      const float invn = 1.0f / (float)N;
      const int lr0 = 8 * wave, row0 = r0 + st * R + lr0;
      f32x4 z[8][2];
      float s[8], v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c) z[i][c] = *reinterpret_cast<const f32x4*>(zb + (lr0 + i) * ZP + 4 * lane + 256 * c);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        s[i] = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          z[i][c] += b4[c];
          s[i] += (z[i][c][0] + z[i][c][1]) + (z[i][c][2] + z[i][c][3]);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] = ((LNF0_ABL & 2) ? s[i] : wave_total(s[i])) * invn;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[i] = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            z[i][c][e] -= s[i];
            v[i] += z[i][c][e] * z[i][c][e];
          }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (LNF0_ABL & 2) ? v[i] : wave_total(v[i]);
      // the totals are wave-uniform: lane i takes row i's, one square root and one division serve the eight rows
      // (the exact fp32 sequences are some 25 instructions), and the rows read their factor back as a scalar
      float mine = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) asm("v_writelane_b32 %0, %1, %2" : "+v"(mine) : "s"(v[i]), "n"(i));
      if (!(LNF0_ABL & 2)) mine = 1.0f / sqrtf(mine * invn + a.eps);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), i));
      if (!(LNF0_ABL & 4) || mine == 1.2345e-30f) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int c = 0; c < 2; ++c) *reinterpret_cast<f32x4*>(xo + (int64_t)(row0 + i) * N + 4 * lane + 256 * c) = z[i][c] * v[i];
        if (lane < 8) rso[row0 + lane] = mine;
      }
    };
    // The loop body is the same on every pass (the loads and the staging past the last step repeat the last step into
    // a stage nobody reads) and uses the prefetched inputs after the LayerNorm stores of the same pass: the wait for
    // the inputs can then be "all but the stores issued since".  With a conditional part, or with the use ahead of the
    // stores, the compiler has to wait for every outstanding store, and their round trip to memory is paid once a step.
    asm volatile("" :: "v"(b4[0]), "v"(b4[1]));         // the bias has arrived before the loop, not inside it
    LNF0_STAMP(0, 0);
    load_in(0);
    stage_in(0);
    load_in(min(1, nsteps - 1));
    LNF0_STAMP(0, 1);
    __syncthreads();                                    // inputs of step 0 are staged
    LNF0_STAMP(0, 2);
    stage_in(1);                                        // the matrix waves multiply step 0
    load_in(min(2, nsteps - 1));
    LNF0_STAMP(0, 3);
    LNF0_STAMP(0, 4);
    __syncthreads();
    LNF0_STAMP(0, 5);
    for (int st = 1; st < nsteps; ++st) {
      // the matrix waves multiply step st
      layernorm(st - 1);
      LNF0_STAMP(0, 3 + 3 * st);
      stage_in(st + 1);                                 // after the stores: the wait for the inputs can then count them
      load_in(min(st + 2, nsteps - 1));
      LNF0_STAMP(0, 4 + 3 * st);
      __syncthreads();                                  // z of step st and the inputs of step st + 1 are complete
      LNF0_STAMP(0, 5 + 3 * st);
    }
    layernorm(nsteps - 1);
    LNF0_STAMP(0, 3 + 3 * nsteps);
  } else {
    // ================= matrix waves =================
    const int mw = wave - 4, l32 = lane & 31, hh = lane >> 5;
    LNF0_STAMP(1, 0);
    const float* __restrict__ Wg = a.W[tw];
    // the planes of W rows 64 mw + 32 i + l32, k = 16 ks + 8 hh .. + 7: [tile i][k step][plane]
    bf16x8 fw[2][4][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int col = 64 * mw + 32 * i + l32, k0 = 16 * ks + 8 * hh;
        f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0;
        if (k0 < a.in) w0 = *reinterpret_cast<const f32x4*>(Wg + (int64_t)col * a.in + k0);
        if (k0 + 4 < a.in) w1 = *reinterpret_cast<const f32x4*>(Wg + (int64_t)col * a.in + k0 + 4);
        u32x2 h0, m0, l0, h1, m1, l1;
        split4(w0, h0, m0, l0);
        split4(w1, h1, m1, l1);
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        fw[i][ks][0] = __builtin_bit_cast(bf16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
        fw[i][ks][1] = __builtin_bit_cast(bf16x8, u32x4{m0[0], m0[1], m1[0], m1[1]});
        fw[i][ks][2] = __builtin_bit_cast(bf16x8, u32x4{l0[0], l0[1], l1[0], l1[1]});
      }
    LNF0_STAMP(1, 1);
    __syncthreads();                                    // inputs of step 0 are staged
    LNF0_STAMP(1, 2);
    for (int st = 0; st < nsteps; ++st) {
      const char* xb = xst(st & 1);
      float* zb = reinterpret_cast<float*>(zst(st & 1));
      f32x16 acc[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 fx[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fx[pl] = *reinterpret_cast<const bf16x8*>(xb + pl * R * XP + l32 * XP + (16 * ks + 8 * hh) * 2);
        constexpr int TW[6] = {2, 0, 1, 1, 0, 0}, TX[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if constexpr (LNF0_ABL & 1) acc[i][term] += (float)fw[i][ks][TW[term]][0] + (float)fx[TX[term]][1];
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[i][ks][TW[term]], fx[TX[term]], acc[i], 0, 0, 0);
          }
      }
      LNF0_STAMP(1, 3 + 3 * st);
      // acc[i][r] = z[batch row l32][column 64 mw + 32 i + (r & 3) + 8 (r >> 2) + 4 hh]: runs of four columns
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 v = {acc[i][4 * j], acc[i][4 * j + 1], acc[i][4 * j + 2], acc[i][4 * j + 3]};
          if (!(LNF0_ABL & 8) || v[0] == 1.2345e-30f) *reinterpret_cast<f32x4*>(zb + l32 * ZP + 64 * mw + 32 * i + 8 * j + 4 * hh) = v;
        }
      LNF0_STAMP(1, 4 + 3 * st);
      __syncthreads();                                  // z of step st is complete; the row waves may refill the other stages
      LNF0_STAMP(1, 5 + 3 * st);
    }
  }
}

}  // namespace mrl
