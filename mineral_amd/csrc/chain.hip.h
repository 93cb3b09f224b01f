// Kernels for batches of at most a few thousand rows (the minibatch of 2,048 rows that cfgs/agent/DFlexAntPPO.yaml ships, the
// SAC batch, the SHAC critic batches), where a launch per layer leaves most of the chip idle and every launch boundary costs
// as much as the layer itself.  gfx950 only.
//
//   chain_fwd_kernel   a block owns 16 batch rows of one tower and walks ALL its layers: the activations stay in LDS, the
//                      weights stream from L2 straight into registers (no other wave of the block needs the same weights, so
//                      an LDS stage would only add a barrier), Linear + bias + LayerNorm + activation per layer, x-hat and
//                      rstd stored for the backward pass.  mineral/nets/mlp.py:50-77.
//   chain_bwd_kernel   the same for the backward chain: dH = dZ * W, activation' and LayerNorm backward per layer, dZ kept
//                      in LDS for the layer below and stored for the weight gradients; column partials (d gamma, d beta,
//                      d bias) per block.  Every step is local to a batch row.
//   wgrad_small_kernel every weight gradient of the step (all layers, heads, towers) in ONE launch: a wave owns a 64 x 64
//                      tile of one dW and one range of batch rows, reads both operands straight from memory (16 bytes per
//                      lane along the feature dimension) and needs no LDS and no barrier; partial slabs per row range.
//
// All three multiply on v_mfma_f32_16x16x4_f32 (exact fp32 fma chains): with 16 rows per block the 32-wide fp32 tile would
// waste half of every instruction, and the weight stream from L2, not the matrix core, is the other bound of these kernels.
#pragma once
#include "gemm.hip.h"

#ifndef CH_ABL
#define CH_ABL 0   // diagnostic builds only (results are garbage): bit 0 no weight / operand loads in the main loops, 1 no MFMAs, 2 no epilogue
#endif

#ifdef CH_TRACE
// diagnostic builds only (tools/chain_probe.hip): shader-clock stamps of every wave of block (8, 0) of the chain kernels
__device__ unsigned long long ch_trace_buf[8][64];
__device__ unsigned long long ch_block_clk[1024][2];   // 100 MHz clock at entry / exit of every block of the traced weight-gradient launch
#define WG_BLOCK(which) do { if (threadIdx.x == 0 && blockIdx.x < 1024) ch_block_clk[blockIdx.x][which] = __builtin_amdgcn_s_memrealtime(); } while (0)
#ifndef CH_TRACE_WG
#define CH_TRACE_WG 16   // the traced block of the weight-gradient kernels
#endif
#define WG_STAMP(slot)                                                                                         \
  do {                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    if (blockIdx.x == CH_TRACE_WG && (threadIdx.x & 63) == 0 && (slot) < 64) ch_trace_buf[threadIdx.x >> 6][(slot)] = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
#define WG_REAL(slot)                                                                                          \
  do {                                                                                                         \
    if (blockIdx.x == CH_TRACE_WG && (threadIdx.x & 63) == 0) ch_trace_buf[threadIdx.x >> 6][(slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define CH_STAMP(slot)                                                                                         \
  do {                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    if (blockIdx.x == 8 && blockIdx.y == 0 && (threadIdx.x & 63) == 0) ch_trace_buf[threadIdx.x >> 6][(slot)] = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
#else
#define CH_STAMP(slot) do { } while (0)
#define WG_STAMP(slot) do { } while (0)
#define WG_REAL(slot) do { } while (0)
#define WG_BLOCK(which) do { } while (0)
#endif

namespace mrl {

constexpr int CH_ROWS = 16;     // batch rows per block of the chain kernels
constexpr int CH_WAVES = 8;     // waves per block
constexpr int CH_MAXL = 4;      // layers per tower (MRL_MAX_LAYERS)
constexpr int CH_RS = 68;       // floats per row of the row-statistics exchange: 32 lane groups x 2 values, + 4 so that the 16 rows
                                // start on different banks
constexpr int CH_RED = CH_ROWS * CH_RS;   // floats of that exchange at the start of the dynamic LDS

// tower kinds, as MRL_TOWER_* of include/mineral_ppo.h
enum { CH_LN_SILU = 0, CH_PLAIN_ELU = 1, CH_LN_ELU = 2, CH_PLAIN_SILU = 3, CH_LN_RELU = 4, CH_PLAIN_RELU = 5, CH_LN_TANH = 6, CH_PLAIN_TANH = 7, CH_NKINDS = 8 };
enum { CH_ACT_SILU = 0, CH_ACT_ELU = 1, CH_ACT_RELU = 2, CH_ACT_TANH = 3 };
constexpr bool ch_kind_norm(int kind) { return kind == CH_LN_SILU || kind == CH_LN_ELU || kind == CH_LN_RELU || kind == CH_LN_TANH; }
constexpr int ch_kind_act(int kind) {
  return (kind == CH_LN_SILU || kind == CH_PLAIN_SILU) ? CH_ACT_SILU
       : (kind == CH_LN_ELU || kind == CH_PLAIN_ELU) ? CH_ACT_ELU
       : (kind == CH_LN_RELU || kind == CH_PLAIN_RELU) ? CH_ACT_RELU : CH_ACT_TANH;
}

struct ChainFwdP {
  const float* x;              // [M][in_dim] rows of the first layer's input, row stride in_dim
  const float* mean;           // running mean / std of the observations (fp32) or null: no input normalisation
  const float* stdv;
  float lo, hi;
  int M, in_dim, n_layers;
  int units[CH_MAXL];
  int ld;                      // LDS row stride in floats: a multiple of 64, at least every width (in_dim rounded up to 32)
  int rows_written;            // rows of x-hat / rstd that must end up finite (the padded row count of the workspace)
  float ln_eps;
  const float* W[2][CH_MAXL];  // weights re-blocked by reblock_kernel (forward order, see chain_fwd_layer)
  const float* b[2][CH_MAXL];
  const float* g[2][CH_MAXL];
  const float* be[2][CH_MAXL];
  float* xhat[2][CH_MAXL];     // [rows_pad][units] normalised pre-activation (plain towers: the pre-activation z)
  float* rstd[2][CH_MAXL];     // [rows_pad]
  float* hact[2][CH_MAXL];     // [rows_pad][units] the layer's output after the activation (operand of the weight gradients), or null
  float* xnorm;                // [rows_pad][in_dim] the normalised input rows (written by the blocks of the first tower), or null
  int natural;                 // W points at the row-major parameter tensors themselves, not at re-blocked copies
};

struct ChainBwdP {
  int M, n_layers, ld;
  int units[CH_MAXL];
  const float* seeds[2];       // [M][32] d loss / d head outputs, zero padded
  const float* Wn[2][CH_MAXL]; // the matrix that carries dZ of the layer above down to layer l (W[l+1], the head for the last layer), re-blocked
                               // by reblock_kernel in backward order (see chain_bwd_layer), rows past the valid ones zero
  const float* xhat[2][CH_MAXL];
  const float* rstd[2][CH_MAXL];
  const float* g[2][CH_MAXL];
  const float* be[2][CH_MAXL];
  float* dz[2][CH_MAXL];       // [rows_pad][units] d loss / d pre-activation
  float* dy[2][CH_MAXL];       // [rows_pad][units] d loss / d LayerNorm output (its column sums are d beta, with x-hat d gamma); null for plain towers
  int natural;                 // Wn points at the row-major parameter tensors themselves ([wn_rows][units]), not at re-blocked copies
  float* colblk[2][CH_MAXL];   // [blocks][3][units] column sums of the block's 16 rows: d gamma (dY * x-hat), d beta (dY), d bias (dZ); or null:
                               // then dY goes to memory (dy) and the weight-gradient launch sums the columns
  int wn_rows[2][CH_MAXL];     // rows of Wn (the width of the layer above; the head's outputs for the last layer)
};

// Activations live in LDS as [16 rows][ld] with the 16-byte chunks of a row XOR-ed with the row number: the MFMA operand
// reads (lane = row + 16 * k-group, one 16-byte chunk each) then touch 16 different 4-bank slots per ds_read_b128 lane group.
__device__ __forceinline__ int ch_off(int row, int chunk, int ld) { return row * ld + ((chunk ^ (row & 15)) << 2); }

// the activations of mineral/nets/mlp.py:5-22 and their derivatives at the pre-activation y
template <int ACT>
__device__ __forceinline__ float ch_act(float y) {
  if constexpr (ACT == CH_ACT_ELU) return y > 0.f ? y : __builtin_amdgcn_exp2f(1.44269504088896340736f * y) - 1.0f;
  else if constexpr (ACT == CH_ACT_RELU) return fmaxf(y, 0.f);
  else if constexpr (ACT == CH_ACT_TANH) {   // 1 - 2 / (1 + e^{2|y|}), sign restored
    const float e = __builtin_amdgcn_exp2f(2.88539008177792681472f * fabsf(y));
    return copysignf(1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e), y);
  } else return y * sigmoid_fast(y);
}
template <int ACT>
__device__ __forceinline__ float ch_dact(float y) {
  if constexpr (ACT == CH_ACT_ELU) return y > 0.f ? 1.0f : __builtin_amdgcn_exp2f(1.44269504088896340736f * y);
  else if constexpr (ACT == CH_ACT_RELU) return y > 0.f ? 1.0f : 0.0f;
  else if constexpr (ACT == CH_ACT_TANH) {
    const float th = ch_act<CH_ACT_TANH>(y);
    return 1.0f - th * th;
  } else {
    const float sg = sigmoid_fast(y);
    return sg * (1.0f + y * (1.0f - sg));
  }
}

__device__ __forceinline__ f32x4 ch_load16(rsrc_t rs, int voff, int soff) {
  if constexpr (CH_ABL & 1) return f32x4{(float)voff, 1.0f, (float)soff, 2.0f};
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
__device__ __forceinline__ f32x4 ch_mfma(float a, float b, f32x4 c) {
  if constexpr (CH_ABL & 2) {
    asm volatile("" ::"v"(a), "v"(b));
    return c;
  }
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// A [16][N] tile of the LDS image to rows m0 .. m0+15 of a row-major [rows][N] matrix: consecutive threads write consecutive
// 16-byte pieces (the MFMA output layout would scatter 16-byte pieces over 16 rows per store instruction).
template <int N>
__device__ __forceinline__ void ch_copy_out(int tid, const float* tile, int ld, float* dst, int m0, int rows_ok) {
  constexpr int CPR = N / 4;
#pragma unroll
  for (int c = tid; c < CH_ROWS * CPR; c += 64 * CH_WAVES) {
    const int r = c / CPR, ch = c % CPR;
    const f32x4 v = *reinterpret_cast<const f32x4*>(tile + ch_off(r, ch, ld));
    if (m0 + r < rows_ok) *reinterpret_cast<f32x4*>(dst + (int64_t)(m0 + r) * N + ch * 4) = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// Product orientation: D[feature][batch row] = W[feature][k] * x[row][k], so a lane holds ONE batch row (lane & 15) and 4
// consecutive features per 16-feature tile: LayerNorm statistics start as in-lane sums, stores are 16 bytes wide.
// Weights come re-blocked (reblock_kernel): per 16-feature tile and 32-k step two runs of 1 KB, [half h][k-group g][feature r][4 k]
// = W[16 tile + r][32 step + 16 h + 4 g + (0..3)], so that a wave's load is 64 consecutive 16-byte pieces (one address register,
// a scalar offset per tile and step).  Element j of a load is the operand of MFMA j; its k index is the one of element j of the
// activation chunk 8 step + 4 h + g read from LDS.  A register group is 8 loads = 32 MFMAs per wave, whatever the layer width.
// ---------------------------------------------------------------------------------------------------------------------
// where a lane finds the weights of (tile, step, half): byte offset voff + base + tile * tile_stride + step * step_stride + half * half_off
struct ChFwdAddr { int voff, base, tile_stride, step_stride, half_off, bytes; };
__device__ __forceinline__ ChFwdAddr chain_fwd_addr(bool natural, int N, int K, int tile0, int lane) {
  const int steps = (K + 31) >> 5;
  ChFwdAddr a;
  if (natural) {   // W[N][K] row-major: lane (r, g) reads W[16 tile + r][32 step + 16 half + 4 g ..]
    a.voff = ((lane & 15) * K + 4 * (lane >> 4)) * 4; a.tile_stride = 16 * K * 4; a.step_stride = 128; a.half_off = 64; a.bytes = N * K * 4;
  } else {
    a.voff = lane * 16; a.tile_stride = steps * 2048; a.step_stride = 2048; a.half_off = 1024; a.bytes = N * steps * 32 * 4;
  }
  a.base = tile0 * a.tile_stride;
  return a;
}
template <int TPW>
__device__ __forceinline__ void chain_fwd_load_group(rsrc_t rsW, const ChFwdAddr& a, int gi, f32x4 (&w)[8]) {
  constexpr int U = 4 / TPW;
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int soff = a.base + i * a.tile_stride + (gi * U + u) * a.step_stride;
      w[(u * TPW + i) * 2] = ch_load16(rsW, a.voff, soff);
      w[(u * TPW + i) * 2 + 1] = ch_load16(rsW, a.voff, soff + a.half_off);
    }
}

// first register group of layer l (input width K) into w0: issued one layer ahead, so that the loads fly during the epilogue
__device__ __forceinline__ void chain_fwd_prefetch(const ChainFwdP& p, int tw, int l, int K, f32x4 (&w0)[8]) {
  const int N = p.units[l], ntile = N >> 4;
  const int tpw = ntile >= CH_WAVES ? ntile / CH_WAVES : 1, nact = ntile >= CH_WAVES ? CH_WAVES : ntile;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // provably wave-uniform: it goes into scalar offsets
  if (wave >= nact) return;
  const ChFwdAddr a = chain_fwd_addr(p.natural != 0, N, K, wave * tpw, lane);
  const rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W[tw][l]), 0, a.bytes, 0x00020000);
  if (tpw == 4) chain_fwd_load_group<4>(rsW, a, 0, w0);
  else if (tpw == 2) chain_fwd_load_group<2>(rsW, a, 0, w0);
  else chain_fwd_load_group<1>(rsW, a, 0, w0);
}

// One layer of N outputs on the block's 16 rows.  xin [16][ld] (K valid columns, zero up to the next multiple of 32); hout
// [16][ld] receives act(LayerNorm(x W^T + b)) unless this is the last layer.  wpre: in, the first register group of this layer;
// out, that of the next one.
template <int N, int KIND>
__device__ __forceinline__ void chain_fwd_layer(const ChainFwdP& p, int tw, int l, int K, float* xin, float* hout, float* red, int m0,
                                                f32x4 (&wpre0)[8]) {
  constexpr int NTILE = N / 16;
  constexpr int TPW = NTILE >= CH_WAVES ? NTILE / CH_WAVES : 1;     // 16-feature tiles per wave
  constexpr int NACT = NTILE >= CH_WAVES ? CH_WAVES : NTILE;        // waves that own tiles
  constexpr int U = 4 / TPW;                                        // 32-k steps per register group
  constexpr bool HAS_NORM = ch_kind_norm(KIND);
  constexpr int ACT = ch_kind_act(KIND);
  // the lane's indices are rebuilt here, behind an opaque copy of the thread id: hoisted to kernel entry, the address arithmetic of
  // all five layer widths would be live through the whole kernel and spill
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, g = lane >> 4;
  const bool active = wave < NACT;
  const bool last = l == p.n_layers - 1;
  const int nw0 = wave * TPW * 16;                                  // first feature of this wave
  const int ld = p.ld;
  f32x4 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this lane's epilogue operands: requested before the main loop, needed after it
  f32x4 b4[TPW], g4[TPW], be4[TPW];
  if (active) {
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int c = nw0 + 16 * i + 4 * g;
      b4[i] = *reinterpret_cast<const f32x4*>(p.b[tw][l] + c);
      if constexpr (HAS_NORM) {
        g4[i] = *reinterpret_cast<const f32x4*>(p.g[tw][l] + c);
        be4[i] = *reinterpret_cast<const f32x4*>(p.be[tw][l] + c);
      }
    }
    const int steps = (K + 31) >> 5;
    const ChFwdAddr wad = chain_fwd_addr(p.natural != 0, N, K, wave * TPW, lane);
    const rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W[tw][l]), 0, wad.bytes, 0x00020000);
    f32x4 (&wa)[8] = wpre0;   // the group prefetched during the layer before
    f32x4 wb[8];
    const int ngroups = (steps + U - 1) / U;
    if (ngroups > 1) chain_fwd_load_group<TPW>(rsW, wad, 1, wb);
    auto compute_group = [&](int gi, const f32x4 (&w)[8]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int step = gi * U + u;
        if (step < steps) {
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(xin + ch_off(r16, step * 8 + g, ld));
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(xin + ch_off(r16, step * 8 + 4 + g, ld));
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < TPW; ++i) acc[i] = ch_mfma(w[(u * TPW + i) * 2][j], x0[j], acc[i]);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < TPW; ++i) acc[i] = ch_mfma(w[(u * TPW + i) * 2 + 1][j], x1[j], acc[i]);
        }
      }
    };
    // two register groups in turn: a group is reloaded (two groups ahead) as soon as its MFMAs have been issued
    for (int gi = 0; gi < ngroups; gi += 2) {
      compute_group(gi, wa);
      if (gi + 2 < ngroups) chain_fwd_load_group<TPW>(rsW, wad, gi + 2, wa);
      if (gi + 1 < ngroups) {
        compute_group(gi + 1, wb);
        if (gi + 3 < ngroups) chain_fwd_load_group<TPW>(rsW, wad, gi + 3, wb);
      }
    }
  }
  CH_STAMP(4 * l + 1);
  if (!last) chain_fwd_prefetch(p, tw, l + 1, N, wpre0);
  if constexpr (CH_ABL & 4) {
#pragma unroll
    for (int i = 0; i < TPW; ++i) asm volatile("" ::"v"(acc[i]));
    __syncthreads();
    return;
  }

  // ---- epilogue: bias, LayerNorm over the row, x-hat / rstd to memory, activation to LDS -------------------------------
  // lane (r16, g) holds row r16, features nw0 + 16 i + 4 g + (0..3) of tile i
  constexpr int NL = 4 * TPW;                                       // features per lane
  float z[TPW][4];
  const int row = m0 + r16;
  float rstd = 1.0f, mean = 0.f;
  if (active) {
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) z[i][e] = acc[i][e] + b4[i][e];
  }
  if constexpr (HAS_NORM) {
    // Every lane reduces its own NL features to (mean, sum of squared deviations); the 4 * NACT lane groups of a row are then
    // combined exactly (equal counts): no cancellation anywhere, one barrier.
    if (active) {
      float sl = 0.f;
#pragma unroll
      for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) sl += z[i][e];
      const float ml = sl * (1.0f / (float)NL);
      float ql = 0.f;
#pragma unroll
      for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) ql += (z[i][e] - ml) * (z[i][e] - ml);
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<f32x2*>(red + r16 * CH_RS + (wave * 4 + g) * 2) = f32x2{ml, ql};
    }
    __syncthreads();
    CH_STAMP(4 * l + 2);
    constexpr int G = 4 * NACT;                                     // lane groups per row
    float ms = 0.f, qs = 0.f;
#pragma unroll
    for (int k = 0; k < G / 2; ++k) {
      const f32x4 st = *reinterpret_cast<const f32x4*>(red + r16 * CH_RS + 4 * k);
      ms += st[0] + st[2];
      qs += st[1] + st[3];
    }
    mean = ms * (1.0f / (float)G);
    float dev = 0.f;
#pragma unroll
    for (int k = 0; k < G / 2; ++k) {
      const f32x4 st = *reinterpret_cast<const f32x4*>(red + r16 * CH_RS + 4 * k);
      const float d0 = st[0] - mean, d1 = st[2] - mean;
      dev += d0 * d0 + d1 * d1;
    }
    const float var = (qs + (float)NL * dev) * (1.0f / (float)N);
    rstd = 1.0f / sqrtf(var + p.ln_eps);
    if (wave == 0 && g == 0 && row < p.rows_written) p.rstd[tw][l][row] = rstd;
  }
  if constexpr (!HAS_NORM) __syncthreads();   // every wave has left the main loop: the input tile may be overwritten
  if (active) {
    // x-hat goes into the input tile and the activated output into hout; both leave as whole rows after the barrier
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int c = nw0 + 16 * i + 4 * g;
      f32x4 xh, h;
      if constexpr (HAS_NORM) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[e] = (z[i][e] - mean) * rstd;
          h[e] = ch_act<ACT>(xh[e] * g4[i][e] + be4[i][e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[e] = z[i][e];
          h[e] = ch_act<ACT>(xh[e]);
        }
      }
      *reinterpret_cast<f32x4*>(xin + ch_off(r16, c >> 2, ld)) = xh;
      *reinterpret_cast<f32x4*>(hout + ch_off(r16, c >> 2, ld)) = h;
    }
  }
  CH_STAMP(4 * l + 3);
  __syncthreads();   // both tiles are complete; the exchange buffer may be reused
  CH_STAMP(4 * l + 4);
  ch_copy_out<N>(tid, xin, ld, p.xhat[tw][l], m0, p.rows_written);
  if (p.hact[tw][l]) ch_copy_out<N>(tid, hout, ld, p.hact[tw][l], m0, p.rows_written);
}

// The forward chain of the block's 16 rows.  On return `cur` is the tile with the last layer's activated output, `nxt` the one with
// its x-hat (both still being copied to memory by some waves: a barrier must come before either is overwritten).
template <int KIND>
__device__ __forceinline__ void chain_fwd_body(const ChainFwdP& p, float* red, float*& cur, float*& nxt, int tw, int m0) {
  const int tid = threadIdx.x;
  f32x4 wpre0[8];
  CH_STAMP(0);
  chain_fwd_prefetch(p, tw, 0, p.in_dim, wpre0);
  {
    // the block's 16 input rows, normalised (mineral/common/normalizers.py:65-68), zero in the columns up to the next
    // multiple of 32; rows past the batch repeat its last row so that everything downstream stays finite
    const int K0 = p.in_dim, cpr = ((K0 + 31) & ~31) >> 2;
    for (int c = tid; c < CH_ROWS * cpr; c += 64 * CH_WAVES) {
      const int r = c / cpr, ch = c - r * cpr;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ch * 4 < K0) {
        const int gr = min(m0 + r, p.M - 1);
        v = *reinterpret_cast<const f32x4*>(p.x + (int64_t)gr * K0 + ch * 4);
        if (p.mean) {
          const f32x4 mu = *reinterpret_cast<const f32x4*>(p.mean + ch * 4);
          const f32x4 sd = *reinterpret_cast<const f32x4*>(p.stdv + ch * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf((v[e] - mu[e]) / sd[e], p.lo), p.hi);
        }
      }
      *reinterpret_cast<f32x4*>(cur + ch_off(r, ch, p.ld)) = v;
      if (p.xnorm && tw == 0 && ch * 4 < K0 && m0 + r < p.rows_written) *reinterpret_cast<f32x4*>(p.xnorm + (int64_t)(m0 + r) * K0 + ch * 4) = v;
    }
  }
  __syncthreads();
  int K = p.in_dim;
  for (int l = 0; l < p.n_layers; ++l) {
    switch (p.units[l]) {
      case 512: chain_fwd_layer<512, KIND>(p, tw, l, K, cur, nxt, red, m0, wpre0); break;
      case 256: chain_fwd_layer<256, KIND>(p, tw, l, K, cur, nxt, red, m0, wpre0); break;
      case 128: chain_fwd_layer<128, KIND>(p, tw, l, K, cur, nxt, red, m0, wpre0); break;
      case 64: chain_fwd_layer<64, KIND>(p, tw, l, K, cur, nxt, red, m0, wpre0); break;
      default: chain_fwd_layer<32, KIND>(p, tw, l, K, cur, nxt, red, m0, wpre0); break;
    }
    float* t = cur; cur = nxt; nxt = t;
    K = p.units[l];
  }
}

template <int KIND>
__global__ __launch_bounds__(64 * CH_WAVES) void chain_fwd_kernel(const ChainFwdP p) {
  extern __shared__ __attribute__((aligned(16))) float ch_lds[];
  float* cur = ch_lds + CH_RED;
  float* nxt = cur + CH_ROWS * p.ld;
  chain_fwd_body<KIND>(p, ch_lds, cur, nxt, blockIdx.y, blockIdx.x * CH_ROWS);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward: dH[row][n] = sum_k dZabove[row][k] * Wn[k][n] for the N units of layer l, then the activation derivative and the
// LayerNorm backward of that layer; dZ (and dY) to memory for the weight-gradient launch, dZ to LDS for the layer below.
// The reduction index is the ROW of Wn, so a lane's load runs along the features: Wn comes re-blocked as
// [feature group of the wave][16-k step][s][k-group g][i][J] = Wn[16 step + 4 g + s][16 J wave + J i + (0..J-1)], every load of
// a wave is 64 consecutive pieces of 4 J bytes, and element jj is the A operand of accumulator jj, whose output row i is
// feature 16 J wave + J i + jj.  A lane (batch row r16, g) then holds 4 J CONSECUTIVE features from 4 J (4 wave + g): e = J r + jj.
// A register group is 32 floats = 32 MFMAs per wave, whatever the layer width.
// ---------------------------------------------------------------------------------------------------------------------
// where a lane finds its J weights of (step, s): byte offset voff + base + step * step_stride + s * s_stride
struct ChBwdAddr { int voff, base, step_stride, s_stride, bytes; };
__device__ __forceinline__ ChBwdAddr chain_bwd_addr(bool natural, int J, int N, int K, int rows, int wave, int lane) {
  const int steps = (K + 15) >> 4;
  ChBwdAddr a;
  if (natural) {   // Wn[rows][N] row-major: lane (i, g) reads Wn[16 step + 4 g + s][16 J wave + J i ..]; rows past the matrix read as zeros
    a.voff = (4 * (lane >> 4) * N + J * (lane & 15)) * 4; a.base = 16 * J * wave * 4; a.step_stride = 16 * N * 4; a.s_stride = N * 4; a.bytes = rows * N * 4;
  } else {
    a.voff = lane * J * 4; a.base = wave * steps * 4 * 64 * J * 4; a.step_stride = 4 * 64 * J * 4; a.s_stride = 64 * J * 4; a.bytes = steps * 16 * N * 4;
  }
  return a;
}
template <int J>
__device__ __forceinline__ void chain_bwd_load_group(rsrc_t rsW, const ChBwdAddr& a, int gi, float (&w)[32]) {
  constexpr int U = 8 / J;
  const int voff = a.voff;
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int soff = a.base + (gi * U + u) * a.step_stride + s * a.s_stride;
      if constexpr (J == 4) {
        const f32x4 t = ch_load16(rsW, voff, soff);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) w[(u * 4 + s) * 4 + jj] = t[jj];
      } else if constexpr (J == 2) {
        typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
        if constexpr (CH_ABL & 1) { w[(u * 4 + s) * 2] = (float)soff; w[(u * 4 + s) * 2 + 1] = 1.0f; }
        else {
          const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(rsW, voff, soff, 0);
          w[(u * 4 + s) * 2] = __uint_as_float(t[0]);
          w[(u * 4 + s) * 2 + 1] = __uint_as_float(t[1]);
        }
      } else {
        if constexpr (CH_ABL & 1) w[u * 4 + s] = (float)soff;
        else w[u * 4 + s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsW, voff, soff, 0));
      }
    }
}

__device__ __forceinline__ void chain_bwd_prefetch(const ChainBwdP& p, int tw, int l, int K, float (&w0)[32]) {
  const int N = p.units[l];
  const int J = N >= 512 ? 4 : (N >= 256 ? 2 : 1);
  const int nact = N / (16 * J) < CH_WAVES ? N / (16 * J) : CH_WAVES;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // provably wave-uniform: it goes into scalar offsets
  if (wave >= nact) return;
  const ChBwdAddr a = chain_bwd_addr(p.natural != 0, J, N, K, p.wn_rows[tw][l], wave, lane);
  const rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Wn[tw][l]), 0, a.bytes, 0x00020000);
  if (J == 4) chain_bwd_load_group<4>(rsW, a, 0, w0);
  else if (J == 2) chain_bwd_load_group<2>(rsW, a, 0, w0);
  else chain_bwd_load_group<1>(rsW, a, 0, w0);
}

template <int N, int KIND>
__device__ __forceinline__ void chain_bwd_layer(const ChainBwdP& p, int tw, int l, int K, float* zin, float* zout, float* red, int m0,
                                                float (&wpre0)[32], float* ptile, float* colred) {
  constexpr int J = N >= 512 ? 4 : (N >= 256 ? 2 : 1);              // features per lane and load
  constexpr int GW = 16 * J;                                        // features per wave
  constexpr int NACT = N / GW < CH_WAVES ? N / GW : CH_WAVES;
  static_assert(N / GW <= CH_WAVES, "one feature group per wave");
  constexpr int U = 8 / J;                                          // 16-k steps per register group
  constexpr int V = 4 * J;                                          // features per lane
  constexpr bool HAS_NORM = ch_kind_norm(KIND);
  constexpr int ACT = ch_kind_act(KIND);
  int tid = threadIdx.x;   // opaque copy: see chain_fwd_layer
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, g = lane >> 4;
  const bool active = wave < NACT;
  const bool first = l == 0;
  const int ld = p.ld;
  f32x4 acc[J];
#pragma unroll
  for (int jj = 0; jj < J; ++jj) acc[jj] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this lane's epilogue operands (its V consecutive features of row r16): requested before the main loop
  const int c0 = V * (4 * wave + g);           // first of this lane's V consecutive features
  const int row = m0 + r16;
  f32x4 x4[J], g4[J], be4[J];
  float rstd = 1.0f;
  if (active) {
    const float* xr = p.xhat[tw][l] + (int64_t)row * N + c0;
#pragma unroll
    for (int q = 0; q < J; ++q) {
      x4[q] = *reinterpret_cast<const f32x4*>(xr + 4 * q);
      if constexpr (HAS_NORM) {
        g4[q] = *reinterpret_cast<const f32x4*>(p.g[tw][l] + c0 + 4 * q);
        be4[q] = *reinterpret_cast<const f32x4*>(p.be[tw][l] + c0 + 4 * q);
      }
    }
    if constexpr (HAS_NORM) rstd = p.rstd[tw][l][row];
    const int steps = (K + 15) >> 4;
    const ChBwdAddr wad = chain_bwd_addr(p.natural != 0, J, N, K, p.wn_rows[tw][l], wave, lane);
    const rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Wn[tw][l]), 0, wad.bytes, 0x00020000);
    float (&wa)[32] = wpre0;   // the group prefetched during the layer before
    float wb[32];
    const int ngroups = (steps + U - 1) / U;
    if (ngroups > 1) chain_bwd_load_group<J>(rsW, wad, 1, wb);
    auto compute_group = [&](int gi, const float (&w)[32]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int step = gi * U + u;
        if (step < steps) {
          const f32x4 zf = *reinterpret_cast<const f32x4*>(zin + ch_off(r16, step * 4 + g, ld));
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int jj = 0; jj < J; ++jj) acc[jj] = ch_mfma(w[(u * 4 + s) * J + jj], zf[s], acc[jj]);
        }
      }
    };
    for (int gi = 0; gi < ngroups; gi += 2) {
      compute_group(gi, wa);
      if (gi + 2 < ngroups) chain_bwd_load_group<J>(rsW, wad, gi + 2, wa);
      if (gi + 1 < ngroups) {
        compute_group(gi + 1, wb);
        if (gi + 3 < ngroups) chain_bwd_load_group<J>(rsW, wad, gi + 3, wb);
      }
    }
  }
  CH_STAMP(4 * (p.n_layers - 1 - l) + 1);
  if (!first) chain_bwd_prefetch(p, tw, l - 1, N, wpre0);
  if constexpr (CH_ABL & 4) {
#pragma unroll
    for (int jj = 0; jj < J; ++jj) asm volatile("" ::"v"(acc[jj]));
    __syncthreads();
    return;
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------------
  float dxh[V], xh[V], dyv[V];
  if (active) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < J; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int e = 4 * q + t;                               // feature c0 + e  =  accumulator e % J, register e / J
        const float dh = acc[e % J][e / J];
        const float x = x4[q][t];
        if constexpr (HAS_NORM) {
          const float dy = dh * ch_dact<ACT>(x * g4[q][t] + be4[q][t]);
          xh[e] = x;
          dyv[e] = dy;
          dxh[e] = dy * g4[q][t];
          s1 += dxh[e];
          s2 += dxh[e] * x;
        } else {
          dxh[e] = dh * ch_dact<ACT>(x);                       // plain towers: dZ = dH * act'(z)
        }
      }
    if constexpr (HAS_NORM) {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<f32x2*>(red + r16 * CH_RS + (wave * 4 + g) * 2) = f32x2{s1, s2};
    }
  }
  if constexpr (HAS_NORM) {
    __syncthreads();
    CH_STAMP(4 * (p.n_layers - 1 - l) + 2);
    constexpr int G = 4 * NACT;
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int k = 0; k < G / 2; ++k) {
      const f32x4 st = *reinterpret_cast<const f32x4*>(red + r16 * CH_RS + 4 * k);
      t1 += st[0] + st[2];
      t2 += st[1] + st[3];
    }
    const float m1 = t1 * (1.0f / (float)N), m2 = t2 * (1.0f / (float)N);
    if (active) {
#pragma unroll
      for (int e = 0; e < V; ++e) dxh[e] = rstd * (dxh[e] - m1 - xh[e] * m2);
    }
  }
  if (active) {
    // dxh now holds dZ of this layer: into zout (the operand of the layer below); dY into the input tile, which every wave has
    // finished reading (the barrier above).  Both leave as whole rows after the next barrier.
#pragma unroll
    for (int q = 0; q < J; ++q) {
      *reinterpret_cast<f32x4*>(zout + ch_off(r16, (c0 >> 2) + q, ld)) = f32x4{dxh[4 * q], dxh[4 * q + 1], dxh[4 * q + 2], dxh[4 * q + 3]};
      if constexpr (HAS_NORM) {
        *reinterpret_cast<f32x4*>(zin + ch_off(r16, (c0 >> 2) + q, ld)) = f32x4{dyv[4 * q], dyv[4 * q + 1], dyv[4 * q + 2], dyv[4 * q + 3]};
        if (p.colblk[tw][l])
          *reinterpret_cast<f32x4*>(ptile + ch_off(r16, (c0 >> 2) + q, ld)) =
              f32x4{dyv[4 * q] * xh[4 * q], dyv[4 * q + 1] * xh[4 * q + 1], dyv[4 * q + 2] * xh[4 * q + 2], dyv[4 * q + 3] * xh[4 * q + 3]};
      }
    }
  }
  CH_STAMP(4 * (p.n_layers - 1 - l) + 3);
  __syncthreads();   // the tiles are complete, the exchange buffer reusable
  CH_STAMP(4 * (p.n_layers - 1 - l) + 4);
  if (!p.colblk[tw][l]) {
    ch_copy_out<N>(tid, zout, ld, p.dz[tw][l], m0, 0x7fffffff);
    if constexpr (HAS_NORM) ch_copy_out<N>(tid, zin, ld, p.dy[tw][l], m0, 0x7fffffff);
    return;
  }
  // dZ to memory as whole rows, and the column sums of the block's 16 rows on the way: a thread's pieces all lie in one 4-column
  // chunk (512 is a multiple of the chunks per row), so it sums them in registers; the threads that share a chunk are combined
  // in a fixed order through LDS
  constexpr int CPR = N / 4, PIECES = CH_ROWS * CPR, NQ = HAS_NORM ? 3 : 1;
  constexpr int TSH = (PIECES < 64 * CH_WAVES ? PIECES : 64 * CH_WAVES) / CPR;   // threads that share a chunk
  f32x4 sz = {0.f, 0.f, 0.f, 0.f}, sb = sz, sg = sz;
#pragma unroll
  for (int c = tid; c < PIECES; c += 64 * CH_WAVES) {
    const int r = c / CPR, ch = c % CPR;
    const f32x4 vz = *reinterpret_cast<const f32x4*>(zout + ch_off(r, ch, ld));
    *reinterpret_cast<f32x4*>(p.dz[tw][l] + (int64_t)(m0 + r) * N + ch * 4) = vz;
    sz += vz;
    if constexpr (HAS_NORM) {
      sb += *reinterpret_cast<const f32x4*>(zin + ch_off(r, ch, ld));
      sg += *reinterpret_cast<const f32x4*>(ptile + ch_off(r, ch, ld));
    }
  }
  f32x4* cr = reinterpret_cast<f32x4*>(colred);   // [NQ][512]
  if (tid < PIECES) {
    cr[tid] = sz;
    if constexpr (HAS_NORM) { cr[512 + tid] = sb; cr[1024 + tid] = sg; }
  }
  __syncthreads();
  if (tid < NQ * CPR) {
    const int q = tid / CPR, ch = tid - q * CPR;
    f32x4 s = cr[q * 512 + ch];
#pragma unroll
    for (int k = 1; k < TSH; ++k) s += cr[q * 512 + k * CPR + ch];
    const int slot = q == 0 ? 2 : (q == 1 ? 1 : 0);   // order of the partials: d gamma, d beta, d bias
    *reinterpret_cast<f32x4*>(p.colblk[tw][l] + ((int64_t)blockIdx.x * 3 + slot) * N + ch * 4) = s;
  }
}

// The backward chain of the block's 16 rows from the head seeds in tile `cur` ([16][32], swizzled like every tile); wpre0 holds the
// first register group of the last layer (chain_bwd_prefetch).
template <int KIND>
__device__ __forceinline__ void chain_bwd_body(const ChainBwdP& p, float* red, float* cur, float* nxt, int tw, int m0, float (&wpre0)[32], float* ptile,
                                               float* colred) {
  int K = 32;
  for (int l = p.n_layers - 1; l >= 0; --l) {
    switch (p.units[l]) {
      case 512: chain_bwd_layer<512, KIND>(p, tw, l, K, cur, nxt, red, m0, wpre0, ptile, colred); break;
      case 256: chain_bwd_layer<256, KIND>(p, tw, l, K, cur, nxt, red, m0, wpre0, ptile, colred); break;
      case 128: chain_bwd_layer<128, KIND>(p, tw, l, K, cur, nxt, red, m0, wpre0, ptile, colred); break;
      case 64: chain_bwd_layer<64, KIND>(p, tw, l, K, cur, nxt, red, m0, wpre0, ptile, colred); break;
      default: chain_bwd_layer<32, KIND>(p, tw, l, K, cur, nxt, red, m0, wpre0, ptile, colred); break;
    }
    float* t = cur; cur = nxt; nxt = t;
    K = p.units[l];
  }
}

template <int KIND>
__global__ __launch_bounds__(64 * CH_WAVES) void chain_bwd_kernel(const ChainBwdP p) {
  extern __shared__ __attribute__((aligned(16))) float ch_lds[];
  float* red = ch_lds;
  float* cur = ch_lds + CH_RED;
  float* nxt = cur + CH_ROWS * p.ld;
  const int tid = threadIdx.x, tw = blockIdx.y, m0 = blockIdx.x * CH_ROWS;
  float wpre0[32];
  CH_STAMP(0);
  chain_bwd_prefetch(p, tw, p.n_layers - 1, 32, wpre0);
  if (tid < CH_ROWS * 8) {
    // head seeds of the block's rows: [16][32]; rows past the batch are zero (their gradients vanish)
    const int r = tid >> 3, ch = tid & 7;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (m0 + r < p.M) v = *reinterpret_cast<const f32x4*>(p.seeds[tw] + (int64_t)(m0 + r) * 32 + ch * 4);
    *reinterpret_cast<f32x4*>(cur + ch_off(r, ch, p.ld)) = v;
  }
  __syncthreads();
  float* ptile = nxt + CH_ROWS * p.ld;   // dY * x-hat of the layer, then the exchange of the column sums (only with colblk)
  chain_bwd_body<KIND>(p, red, cur, nxt, tw, m0, wpre0, ptile, ptile + CH_ROWS * p.ld);
}

// ---------------------------------------------------------------------------------------------------------------------
// PPO minibatch step: forward chain, heads with the losses and their derivatives, backward chain -- one launch.  Every step is
// local to a batch row, so the block that owns 16 rows never has to leave its CU between the passes: the last layer's output
// is still in LDS when the heads multiply it, and the seeds d loss / d head go to the backward chain through LDS.
// (mineral/agents/ppo/ppo.py:277-303 on mineral/agents/ppo/models.py:93-107.)
// ---------------------------------------------------------------------------------------------------------------------
struct ChainPpoP {
  ChainFwdP f;
  ChainBwdP b;
  HeadsP h;                    // metric_part [blocks][2][MRL_NMETRIC] is per 16-row block here; headb_part is not written (the weight-gradient
                               // launch sums the seed columns)
  const float* Wh[2];          // head weights re-blocked in forward order as a [32][U] matrix (rows past the head's outputs zero)
  const float* bh[2];
  int head_n[2];               // head outputs per tower (2 A or A for the actor, 1 for the critic)
};

constexpr int CH_TS = 33;                                                       // row stride of the head-output tile
constexpr int CH_HEADS_F = CH_ROWS * CH_TS + 16 * MRL_NMETRIC * 2;    // floats of LDS behind the two activation tiles
constexpr int CH_COLRED_F = 3 * 512 * 4;                              // floats of the column-sum exchange of the backward chain

// head pre-activations of the block's rows: T[row][col] = h[row] . Wh[col] + bias[col], 32 columns (zero weights past the head)
__device__ __forceinline__ void chain_heads_gemm(const float* WhB, bool natural, const float* bias, int Nv, int K, const float* hin, int ld, float* T) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, g = lane >> 4;
  if (wave >= 2) return;
  const int steps = K >> 5;   // layer widths are multiples of 32
  const ChFwdAddr a = chain_fwd_addr(natural, natural ? Nv : 32, K, wave, lane);   // natural: rows past the head's outputs read as zeros
  const rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(WhB), 0, a.bytes, 0x00020000);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < steps; s0 += 4) {
    f32x4 w[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int soff = a.base + min(s0 + u, steps - 1) * a.step_stride;
      w[2 * u] = ch_load16(rsW, a.voff, soff);
      w[2 * u + 1] = ch_load16(rsW, a.voff, soff + a.half_off);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int step = s0 + u;
      if (step < steps) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(hin + ch_off(r16, step * 8 + g, ld));
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(hin + ch_off(r16, step * 8 + 4 + g, ld));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = ch_mfma(w[2 * u][j], x0[j], acc);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = ch_mfma(w[2 * u + 1][j], x1[j], acc);
      }
    }
  }
  const int c = 16 * wave + 4 * g;
#pragma unroll
  for (int e = 0; e < 4; ++e) T[r16 * CH_TS + c + e] = acc[e] + (c + e < Nv ? bias[c + e] : 0.f);
}

// Losses, metrics and seeds of the block's 16 rows from the head outputs T.  Actor blocks: wave w < 4 takes rows 4 w .. 4 w + 3,
// one per 16-lane group, lane j of the group action j.  Critic blocks: lane l < 16 of wave 0 takes row l.  The seed rows go to the
// LDS tile `seedt` (the input of the backward chain; rows past the batch zero) and to memory (the heads' weight and bias gradients).
__device__ __forceinline__ void chain_heads_train(const HeadsP& hp, int tw, int m0, int M, const float* T, float* seedt, int ld, double* red2) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int A = hp.act_dim;
  auto fexp = [](float x) { return __builtin_amdgcn_exp2f(1.44269504088896340736f * x); };
  auto flog = [](float x) { return 0.69314718055994530942f * __builtin_amdgcn_logf(x); };
  auto frcp = [](float x) { return __builtin_amdgcn_rcpf(x); };
  auto ftanh = [&](float x) {
    const float e = __builtin_amdgcn_exp2f(2.88539008177792681472f * fabsf(x));
    return copysignf(1.0f - 2.0f * frcp(1.0f + e), x);
  };
  if (tw == 0) {
    if (wave >= 4) return;
    // mineral/agents/ppo/models.py:93-107,140-151 and mineral/agents/ppo/ppo.py:277-303
    const int g = lane >> 4, j = lane & 15, lr = wave * 4 + g, row = m0 + lr;
    const bool jv = j < A, rv = row < M;
    const int rowc = rv ? row : 0;
    const float a = jv ? hp.actions[(int64_t)rowc * A + j] : 0.f;
    const float omu = jv ? hp.mu[(int64_t)rowc * A + j] : 0.f;
    const float osg = jv ? hp.sigma[(int64_t)rowc * A + j] : 1.f;
    const float oldnl = hp.old_neglogp[rowc], adv = hp.adv[rowc];
    const float o = T[lr * CH_TS + j];
    // (two separate loads: a select between the parameter's address and the LDS tile's would become one flat load)
    const float ls_fixed = hp.fixed_logstd ? hp.fixed_logstd[jv ? j : 0] : 0.f;
    const float ls = hp.fixed_logstd ? ls_fixed : T[lr * CH_TS + ((A + j) & 31)];
    float mu, sg = 0.f, sigma;
    if (hp.dist_type == DIST_DREAMERV3) {      // mineral/nets/dist.py:42-47
      mu = ftanh(o);
      sg = sigmoid_fast(ls + 2.0f);
      sigma = (hp.maxstd - hp.minstd) * sg + hp.minstd;
    } else {                                   // mineral/nets/dist.py:32-34
      mu = o;
      sigma = fexp(ls);
    }
    if (!jv) { mu = 0.f; sigma = 1.f; }
    const float logsig = flog(sigma), rsig = frcp(sigma);
    const float d = a - mu, rvar = rsig * rsig;
    float nl = jv ? (d * d) * (0.5f * rvar) + logsig + MRL_LOG_SQRT_2PI : 0.f;
    float ent = jv ? 0.5f + MRL_LOG_SQRT_2PI + logsig : 0.f;
    float bl = 0.f, dbl = 0.f;
    if (jv) {
      if (hp.bounds_type == 0) {               // 'bound' (ppo.py:480-484)
        const float hi = fmaxf(mu - 1.1f, 0.f), lo = fminf(mu + 1.1f, 0.f);
        bl = lo * lo + hi * hi;
        dbl = 2.0f * hi + 2.0f * lo;
      } else {                                 // 'reg' (ppo.py:485-487)
        bl = mu * mu;
        dbl = 2.0f * mu;
      }
    }
    float kl = 0.f;
    if (jv) {                                  // ppo.py:492-498, p0 = new, p1 = stored
      const float dm = omu - mu;
      kl = flog(osg * rsig + 1e-5f) + (sigma * sigma + dm * dm) * (0.5f * frcp(osg * osg + 1e-5f)) - 0.5f;
    }
    nl = group_sum16(nl);
    ent = group_sum16(ent);
    bl = group_sum16(bl);
    kl = group_sum16(kl);
    const float ratio = fexp(oldnl - nl);                  // ppo.py:455
    const float lo_r = 1.0f - hp.e_clip, hi_r = 1.0f + hp.e_clip;
    float rc = fminf(fmaxf(ratio, lo_r), hi_r);
    float inr = (ratio >= lo_r && ratio <= hi_r) ? 1.0f : 0.0f;      // d clamp / d ratio
    if (hp.smooth_clamp) {                                            // ppo.py:449-450: a sigmoid between the bounds
      const float sc = frcp(1.0f + fexp((-(ratio - lo_r) * frcp(hi_r - lo_r) + 0.5f) * 4.0f));
      rc = sc * (hi_r - lo_r) + lo_r;
      inr = 4.0f * sc * (1.0f - sc);
    }
    const float x1 = -(adv * ratio), x2 = -(adv * rc);
    const float al = fmaxf(x1, x2);
    const float dratio = (x1 > x2) ? -adv : (x1 < x2) ? -adv * inr : 0.5f * (-adv) + 0.5f * (-adv * inr);
    const float dnl = hp.inv_mb * dratio * (-ratio);
    const float dmu = dnl * (-d * rvar) + hp.bounds_coef * hp.inv_mb * dbl;
    const float dsig = dnl * (-(d * d) * (rvar * rsig) + rsig) - hp.entropy_coef * hp.inv_mb * rsig;
    float s_mu, s_ls;
    if (hp.dist_type == DIST_DREAMERV3) {
      s_mu = dmu * (1.0f - mu * mu);
      s_ls = dsig * (hp.maxstd - hp.minstd) * sg * (1.0f - sg);
    } else {
      s_mu = dmu;
      s_ls = dsig * sigma;
    }
    if (!rv || !jv) { s_mu = 0.f; s_ls = 0.f; }
    auto seed_at = [&](int c) -> float& { return seedt[ch_off(lr, c >> 2, ld) + (c & 3)]; };
    if (jv) {
      seed_at(j) = s_mu;
      seed_at(A + j) = s_ls;
    }
    for (int c = 2 * A + j; c < 32; c += 16) seed_at(c) = 0.f;
    if (rv) {
      float* sd = hp.seeds[0] + (int64_t)row * 32;
      if (jv) {
        sd[j] = s_mu;
        sd[A + j] = s_ls;
        hp.mu[(int64_t)row * A + j] = mu;                  // experience.py:83-87 write-back
        hp.sigma[(int64_t)row * A + j] = sigma;
      }
      for (int c = 2 * A + j; c < 32; c += 16) sd[c] = 0.f;
    }
    const int part = wave * 4 + g;
    if (j == 0 && rv) {
      double* m = red2 + part * MRL_NMETRIC;
      m[0] = (double)al; m[1] = (double)ent; m[2] = (double)bl; m[3] = (double)kl;
      m[4] = (fabsf(ratio - 1.0f) > hp.e_clip) ? 1.0 : 0.0;
    }
  } else {
    if (wave >= 1) return;
    // critic: column 0 of the tile is the value (models.py:147-149); ppo.py:466-473
    const int lr = lane & 15, row = m0 + lr;
    const bool rv = row < M, mine = rv && lane < 16;
    const int rowc = rv ? row : 0;
    const float v = T[lr * CH_TS];
    const float ret = hp.returns[rowc], vp = hp.value_preds[rowc];
    const float e1 = v - ret;
    float cl = e1 * e1, dv = 2.0f * e1;
    if (hp.clip_value_loss) {
      const float dvp = v - vp;
      const float vc = vp + fminf(fmaxf(dvp, -hp.e_clip), hp.e_clip);
      const float e2 = vc - ret;
      const float c2v = e2 * e2;
      const float in2 = (dvp >= -hp.e_clip && dvp <= hp.e_clip) ? 1.0f : 0.0f;
      const float d2 = 2.0f * e2 * in2;
      dv = (cl > c2v) ? dv : (cl < c2v) ? d2 : 0.5f * dv + 0.5f * d2;
      cl = fmaxf(cl, c2v);
    }
    const float s_v = mine ? hp.inv_mb * 0.5f * hp.critic_coef * dv : 0.f;   // ppo.py:285
    if (lane < 16) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f}, first = {s_v, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) *reinterpret_cast<f32x4*>(seedt + ch_off(lr, ch, ld)) = ch == 0 ? first : z;
      if (rv) {
        f32x4* sd = reinterpret_cast<f32x4*>(hp.seeds[1] + (int64_t)row * 32);
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) sd[ch] = ch == 0 ? first : z;
      }
    }
    const double dr = mine ? (double)ret - (double)v : 0.0;
    double q[5] = {mine ? (double)cl : 0.0, dr, dr * dr, mine ? (double)ret : 0.0, mine ? (double)ret * ret : 0.0};
#pragma unroll
    for (int msk = 1; msk < 16; msk <<= 1) {
#pragma unroll
      for (int i = 0; i < 5; ++i) q[i] += __shfl_xor(q[i], msk);
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 5; ++i) red2[5 + i] = q[i];
    }
  }
}

template <int KIND>
__global__ __launch_bounds__(64 * CH_WAVES) void ppo_chain_kernel(const ChainPpoP p) {
  extern __shared__ __attribute__((aligned(16))) float ch_lds[];
  float* red = ch_lds;
  float* cur = ch_lds + CH_RED;
  float* nxt = cur + CH_ROWS * p.f.ld;
  float* T = nxt + CH_ROWS * p.f.ld;
  double* red2 = reinterpret_cast<double*>(T + CH_ROWS * CH_TS);   // [16 (wave, group) partials][MRL_NMETRIC]
  const int tid = threadIdx.x, tw = blockIdx.y, m0 = blockIdx.x * CH_ROWS, ld = p.f.ld;
  chain_fwd_body<KIND>(p.f, red, cur, nxt, tw, m0);
  if (m0 >= p.f.M) return;   // blocks of padding rows (the activation tensors are kept finite up to a whole 128-row tile): forward only
  float wpre0[32];
  chain_bwd_prefetch(p.b, tw, p.b.n_layers - 1, 32, wpre0);
  for (int i = tid; i < 16 * MRL_NMETRIC; i += 64 * CH_WAVES) red2[i] = 0.0;
  chain_heads_gemm(p.Wh[tw], p.f.natural != 0, p.bh[tw], p.head_n[tw], p.f.units[p.f.n_layers - 1], cur, ld, T);
  __syncthreads();   // T is complete, and every wave has finished copying the last layer's tiles to memory
  chain_heads_train(p.h, tw, m0, p.f.M, T, nxt, ld, red2);
  __syncthreads();
  if (tid < MRL_NMETRIC) {
    double s = 0.0;
    for (int w = 0; w < 16; ++w) s += red2[w * MRL_NMETRIC + tid];
    p.h.metric_part[((int64_t)blockIdx.x * 2 + tw) * MRL_NMETRIC + tid] = s;
  }
  float* ptile = reinterpret_cast<float*>(red2 + 16 * MRL_NMETRIC);
  chain_bwd_body<KIND>(p.b, red, nxt, cur, tw, m0, wpre0, ptile, ptile + CH_ROWS * ld);
}

// ---------------------------------------------------------------------------------------------------------------------
// Re-blocking of the weight matrices for the two chain kernels (the parameters change with every optimizer step, so this
// runs once per step: 1.6 MB read, 3 MB written).  One thread per 16-byte piece of the destination; the four floats of a
// piece are consecutive in the source too.
//   forward order  (kind 0) of W[N][K]:  piece ((tile * steps + step) * 2 + h) * 64 + 16 g + r  =  W[16 tile + r][32 step + 16 h + 4 g .. + 3]
//   backward order (kind 1) of Wn[rows][N], J floats per lane:  float ((((nw * steps + step) * 4 + s) * 4 + g) * 16 + i) * J + jj
//                                                               =  Wn[16 step + 4 g + s][16 J nw + J i + jj]
// Rows / columns past the matrix are zeros.
// ---------------------------------------------------------------------------------------------------------------------
struct ReblockDesc { const float* src; float* dst; int kind, N, K, rows, piece0; };   // K: columns of src (fwd); rows: valid rows of src
constexpr int RB_MAXDESC = 18;   // (4 layers x 2 orders + head) x 2 towers
// Merge of a batch's moments into the running statistics of the observations (mineral/common/normalizers.py:44-55), and the
// fp32 mean / standard deviation the kernels normalise with.  One block; the shared count is read before anyone writes it.
struct RmsMergeArgs { double *mean, *var, *count; const float *bmean, *bvar; double bcount; int n; float eps; float *mean_f, *std_f; };
__device__ __forceinline__ void rms_merge_body(const RmsMergeArgs& a) {
  const double cnt = a.count[0];
  __syncthreads();
  for (int i = threadIdx.x; i < a.n; i += blockDim.x) {
    const double m = a.mean[i], v = a.var[i];
    const double delta = (double)a.bmean[i] - m;
    const double tot = cnt + a.bcount;
    const double nm = m + delta * a.bcount / tot;
    const double m2 = v * cnt + (double)a.bvar[i] * a.bcount + delta * delta * cnt * a.bcount / tot;
    const double nv = m2 / tot;
    a.mean[i] = nm;
    a.var[i] = nv;
    if (a.mean_f) a.mean_f[i] = (float)nm;
    if (a.std_f) a.std_f[i] = sqrtf((float)nv + a.eps);
  }
  if (threadIdx.x == 0) a.count[0] = cnt + a.bcount;
}

struct ReblockP { ReblockDesc d[RB_MAXDESC]; int nd, npieces, has_merge; RmsMergeArgs merge; };

__global__ __launch_bounds__(256) void reblock_kernel(const ReblockP p) {
  if (p.has_merge && blockIdx.x == gridDim.x - 1) {   // one more block: the statistics merge of the step
    rms_merge_body(p.merge);
    return;
  }
  const int piece = blockIdx.x * 256 + threadIdx.x;
  if (piece >= p.npieces) return;
  int di = 0;
  for (int q = 1; q < p.nd; ++q)
    if (piece >= p.d[q].piece0) di = q;
  const ReblockDesc d = p.d[di];
  const int lp = piece - d.piece0;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (d.kind == 0) {
    const int steps = (d.K + 31) >> 5;
    const int r = lp & 15, g = (lp >> 4) & 3, h = (lp >> 6) & 1, ts = lp >> 7;
    const int tile = ts / steps, step = ts - tile * steps;
    const int k = 32 * step + 16 * h + 4 * g;
    if (k < d.K && 16 * tile + r < d.rows) v = *reinterpret_cast<const f32x4*>(d.src + (int64_t)(16 * tile + r) * d.K + k);
  } else {
    const int J = d.N >= 512 ? 4 : (d.N >= 256 ? 2 : 1);
    const int f = lp * 4;                       // first float of the piece
    const int ij = f % (16 * J), i = ij / J;    // jj == 0: a piece starts a lane (J == 4) or covers 2 / 4 whole lanes
    const int rest = f / (16 * J);
    const int g = rest & 3, s = (rest >> 2) & 3, ws = rest >> 4;
    const int steps = (d.K + 15) >> 4;
    const int nw = ws / steps, step = ws - nw * steps;
    const int k = 16 * step + 4 * g + s;
    if (k < d.rows) v = *reinterpret_cast<const f32x4*>(d.src + (int64_t)k * d.N + 16 * J * nw + J * i);
  }
  *reinterpret_cast<f32x4*>(d.dst + (int64_t)lp * 4) = v;
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradients: slab[split][m][n] = sum over the split's batch rows k of dZ[k][m] * f(X[k][n]), f = an operand prologue
// (the chain path passes the stored activations, PRO_NONE; the others serve callers that only have x-hat).
// A task = one 64 x 64 tile of one dW and one row range, computed by a PAIR of waves (the halves `hf` take rows
// m0 + 4 i + 2 hf + (0, 1) of the tile): lane (i, g) loads dZ[k0 + g][m0 + 4 i + 2 hf .. + 1] (8 bytes) and
// X[k0 + g][n0 + 4 i .. + 3] (16 bytes; 256 contiguous bytes per row and wave), and the 2 x 4 combinations of their elements
// feed 8 accumulators: output row m0 + 4 i + 2 hf + a, column n0 + 4 j + b.  No LDS, no barrier.  Blocks are 8 waves = 4 tasks,
// two waves per SIMD: what one wave spends on loads, address arithmetic and stores runs under the other's MFMAs.
// The tasks of the first column tile also take the column sums of their 64 features over their row range: d bias from the dZ
// they load anyway, and for LayerNorm layers d gamma = sum dY * x-hat and d beta = sum dY (COLS 1 / 2).
// ---------------------------------------------------------------------------------------------------------------------
struct WgDesc {
  const float* dz;       // [M][ldz]
  const float* xin;      // [M][in]
  const float* pro0;     // per-feature prologue vectors of xin's columns (see PRO_*), or null
  const float* pro1;
  float* slab;           // [nsplit][slab_rows][in]
  const float* dy;       // [M][u] d loss / d LayerNorm output of this layer, or null
  const float* xout;     // [M][u] x-hat of this layer (with dy)
  float* colpart;        // [nsplit][3][u] d gamma, d beta, d bias partial column sums, or null (heads)
  int u, in, ldz, slab_rows, pro;
  int tiles_n, tiles, task0;   // 64-wide column tiles, tiles of this gradient, id of its first task
};
constexpr int WG_MAXDESC = 10;   // (4 layers + heads) x 2 towers
struct WgradSmallP {
  WgDesc d[WG_MAXDESC];
  int nd, ntasks, M, nsplit, ksplit;
  float lo, hi;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 ch_load8(rsrc_t rs, int voff, int soff) {
  if constexpr (CH_ABL & 1) return f32x2{(float)voff, 1.0f};
  typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
}

template <int PRO, int COLS>
__device__ __forceinline__ void wgrad_small_task(const WgradSmallP& p, const WgDesc& d, int split, int m0, int n0, int hf) {
  const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
  const int k0 = split * p.ksplit, k1 = min(p.M, k0 + p.ksplit);
  // rows past the batch read as zeros (descriptor bounds); columns past a matrix's width wrap into the next row and only
  // feed outputs that are not stored
  const rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.dz), 0, p.M * d.ldz * 4, 0x00020000);
  const rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.xin), 0, p.M * d.in * 4, 0x00020000);
  const rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(COLS == 2 ? d.dy : d.dz), 0, p.M * d.ldz * 4, 0x00020000);
  const rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(COLS == 2 ? d.xout : d.dz), 0, p.M * d.ldz * 4, 0x00020000);
  const int voffZ = (g * d.ldz + m0 + 4 * r16 + 2 * hf) * 4, voffX = (g * d.in + n0 + 4 * r16) * 4;
  const int strideZ = d.ldz * 4 * 4, strideX = d.in * 4 * 4;   // bytes per 4 rows
  f32x4 pv0 = {0.f, 0.f, 0.f, 0.f}, pv1 = {1.f, 1.f, 1.f, 1.f};
  if constexpr (PRO == PRO_NORM || PRO == PRO_LNSILU || PRO == PRO_LNELU) {
    const int c = min(n0 + 4 * r16, last_chunk(d.in));
    pv0 = *reinterpret_cast<const f32x4*>(d.pro0 + c);
    pv1 = *reinterpret_cast<const f32x4*>(d.pro1 + c);
  }
  auto transform = [&](f32x4 v) {
    f32x4 o;
    if constexpr (PRO == PRO_NONE) return v;
    else if constexpr (PRO == PRO_NORM) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fminf(fmaxf((v[e] - pv0[e]) / pv1[e], p.lo), p.hi);
      return o;
    } else if constexpr (PRO == PRO_ELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = ch_act<CH_ACT_ELU>(v[e]);
      return o;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = ch_act<PRO == PRO_LNELU ? CH_ACT_ELU : CH_ACT_SILU>(v[e] * pv0[e] + pv1[e]);
      return o;
    }
  };

  f32x4 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x2 cz = {0.f, 0.f}, cg = cz, cb = cz;
  constexpr int U = 8;                                   // 4-row steps per register group (32 rows): 64 MFMAs per wave
  constexpr int UC = COLS == 2 ? U : 1;
  // row ranges are multiples of 32 rows, so every group is whole (rows past the batch read as zeros) and the unrolled group
  // has no branch in it
  const int ngroups = (k1 - k0 + 4 * U - 1) / (4 * U);
  f32x2 za[U], zb[U], ya[UC], yb[UC], oa[UC], ob[UC];
  f32x4 xa[U], xb[U];
  auto load_group = [&](int gi, f32x2 (&z)[U], f32x4 (&x)[U], f32x2 (&y)[UC], f32x2 (&o)[UC]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int st = (k0 >> 2) + gi * U + u;
      z[u] = ch_load8(rsZ, voffZ, st * strideZ);
      x[u] = ch_load16(rsX, voffX, st * strideX);
      if constexpr (COLS == 2) {
        y[u] = ch_load8(rsY, voffZ, st * strideZ);
        o[u] = ch_load8(rsO, voffZ, st * strideZ);
      }
    }
  };
  auto compute_group = [&](const f32x2 (&z)[U], const f32x4 (&x)[U], const f32x2 (&y)[UC], const f32x2 (&o)[UC]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const f32x2 zv = z[u];
      const f32x4 xv = transform(x[u]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = ch_mfma(zv[a], xv[b], acc[a][b]);
      if constexpr (COLS >= 1) cz += zv;
      if constexpr (COLS == 2) {
        cg += y[u] * o[u];
        cb += y[u];
      }
    }
  };
  WG_STAMP(0);
  load_group(0, za, xa, ya, oa);
  for (int gi = 0; gi < ngroups; gi += 2) {
    if (gi + 1 < ngroups) load_group(gi + 1, zb, xb, yb, ob);
    WG_STAMP(1 + 2 * gi);
    compute_group(za, xa, ya, oa);
    WG_STAMP(2 + 2 * gi);
    if (gi + 1 < ngroups) {
      if (gi + 2 < ngroups) load_group(gi + 2, za, xa, ya, oa);
      WG_STAMP(3 + 2 * gi);
      compute_group(zb, xb, yb, ob);
      WG_STAMP(4 + 2 * gi);
    }
  }
  // lane (j = r16, g) holds rows m0 + 16 g + 4 r + 2 hf + a, columns n0 + 4 j + (0..3) in acc[a][0..3][r]
  float* out = d.slab + (int64_t)split * d.slab_rows * d.in;
  const int col = n0 + 4 * r16;
  if (col < d.in) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int orow = m0 + 16 * g + 4 * r + 2 * hf + a;
        if (orow < d.slab_rows) *reinterpret_cast<f32x4*>(out + (int64_t)orow * d.in + col) = f32x4{acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
      }
  }
  WG_STAMP(40);
  if constexpr (COLS >= 1) {
    // column sums: lane (i, g) summed rows k0 + g + 4 step; the four k-groups are combined in a fixed order
    float* cp = d.colpart + (int64_t)split * 3 * d.u;
    const int c = m0 + 4 * r16 + 2 * hf;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      cz[e] = add_xor32(add_xor16(cz[e]));
      if constexpr (COLS == 2) {
        cg[e] = add_xor32(add_xor16(cg[e]));
        cb[e] = add_xor32(add_xor16(cb[e]));
      }
    }
    if (g == 0 && c < d.u) {
      *reinterpret_cast<f32x2*>(cp + 2 * d.u + c) = cz;
      if constexpr (COLS == 2) {
        *reinterpret_cast<f32x2*>(cp + c) = cg;
        *reinterpret_cast<f32x2*>(cp + d.u + c) = cb;
      }
    }
  }
}

template <int PRO>
__device__ __forceinline__ void wgrad_small_cols(const WgradSmallP& p, const WgDesc& d, int split, int m0, int n0, int hf) {
  const int cols = (n0 == 0 && d.colpart) ? (d.dy ? 2 : 1) : 0;
  if (cols == 2) wgrad_small_task<PRO, 2>(p, d, split, m0, n0, hf);
  else if (cols == 1) wgrad_small_task<PRO, 1>(p, d, split, m0, n0, hf);
  else wgrad_small_task<PRO, 0>(p, d, split, m0, n0, hf);
}

__global__ __launch_bounds__(512) void wgrad_small_kernel(const WgradSmallP p) {
  // Blocks are handed to the 8 XCDs in turn (each with an L2 of its own): block ids are remapped so that every XCD works on
  // one contiguous eighth of the task list, i.e. on tiles that read the same rows of the same matrices.  Placement only
  // decides what is found in L2; results do not depend on it.
  const int nwg = gridDim.x, xq = nwg >> 3, xr = nwg & 7, xcd = blockIdx.x & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int task = wg * 4 + (wave >> 1), hf = wave & 1;
  if (task >= p.ntasks) return;
  int di = 0;
  for (int q = 1; q < p.nd; ++q)
    if (task >= p.d[q].task0) di = q;
  const WgDesc& d = p.d[di];
  const int local = task - d.task0;
  const int split = local / d.tiles, tile = local - split * d.tiles;
  const int m0 = (tile / d.tiles_n) * 64, n0 = (tile % d.tiles_n) * 64;
  switch (d.pro) {
    case PRO_NORM: wgrad_small_cols<PRO_NORM>(p, d, split, m0, n0, hf); break;
    case PRO_LNSILU: wgrad_small_cols<PRO_LNSILU>(p, d, split, m0, n0, hf); break;
    case PRO_ELU: wgrad_small_cols<PRO_ELU>(p, d, split, m0, n0, hf); break;
    case PRO_LNELU: wgrad_small_cols<PRO_LNELU>(p, d, split, m0, n0, hf); break;
    default: wgrad_small_cols<PRO_NONE>(p, d, split, m0, n0, hf); break;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradients, block tiles: the same slabs as wgrad_small_kernel from a block of 8 waves per 128 x 128 (or 128 x 64)
// tile of one dW and one range of batch rows.  Both operands are staged through LDS 32 rows at a time (every thread loads 16
// bytes per operand and 16 rows, coalesced along the feature dimension; the loads of the stage after next fly during the
// MFMAs of this one), so a CU reads each operand element once per tile instead of once per wave, and the matrix core
// operands come from LDS: lane (i, g) takes dZ[k0 + g][m0 + MT i .. + MT-1] and X[k0 + g][n0 + 2 i, + 1] per 4-row step and
// feeds MT x 2 accumulators.  Operands as stored by the forward chain (no prologue).
// The threads that load dZ also sum what they load: column sums of dZ (d bias), and with dY / x-hat of the layer, of dY
// (d beta) and dY * x-hat (d gamma), for the tiles of the first column block.
// Row ranges are whole 32-row stages, dealt out evenly: split s of n takes stages [s G / n, (s + 1) G / n) of the G stages.
// ---------------------------------------------------------------------------------------------------------------------
struct WgTileDesc {
  const float* dz;       // [M][u]
  const float* xin;      // [M][in]
  float* slab;           // [nsplit][u][in]
  const float* dy;       // [M][u] or null
  const float* xout;     // [M][u] x-hat of this layer (with dy)
  float* colpart;        // [nsplit][3][u] d gamma, d beta, d bias partial column sums, or null
  int u, in, bm, bn;     // tile rows x columns: 128 x 128, 128 x 64 or 32 x 128
  int tiles_n, tiles, nsplit, task0;
};
// Tasks behind the tiles: the column sums the backward chain left per 16-row block ([nblk][3 u] per layer) are summed in `groups`
// groups of consecutive blocks, so that the reduction that builds the gradient vector sees a few slabs per tensor.
struct WgColDesc { const float* src; float* dst; int n4, nblk, groups, task0; };   // n4: 16-byte chunks of a slab (3 u / 4)
struct WgTileP { WgTileDesc d[WG_MAXDESC]; int nd, ntasks, M; WgColDesc cd[2 * CH_MAXL]; int ncd; };
constexpr int WT_LD = 132;                       // floats per staged row: 128 + 4, so that the four k-groups of a read start on different banks
constexpr int WT_ROWS = 32;                      // rows per stage
constexpr int WT_LDS_F = 2 * 2 * WT_ROWS * WT_LD;   // floats: two stages of two operands

// Tile shapes (rows x columns): 128 x 128 (MT 4, NT 2), 128 x 64 (MT 2, NT 2), 32 x 128 (MT 2, NT 1: the heads' 32 rows).
template <int BN, int MT, int NT>
__device__ __forceinline__ void wgrad_tile_task(const WgTileP& p, const WgTileDesc& d, int split, int m0, int n0, float* lds) {
  constexpr int WN = BN / (16 * NT), WM = 8 / WN;      // waves along the columns / rows of the tile; wave tile (16 MT) x (16 NT)
  constexpr int BM = WM * 16 * MT;
  float* As = lds;
  float* Bs = lds + 2 * WT_ROWS * WT_LD;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i16 = lane & 15, kg = lane >> 4;
  const int wm0 = (wave / WN) * (16 * MT), wn0 = (wave % WN) * (16 * NT);
  const int G = (p.M + WT_ROWS - 1) / WT_ROWS;
  const int s0 = (int)((int64_t)split * G / d.nsplit), s1 = (int)((int64_t)(split + 1) * G / d.nsplit);
  const int nst = s1 - s0, k0 = s0 * WT_ROWS;
  const int cols = (n0 == 0 && d.colpart) ? (d.dy ? 2 : 1) : 0;
  // loader role of this thread: rows lr and lr + 16 of a stage, 4 columns from col4
  const int lr = tid >> 5, col4 = (tid & 31) * 4;
  const bool loadsA = col4 < BM, loadsB = col4 < BN;
  // rows past the batch read as zeros (descriptor bounds); columns past a matrix's width wrap into the next row and only feed
  // outputs that are not stored
  const rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.dz), 0, p.M * d.u * 4, 0x00020000);
  const rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.xin), 0, p.M * d.in * 4, 0x00020000);
  const rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.dy ? d.dy : d.dz), 0, p.M * d.u * 4, 0x00020000);
  const rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.dy ? d.xout : d.dz), 0, p.M * d.u * 4, 0x00020000);
  const int voffZ = (lr * d.u + m0 + col4) * 4, voffX = (lr * d.in + n0 + col4) * 4;
  const int h16Z = 16 * d.u * 4, h16X = 16 * d.in * 4;     // bytes per 16 rows
  f32x4 acc[MT][NT];
#pragma unroll
  for (int j = 0; j < MT; ++j)
#pragma unroll
    for (int t2 = 0; t2 < NT; ++t2) acc[j][t2] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 cz = {0.f, 0.f, 0.f, 0.f}, cg = cz, cb = cz;
  // two register sets: the loads of a stage are issued two stages before its rows go to LDS
  struct Regs { f32x4 a[2], b[2], y[2], o[2]; };
  Regs r0, r1;
  auto load_stage = [&](int s, Regs& r) {
    const int sz = (k0 + s * WT_ROWS) * d.u * 4, sx = (k0 + s * WT_ROWS) * d.in * 4;
    if (loadsA) {
      r.a[0] = ch_load16(rsZ, voffZ, sz);
      r.a[1] = ch_load16(rsZ, voffZ, sz + h16Z);
    }
    if (loadsB) {
      r.b[0] = ch_load16(rsX, voffX, sx);
      r.b[1] = ch_load16(rsX, voffX, sx + h16X);
    }
    if (cols == 2) {
      r.y[0] = ch_load16(rsY, voffZ, sz);
      r.y[1] = ch_load16(rsY, voffZ, sz + h16Z);
      r.o[0] = ch_load16(rsO, voffZ, sz);
      r.o[1] = ch_load16(rsO, voffZ, sz + h16Z);
    }
  };
  auto store_stage = [&](int buf, const Regs& r) {
    if (loadsA) {
      float* a = As + buf * WT_ROWS * WT_LD + lr * WT_LD + col4;
      *reinterpret_cast<f32x4*>(a) = r.a[0];
      *reinterpret_cast<f32x4*>(a + 16 * WT_LD) = r.a[1];
    }
    if (loadsB) {
      float* b = Bs + buf * WT_ROWS * WT_LD + lr * WT_LD + col4;
      *reinterpret_cast<f32x4*>(b) = r.b[0];
      *reinterpret_cast<f32x4*>(b + 16 * WT_LD) = r.b[1];
    }
    if (cols >= 1 && loadsA) cz += r.a[0] + r.a[1];
    if (cols == 2) {
      cg += r.y[0] * r.o[0] + r.y[1] * r.o[1];
      cb += r.y[0] + r.y[1];
    }
  };
  auto compute_stage = [&](int buf) {
    const float* a = As + buf * WT_ROWS * WT_LD + kg * WT_LD + wm0 + MT * i16;
    const float* b = Bs + buf * WT_ROWS * WT_LD + kg * WT_LD + wn0 + NT * i16;
#pragma unroll
    for (int ks = 0; ks < WT_ROWS / 4; ++ks) {
      float av[MT];
      if constexpr (MT == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(a + ks * 4 * WT_LD);
#pragma unroll
        for (int j = 0; j < 4; ++j) av[j] = v[j];
      } else {
        const f32x2 v = *reinterpret_cast<const f32x2*>(a + ks * 4 * WT_LD);
        av[0] = v[0]; av[1] = v[1];
      }
      float bv[NT];
      if constexpr (NT == 2) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(b + ks * 4 * WT_LD);
        bv[0] = v[0]; bv[1] = v[1];
      } else {
        bv[0] = b[ks * 4 * WT_LD];
      }
#pragma unroll
      for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int t2 = 0; t2 < NT; ++t2) acc[j][t2] = ch_mfma(av[j], bv[t2], acc[j][t2]);
    }
  };
  WG_STAMP(0);
  WG_REAL(50);
  if (nst > 0) {
    load_stage(0, r0);
    if (nst > 1) load_stage(1, r1);
    store_stage(0, r0);
    if (nst > 2) load_stage(2, r0);
    __syncthreads();
    WG_STAMP(1);
    // stage s is multiplied from buffer s & 1 while stage s + 1 goes from registers into the other buffer (every wave left that
    // one at the barrier that ended stage s - 1) and the loads of stage s + 3 are issued; the late waves of a stage go from their
    // last MFMA straight to the barrier
    // The two waves of a SIMD take the register-to-LDS step at different ends of the stage (waves 0-3 before their MFMAs,
    // waves 4-7 after), so that one of them always has MFMAs to issue.
    const bool early = wave < 4;
    for (int s = 0; s < nst; s += 2) {
      auto refill1 = [&] {
        if (s + 1 < nst) {
          store_stage(1, r1);
          if (s + 3 < nst) load_stage(s + 3, r1);
        }
      };
      auto refill0 = [&] {
        if (s + 2 < nst) {
          store_stage(0, r0);
          if (s + 4 < nst) load_stage(s + 4, r0);
        }
      };
      if (early) refill1();
      compute_stage(0);
      if (!early) refill1();
      WG_STAMP(2 + 2 * s);
      __syncthreads();
      WG_STAMP(3 + 2 * s);
      if (s + 1 < nst) {
        if (early) refill0();
        compute_stage(1);
        if (!early) refill0();
        WG_STAMP(4 + 2 * s);
        __syncthreads();
        WG_STAMP(5 + 2 * s);
      }
    }
  }
  // lane (c = i16, kg) holds rows m0 + wm0 + MT (4 kg + r) + j, columns n0 + wn0 + NT c + (0 .. NT-1) in acc[j][..][r]
  float* out = d.slab + (int64_t)split * d.u * d.in;
  const int col = n0 + wn0 + NT * i16;
  if (col < d.in) {
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int orow = m0 + wm0 + MT * (4 * kg + r) + j;
        if (orow < d.u) {
          if constexpr (NT == 2) *reinterpret_cast<f32x2*>(out + (int64_t)orow * d.in + col) = f32x2{acc[j][0][r], acc[j][1][r]};
          else out[(int64_t)orow * d.in + col] = acc[j][0][r];
        }
      }
  }
  WG_STAMP(40);
  WG_REAL(51);
  if (cols >= 1) {
    // column sums: the 16 loader rows of a column chunk are combined in a fixed order through LDS (all waves are past the last barrier)
    f32x4* red = reinterpret_cast<f32x4*>(lds);            // [3][16][32 chunks]
    red[(0 * 16 + lr) * 32 + (tid & 31)] = cz;
    if (cols == 2) {
      red[(1 * 16 + lr) * 32 + (tid & 31)] = cg;
      red[(2 * 16 + lr) * 32 + (tid & 31)] = cb;
    }
    __syncthreads();
    const int q = tid >> 5, ch = tid & 31;                 // quantity 0 d bias, 1 d gamma, 2 d beta
    if (q < (cols == 2 ? 3 : 1) && 4 * ch < BM && m0 + 4 * ch < d.u) {
      f32x4 s = red[(q * 16) * 32 + ch];
#pragma unroll
      for (int r = 1; r < 16; ++r) s += red[(q * 16 + r) * 32 + ch];
      float* cp = d.colpart + (int64_t)split * 3 * d.u;
      const int slot = q == 0 ? 2 : (q == 1 ? 0 : 1);      // colpart order: d gamma, d beta, d bias
      *reinterpret_cast<f32x4*>(cp + slot * d.u + m0 + 4 * ch) = s;
    }
  }
}

// the tasks behind the tiles: sums of the backward chain's per-block column sums (WgColDesc); a block of 512 threads takes
// 1,024 (chunk, group) outputs, two per thread
__device__ __forceinline__ void wgrad_col_task(const WgTileP& p, int task) {
  int ci = 0;
  for (int q = 1; q < p.ncd; ++q)
    if (task >= p.cd[q].task0) ci = q;
  const WgColDesc& cdsc = p.cd[ci];
  const int per = (cdsc.nblk + cdsc.groups - 1) / cdsc.groups;
  const int64_t stride = 4 * (int64_t)cdsc.n4;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int idx = (task - cdsc.task0) * 1024 + h * 512 + threadIdx.x;
    const int grp = idx / cdsc.n4, ch = idx - grp * cdsc.n4;
    if (grp >= cdsc.groups) continue;
    const int b0 = grp * per, b1 = min(cdsc.nblk, b0 + per);
    const float* src = cdsc.src + 4 * (int64_t)ch;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int b = b0;
    for (; b + 16 <= b1; b += 16) {
      f32x4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (b + u) * stride);
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; b < b1; ++b) s += *reinterpret_cast<const f32x4*>(src + b * stride);
    *reinterpret_cast<f32x4*>(cdsc.dst + grp * stride + 4 * ch) = s;
  }
}

__global__ __launch_bounds__(512) void wgrad_tile_kernel(const WgTileP p) {
  extern __shared__ __attribute__((aligned(16))) float wt_lds[];
  WG_BLOCK(0);
  // blocks are handed to the 8 XCDs in turn: remapped so that every XCD works on one contiguous eighth of the task list (the
  // tiles of one row range read the same rows)
  const int nwg = gridDim.x, xq = nwg >> 3, xr = nwg & 7, xcd = blockIdx.x & 7;
  const int task = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  if (task >= p.ntasks) {
    wgrad_col_task(p, task);
    return;
  }
  int di = 0;
  for (int q = 1; q < p.nd; ++q)
    if (task >= p.d[q].task0) di = q;
  const WgTileDesc& d = p.d[di];
  const int local = task - d.task0;
  const int split = local / d.tiles, tile = local - split * d.tiles;
  const int m0 = (tile / d.tiles_n) * d.bm, n0 = (tile % d.tiles_n) * d.bn;
  if (d.bm == 32) wgrad_tile_task<128, 2, 1>(p, d, split, m0, n0, wt_lds);
  else if (d.bn == 128) wgrad_tile_task<128, 4, 2>(p, d, split, m0, n0, wt_lds);
  else wgrad_tile_task<64, 2, 2>(p, d, split, m0, n0, wt_lds);
  WG_BLOCK(1);
}

}  // namespace mrl
