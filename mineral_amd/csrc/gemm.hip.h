// fp32 MFMA GEMM family for the PPO actor/critic towers (gfx950 only).
//
// One kernel template covers the three products of a Linear layer
//   forward   Z[m][n]  = sum_k X[m][k]  * W[n][k]      (A: k contiguous, B: k contiguous)
//   dgrad     dX[m][k] = sum_n dZ[m][n] * W[n][k]      (A: reduction contiguous, B: reduction is the row index)
//   wgrad     dW[n][k] = sum_m dZ[m][n] * X[m][k]      (A and B: reduction is the row index)
// on v_mfma_f32_32x32x2_f32 (exact fp32 fma chains), with the element-wise work of the layer fused in:
//   A/B prologue  : running-mean/std normalisation of raw observations, or LayerNorm-affine + SiLU of the
//                   stored normalised activations (activations are kept in HBM as xhat = (z-mean)*rstd only)
//   epilogue      : bias + LayerNorm statistics (forward), SiLU'/LayerNorm backward + column partial sums
//                   (dgrad), plain partial slabs (wgrad, split over the batch), or the policy/value heads with
//                   the PPO losses and their gradients.
// Mirrors mineral/nets/mlp.py:50-77 (Linear -> LayerNorm -> SiLU) and torch autograd's backward of it.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef MRL_ABLATE_FWD
#define MRL_ABLATE_FWD 0   // diagnostic builds only: bit 0 no activation stores, 1 no split, 2 no operand prologue, 3 one step (split-bf16 forward loop)
#endif

namespace mrl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { LAY_K = 0, LAY_R = 1 };           // LAY_K: reduction index contiguous; LAY_R: reduction index = row
// PRO_ELU: plain towers (no normalisation layer), operand = ELU(stored pre-activation); PRO_LNELU: LayerNorm affine, then ELU
enum { PRO_NONE = 0, PRO_NORM = 1, PRO_LNSILU = 2, PRO_ELU = 3, PRO_LNELU = 4 };
enum { EPI_LNFWD = 0, EPI_LNBWD = 1, EPI_SLAB = 2, EPI_HEADS_TRAIN = 3, EPI_HEADS_ACT = 4, EPI_BIASFWD = 5, EPI_ACTBWD = 6 };
enum { DIST_NORMAL = 0, DIST_DREAMERV3 = 1, DIST_SQUASHED = 2 };   // DIST_SQUASHED: rollout-style sampling only (SAC)

#define MRL_LOG_SQRT_2PI 0.91893853320467274178f
#define MRL_NMETRIC 12

struct HeadsP {
  // training (EPI_HEADS_TRAIN) inputs, all indexed by minibatch row
  const float* actions;      // [M][A]
  const float* old_neglogp;  // [M]
  const float* adv;          // [M]
  const float* returns;      // [M]
  const float* value_preds;  // [M]
  float* mu;                 // [M][A] in: old mu, out: new mu   (row stride mu_rs)
  float* sigma;              // [M][A]
  float* seeds[2];           // [M][32] per tower: d loss / d head pre-activations (zero padded)
  double* metric_part;       // [gridDim.x][2][MRL_NMETRIC]
  float* headb_part;         // [gridDim.x][2][32] column sums of the seeds (head bias gradients)
  float inv_mb, e_clip, critic_coef, entropy_coef, bounds_coef;
  int bounds_type, clip_value_loss, smooth_clamp;
  // rollout (EPI_HEADS_ACT)
  const float* noise;        // [M][A] standard normal
  float* act_out;            // actions  -> act_out[row*out_rs + out_off + j]
  float* act_clamped;        // [M][A] clamp(action,-1,1) or null
  float* neglogp_out;        // [M]
  float* value_out;          // [M]
  const float* fixed_logstd; // [A] state-independent log-std parameter (models.py:66-67,151) or null: log-std from the head
  int raw_mu;                // rollout mode: store only the mu head's raw output (models.py:112-114, act(sample=False)) in `mu`
  const double* vrms_mean;   // value_rms state (fp64) or null when normalize_value is off
  const double* vrms_var;
  float vrms_eps;
  int64_t out_rs, out_off;   // row stride / offset (floats) of act_out, mu, sigma in rollout mode
  // shared
  int act_dim, dist_type;
  float minstd, maxstd;
};

struct GemmP {
  const float* A[2];
  const float* B[2];
  int64_t lda, ldb;
  int M, N, K;                 // GEMM sizes (M rows, N cols, K reduction); all loads are guarded by these
  int Nt[2], Kt[2];            // per-tower valid N / K (the heads differ: 2*act_dim vs 1); 0 = use N / K
  int tw0;                     // role of blockIdx.z == 0: 0 actor, 1 critic (critic-only launches)
  // weights pre-split into three bf16 planes (same row-major element order as B; plane q of tower z starts at
  // Bpl[z] + q * pl_stride); read instead of B by dgrad_split_kernel<true>
  const unsigned short* Bpl[2];
  int64_t pl_stride;
  const float* pro0[2];        // PRO_NORM: mean[f]   PRO_LNSILU / PRO_LNELU: gamma[f]
  const float* pro1[2];        // PRO_NORM: std[f]    PRO_LNSILU / PRO_LNELU: beta[f]
  float clamp_lo, clamp_hi;
  // EPI_SLAB launches use a 1-D grid of 8*ceil(groups/8)*group_size blocks.  A group is the set of tiles that read the
  // same operand rows (dgrad: the N tiles of one row block; wgrad: all tiles of one batch split); its blocks are given
  // ids that differ by 8 so that they land on one XCD and share its L2 (placement is a speed matter only).
  int ksplit;                  // reduction rows per split
  int m_tiles, n_tiles;        // tile grid of the output
  int n_groups, group_mode;    // group_mode 0: group = row tile (dgrad), 1: group = split (wgrad)
  int nt_hint;                 // host side only (launch_dgrad_split): 1 the streaming LayerNorm backward reads the output next, 2 the fused first-layer kernel
  // EPI_LNFWD: out = xhat [M][N], rstd [M], bias [N]
  // EPI_LNBWD: out = dz [M][N]; xhat/rstd/gamma/beta of the layer being differentiated; colpart slab
  // EPI_SLAB : out = slab [n_split][M][N]
  float* out[2];
  int64_t ldo;
  const float* bias[2];
  float* rstd[2];              // LNFWD: written; LNBWD: read
  const float* xhat[2];        // LNBWD
  const float* gamma[2];       // LNBWD
  const float* beta[2];        // LNBWD
  float* colpart[2];           // LNBWD: [gridDim.x][3][N]  (dgamma, dbeta, dbias partial column sums)
  float ln_eps;
  int act_elu;                 // LN backward: the activation after the LayerNorm is ELU (else SiLU)
  HeadsP h;
#ifdef MRL_KTRACE
  int ktrace;                  // instrumented diagnostic build only: 1 + z = block 8 of tower z records cycle stamps
  int ablate;                  // bit 0: no staging, bit 1: no fragment reads, bit 2: no barrier, bit 3: no MFMA (results are garbage)
#endif
};

#ifdef MRL_KTRACE
// Instrumented diagnostic build (tools/ktrace_run.py): cycle-counter stamps of the four waves of one block.
constexpr int KTRACE_SLOTS = 1024;
__device__ unsigned long long g_ktrace[8][KTRACE_SLOTS];   // one row per wave of the traced block (up to 8)
constexpr int KTRACE_BLOCKS = 8192;
__device__ unsigned long long g_kblock[KTRACE_BLOCKS][2];   // 100 MHz clock at entry / exit of every block
#ifdef MRL_KTRACE_ABLATE   // switches inside the main loop change its code; a build of its own keeps the stamps honest
#define MRL_ABLATE(bit) ((p.ablate >> (bit)) & 1)
#else
#define MRL_ABLATE(bit) false
#endif
#define MRL_KB(which)                                                                                    \
  do {                                                                                                   \
    const unsigned kb_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                 \
    if (p.ktrace && threadIdx.x == 0 && kb_ < KTRACE_BLOCKS) g_kblock[kb_][which] = wall_clock64();       \
  } while (0)
#define MRL_KT(slot)                                                                                     \
  do {                                                                                                   \
    if (p.ktrace && blockIdx.x == 8 && blockIdx.z == p.ktrace - 1 && (threadIdx.x & 63) == 0 && (slot) < KTRACE_SLOTS) \
      g_ktrace[threadIdx.x >> 6][(slot)] = __builtin_readcyclecounter();                                 \
  } while (0)
#else
#define MRL_KT(slot) do { } while (0)
#define MRL_ABLATE(bit) false
#define MRL_KB(which) do { } while (0)
#endif

// fast sigmoid on the hardware exp2 / rcp (about 1 ulp each); used where the result feeds fp32 GEMM operands
__device__ __forceinline__ float sigmoid_fast(float y) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * y));
}

__device__ __forceinline__ float silu_f(float y) { return y / (1.0f + expf(-y)); }
__device__ __forceinline__ float sigmoid_f(float y) { return 1.0f / (1.0f + expf(-y)); }

// Exact three-term bf16 split of fp32 values for the split-bf16 GEMMs: x = h + m + l with h = bf16(x), m = bf16(x - h),
// l = x - h - m, each rounded to nearest even (v_cvt_pk_bf16_f32 converts and packs two values at once).  x - h and
// (x - h) - m are exact in fp32 and the last remainder has at most 8 significant bits, so nothing is lost; rounding to
// nearest keeps |m| <= 2^-8 |x| and |l| <= 2^-16 |x| with either sign, so the three product terms the GEMMs drop (ml, lm,
// ll) stay below 2^-23 |a b| and do not bias the sum.  |x| must stay below 3.38e38 (beyond it h rounds to infinity).
typedef __bf16 split_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int split_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned split_pack2(float a, float b) {
  const split_bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
// the fp32 values of the two halves of a packed pair (v_perm keeps the compiler from converting the low element twice)
__device__ __forceinline__ float split_lo(unsigned pk) { return __uint_as_float(__builtin_amdgcn_perm(pk, 0u, 0x05040c0cu)); }
__device__ __forceinline__ float split_hi(unsigned pk) { return __uint_as_float(pk & 0xffff0000u); }
__device__ __forceinline__ void split_rne4(const f32x4 x, split_u32x2& h, split_u32x2& m, split_u32x2& l) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float x0 = x[2 * e], x1 = x[2 * e + 1];
    const unsigned hp = split_pack2(x0, x1);
    const float r0 = x0 - split_lo(hp), r1 = x1 - split_hi(hp);
    const unsigned mp = split_pack2(r0, r1);
    const float s0 = r0 - split_lo(mp), s1 = r1 - split_hi(mp);
    h[e] = hp;
    m[e] = mp;
    l[e] = split_pack2(s0, s1);
  }
}

// Tile-relative buffer addressing for the epilogues: one descriptor per block tile (uniform), one 32-bit byte offset
// per lane, and a compile-time scalar offset per element -- no per-element 64-bit address registers.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t tile_rsrc(const float* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store(float v, rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

// start of the last 4-float chunk that holds a valid column when `lim` columns are valid (lim >= 1)
__device__ __forceinline__ int last_chunk(int lim) { return ((lim - 1) >> 2) << 2; }

// Reduce V (16 or 32) per-lane values over the 32 lanes of each half-wave with recursive halving: V-1 shuffles
// instead of 5V.  Returns the total of slot `own` (own = l32 for V == 32, l32 >> 1 for V == 16); v is clobbered.
template <int N, int MASK>
struct HalvingStep {
  // one level: lanes with (l32 & MASK) keep the upper half of the N live values, the others the lower half
  static __device__ __forceinline__ void run(float* v, int l32) {
    if constexpr (N > 1) {
      const bool up = (l32 & MASK) != 0;
#pragma unroll
      for (int j = 0; j < N / 2; ++j) {
        const float send = up ? v[j] : v[j + N / 2];
        const float keep = up ? v[j + N / 2] : v[j];
        v[j] = keep + __shfl_xor(send, MASK);
      }
      HalvingStep<N / 2, MASK / 2>::run(v, l32);
    }
  }
};

template <int V>
__device__ __forceinline__ float halving_sum32(float (&v)[V], int l32) {
  static_assert(V == 16 || V == 32, "one or two 32x32 accumulator tiles per wave row");
  HalvingStep<V, 16>::run(v, l32);
  float r = v[0];
  if (V == 16) r += __shfl_xor(r, 1);
  return r;
}

// Cross-lane sums on DPP adds and lane swaps instead of __shfl_xor, which hipcc lowers to ds_bpermute_b32 (an LDS-crossbar round
// trip of > 100 cycles per step).  The values added are those of the xor butterfly (v + v[lane ^ 1], ^ 2, ^ 4, ^ 8, ^ 16: after
// every step the lanes of a group hold one common value, so "the lane mirrored inside the group" supplies the same addend as
// "the lane ^ step"), so the results are bit-identical to it.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float v) {   // the DPP-selected lane's value; 0 in rows that ROW_MASK leaves out
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// sum over each group of 16 consecutive lanes (all lanes of the group receive the total)
__device__ __forceinline__ float group_sum16(float v) {
  v += dpp_take<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += dpp_take<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
  v += dpp_take<0x141, 0xf>(v);   // row_half_mirror
  v += dpp_take<0x140, 0xf>(v);   // row_mirror
  return v;
}
// v + v[lane ^ 16]: v_permlane16_swap of (v, v) leaves {row 0, row 0, row 2, row 2} and {row 1, row 1, row 3, row 3}
__device__ __forceinline__ float add_xor16(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto sw = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
}
// v + v[lane ^ 32]: v_permlane32_swap of (v, v) leaves {low half, low half} and {high half, high half}
__device__ __forceinline__ float add_xor32(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
}
// Sum over the wave with the addends of the descending xor butterfly (v + v[lane ^ 32], ^ 16, ^ 8, ^ 4, ^ 2, ^ 1), bit-identical
// to it: after the ^ 32 and ^ 16 steps the four rows of 16 lanes are equal, and a rotation of a row by 8, 4, 2, 1 lanes then
// supplies the value the xor partner holds (after the step of width w the row has period w).
__device__ __forceinline__ float wave_sum_desc(float v) {
  v = add_xor16(add_xor32(v));
  v += dpp_take<0x128, 0xf>(v);   // row_ror:8
  v += dpp_take<0x124, 0xf>(v);   // row_ror:4
  v += dpp_take<0x122, 0xf>(v);   // row_ror:2
  v += dpp_take<0x121, 0xf>(v);   // row_ror:1
  return v;
}
// sum over the 32 lanes of a half-wave (all lanes of the half receive the total)
__device__ __forceinline__ float half_sum32(float v) { return add_xor16(group_sum16(v)); }

template <int BM, int BN, int BK, int WGM, int WGN, int LA, int LB, int EPI>
struct GemmCfg {
  static constexpr int TM = BM / WGM / 32;
  static constexpr int TN = BN / WGN / 32;
  static constexpr int RA = (LA == LAY_K) ? BM : BK;   // rows / cols of the staged A tile
  static constexpr int CA = (LA == LAY_K) ? BK : BM;
  static constexpr int RB = (LB == LAY_K) ? BN : BK;
  static constexpr int CB = (LB == LAY_K) ? BK : BN;
  static constexpr int SA = CA + 4;                    // padded LDS row strides (floats)
  static constexpr int SB = CB + 4;
  static constexpr int NT = 64 * WGM * WGN;             // threads per block: one wave per 32-row x (BN / WGN)-column slice
  static constexpr int A_F4 = (RA * CA / 4 + NT - 1) / NT;
  static constexpr int B_F4 = (RB * CB / 4 + NT - 1) / NT;
  static constexpr int LDS_TILES = RA * SA + RB * SB;
  static constexpr int LDS_EPI = 2 * BM * WGN + 3 * BN * WGM + 64;   // cross-wave reduction scratch
  static constexpr int LDS_FLOATS = LDS_TILES > LDS_EPI ? LDS_TILES : LDS_EPI;
  static_assert(WGM * WGN == 4 || WGM * WGN == 8, "4 waves (256 threads) or 8 waves (512 threads) per block");
  static_assert(BM % (WGM * 32) == 0 && BN % (WGN * 32) == 0, "wave tiles are multiples of 32x32");
  static_assert(BK % 8 == 0, "k sub-step is 8");
};

// PRO: prologue applied while staging — to operand A in the forward kernels (LA == LAY_K), to operand B in wgrad
// (EPI_SLAB).  The feature index of an element is its column in the stored matrix in both cases.
// blocks per CU asked of the register allocator: 3 for the plain-store kernels whose accumulators take at most 64
// registers per lane (the LayerNorm / heads epilogues need the larger budget)
constexpr int gemm_min_waves(int bm, int bn, int epi) { return (epi == EPI_SLAB && bm * bn / 256 <= 64) ? 3 : 2; }

// MM 1: the main loop multiplies on the bf16 matrix cores from exact three-term splits of the fp32 operands (fp32
// accuracy, see the MM branch of the main loop); built for forward tiles with both operands k-contiguous and K a multiple
// of 16.  (Reading the weights from planes split beforehand, as dgrad_split_kernel<true> does, was measured here too: 125
// against 120 us for the 512 -> 256 layer, so the forward splits both operands itself.)
template <int BM, int BN, int BK, int WGM, int WGN, int LA, int LB, int EPI, int PRO, int NBUF, int MM = 0>
__global__ __launch_bounds__(64 * WGM * WGN, gemm_min_waves(BM, BN, EPI)) void gemm_kernel(const GemmP p) {
  using C = GemmCfg<BM, BN, BK, WGM, WGN, LA, LB, EPI>;
  constexpr int TM = C::TM, TN = C::TN, SA = C::SA, SB = C::SB;
  constexpr bool PRO_ON_A = (PRO != PRO_NONE) && (EPI != EPI_SLAB);
  constexpr bool PRO_ON_B = (PRO != PRO_NONE) && (EPI == EPI_SLAB);
  static_assert(!PRO_ON_A || LA == LAY_K, "forward prologue expects a k-contiguous A");
  static_assert(!PRO_ON_B || LB == LAY_R, "wgrad prologue expects a row-reduction B");
  constexpr int NT = C::NT;
  static_assert(NT % (C::CA / 4) == 0 && NT % (C::CB / 4) == 0, "each thread stages one fixed column chunk");
  static_assert(NT == 256 || EPI == EPI_LNFWD || EPI == EPI_BIASFWD, "8-wave blocks are built for the forward kernels");
  static_assert(NBUF == 1 || NBUF == 2, "single- or double-buffered LDS tiles");
  static_assert(MM == 0 || (LA == LAY_K && LB == LAY_K && BK == 16 && NT == 256 && EPI != EPI_SLAB), "split main loop: forward tiles, 16 k per step");
  static_assert(MM >= 0 && MM <= 2, "MM 2: the split main loop with the weights read from planes split beforehand (GemmP::Bpl)");
  // split main loop: per stage three bf16 planes of [BM + BN rows][16 k] (32 B per row), two stages
  constexpr int LDS_MAIN = MM ? NBUF * 3 * (BM + BN) * 8 : NBUF * C::LDS_TILES;   // floats
  constexpr int LDS_ALL = (LDS_MAIN > C::LDS_EPI) ? LDS_MAIN : C::LDS_EPI;
  __shared__ __attribute__((aligned(16))) float lds[LDS_ALL];

  MRL_KT(1000);
  MRL_KB(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, h = lane >> 5;
  const int wm = wave / WGN, wn = wave % WGN;
  const int tw = blockIdx.z;                 // index into the per-tower pointer arrays
  const int Nv = p.Nt[tw] ? p.Nt[tw] : p.N;  // valid output columns / reduction length of this tower
  int m0, n0, kbeg = 0, kend = p.Kt[tw] ? p.Kt[tw] : p.K, split = 0;
  const int kend_full = kend;
  if (EPI == EPI_SLAB) {
    const int gsz = p.group_mode ? p.m_tiles * p.n_tiles : p.n_tiles;
    const int lin = blockIdx.x, t = lin >> 3, j = t % gsz;
    const int grp = (t / gsz) * 8 + (lin & 7);
    if (grp >= p.n_groups) return;           // padding blocks of the last octet of groups
    if (p.group_mode) {
      split = grp;
      m0 = (j % p.m_tiles) * BM;
      n0 = (j / p.m_tiles) * BN;
    } else {
      m0 = grp * BM;
      n0 = j * BN;
    }
    kbeg = split * p.ksplit;
    kend = min(kend, kbeg + p.ksplit);
  } else {
    m0 = blockIdx.x * BM;
    n0 = blockIdx.y * BN;
  }
  const float* __restrict__ Ag = p.A[tw];
  const float* __restrict__ Bg = p.B[tw];
  const float* pro0 = p.pro0[tw];
  const float* pro1 = p.pro1[tw];

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // staging: thread `tid` owns column chunk cA / cB (4 floats) of rows rA0 + i*RSTEP of each tile
  constexpr int A_RSTEP = NT / (C::CA / 4), B_RSTEP = NT / (C::CB / 4);
  const int cA = (tid % (C::CA / 4)) * 4, rA0 = tid / (C::CA / 4);
  const int cB = (tid % (C::CB / 4)) * 4, rB0 = tid / (C::CB / 4);
  constexpr bool A_FULL = (C::RA * C::CA / 4) % NT == 0, B_FULL = (C::RB * C::CB / 4) % NT == 0;
  f32x4 ra[C::A_F4], rb[C::B_F4];
  f32x4 pv0 = {0.f, 0.f, 0.f, 0.f}, pv1 = {1.f, 1.f, 1.f, 1.f};   // prologue vectors of this thread's column chunk
  constexpr bool PRO_VEC = PRO == PRO_NORM || PRO == PRO_LNSILU || PRO == PRO_LNELU;   // prologues with per-feature vectors
  if constexpr (PRO_ON_B && PRO_VEC) {   // wgrad: the feature index n0 + cB does not change along the reduction
    const int gc = min(n0 + cB, last_chunk(Nv));
    pv0 = *reinterpret_cast<const f32x4*>(pro0 + gc);
    pv1 = *reinterpret_cast<const f32x4*>(pro1 + gc);
  }

  // Operand tiles are read through one buffer descriptor per operand whose base is this block's first element: a
  // lane's byte offset inside the tile never changes (one register per staging unit, clamped into the matrix along
  // the dimension that does not move), and the reduction offset of the step is a scalar.  Reads past the end of the
  // matrix return zeros instead of faulting, so the loads are unconditional -- also for the tiles prefetched past the
  // end of the reduction; what lies outside the valid rows / columns is zeroed when the tile is written to LDS.
  // A staging unit is one 16-byte chunk per thread: units 0 .. A_F4-1 belong to the A tile, the rest to the B tile.
  constexpr int NSTAGE = C::A_F4 + C::B_F4;
  const int kfull = kend_full;   // reduction length of the whole matrix (the rows that exist in a LAY_R operand)
  auto operand_rsrc = [&](const float* base, int lay, int64_t ld, int fixed0, int fixed_n) {
    // lay LAY_K: rows fixed0.. of fixed_n rows, columns = reduction;  LAY_R: rows = reduction, columns of fixed_n
    const int64_t origin = (lay == LAY_K) ? (int64_t)fixed0 * ld + kbeg : (int64_t)kbeg * ld;
    const int64_t extent = (lay == LAY_K) ? (int64_t)(fixed_n - 1) * ld + kfull : (int64_t)(kfull - 1) * ld + fixed_n;
    // a row whose length is not a multiple of 4 ends inside a 4-float chunk: the chunk's tail must stay readable
    const int ncols = (lay == LAY_K) ? kfull : fixed_n;
    int64_t bytes = (extent - origin + ((4 - (ncols & 3)) & 3)) * 4;
    bytes = bytes < 0 ? 0 : (bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + origin), 0, (int)(unsigned)bytes, 0x00020000);
  };
  const rsrc_t rsA = operand_rsrc(Ag, LA, p.lda, m0, p.M);
  const rsrc_t rsB = operand_rsrc(Bg, LB, p.ldb, n0, Nv);
  int voffA[C::A_F4], voffB[C::B_F4];
  bool inA[C::A_F4], inB[C::B_F4];   // the part of the validity test that does not depend on the reduction offset
#pragma unroll
  for (int i = 0; i < C::A_F4; ++i) {
    const int r = rA0 + i * A_RSTEP;
    if (LA == LAY_K) {
      voffA[i] = (int)(((int64_t)(min(m0 + r, p.M - 1) - m0) * p.lda + cA) * 4);
      inA[i] = m0 + r < p.M;
    } else {
      voffA[i] = (int)(((int64_t)r * p.lda + min(m0 + cA, last_chunk(p.M))) * 4);
      inA[i] = m0 + cA < p.M;
    }
  }
#pragma unroll
  for (int i = 0; i < C::B_F4; ++i) {
    const int r = rB0 + i * B_RSTEP;
    if (LB == LAY_K) {
      voffB[i] = (int)(((int64_t)(min(n0 + r, Nv - 1) - n0) * p.ldb + cB) * 4);
      inB[i] = n0 + r < Nv;
    } else {
      voffB[i] = (int)(((int64_t)r * p.ldb + min(n0 + cB, last_chunk(Nv))) * 4);
      inB[i] = n0 + cB < Nv;
    }
  }
  const int kstepA = (LA == LAY_K) ? 4 : (int)(p.lda * 4), kstepB = (LB == LAY_K) ? 4 : (int)(p.ldb * 4);   // bytes per reduction index
  rsrc_t rsP0, rsP1;
  if constexpr (PRO_ON_A && PRO_VEC) {   // forward prologue vectors follow the reduction offset of the A tile
    const int bytes = (kfull - kbeg) * 4;
    rsP0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pro0 + kbeg), 0, bytes, 0x00020000);
    rsP1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pro1 + kbeg), 0, bytes, 0x00020000);
  }
  auto load16 = [](rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
  };
  auto load_unit = [&](int u, int k0) {
    if (u < C::A_F4) {
      const int i = u;
      if (A_FULL || rA0 + i * A_RSTEP < C::RA) ra[i] = load16(rsA, voffA[i], (k0 - kbeg) * kstepA);
    } else {
      const int i = u - C::A_F4;
      if (B_FULL || rB0 + i * B_RSTEP < C::RB) rb[i] = load16(rsB, voffB[i], (k0 - kbeg) * kstepB);
    }
  };
  auto load_pro = [&](int k0) {
    if constexpr (PRO_ON_A && PRO_VEC) {
      pv0 = load16(rsP0, cA * 4, (k0 - kbeg) * 4);
      pv1 = load16(rsP1, cA * 4, (k0 - kbeg) * 4);
    }
  };
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int u = 0; u < NSTAGE; ++u) load_unit(u, k0);
    load_pro(k0);
  };
  auto transform = [&](f32x4 v) {
    f32x4 o;
    if constexpr (PRO == PRO_NORM) {           // mineral/common/normalizers.py:65-68
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fminf(fmaxf((v[e] - pv0[e]) / pv1[e], p.clamp_lo), p.clamp_hi);
    } else if constexpr (PRO == PRO_ELU) {     // F.elu of the stored pre-activation (alpha = 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = v[e] > 0.f ? v[e] : __builtin_amdgcn_exp2f(1.44269504088896340736f * v[e]) - 1.0f;
    } else if constexpr (PRO == PRO_LNELU) {   // LayerNorm affine + ELU of the stored normalised activation
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y = v[e] * pv0[e] + pv1[e];
        o[e] = y > 0.f ? y : __builtin_amdgcn_exp2f(1.44269504088896340736f * y) - 1.0f;
      }
    } else {                                   // LayerNorm affine + SiLU of the stored normalised activation
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y = v[e] * pv0[e] + pv1[e];
        o[e] = y * sigmoid_fast(y);
      }
    }
    return o;
  };
  // prologue + zero fill + write to LDS of one unit of the tile that was loaded for reduction offset k0
  auto store_unit = [&](int u, int k0, int buf) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    float* As = lds + buf * C::LDS_TILES;
    float* Bs = As + C::RA * SA;
    const int kleft = kend - k0;   // valid reduction indices from k0 on (may be <= 0 for a tile past the end)
    if (u < C::A_F4) {
      const int i = u, r = rA0 + i * A_RSTEP;
      if (A_FULL || r < C::RA) {
        const bool ok = inA[i] && (((LA == LAY_K) ? cA : r) < kleft);
        f32x4 v = ra[i];
        if constexpr (PRO_ON_A) v = transform(v);
        *reinterpret_cast<f32x4*>(As + r * SA + cA) = ok ? v : zero;
      }
    } else {
      const int i = u - C::A_F4, r = rB0 + i * B_RSTEP;
      if (B_FULL || r < C::RB) {
        const bool ok = inB[i] && (((LB == LAY_K) ? cB : r) < kleft);
        f32x4 v = rb[i];
        if constexpr (PRO_ON_B) v = transform(v);
        *reinterpret_cast<f32x4*>(Bs + r * SB + cB) = ok ? v : zero;
      }
    }
  };
  auto store_tiles = [&](int k0, int buf) {
#pragma unroll
    for (int u = 0; u < NSTAGE; ++u) store_unit(u, k0, buf);
  };

  // fragments of k sub-step kk (8 reduction indices: lane half h holds 4*h .. 4*h+3) of the tile in LDS buffer `buf`
  auto frag_load = [&](int buf, int kk, f32x4 (&af)[TM], f32x4 (&bf)[TN]) {
    const float* As = lds + buf * C::LDS_TILES;
    const float* Bs = As + C::RA * SA;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int row = wm * TM * 32 + i * 32 + l32;
      if (LA == LAY_K) {
        af[i] = *reinterpret_cast<const f32x4*>(As + row * SA + kk * 8 + 4 * h);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) af[i][j] = As[(kk * 8 + 4 * h + j) * SA + row];
      }
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      int col = wn * TN * 32 + i * 32 + l32;
      if (LB == LAY_K) {
        bf[i] = *reinterpret_cast<const f32x4*>(Bs + col * SB + kk * 8 + 4 * h);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[i][j] = Bs[(kk * 8 + 4 * h + j) * SB + col];
      }
    }
  };
  auto mfma_j = [&](int j, const f32x4 (&af)[TM], const f32x4 (&bf)[TN]) {
#pragma unroll
    for (int im = 0; im < TM; ++im)
#pragma unroll
      for (int in = 0; in < TN; ++in)
        if constexpr (EPI == EPI_LNFWD || EPI == EPI_BIASFWD)   // transposed product: a lane holds one batch row and 16 features per 32x32 tile
          acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[in][j], af[im][j], acc[im][in], 0, 0, 0);
        else
          acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[im][j], bf[in][j], acc[im][in], 0, 0, 0);
  };
  auto mfma_group = [&](const f32x4 (&af)[TM], const f32x4 (&bf)[TN]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) mfma_j(j, af, bf);
  };

  constexpr int NG = BK / 8;   // k sub-steps per tile
  if constexpr (MM >= 1) {
    // Every fp32 operand element is split exactly into three bf16 terms x = h + m + l (split_rne4) and each product is
    // accumulated in fp32 from the six terms hh, hm, mh, mm, hl, lh; the dropped terms are below 2^-23 of |a*b|.  Planes
    // live in LDS as [rows][16 k] bf16 with the two 16-byte halves of a row swapped on every other group of 8 rows (a
    // quarter-wave's 16-byte fragment reads then cover all banks).  Per step: the MFMAs of step st from stage st & 1, then
    // the operands of step st+1 (already in registers) are split into the other stage and the next loads issued; one
    // barrier per step (NBUF 2), or one stage and two barriers per step for short reductions (NBUF 1).
    using sp_bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
    constexpr int ROWS = BM + BN, SPLANE = ROWS * 32, SSTAGE = 3 * SPLANE;
    constexpr int A_CH = BM * 4 / NT, B_CH = BN * 4 / NT;           // 16-byte chunks per thread and step
    static_assert(BM * 4 % NT == 0 && BN * 4 % NT == 0, "whole chunks per thread");
    char* sl = reinterpret_cast<char*>(lds);
    const int K = kend;                                              // kbeg == 0 in the forward kernels
    const int nsteps = (MRL_ABLATE_FWD & 8) ? 1 : K / 16;
    const int ca = tid & 3, row0 = tid >> 2;                         // chunk ca of rows row0 + 64 i
    auto r_off = [](int row, int half) { return row * 32 + ((half ^ ((row >> 3) & 1)) * 16); };
    // A (activations, from HBM) is loaded three steps ahead into alternating register sets; the weights and the prologue
    // vectors (L2-resident, every block reads the same) one step ahead into a single set
    // MM 2: the weights come as three bf16 planes split beforehand, in k-blocked order (element (n, k) of the [Nv][K] matrix
    // at (k / 16) * Nv * 16 + n * 16 + k % 16): 16-byte chunks of 8 k go to LDS as they are, chunk (row, half) = thread
    // (tid >> 1, tid & 1) plus 128 rows per further chunk
    constexpr bool WPL = MM == 2;
    constexpr int BP_CH = BN * 2 / NT;
    static_assert(!WPL || BN * 2 % NT == 0, "whole plane chunks per thread");
    typedef unsigned int sp_u32x4 __attribute__((ext_vector_type(4)));
    const unsigned short* __restrict__ Wpl = p.Bpl[tw];
    const int rowp = tid >> 1, halfp = tid & 1;
    f32x4 sa[2][A_CH], sb[WPL ? 1 : B_CH], spv0, spv1;
    sp_u32x4 sw[WPL ? BP_CH : 1][3];
    auto sp_load = [&](int st, auto SET) {
      constexpr int S = decltype(SET)::value;
      const int k0 = min(st * 16, K - 16) + 4 * ca;                  // steps past the end re-read the last one; never multiplied
#pragma unroll
      for (int i = 0; i < A_CH; ++i) sa[S][i] = *reinterpret_cast<const f32x4*>(Ag + (int64_t)min(m0 + row0 + 64 * i, p.M - 1) * p.lda + k0);
    };
    auto sp_load_w = [&](int st) {
      const int k0 = min(st * 16, K - 16) + 4 * ca;
      if constexpr (WPL) {
        // k-blocked planes: the [Nv rows][16 k] tile of step st is one contiguous run, thread tid takes its 16-byte chunk tid
        const int64_t kb = (int64_t)min(st, K / 16 - 1) * Nv * 16 + 8 * halfp;
#pragma unroll
        for (int i = 0; i < BP_CH; ++i)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            sw[i][pl] = *reinterpret_cast<const sp_u32x4*>(Wpl + pl * p.pl_stride + kb + (int64_t)min(n0 + rowp + 128 * i, Nv - 1) * 16);
      } else {
#pragma unroll
      for (int i = 0; i < B_CH; ++i) sb[i] = *reinterpret_cast<const f32x4*>(Bg + (int64_t)min(n0 + row0 + 64 * i, Nv - 1) * p.ldb + k0);
      }
      if constexpr (PRO_ON_A && PRO_VEC) {
        spv0 = *reinterpret_cast<const f32x4*>(pro0 + k0);
        spv1 = *reinterpret_cast<const f32x4*>(pro1 + k0);
      }
    };
    auto sp_split = [&](const f32x4 x, char* q) {
      split_u32x2 h, m, l;
      if constexpr (MRL_ABLATE_FWD & 2) {
        h = split_u32x2{__float_as_uint(x[0]), __float_as_uint(x[1])};
        m = split_u32x2{__float_as_uint(x[2]), __float_as_uint(x[3])};
        l = h;
      } else
      split_rne4(x, h, m, l);
      *reinterpret_cast<split_u32x2*>(q) = h;
      *reinterpret_cast<split_u32x2*>(q + SPLANE) = m;
      *reinterpret_cast<split_u32x2*>(q + 2 * SPLANE) = l;
    };
    auto sp_stage = [&](auto SET) {
      constexpr int S = decltype(SET)::value;
      char* buf = sl + S * SSTAGE;
      if constexpr (PRO_ON_A && PRO_VEC) { pv0 = spv0; pv1 = spv1; }
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        f32x4 v = sa[S][i];
        if constexpr (PRO_ON_A && !(MRL_ABLATE_FWD & 4)) v = transform(v);
        sp_split(v, buf + r_off(row0 + 64 * i, ca >> 1) + (ca & 1) * 8);
      }
      if constexpr (WPL) {
#pragma unroll
        for (int i = 0; i < BP_CH; ++i) {
          const bool ok = n0 + rowp + 128 * i < Nv;                  // weight rows past the valid outputs contribute zeros
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<sp_u32x4*>(buf + pl * SPLANE + r_off(BM + rowp + 128 * i, halfp)) = ok ? sw[i][pl] : sp_u32x4{0u, 0u, 0u, 0u};
        }
      } else {
#pragma unroll
      for (int i = 0; i < B_CH; ++i) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const bool ok = n0 + row0 + 64 * i < Nv;                    // weight rows past the valid outputs contribute zeros
        sp_split(ok ? sb[i] : zero, buf + r_off(BM + row0 + 64 * i, ca >> 1) + (ca & 1) * 8);
      }
      }
    };
    int fa_off[TM], fb_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa_off[i] = r_off(wm * TM * 32 + i * 32 + l32, h);
#pragma unroll
    for (int i = 0; i < TN; ++i) fb_off[i] = r_off(BM + wn * TN * 32 + i * 32 + l32, h);
    auto sp_mfmas = [&](auto SET) {
      constexpr int S = decltype(SET)::value;
      const char* cur = sl + S * SSTAGE;
      constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};   // (a plane, b plane) of the six terms, smallest first
      sp_bf16x8 fa[TM][3], fb[2][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i][pl] = *reinterpret_cast<const sp_bf16x8*>(cur + pl * SPLANE + fa_off[i]);
        fb[0][pl] = *reinterpret_cast<const sp_bf16x8*>(cur + pl * SPLANE + fb_off[0]);
      }
#pragma unroll
      for (int in = 0; in < TN; ++in) {
        if (in + 1 < TN) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) fb[(in + 1) & 1][pl] = *reinterpret_cast<const sp_bf16x8*>(cur + pl * SPLANE + fb_off[in + 1]);
        }
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
          for (int im = 0; im < TM; ++im)
            if constexpr (EPI == EPI_LNFWD || EPI == EPI_BIASFWD)   // transposed product, as in the fp32 loop
              acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[in & 1][TB[term]], fa[im][TA[term]], acc[im][in], 0, 0, 0);
            else
              acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[im][TA[term]], fb[in & 1][TB[term]], acc[im][in], 0, 0, 0);
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto sp_step = [&](int st, auto PAR) {
      constexpr int P = decltype(PAR)::value;
      using Q = std::integral_constant<int, P ^ 1>;
      sp_mfmas(PAR);
      sp_stage(Q{});
      sp_load(st + 3, Q{});
      sp_load_w(st + 2);
      __syncthreads();
    };
    if constexpr (NBUF == 1) {                                       // short reductions (first layer): one stage, two barriers per step
      sp_load(0, I0{});
      sp_load_w(0);
      for (int st = 0; st < nsteps; ++st) {
        sp_stage(I0{});
        __syncthreads();
        sp_load(st + 1, I0{});
        sp_load_w(st + 1);
        sp_mfmas(I0{});
        __syncthreads();
      }
    } else {
      sp_load(0, I0{});
      sp_load(1, I1{});
      sp_load_w(0);
      sp_stage(I0{});
      sp_load(2, I0{});
      sp_load_w(1);
      __syncthreads();
      int st = 0;
      for (; st + 1 < nsteps; st += 2) {
        sp_step(st, I0{});
        sp_step(st + 1, I1{});
      }
      if (st < nsteps) sp_step(st, I0{});
    }
  } else if constexpr (NBUF == 1) {
    f32x4 af[TM], bf[TN];
    if (kbeg < kend) load_tiles(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      __syncthreads();                 // previous tile fully consumed
      store_tiles(k0, 0);
      __syncthreads();
      if (k0 + BK < kend) load_tiles(k0 + BK);   // raw prefetch into registers; consumed after the MFMAs below
#pragma unroll
      for (int kk = 0; kk < NG; ++kk) {
        frag_load(0, kk, af, bf);
        mfma_group(af, bf);
      }
    }
  } else {
    // Double-buffered and software-pipelined, one barrier per tile.  While the MFMAs of sub-step g of tile s issue,
    // the fragments of sub-step g+1 are read from LDS; during sub-step 0, tile s+1 goes from registers into the other
    // LDS buffer and tile s+2 from memory into registers.  The barrier sits before the last sub-step, whose MFMAs
    // (operands already in registers) cover the fragment reads of the next tile.  Staging is unconditional: loads
    // are clamped into the matrix and a tile past the end of the reduction is zeros that nobody reads.
    static_assert(NG % 2 == 0, "fragment registers alternate by sub-step parity");
    const int nsteps = (kend - kbeg + BK - 1) / BK;
    f32x4 af[2][TM], bf[2][TN];
    if (nsteps > 0) {
      load_tiles(kbeg);
      store_tiles(kbeg, 0);
      load_tiles(kbeg + BK);
      __syncthreads();
      frag_load(0, 0, af[0], bf[0]);
    }
    MRL_KT(0);
    for (int st = 0; st < nsteps; ++st) {
      const int k0 = kbeg + st * BK, cur = st & 1;
      MRL_KT(400 + st);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
          if (!MRL_ABLATE(1)) frag_load(cur, g + 1, af[(g + 1) & 1], bf[(g + 1) & 1]);
        } else {
          MRL_KT(1 + st * 2);
          if (!MRL_ABLATE(2)) __syncthreads();
          MRL_KT(2 + st * 2);
          if (!MRL_ABLATE(1)) frag_load(cur ^ 1, 0, af[0], bf[0]);
        }
        if (g == 0) {
          // the staging units are spread over the four MFMA batches of this sub-step; the scheduler may interleave
          // inside a batch but not move work from one batch to another
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (!MRL_ABLATE(3)) mfma_j(j, af[0], bf[0]);
            if (!MRL_ABLATE(0)) {
#pragma unroll
              for (int u = j * NSTAGE / 4; u < (j + 1) * NSTAGE / 4; ++u) {
                store_unit(u, k0 + BK, cur ^ 1);
                load_unit(u, k0 + 2 * BK);
              }
              if (j == 3) load_pro(k0 + 2 * BK);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          if (!MRL_ABLATE(3)) mfma_group(af[g & 1], bf[g & 1]);
        }
      }
    }
  }
  __syncthreads();   // LDS is reused as reduction scratch below
  MRL_KT(1001);

  // element (im, in, r) of this lane: row = rbase(im) + (r&3) + 8*(r>>2), col = cbase(in)
  auto rowof = [&](int im, int r) { return wm * TM * 32 + im * 32 + (r & 3) + 8 * (r >> 2) + 4 * h; };
  auto colof = [&](int in) { return wn * TN * 32 + in * 32 + l32; };

  if constexpr (EPI == EPI_SLAB) {
    float* out = p.out[tw] + (int64_t)split * p.M * p.ldo;
#pragma unroll
    for (int im = 0; im < TM; ++im)
#pragma unroll
      for (int in = 0; in < TN; ++in) {
        int col = n0 + colof(in);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int row = m0 + rowof(im, r);
          if (row < p.M && col < p.N) out[(int64_t)row * p.ldo + col] = acc[im][in][r];
        }
      }
  } else if constexpr (EPI == EPI_LNFWD) {
    // The forward kernels issue their MFMAs with the operands swapped, so lane (l32, h) of tile (im, in) holds batch row
    // wm*TM*32 + im*32 + l32 and the 16 features wn*TN*32 + in*32 + 8*g + 4*h + (0..3), g = 0..3: the LayerNorm sums
    // over a row are plain adds in the lane plus one exchange between the half-waves, and every 4 consecutive
    // accumulator registers leave as one 16-byte store.
    // work on scalars from here on: element-wise updates of the 16-wide accumulator vectors cost whole-vector copies
    float ea[TM][TN][16];
    const float* bias = p.bias[tw];
#pragma unroll
    for (int in = 0; in < TN; ++in)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + n0 + wn * TN * 32 + in * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int im = 0; im < TM; ++im)
#pragma unroll
          for (int e = 0; e < 4; ++e) ea[im][in][g * 4 + e] = acc[im][in][g * 4 + e] + b4[e];
      }
    // bias, two-pass LayerNorm statistics over the full row (BN == N), store xhat and rstd
    float* partA = lds;                 // [WGN][BM] row sums per wave column
    float* partB = lds + BM * WGN;      // [WGN][BM] sums of squared deviations
    const float invn = 1.0f / (float)BN;
#pragma unroll
    for (int im = 0; im < TM; ++im) {
      float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
      for (int in = 0; in < TN; ++in)
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
          t0 += ea[im][in][r]; t1 += ea[im][in][r + 1]; t2 += ea[im][in][r + 2]; t3 += ea[im][in][r + 3];
        }
      float t = (t0 + t1) + (t2 + t3);
      t = add_xor32(t);
      if (h == 0) partA[wn * BM + wm * TM * 32 + im * 32 + l32] = t;
    }
    MRL_KT(1003);
    __syncthreads();
    MRL_KT(1004);
#pragma unroll
    for (int im = 0; im < TM; ++im) {
      const int row = wm * TM * 32 + im * 32 + l32;
      float m = partA[row];
#pragma unroll
      for (int w = 1; w < WGN; ++w) m += partA[w * BM + row];
      const float mean = m * invn;
      float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
      for (int in = 0; in < TN; ++in)
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
          const float d0 = ea[im][in][r] - mean, d1 = ea[im][in][r + 1] - mean, d2 = ea[im][in][r + 2] - mean, d3 = ea[im][in][r + 3] - mean;
          ea[im][in][r] = d0; ea[im][in][r + 1] = d1; ea[im][in][r + 2] = d2; ea[im][in][r + 3] = d3;
          q0 += d0 * d0; q1 += d1 * d1; q2 += d2 * d2; q3 += d3 * d3;
        }
      float q = (q0 + q1) + (q2 + q3);
      q = add_xor32(q);
      if (h == 0) partB[wn * BM + row] = q;
    }
    MRL_KT(1005);
    __syncthreads();
    MRL_KT(1006);
    // rows >= M of the last block are written too: the activation workspace is padded to whole row tiles
    float* outp = p.out[tw] + (int64_t)m0 * BN;
    float* rs = p.rstd[tw];
#pragma unroll
    for (int im = 0; im < TM; ++im) {
      const int row = wm * TM * 32 + im * 32 + l32;
      float v = partB[row];
#pragma unroll
      for (int w = 1; w < WGN; ++w) v += partB[w * BM + row];
      const float rstd = 1.0f / sqrtf(v * invn + p.ln_eps);
      if (wn == 0 && h == 0) rs[m0 + row] = rstd;
      float* orow = outp + (int64_t)row * BN + wn * TN * 32 + 4 * h;
#pragma unroll
      for (int in = 0; in < TN; ++in)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 o4 = {ea[im][in][g * 4] * rstd, ea[im][in][g * 4 + 1] * rstd, ea[im][in][g * 4 + 2] * rstd, ea[im][in][g * 4 + 3] * rstd};
          if (!(MRL_ABLATE_FWD & 1) || o4[0] == 1.2345e-30f) *reinterpret_cast<f32x4*>(orow + in * 32 + 8 * g) = o4;
        }
    }
  } else if constexpr (EPI == EPI_LNBWD) {
    // work on scalars from here on: element-wise updates of the 16-wide accumulator vectors cost whole-vector copies
    float ea[TM][TN][16];
#pragma unroll
    for (int im = 0; im < TM; ++im)
#pragma unroll
      for (int in = 0; in < TN; ++in)
#pragma unroll
        for (int r = 0; r < 16; ++r) ea[im][in][r] = acc[im][in][r];
    // acc = d loss / d h with h = silu(y), y = xhat*gamma + beta, xhat = LayerNorm-normalised z.
    // dy = dh * silu'(y); dxhat = dy*gamma; dz = rstd * (dxhat - mean(dxhat) - xhat*mean(dxhat*xhat)).
    float* part1 = lds;                        // [WGN][BM]  sum dxhat
    float* part2 = lds + BM * WGN;             // [WGN][BM]  sum dxhat*xhat
    float* cred = lds + 2 * BM * WGN;          // [WGM][3][BN] column partials (dgamma, dbeta, dbias)
    const rsrc_t xh_rs = tile_rsrc(p.xhat[tw] + (int64_t)m0 * BN, BM * BN * 4);
    const rsrc_t out_rs = tile_rsrc(p.out[tw] + (int64_t)m0 * BN, BM * BN * 4);
    const int lane_boff = (((wm * TM * 32 + 4 * h) * BN) + wn * TN * 32 + l32) * 4;
    const float* gam = p.gamma[tw];
    const float* bet = p.beta[tw];
    const float invn = 1.0f / (float)BN;
    const int own = l32 >> 1;
    const int own_row = wm * TM * 32 + (own & 3) + 8 * (own >> 2) + 4 * h;   // + im*32
    const bool owner = (l32 & 1) == 0;
    float gv[TN], bv[TN], sg[TN], sb[TN];
#pragma unroll
    for (int in = 0; in < TN; ++in) {
      gv[in] = gam[n0 + colof(in)];
      bv[in] = bet[n0 + colof(in)];
      sg[in] = sb[in] = 0.f;
    }
    // pass 1, software-pipelined by hand: batch b = (im, in, half) covers 8 elements; the loads of batch b+1 are in
    // flight while batch b is computed.  Small batches keep the per-element temporaries within the register budget.
    constexpr int NB1 = TM * TN * 2;
    auto x_off1 = [&](int b, int e) {
      const int im = b / (TN * 2), in = (b / 2) % TN, r = (b & 1) * 8 + e;
      return ((im * 32 + (r & 3) + 8 * (r >> 2)) * BN + in * 32) * 4;
    };
    float xb[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) xb[0][e] = buf_load(xh_rs, lane_boff, x_off1(0, e));
    float c1[16], c2[16];
#pragma unroll
    for (int b = 0; b < NB1; ++b) {
      const int im = b / (TN * 2), in = (b / 2) % TN, hf = b & 1;
      if (in == 0 && hf == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) c1[r] = c2[r] = 0.f;
      }
      if (b + 1 < NB1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) xb[(b + 1) & 1][e] = buf_load(xh_rs, lane_boff, x_off1(b + 1, e));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // no row guard: rows >= M carry dh == 0 (their A rows were zero-filled) and the workspace is padded
        const int r = hf * 8 + e;
        const float x = xb[b & 1][e];
        const float y = x * gv[in] + bv[in];
        float dact;                            // derivative of the activation at y
        if (p.act_elu) dact = y > 0.f ? 1.0f : __builtin_amdgcn_exp2f(1.44269504088896340736f * y);
        else {
          const float sgm = sigmoid_fast(y);
          dact = sgm * (1.0f + y * (1.0f - sgm));
        }
        const float dy = ea[im][in][r] * dact;
        sg[in] += dy * x;
        sb[in] += dy;
        const float dxh = dy * gv[in];
        ea[im][in][r] = dxh;
        c1[r] += dxh;
        c2[r] += dxh * x;
      }
      __builtin_amdgcn_sched_barrier(0);
      if (in == TN - 1 && hf == 1) {
        const float t1 = halving_sum32<16>(c1, l32);
        const float t2 = halving_sum32<16>(c2, l32);
        if (owner) {
          part1[wn * BM + own_row + im * 32] = t1;
          part2[wn * BM + own_row + im * 32] = t2;
        }
      }
    }
#pragma unroll
    for (int in = 0; in < TN; ++in) {
      const float a = add_xor32(sg[in]);
      const float b = add_xor32(sb[in]);
      if (h == 0) {
        cred[(wm * 3 + 0) * BN + colof(in)] = a;
        cred[(wm * 3 + 1) * BN + colof(in)] = b;
      }
    }
    __syncthreads();
    const float* rs = p.rstd[tw];
    float cz[TN];
#pragma unroll
    for (int in = 0; in < TN; ++in) cz[in] = 0.f;
    // pass 2, same pipelining: group q = (im, g) is 4 rows x TN columns; loads of group q+1 fly during group q
    constexpr int NG2 = TM * 4;
    float xs[2][4][TN];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int in = 0; in < TN; ++in) xs[0][rr][in] = buf_load(xh_rs, lane_boff, (rr * BN + in * 32) * 4);
#pragma unroll
    for (int q = 0; q < NG2; ++q) {
      const int im = q / 4, g = q % 4;
      const int base = wm * TM * 32 + im * 32 + 8 * g + 4 * h;
      f32x4 t1 = *reinterpret_cast<const f32x4*>(part1 + base);
      f32x4 t2 = *reinterpret_cast<const f32x4*>(part2 + base);
#pragma unroll
      for (int w = 1; w < WGN; ++w) {
        t1 += *reinterpret_cast<const f32x4*>(part1 + w * BM + base);
        t2 += *reinterpret_cast<const f32x4*>(part2 + w * BM + base);
      }
      const f32x4 rs4 = *reinterpret_cast<const f32x4*>(rs + m0 + base);
      if (q + 1 < NG2) {
        const int im2 = (q + 1) / 4, g2 = (q + 1) % 4;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int in = 0; in < TN; ++in)
            xs[(q + 1) & 1][rr][in] = buf_load(xh_rs, lane_boff, ((im2 * 32 + 8 * g2 + rr) * BN + in * 32) * 4);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float rstd = rs4[rr];
        const float m1 = t1[rr] * invn, m2 = t2[rr] * invn;
#pragma unroll
        for (int in = 0; in < TN; ++in) {
          const float dz = rstd * (ea[im][in][g * 4 + rr] - m1 - xs[q & 1][rr][in] * m2);
          buf_store(dz, out_rs, lane_boff, ((im * 32 + 8 * g + rr) * BN + in * 32) * 4);
          cz[in] += dz;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int in = 0; in < TN; ++in) {
      float z = add_xor32(cz[in]);
      if (h == 0) cred[(wm * 3 + 2) * BN + colof(in)] = z;
    }
    __syncthreads();
    float* cp = p.colpart[tw] + (int64_t)blockIdx.x * 3 * BN;
    for (int i = tid; i < 3 * BN; i += 256) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < WGM; ++w) sum += cred[w * 3 * BN + i];
      cp[i] = sum;
    }
  } else if constexpr (EPI == EPI_BIASFWD) {
    // plain towers (mineral/nets/mlp.py:50-77 with norm_type None): the layer output z = x W^T + b is stored as it is and
    // the activation is applied by whoever reads it.  Same orientation as EPI_LNFWD: 16-byte stores, one batch row per lane.
    const float* bias = p.bias[tw];
    float* outp = p.out[tw] + (int64_t)m0 * BN;
#pragma unroll
    for (int im = 0; im < TM; ++im) {
      const int row = wm * TM * 32 + im * 32 + l32;
      float* orow = outp + (int64_t)row * BN + wn * TN * 32 + 4 * h;
#pragma unroll
      for (int in = 0; in < TN; ++in)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + n0 + wn * TN * 32 + in * 32 + 8 * g + 4 * h);
          const f32x4 o4 = {acc[im][in][g * 4] + b4[0], acc[im][in][g * 4 + 1] + b4[1], acc[im][in][g * 4 + 2] + b4[2], acc[im][in][g * 4 + 3] + b4[3]};
          *reinterpret_cast<f32x4*>(orow + in * 32 + 8 * g) = o4;
        }
    }
  } else if constexpr (EPI == EPI_ACTBWD) {
    // plain towers: acc = d loss / d h with h = elu(z); dz = dh * elu'(z), elu'(z) = 1 for z > 0 and exp(z) otherwise.
    // Column sums of dz (the bias gradient) go to slot 2 of the column partials, as in EPI_LNBWD.
    // no row guards: rows >= M carry dh == 0 (their A rows were zero-filled) and the workspace is padded to whole tiles
    float* cred = lds;                         // [WGM][BN]
    const rsrc_t z_rs = tile_rsrc(p.xhat[tw] + (int64_t)m0 * BN, BM * BN * 4);
    const rsrc_t out_rs = tile_rsrc(p.out[tw] + (int64_t)m0 * BN, BM * BN * 4);
    const int lane_boff = (((wm * TM * 32 + 4 * h) * BN) + wn * TN * 32 + l32) * 4;
#pragma unroll
    for (int in = 0; in < TN; ++in) {
      float cz = 0.f;
#pragma unroll
      for (int im = 0; im < TM; ++im) {
        float zv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) zv[r] = buf_load(z_rs, lane_boff, ((im * 32 + (r & 3) + 8 * (r >> 2)) * BN + in * 32) * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = zv[r] > 0.f ? 1.0f : __builtin_amdgcn_exp2f(1.44269504088896340736f * zv[r]);
          const float dz = acc[im][in][r] * d;
          buf_store(dz, out_rs, lane_boff, ((im * 32 + (r & 3) + 8 * (r >> 2)) * BN + in * 32) * 4);
          cz += dz;
        }
      }
      cz = add_xor32(cz);
      if (h == 0) cred[wm * BN + colof(in)] = cz;
    }
    __syncthreads();
    float* cp = p.colpart[tw] + (int64_t)blockIdx.x * 3 * BN + 2 * BN;
    for (int i = tid; i < BN; i += 256) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < WGM; ++w) sum += cred[w * BN + i];
      cp[i] = sum;
    }
  } else if constexpr (EPI == EPI_HEADS_TRAIN || EPI == EPI_HEADS_ACT) {
    static_assert(TN == 1 && WGN == 1 && TM == 1, "heads use a 32-wide tile, one 32x32 block per wave");
    const HeadsP& hp = p.h;
    const int A = hp.act_dim;
    const bool actor = (tw + p.tw0 == 0);
    // The MFMA layout gives a lane 16 rows of ONE column, and only the first 2A columns mean anything.  The wave's tile
    // goes through LDS once so that afterwards a lane owns (row, action) pairs: 16-lane groups take one row each (8
    // rows per group), the per-row sums over actions stay 16-lane reductions, and the critic takes one row per lane.
    constexpr int TS = 33;                                   // row stride of the transposed tile
    constexpr int NPART = 16;                                // (wave, group) partials
    float* T = lds + wave * (32 * TS);
    double* red = reinterpret_cast<double*>(lds + 4 * 32 * TS);          // [NPART][MRL_NMETRIC]
    float* csum = lds + 4 * 32 * TS + NPART * MRL_NMETRIC * 2;           // [NPART][32] column sums of the seeds
    static_assert(4 * 32 * TS + NPART * MRL_NMETRIC * 2 + NPART * 32 <= LDS_ALL, "heads epilogue scratch");
    {
      const float bias = (l32 < Nv) ? p.bias[tw][l32] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * h) * TS + l32] = acc[0][0][r] + bias;
      if constexpr (EPI == EPI_HEADS_TRAIN) {
        for (int i = tid; i < NPART * MRL_NMETRIC; i += 256) red[i] = 0.0;
        for (int i = tid; i < NPART * 32; i += 256) csum[i] = 0.f;
      }
    }
    __syncthreads();
    // hardware exp2 / log2 / rcp (about 1 ulp each) instead of the libm routines: the tolerances of the parity tests
    // are three orders of magnitude wider
    auto fexp = [](float x) { return __builtin_amdgcn_exp2f(1.44269504088896340736f * x); };
    auto flog = [](float x) { return 0.69314718055994530942f * __builtin_amdgcn_logf(x); };
    auto frcp = [](float x) { return __builtin_amdgcn_rcpf(x); };
    auto ftanh = [&](float x) {                              // 1 - 2 / (1 + e^{2|x|}), sign restored
      const float e = __builtin_amdgcn_exp2f(2.88539008177792681472f * fabsf(x));
      return copysignf(1.0f - 2.0f * frcp(1.0f + e), x);
    };
    const int wrow0 = m0 + wm * 32;                          // first row of this wave's tile
    if (actor) {
      // lanes j < A of a 16-lane group hold action j of the group's row: mu_head[j] and logstd[j]
      // (mineral/agents/ppo/models.py:140-151)
      const int g = lane >> 4, j = lane & 15;
      const bool jv = j < A;
      float cs0 = 0.f, cs1 = 0.f;                            // column sums of the seeds (bias gradients)
      float m_al = 0.f, m_ent = 0.f, m_bl = 0.f, m_kl = 0.f, m_clip = 0.f;
      // every per-row input of the 8 iterations is requested up front: one memory round trip instead of eight
      float in_a[8], in_mu[8], in_sg[8], in_nl[8], in_adv[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = wrow0 + it * 4 + g;
        const int rowc = row < p.M ? row : 0;
        if constexpr (EPI == EPI_HEADS_TRAIN) {
          in_a[it] = jv ? hp.actions[(int64_t)rowc * A + j] : 0.f;
          in_mu[it] = jv ? hp.mu[(int64_t)rowc * A + j] : 0.f;
          in_sg[it] = jv ? hp.sigma[(int64_t)rowc * A + j] : 1.f;
          in_nl[it] = hp.old_neglogp[rowc];
          in_adv[it] = hp.adv[rowc];
        } else {
          in_a[it] = (jv && hp.noise) ? hp.noise[(int64_t)rowc * A + j] : 0.f;
          in_mu[it] = in_sg[it] = in_nl[it] = in_adv[it] = 0.f;
        }
      }
      // read once: a select between the parameter's address and the LDS tile's inside the loop would be a flat load
      // that waits for every outstanding store of the wave on each pass
      const float ls_fixed = hp.fixed_logstd ? hp.fixed_logstd[jv ? j : 0] : 0.f;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int lr = it * 4 + g;
        const int row = wrow0 + lr;
        const bool rv = row < p.M;
        const float o = T[lr * TS + j];
        const float ls = hp.fixed_logstd ? ls_fixed : T[lr * TS + ((A + j) & 31)];
        float mu, sg = 0.f, sigma;
        if (hp.dist_type == DIST_DREAMERV3) {      // mineral/nets/dist.py:42-47
          mu = ftanh(o);
          sg = sigmoid_fast(ls + 2.0f);
          sigma = (hp.maxstd - hp.minstd) * sg + hp.minstd;
        } else if (hp.dist_type == DIST_SQUASHED) {  // mineral/nets/dist.py:39-43; minstd / maxstd carry the logstd clamp
          mu = o;
          sigma = fexp(fminf(fmaxf(ls, hp.minstd), hp.maxstd));
        } else {                                   // mineral/nets/dist.py:32-34
          mu = o;
          sigma = fexp(ls);
        }
        if (!jv) { mu = 0.f; sigma = 1.f; }
        const float logsig = flog(sigma);
        const float rsig = frcp(sigma);
        if (EPI == EPI_HEADS_ACT && hp.dist_type == DIST_SQUASHED) {
          // rsample + log_prob of the tanh-squashed normal (mineral/nets/distributions.py:10-52, sac.py:118-136):
          // u = mu + sigma*eps, a = tanh(u), log p = sum_j [ log N(u_j) - 2*(log 2 - u_j - softplus(-2 u_j)) ]
          const float eps = in_a[it];
          const float u = mu + sigma * eps;
          const float a = ftanh(u);
          const float x2 = -2.0f * u;
          const float sp = x2 > 20.0f ? x2 : flog(1.0f + fexp(x2));
          float lp = jv ? (-0.5f * eps * eps - logsig - MRL_LOG_SQRT_2PI) - 2.0f * (0.69314718055994530942f - u - sp) : 0.f;
          lp = group_sum16(lp);
          if (rv && jv) {
            const int64_t o_idx = (int64_t)row * hp.out_rs + hp.out_off + j;
            hp.act_out[o_idx] = a;
            hp.mu[o_idx] = ls;          // the raw log-std: the backward needs to know where the clamp was active
            hp.sigma[o_idx] = sigma;
          }
          if (rv && j == 0) hp.neglogp_out[row] = lp;   // the log-probability itself
        } else if (EPI == EPI_HEADS_ACT && hp.raw_mu) {
          if (rv && jv) hp.mu[(int64_t)row * hp.out_rs + hp.out_off + j] = o;   // models.py:112-114: mu before the distribution's transform
        } else if constexpr (EPI == EPI_HEADS_ACT) {
          // mineral/agents/ppo/models.py:115-117 with sample = mu + sigma * noise
          const float eps = in_a[it];
          const float a = mu + sigma * eps;
          const float d = a - mu;
          float nl = jv ? (d * d) * (0.5f * rsig * rsig) + logsig + MRL_LOG_SQRT_2PI : 0.f;
          nl = group_sum16(nl);
          if (rv && jv) {
            const int64_t o_idx = (int64_t)row * hp.out_rs + hp.out_off + j;
            hp.act_out[o_idx] = a;
            hp.mu[o_idx] = mu;
            hp.sigma[o_idx] = sigma;
            if (hp.act_clamped) hp.act_clamped[(int64_t)row * A + j] = fminf(fmaxf(a, -1.0f), 1.0f);
          }
          if (rv && j == 0) hp.neglogp_out[row] = nl;
        } else {
          // mineral/agents/ppo/models.py:93-107 and mineral/agents/ppo/ppo.py:277-303
          const float a = in_a[it], omu = in_mu[it], osg = in_sg[it];
          const float d = a - mu;
          const float rvar = rsig * rsig;
          float nl = jv ? (d * d) * (0.5f * rvar) + logsig + MRL_LOG_SQRT_2PI : 0.f;
          float ent = jv ? 0.5f + MRL_LOG_SQRT_2PI + logsig : 0.f;
          float bl = 0.f, dbl = 0.f;
          if (jv) {
            if (hp.bounds_type == 0) {             // 'bound' (ppo.py:480-484)
              const float hi = fmaxf(mu - 1.1f, 0.f), lo = fminf(mu + 1.1f, 0.f);
              bl = lo * lo + hi * hi;
              dbl = 2.0f * hi + 2.0f * lo;
            } else {                               // 'reg' (ppo.py:485-487)
              bl = mu * mu;
              dbl = 2.0f * mu;
            }
          }
          float kl = 0.f;
          if (jv) {                                // ppo.py:492-498, p0 = new, p1 = stored
            const float dm = omu - mu;
            kl = flog(osg * rsig + 1e-5f) + (sigma * sigma + dm * dm) * (0.5f * frcp(osg * osg + 1e-5f)) - 0.5f;
          }
          nl = group_sum16(nl);
          ent = group_sum16(ent);
          bl = group_sum16(bl);
          kl = group_sum16(kl);
          const float oldnl = in_nl[it], adv = in_adv[it];
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
          if (rv) {
            float* sd = hp.seeds[0] + (int64_t)row * 32;
            if (jv) {
              sd[j] = s_mu;
              sd[A + j] = s_ls;
              cs0 += s_mu;
              cs1 += s_ls;
              hp.mu[(int64_t)row * A + j] = mu;                // experience.py:83-87 write-back
              hp.sigma[(int64_t)row * A + j] = sigma;
            }
            for (int c = 2 * A + j; c < 32; c += 16) sd[c] = 0.f;   // zero padding of the seed row
            if (j == 0) {
              m_al += al;
              m_ent += ent;
              m_bl += bl;
              m_kl += kl;
              m_clip += (fabsf(ratio - 1.0f) > hp.e_clip) ? 1.0f : 0.0f;
            }
          }
        }
      }
      if constexpr (EPI == EPI_HEADS_TRAIN) {
        const int part = wave * 4 + g;
        if (jv) {
          csum[part * 32 + j] = cs0;
          csum[part * 32 + A + j] = cs1;
        }
        if (j == 0) {
          double* m = red + part * MRL_NMETRIC;
          m[0] = (double)m_al; m[1] = (double)m_ent; m[2] = (double)m_bl; m[3] = (double)m_kl; m[4] = (double)m_clip;
        }
      }
    } else {
      // critic: column 0 of the tile is the value (models.py:147-149); lane l32 takes row l32, the two half-waves share
      // the stores of the zero-padded seed row
      const int lr = l32;
      const int row = wrow0 + lr;
      const bool rv = row < p.M;
      const int rowc = rv ? row : 0;
      const float o = T[lr * TS];
      if constexpr (EPI == EPI_HEADS_ACT) {
        if (rv && h == 0) {
          float v = o;
          if (hp.vrms_mean) {                    // normalizers.py:70-73 (clamp first, then un-scale)
            v = fminf(fmaxf(v, p.clamp_lo), p.clamp_hi);
            v = sqrtf((float)hp.vrms_var[0] + hp.vrms_eps) * v + (float)hp.vrms_mean[0];
          }
          hp.value_out[row] = v;
        }
      } else {
        const float ret = hp.returns[rowc], vp = hp.value_preds[rowc];
        const float v = o;
        const float e1 = v - ret;
        float cl = e1 * e1, dv = 2.0f * e1;      // ppo.py:466-473
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
        const float s_v = hp.inv_mb * 0.5f * hp.critic_coef * dv;   // ppo.py:285
        if (rv) {
          f32x4* sd = reinterpret_cast<f32x4*>(hp.seeds[1] + (int64_t)row * 32 + h * 16);
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          sd[0] = (h == 0) ? f32x4{s_v, 0.f, 0.f, 0.f} : z;
          sd[1] = z; sd[2] = z; sd[3] = z;
        }
        const bool mine = rv && h == 0;
        double dr = mine ? (double)ret - (double)v : 0.0;
        double q[5] = {mine ? (double)cl : 0.0, dr, dr * dr, mine ? (double)ret : 0.0, mine ? (double)ret * ret : 0.0};
        float cs = mine ? s_v : 0.f;
#pragma unroll
        for (int msk = 1; msk < 32; msk <<= 1) {
#pragma unroll
          for (int i = 0; i < 5; ++i) q[i] += __shfl_xor(q[i], msk);
          cs += __shfl_xor(cs, msk);
        }
        if (lane == 0) {
          double* m = red + (wave * 4) * MRL_NMETRIC;
#pragma unroll
          for (int i = 0; i < 5; ++i) m[5 + i] = q[i];
          csum[(wave * 4) * 32] = cs;
        }
      }
    }
    if constexpr (EPI == EPI_HEADS_TRAIN) {
      __syncthreads();
      if (tid < MRL_NMETRIC) {
        double s = 0.0;
        for (int w = 0; w < NPART; ++w) s += red[w * MRL_NMETRIC + tid];
        hp.metric_part[((int64_t)blockIdx.x * 2 + tw + p.tw0) * MRL_NMETRIC + tid] = s;   // slot of the tower's role (single-tower launches)
      }
      if (tid < 32) {
        float s = 0.f;
        for (int w = 0; w < NPART; ++w) s += csum[w * 32 + tid];
        hp.headb_part[((int64_t)blockIdx.x * 2 + tw + p.tw0) * 32 + tid] = s;
      }
    }
  }
  MRL_KT(1002);
  MRL_KB(1);
}

}  // namespace mrl
