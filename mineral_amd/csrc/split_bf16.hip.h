// fp32-accurate GEMMs on the bf16 matrix cores of gfx950: the weight gradients and the input gradients of the wide layers.
//
//   wgrad:  slab[split][m][n] = sum over the split's batch rows k of A[k][m] * f(B[k][n])    A = dZ, B = the layer input, f = prologue
//   dgrad:  out[m][n] = sum_k A[m][k] * B[k][n]                                              A = dZ of the layer above, B = its weights
//
// Every fp32 operand is split exactly into three bf16 terms x = h + m + l (8 significant bits each, rounded to nearest; bf16
// has the exponent range of fp32, so small values do not underflow), and each product is accumulated in fp32 from the six
// terms hh, hm, mh, mm, hl, lh; the three dropped terms are below 2^-23 of |a*b| with either sign, the size of the rounding
// of an fp32 product itself.  Against
// an fp64 sum the result is closer than a sequential fp32 fma chain (1.4e-8 against 1.5e-7 of sum|a*b| over 32,768 rows,
// tools/bf16x3_split_probe.hip).  Six v_mfma_f32_32x32x16_bf16 (16 k each) replace eight v_mfma_f32_32x32x2_f32 twice as long.
// Non-finite operands: an infinite element splits into (inf, NaN, NaN), so it contributes NaN where the fp32 MFMA would give
// +-inf; NaN stays NaN.  Finite operands below 3.38e38 in magnitude are exact.
//
// Tiles: 128 x 128 per 256-thread block (2 x 2 waves of 64 x 64), 16 reduction indices per step.  An operand whose rows are
// the reduction index keeps each plane as an LDS image [16 rows][128 columns] of bf16 in its own row-major order, and the
// k-contiguous MFMA fragments come out of it through ds_read_b64_tr_b16; an operand with the reduction index along its rows
// keeps [128 rows][16 k] and is read with ds_read_b128.  Per step the fragments of step st are read while the MFMAs of step
// st-1 run from the other fragment set and the operands of step st+1 are split and written to the other LDS stage; global
// loads run three steps ahead.  One barrier per step.
//
// Grids, split ranges and output layouts are those of the EPI_SLAB launches of gemm_kernel (group_mode 1 for wgrad, 0 for
// dgrad), so plans, workspace and the slab reduction do not change.  dgrad needs N to be a multiple of 128 and K of 16; wgrad
// needs M to be a multiple of 128 and N a multiple of 128 or equal to 64.
#pragma once

#include <type_traits>

namespace mrl {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace wsplit {
constexpr int BM = 128, BN = 128, BK = 16;
constexpr int ROWB = 320;           // bytes per row of a plane: 128 bf16 + 64 B, so that four consecutive rows cover the 64 banks
constexpr int PLANE = BK * ROWB;    // 5,120 B
constexpr int OPER = 3 * PLANE;     // three planes per operand

// x (4 floats) -> three packed bf16 quadruples with x = h + m + l exactly (split_rne4, gemm.hip.h)
__device__ __forceinline__ void split4(const f32x4 x, u32x2& h, u32x2& m, u32x2& l) { split_rne4(x, h, m, l); }

// Two transposed 4-row blocks (rows k .. k+3 and k+4 .. k+7 of this lane's column) make one MFMA fragment.  Every lane of
// the wave must take part (the gather crosses lanes): callers keep the whole block active.
__device__ __forceinline__ bf16x8 tr_frag(const char* base) {
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + 4 * ROWB));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
}  // namespace wsplit

// BNT: output columns per tile, 128 (waves of 64 x 64) or 64 (waves of 64 x 32, for a first layer on 64 observations; the B
// planes then take a 192-byte row pitch, with which four consecutive rows still cover all banks).
// BMT: output rows per tile, 128 (2 x 2 waves) or 256 (4 x 2 waves, 512 threads: a tile of the layer input is then transformed
// and split once for 256 output rows instead of once per 128, which is where most of this kernel's VALU work goes).
#ifndef MRL_SLAB_NT
#define MRL_SLAB_NT 1     // 0: ordinary stores for the weight-gradient slabs (A/B builds)
#endif
#ifndef MRL_ABLATE_WGRAD
#define MRL_ABLATE_WGRAD 0   // diagnostic builds only: bit 0 no slab stores, 1 no prologue / split, 3 one step
#endif
template <int PRO, int BNT, int BMT>
__global__ __launch_bounds__(BMT * 2, 2) void wgrad_split_kernel(const GemmP p) {
  using namespace wsplit;
  constexpr int WABL = MRL_ABLATE_WGRAD;
  static_assert(BNT == 128 || BNT == 64, "column tiles of 128 or 64");
  static_assert(BMT == 128 || (BMT == 256 && BNT == 128), "row tiles of 128, or 256 x 128");
  constexpr int NTH = BMT * 2;                          // threads per block
  constexpr int TN = BNT / 64;                          // 32-column MFMA tiles per wave
  constexpr int AROW = BMT == 128 ? ROWB : 576;         // row pitch of an A plane (256 columns: 512 B + 64)
  constexpr int APLANE = BK * AROW;
  constexpr int BROW = BNT == 128 ? ROWB : 192;         // row pitch of a B plane
  constexpr int BPLANE = BK * BROW;
  constexpr int AOPER = 3 * APLANE;
  constexpr int WSTAGE = AOPER + 3 * BPLANE;            // A planes, then B planes
  constexpr int A_CPR = BMT / 4, B_CPR = BNT / 4;       // 4-float chunks per row
  constexpr int A_F4 = BK * A_CPR / NTH;                // A chunks per thread and step: 2
  constexpr int B_F4 = BK * B_CPR / NTH;                // B chunks per thread and step: 2 or 1
  static_assert(A_F4 * NTH == BK * A_CPR && B_F4 * NTH == BK * B_CPR && B_F4 >= 1, "whole chunks per thread");
  __shared__ __attribute__((aligned(16))) char lds[2 * WSTAGE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tw = blockIdx.z;
  // block -> (split, tile) as in gemm_kernel: the tiles of one split get ids that differ by 8 and share an XCD's L2
  const int gsz = p.m_tiles * p.n_tiles;
  const int lin = blockIdx.x, tq = lin >> 3, j = tq % gsz;
  const int split = (tq / gsz) * 8 + (lin & 7);
  if (split >= p.n_groups) return;                     // padding blocks of the last octet of splits (whole blocks leave)
  const int m0 = (j % p.m_tiles) * BMT, n0 = (j / p.m_tiles) * BNT;
  const int kfull = p.Kt[tw] ? p.Kt[tw] : p.K;
  const int kbeg = split * p.ksplit, kend = min(kfull, kbeg + p.ksplit);
  const int nsteps = (WABL & 8) ? 1 : (kend - kbeg + BK - 1) / BK;
  const float* __restrict__ Ag = p.A[tw];
  const float* __restrict__ Bg = p.B[tw];

  // staging map: A chunk c (4 columns) of rows kr + (NTH / A_CPR) i of each 16-row step, B chunk cb of rows krb + (NTH / B_CPR) i
  const int c = t % A_CPR, kr = t / A_CPR;
  const int cb = t % B_CPR, krb = t / B_CPR;
  constexpr bool PRO_VEC = PRO == PRO_NORM || PRO == PRO_LNSILU || PRO == PRO_LNELU;
  f32x4 pv0 = {0.f, 0.f, 0.f, 0.f}, pv1 = {1.f, 1.f, 1.f, 1.f};
  if constexpr (PRO_VEC) {                              // the feature index of a B column does not change along the reduction
    pv0 = *reinterpret_cast<const f32x4*>(p.pro0[tw] + n0 + 4 * cb);
    pv1 = *reinterpret_cast<const f32x4*>(p.pro1[tw] + n0 + 4 * cb);
  }
  auto transform = [&](f32x4 v) {
    f32x4 o = v;
    if constexpr (PRO == PRO_NORM) {                    // mineral/common/normalizers.py:65-68
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fminf(fmaxf((v[e] - pv0[e]) / pv1[e], p.clamp_lo), p.clamp_hi);
    } else if constexpr (PRO == PRO_ELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = v[e] > 0.f ? v[e] : __builtin_amdgcn_exp2f(1.44269504088896340736f * v[e]) - 1.0f;
    } else if constexpr (PRO == PRO_LNELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y = v[e] * pv0[e] + pv1[e];
        o[e] = y > 0.f ? y : __builtin_amdgcn_exp2f(1.44269504088896340736f * y) - 1.0f;
      }
    } else if constexpr (PRO == PRO_LNSILU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y = v[e] * pv0[e] + pv1[e];
        o[e] = y * sigmoid_fast(y);
      }
    }
    return o;
  };

  f32x4 ra[2][A_F4], rb[2][B_F4];                       // [register set = step parity][row]
  auto gload = [&](int st, auto SET) {
    constexpr int S = decltype(SET)::value;
    const int k = kbeg + st * BK;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int64_t kk = min(k + kr + (NTH / A_CPR) * i, kfull - 1);   // rows past the end are loaded clamped and zeroed when staged
      ra[S][i] = *reinterpret_cast<const f32x4*>(Ag + kk * p.lda + m0 + 4 * c);
    }
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      const int64_t kk = min(k + krb + (NTH / B_CPR) * i, kfull - 1);
      rb[S][i] = *reinterpret_cast<const f32x4*>(Bg + kk * p.ldb + n0 + 4 * cb);
    }
  };
  auto stage = [&](int st, auto SET) {
    constexpr int S = decltype(SET)::value;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    char* buf = lds + S * WSTAGE;
    const int k = kbeg + st * BK;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      u32x2 h, m, l;
      const int row = kr + (NTH / A_CPR) * i;
      char* q = buf + row * AROW + c * 8;
      if constexpr (WABL & 2) {
        h = u32x2{__float_as_uint(ra[S][i][0]), __float_as_uint(ra[S][i][1])};
        m = u32x2{__float_as_uint(ra[S][i][2]), __float_as_uint(ra[S][i][3])};
        l = h;
      } else
      split4(k + row < kend ? ra[S][i] : zero, h, m, l);
      *reinterpret_cast<u32x2*>(q) = h;
      *reinterpret_cast<u32x2*>(q + APLANE) = m;
      *reinterpret_cast<u32x2*>(q + 2 * APLANE) = l;
    }
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      u32x2 h, m, l;
      const int row = krb + (NTH / B_CPR) * i;
      char* q = buf + AOPER + row * BROW + cb * 8;
      f32x4 v = rb[S][i];
      if constexpr (WABL & 2) {
        h = u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])};
        m = u32x2{__float_as_uint(v[2]), __float_as_uint(v[3])};
        l = h;
      } else {
      if constexpr (PRO != PRO_NONE) v = transform(v);
      split4(k + row < kend ? v : zero, h, m, l);
      }
      *reinterpret_cast<u32x2*>(q) = h;
      *reinterpret_cast<u32x2*>(q + BPLANE) = m;
      *reinterpret_cast<u32x2*>(q + 2 * BPLANE) = l;
    }
  };

  // fragment map of v_mfma_f32_32x32x16_bf16 through transposed reads: 16-lane group g takes columns 16 (g & 1) .. +15 and
  // rows 8 (g >> 1) .. +7 of the step; lane 4q + p of the group supplies the address of row q, columns 4p .. 4p+3
  const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int fa = (8 * (g >> 1) + q4) * AROW + (16 * (g & 1) + 4 * p4) * 2 + (wm * 64) * 2;
  const int fb = AOPER + (8 * (g >> 1) + q4) * BROW + (16 * (g & 1) + 4 * p4) * 2 + (wn * 32 * TN) * 2;
  auto tr_frag_p = [](const char* base, int pitch) {    // tr_frag with a plane's own row pitch
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + 4 * pitch));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  bf16x8 fra[2][2][3], frb[2][TN][3];                  // [fragment set = step parity][tile][plane]
  auto frags = [&](auto SET) {
    constexpr int S = decltype(SET)::value;
    const char* cur = lds + S * WSTAGE;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int i = 0; i < 2; ++i) fra[S][i][pl] = tr_frag_p(cur + fa + pl * APLANE + i * 64, AROW);
#pragma unroll
      for (int i = 0; i < TN; ++i) frb[S][i][pl] = tr_frag_p(cur + fb + pl * BPLANE + i * 64, BROW);
    }
  };

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;
  auto mfmas = [&](auto SET) {
    constexpr int S = decltype(SET)::value;
    // six terms per accumulator, smallest first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[S][i][TA[term]], frb[S][jn][TB[term]], acc[i][jn], 0, 0, 0);
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // Step st (parity P): read the fragments of step st from stage P, run the MFMAs of step st-1 from the other fragment set,
  // split the operands of step st+1 (register set P^1) into stage P^1, whose fragments were read a step ago, then refill
  // that register set with step st+3.  Steps past the end of the split stage zeros and are never multiplied.
  auto steady = [&](int st, auto PAR) {
    constexpr int P = decltype(PAR)::value;
    using Q = std::integral_constant<int, P ^ 1>;
    frags(PAR);
    mfmas(Q{});
    stage(st + 1, Q{});
    gload(st + 3, Q{});
    __syncthreads();
  };

  gload(0, I0{});
  gload(1, I1{});
  stage(0, I0{});
  gload(2, I0{});
  __syncthreads();
  frags(I0{});
  stage(1, I1{});
  gload(3, I1{});
  __syncthreads();
  int st = 1;
  for (; st + 1 < nsteps; st += 2) {
    steady(st, I1{});
    steady(st + 1, I0{});
  }
  if (st < nsteps) {
    steady(st, I1{});
    mfmas(I1{});
  } else {
    mfmas(I0{});
  }

  float* out = p.out[tw] + (int64_t)split * p.M * p.ldo;
  const int col = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (!(WABL & 1) || acc[i][jn][r] == 1.2345e-30f) {
          // The slabs are read once, by the slab reduction at the end of the step, after several hundred MB of other traffic:
          // non-temporal stores keep them from pushing the operands of the launches in between out of the caches (update
          // 208.7 -> 205.9 ms on one box; the reduction itself 28.2 -> 29.5 us).  A run-time choice between the two kinds of
          // store is folded into one ordinary store by the compiler, hence the build-time switch.
          if constexpr (MRL_SLAB_NT) __builtin_nontemporal_store(acc[i][jn][r], &out[(int64_t)row * p.ldo + n0 + wn * 32 * TN + jn * 32 + col]);
          else out[(int64_t)row * p.ldo + n0 + wn * 32 * TN + jn * 32 + col] = acc[i][jn][r];
        }
      }
}

// Input gradient of a wide layer: out[m][n] = sum_k A[m][k] B[k][n] with A = dZ of the layer above ([rows][K], K along the
// row) and B = that layer's weights ([K][N]).  K must be a multiple of 16; rows past M are computed from clamped loads and
// not stored.
// WPL: the weights come from planes split beforehand (GemmP::Bpl) and go to LDS as they are.
// NT: the output leaves with non-temporal stores.  Chosen per launch by what reads it next: the streaming LayerNorm backward
// (one pass, in place) runs faster behind such stores (38.2 -> 35.2 us) and the launch itself no slower; the fused first-layer
// kernel, which reads the 134 MB right away, wants them in the caches (66 -> 76 us behind non-temporal stores).
#ifndef MRL_ABLATE_DGRAD
#define MRL_ABLATE_DGRAD 0   // diagnostic builds only (tools/ablate_dgrad.sh): bit 0 no output stores, 1 no split of A, 2 no MFMAs, 3 one step, 4 all blocks store to the same 128 rows
#endif
template <bool WPL, bool NT = false>
__global__ __launch_bounds__(256, 2) void dgrad_split_kernel(const GemmP p) {
  using namespace wsplit;
  constexpr int ABL = MRL_ABLATE_DGRAD;
  constexpr int APLANE = BM * 32;          // [128 rows][16 k] bf16, 32 B per row; the two 16-byte halves of a row swap places
  constexpr int AOPER = 3 * APLANE;        // on every other group of 8 rows, which spreads a quarter-wave's 16-byte reads over all banks
  constexpr int DSTAGE = AOPER + OPER;     // 27,648 B; two stages
  __shared__ __attribute__((aligned(16))) char lds[2 * DSTAGE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tw = blockIdx.z;
  // block -> (row tile, column tile): the column tiles of one row tile get ids that differ by 8 and share an XCD's L2
  const int gsz = p.n_tiles;
  const int lin = blockIdx.x, tq = lin >> 3, j = tq % gsz;
  const int grp = (tq / gsz) * 8 + (lin & 7);
  if (grp >= p.n_groups) return;                       // padding blocks of the last octet of row tiles (whole blocks leave)
  const int m0 = grp * BM, n0 = j * BN;
  const int K = p.Kt[tw] ? p.Kt[tw] : p.K;
  const int nsteps = (ABL & 8) ? 1 : K / BK;
  const float* __restrict__ Ag = p.A[tw];
  const float* __restrict__ Bg = p.B[tw];

  // A staging map: 4-float chunk ca of rows (t >> 2) and (t >> 2) + 64; B: chunk c of rows (t >> 5) and (t >> 5) + 8
  const int ca = t & 3, rowa = t >> 2;
  const int c = t & 31, kr = t >> 5;
  auto a_off = [](int row, int half) { return row * 32 + ((half ^ ((row >> 3) & 1)) * 16); };
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  f32x4 ra[2][2], rb[2][WPL ? 1 : 2];                   // [register set = step parity][row]
  u32x4 rw[2][WPL ? 3 : 1];                             // WPL: one 16-byte chunk (8 columns) of row t >> 4 per plane
  const unsigned short* Wpl = p.Bpl[tw];
  const int wrow = t >> 4, wch = t & 15;
  auto gload = [&](int st, auto SET) {
    constexpr int S = decltype(SET)::value;
    const int k0 = min(st * BK, K - BK);                // steps past the end re-read the last one; they are never multiplied
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t row = min(m0 + rowa + 64 * i, p.M - 1);
      ra[S][i] = *reinterpret_cast<const f32x4*>(Ag + row * p.lda + k0 + 4 * ca);
      if constexpr (!WPL) rb[S][i] = *reinterpret_cast<const f32x4*>(Bg + (int64_t)(k0 + kr + 8 * i) * p.ldb + n0 + 4 * c);
    }
    if constexpr (WPL) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        rw[S][pl] = *reinterpret_cast<const u32x4*>(Wpl + pl * p.pl_stride + (int64_t)(k0 + wrow) * p.ldb + n0 + 8 * wch);
    }
  };
  auto stage = [&](auto SET) {
    constexpr int S = decltype(SET)::value;
    char* buf = lds + S * DSTAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x2 h, m, l;
      char* qa = buf + a_off(rowa + 64 * i, ca >> 1) + (ca & 1) * 8;
      if constexpr (ABL & 2) {
        h = u32x2{__float_as_uint(ra[S][i][0]), __float_as_uint(ra[S][i][1])};
        m = u32x2{__float_as_uint(ra[S][i][2]), __float_as_uint(ra[S][i][3])};
        l = h;
      } else
      split4(ra[S][i], h, m, l);
      *reinterpret_cast<u32x2*>(qa) = h;
      *reinterpret_cast<u32x2*>(qa + APLANE) = m;
      *reinterpret_cast<u32x2*>(qa + 2 * APLANE) = l;
      if constexpr (!WPL) {
        char* qb = buf + AOPER + (kr + 8 * i) * ROWB + c * 8;
        split4(rb[S][i], h, m, l);
        *reinterpret_cast<u32x2*>(qb) = h;
        *reinterpret_cast<u32x2*>(qb + PLANE) = m;
        *reinterpret_cast<u32x2*>(qb + 2 * PLANE) = l;
      }
    }
    if constexpr (WPL) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(buf + AOPER + pl * PLANE + wrow * ROWB + wch * 16) = rw[S][pl];
    }
  };

  // fragments: A rows straight (lane l32 = row, half-wave = k half), B through transposed reads as in the weight gradient
  const int l32 = lane & 31, hh = lane >> 5;
  const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int fb = AOPER + (8 * (g >> 1) + q4) * ROWB + (16 * (g & 1) + 4 * p4) * 2 + (wn * 64) * 2;
  int fa[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) fa[i] = a_off(wm * 64 + i * 32 + l32, hh);
  bf16x8 fra[2][2][3], frb[2][2][3];                   // [fragment set = step parity][tile][plane]
  auto frags = [&](auto SET) {
    constexpr int S = decltype(SET)::value;
    const char* cur = lds + S * DSTAGE;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fra[S][i][pl] = *reinterpret_cast<const bf16x8*>(cur + pl * APLANE + fa[i]);
        frb[S][i][pl] = tr_frag(cur + fb + pl * PLANE + i * 64);
      }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;
  auto mfmas = [&](auto SET) {
    constexpr int S = decltype(SET)::value;
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
    if ((ABL & 4) && p.M > 0) {                         // keep the fragments alive without multiplying
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          acc[i][0][pl] += (float)fra[S][i][pl][0];
          acc[i][1][pl] += (float)frb[S][i][pl][0];
        }
      return;
    }
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[S][i][TA[term]], frb[S][jn][TB[term]], acc[i][jn], 0, 0, 0);
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto steady = [&](int st, auto PAR) {                 // the pipeline of wgrad_split_kernel
    constexpr int P = decltype(PAR)::value;
    using Q = std::integral_constant<int, P ^ 1>;
    frags(PAR);
    mfmas(Q{});
    stage(Q{});
    gload(st + 3, Q{});
    __syncthreads();
  };

  gload(0, I0{});
  gload(1, I1{});
  stage(I0{});
  gload(2, I0{});
  __syncthreads();
  frags(I0{});
  stage(I1{});
  gload(3, I1{});
  __syncthreads();
  int st = 1;
  for (; st + 1 < nsteps; st += 2) {
    steady(st, I1{});
    steady(st + 1, I0{});
  }
  if (st < nsteps) {
    steady(st, I1{});
    mfmas(I1{});
  } else {
    mfmas(I0{});
  }

  float* out = p.out[tw];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int srow = (ABL & 16) ? (row & 127) : row;   // timing-only: every block stores to the same 128 rows (cache-resident)
        if ((ABL & 1) ? (row < p.M && acc[i][jn][r] == 1.2345e-30f) : (row < p.M)) {
          if constexpr (NT) __builtin_nontemporal_store(acc[i][jn][r], &out[(int64_t)srow * p.ldo + n0 + wn * 64 + jn * 32 + l32]);
          else out[(int64_t)srow * p.ldo + n0 + wn * 64 + jn * 32 + l32] = acc[i][jn][r];
        }
      }
}

}  // namespace mrl
