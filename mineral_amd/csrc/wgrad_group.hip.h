// Every weight gradient of a small-batch step in ONE launch on the bf16 matrix cores: the tiles, row ranges and slabs of
// wgrad_tile_kernel (chain.hip.h) with the arithmetic of wgrad_split_kernel (split_bf16.hip.h): each fp32 operand element is
// split exactly into three bf16 terms and every product is accumulated in fp32 from six of the nine term products, which is
// closer to the exact sum than an fp32 fma chain and costs 2.7 times fewer matrix-core cycles than v_mfma_f32_16x16x4_f32.
//
// A block is 8 waves = two groups of 4 (2 x 2 waves of 64 x 64, or 64 x 32 for 64-column tiles) that work on the two halves of
// the block's row range with LDS stages of their own, so that every SIMD holds two waves (one wave's operand splitting runs
// under the other's MFMAs) without doubling the number of slabs; the second group hands its accumulators to the first through
// LDS at the end.  Per 16-row step a group stages both operands as three bf16 planes ([16 rows][columns] each, the reduction
// index along the rows) and reads the k-contiguous MFMA fragments back with ds_read_b64_tr_b16.
//
// Shapes: output rows in tiles of 128 (rows past the matrix are computed from whatever follows it in the workspace and not
// stored), columns a multiple of 128, or 64.  Operands as stored by the forward chain (no prologue).
#pragma once
#include "chain.hip.h"
#include "split_bf16.hip.h"

namespace mrl {

constexpr int WGG_LDS = 2 * 2 * (2 * 3 * wsplit::PLANE);   // bytes: two groups x two stages x (A planes + B planes) = 122,880

template <int BNT>
__device__ __forceinline__ void wgrad_group_task(const WgTileP& p, const WgTileDesc& d, int split, int m0, int n0, char* lds_all) {
  using namespace wsplit;
  constexpr int NTH = 256;                              // threads per group
  constexpr int TN = BNT / 64;                          // 32-column MFMA tiles per wave
  constexpr int AROW = ROWB, APLANE = BK * AROW;
  constexpr int BROW = BNT == 128 ? ROWB : 192, BPLANE = BK * BROW;
  constexpr int AOPER = 3 * APLANE;
  constexpr int WSTAGE = 2 * OPER;                      // a stage of a group (B planes of 64-column tiles leave part of it unused)
  constexpr int A_CPR = 32, B_CPR = BNT / 4;            // 4-float chunks per row
  constexpr int A_F4 = BK * A_CPR / NTH;                // A chunks per thread and step: 2
  constexpr int B_F4 = BK * B_CPR / NTH;                // B chunks per thread and step: 2 or 1
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int t = tid & 255, lane = tid & 63;
  const int grp = __builtin_amdgcn_readfirstlane(tid >> 8), wave = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
  const int wm = wave >> 1, wn = wave & 1;
  char* lds = lds_all + grp * 2 * WSTAGE;

  // the block's row range in 16-row steps, halved between the groups (both run `nsteps` steps; the second may end a step early,
  // rows past a group's end are staged as zeros)
  const int G = (p.M + BK - 1) / BK;
  const int s0 = (int)((int64_t)split * G / d.nsplit), s1 = (int)((int64_t)(split + 1) * G / d.nsplit);
  const int nsteps = (s1 - s0 + 1) >> 1;
  const int kbeg = (s0 + grp * nsteps) * BK;
  const int kend = grp == 0 ? min(p.M, (s0 + nsteps) * BK) : min(p.M, s1 * BK);
  const bool colsum = n0 == 0 && d.colpart != nullptr;   // the heads: column sums of the seeds

  const int c = t % A_CPR, kr = t / A_CPR;
  const int cb = t % B_CPR, krb = t / B_CPR;
  const rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.dz), 0, p.M * d.u * 4, 0x00020000);
  const rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.xin), 0, p.M * d.in * 4, 0x00020000);
  const int voffA = (kr * d.u + m0 + 4 * c) * 4, voffB = (krb * d.in + n0 + 4 * cb) * 4;
  f32x4 ra[2][A_F4], rb[2][B_F4];                       // [register set = step parity][row]
  f32x4 cz = {0.f, 0.f, 0.f, 0.f};
  auto gload = [&](int st, auto SET) {
    constexpr int S = decltype(SET)::value;
    const int k = kbeg + st * BK;                       // rows past the batch read as zeros (descriptor bounds)
#pragma unroll
    for (int i = 0; i < A_F4; ++i) ra[S][i] = ch_load16(rsA, voffA, (k + (NTH / A_CPR) * i) * d.u * 4);
#pragma unroll
    for (int i = 0; i < B_F4; ++i) rb[S][i] = ch_load16(rsB, voffB, (k + (NTH / B_CPR) * i) * d.in * 4);
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
      const f32x4 v = k + row < kend ? ra[S][i] : zero;
      if (colsum) cz += v;
      split4(v, h, m, l);
      *reinterpret_cast<u32x2*>(q) = h;
      *reinterpret_cast<u32x2*>(q + APLANE) = m;
      *reinterpret_cast<u32x2*>(q + 2 * APLANE) = l;
    }
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      u32x2 h, m, l;
      const int row = krb + (NTH / B_CPR) * i;
      char* q = buf + AOPER + row * BROW + cb * 8;
      split4(k + row < kend ? rb[S][i] : zero, h, m, l);
      *reinterpret_cast<u32x2*>(q) = h;
      *reinterpret_cast<u32x2*>(q + BPLANE) = m;
      *reinterpret_cast<u32x2*>(q + 2 * BPLANE) = l;
    }
  };

  // fragment map of v_mfma_f32_32x32x16_bf16 through transposed reads (see wgrad_split_kernel): 16-lane group g takes columns
  // 16 (g & 1) .. +15 and rows 8 (g >> 1) .. +7 of the step; lane 4 q + p of the group supplies the address of row q, columns
  // 4 p .. 4 p + 3
  const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int fa = (8 * (g >> 1) + q4) * AROW + (16 * (g & 1) + 4 * p4) * 2 + (wm * 64) * 2;
  const int fb = AOPER + (8 * (g >> 1) + q4) * BROW + (16 * (g & 1) + 4 * p4) * 2 + (wn * 32 * TN) * 2;
  auto tr_frag_p = [](const char* base, int pitch) {
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
  // Step st (parity P): read the fragments of step st from stage P, run the MFMAs of step st-1 from the other fragment set, split
  // the operands of step st+1 (register set P^1) into stage P^1, whose fragments were read a step ago, then refill that register
  // set with step st+3.  Steps past the end stage zeros and are never multiplied.  Both groups run the same number of steps, so
  // the barriers are the block's.
  auto steady = [&](int st, auto PAR) {
    constexpr int P = decltype(PAR)::value;
    using Q = std::integral_constant<int, P ^ 1>;
    frags(PAR);
    mfmas(Q{});
    stage(st + 1, Q{});
    gload(st + 3, Q{});
    __syncthreads();
  };
  if (nsteps > 0) {
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
  }
  __syncthreads();   // every wave is done with the stages: they become the exchange buffers

  // the second group's accumulators (and column sums) to the first, element e of thread t at [e][t]
  float* xbuf = reinterpret_cast<float*>(lds_all);            // 64 x 256 floats
  f32x4* cbuf = reinterpret_cast<f32x4*>(lds_all + 64 * 256 * 4);   // [2 groups][8 rows][32 chunks]
  if (grp == 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn)
#pragma unroll
        for (int r = 0; r < 16; ++r) xbuf[((i * TN + jn) * 16 + r) * 256 + t] = acc[i][jn][r];
  }
  if (colsum) cbuf[(grp * 8 + kr) * 32 + c] = cz;
  __syncthreads();
  if (grp == 0) {
    float* out = d.slab + (int64_t)split * d.u * d.in;
    const int col = n0 + wn * 32 * TN + (lane & 31), hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          const float v = acc[i][jn][r] + xbuf[((i * TN + jn) * 16 + r) * 256 + t];
          if (row < d.u && col + jn * 32 < d.in) out[(int64_t)row * d.in + col + jn * 32] = v;
        }
    if (colsum && t < 32 && m0 + 4 * t < d.u) {
      f32x4 s = cbuf[t];
#pragma unroll
      for (int k = 1; k < 16; ++k) s += cbuf[k * 32 + t];
      *reinterpret_cast<f32x4*>(d.colpart + (int64_t)split * 3 * d.u + 2 * d.u + m0 + 4 * t) = s;   // slot 2: d bias
    }
  }
}

__global__ __launch_bounds__(512) void wgrad_group_kernel(const WgTileP p) {
  extern __shared__ __attribute__((aligned(16))) char wgg_lds[];
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
  const int m0 = (tile / d.tiles_n) * 128, n0 = (tile % d.tiles_n) * d.bn;
  if (d.bn == 128) wgrad_group_task<128>(p, d, split, m0, n0, wgg_lds);
  else wgrad_group_task<64>(p, d, split, m0, n0, wgg_lds);
}

}  // namespace mrl
