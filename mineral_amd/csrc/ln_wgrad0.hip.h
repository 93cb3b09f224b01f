// LayerNorm + activation backward of a 512-wide FIRST layer fused with that layer's weight gradient.
//
// The first layer has no input gradient, so its dZ has exactly one consumer, the weight gradient dW0 = dZ^T * f(obs)
// (f = the input normalisation).  The streaming LayerNorm backward (ln_bwd_kernel<512>) reads dH and xhat and writes dZ, and
// the weight-gradient launch reads dZ again: 5 floats of traffic per element where 2 are needed.  Here a block of 12 waves
// walks a range of batch rows in steps of 16: it reads dH and xhat of the step (64 KB), finishes the LayerNorm backward in
// registers, splits dZ and the normalised observations into three bf16 planes each in LDS (the exact split of
// split_bf16.hip.h) and multiplies them on v_mfma_f32_32x32x16_bf16 into a [512][IN] accumulator that stays in registers
// for the whole range.  dZ never goes to memory (unless the caller asks for it: dz_out, for an input-gradient launch).
//
// The waves of a block have two roles (three waves per SIMD: two row waves and a matrix wave):
//   row waves 0 .. 7      wave w owns rows 2 w, 2 w + 1 of every step, a lane two 4-column chunks of each: dy = dh * act'(y),
//                         column sums of dy * xhat and dy, dxh = dy * gamma, the row sums of dxh and dxh * xhat over the wave
//                         (DPP adds: the wave holds whole rows), dz = rstd * (dxh - mean(dxh) - xhat * mean(dxh * xhat)),
//                         column sums of dz, split -> A planes of the step's LDS stage; the threads of waves 0 .. 3 also
//                         normalise and split one chunk of the step's input tile -> B planes.  A row's registers are
//                         refilled with the next step's row as soon as the row is done.
//   matrix waves 8 .. 11  wave m owns dZ columns 128 m .. +127: fragments of the stage finished one step earlier
//                         (transposed LDS reads), 24 TN MFMAs per step into 64 TN accumulator registers.
// The VALU work of the row waves and the MFMAs of the matrix wave of a SIMD run side by side; the two LDS stages alternate,
// one barrier per step.  Each role has its own loop, so neither holds the other's registers (168 per wave).
// Measured at 32,768 rows x 2 towers (tools/lnwg0_probe.hip): 63 us, 4.3 TB/s of dH + xhat, against 68 + 48 us for the two
// launches it replaces; without the MFMAs 58 us, with all arithmetic compiled out 46 us (the streaming rate of this shape).
//
// Outputs: slab[block][512][IN] (summed in block order by the slab reduction) and colpart[block][3][512] (d gamma, d beta,
// d bias), the layouts of wgrad_split_kernel and ln_bwd_kernel.
#pragma once

#ifndef LNWG0_ABL
#define LNWG0_ABL 0   // timing-only builds (tools/lnwg0_probe.hip): bit 0 no MFMAs, 1 no wave reductions, 2 no loads inside the loop, 3 no activation derivative, 4 no split of dZ
#endif

namespace mrl {

struct LnWg0Args {
  const float* dh[2];       // [rows][512] d loss / d layer output (what dgrad wrote)
  float* dz_out[2];         // [rows][512] or nullptr
  const float* xhat[2];
  const float* rstd[2];
  const float* gamma[2];
  const float* beta[2];
  const float* x;           // [rows][in] layer input
  int in;                   // inputs per row: a multiple of 4, at most IN (the columns up to IN are zero operands)
  const float* mean;        // input normalisation (nullptr: the input is used as it is)
  const float* std;
  float clamp_lo, clamp_hi;
  float* slab[2];           // [gridDim.x][512][in]
  float* colpart[2];        // [gridDim.x][3][512]
  int M, rows_per_block;    // rows_per_block a multiple of 16
};

// sum over the 64 lanes of the wave, in every lane
__device__ __forceinline__ float wave_total(float v) {
  v += dpp_take<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += dpp_take<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
  v += dpp_take<0x141, 0xf>(v);   // row_half_mirror
  v += dpp_take<0x140, 0xf>(v);   // row_mirror: every lane has the sum of its row of 16
  v += dpp_take<0x142, 0xa>(v);   // row_bcast15 into rows 1 and 3
  v += dpp_take<0x143, 0xc>(v);   // row_bcast31 into rows 2 and 3: lanes 48 .. 63 hold the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

#ifndef LNWG0_D_NT
#define LNWG0_D_NT 1      // 0: ordinary loads of dH (A/B builds)
#endif
constexpr int LNWG0_N = 512;
constexpr int LNWG0_RW = 8;                            // row waves (two rows of every step each), then four matrix waves
constexpr int LNWG0_THREADS = 64 * (LNWG0_RW + 4);
template <int IN>
struct LnWg0Lds {
  static constexpr int AROW = LNWG0_N * 2 + 64;        // row pitch of an A plane: four consecutive rows cover the 64 banks
  static constexpr int APLANE = wsplit::BK * AROW;
  static constexpr int AOPER = 3 * APLANE;
  static constexpr int BROW = IN * 2 + 64;
  static constexpr int BPLANE = wsplit::BK * BROW;
  static constexpr int WSTAGE = AOPER + 3 * BPLANE;
  static constexpr int BYTES = 2 * WSTAGE;             // (the column sums of the row waves pass through the stages at the end)
};

template <int IN, bool ELU>
__global__ __launch_bounds__(LNWG0_THREADS, 1) void lnbwd_wgrad0_kernel(const LnWg0Args a) {
  using namespace wsplit;
  using Ld = LnWg0Lds<IN>;
  constexpr int N = LNWG0_N, TN = IN / 32;
  constexpr int AROW = Ld::AROW, APLANE = Ld::APLANE, AOPER = Ld::AOPER, BROW = Ld::BROW, BPLANE = Ld::BPLANE, WSTAGE = Ld::WSTAGE;
  constexpr int B_CPR = IN / 4;                        // 4-float chunks per input row
  static_assert(IN == 64, "up to 64 observations per row: one chunk per thread of the first four row waves");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int t = threadIdx.x, lane = t & 63, tw = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r0 = blockIdx.x * a.rows_per_block, r1 = min(a.M, r0 + a.rows_per_block);
  const int nsteps = (r1 - r0 + BK - 1) / BK;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  constexpr int RW = LNWG0_RW, RPW = BK / RW;          // rows of a step per row wave
  static_assert(RW * RPW * 128 * 3 * 16 <= 2 * Ld::WSTAGE * RPW, "column sums fit the stages");
  if (wave < RW) {
    // ================= row waves: LayerNorm backward of rows 4 wave .. +3 of every step, operand planes into LDS =================
    const float* __restrict__ xh = a.xhat[tw];
    const float* __restrict__ dh = a.dh[tw];
    const float* __restrict__ rs = a.rstd[tw];
    const float* __restrict__ xin = a.x;
    float* __restrict__ dzo = a.dz_out[tw];
    f32x4 g[2], b[2], cgr[2], cbr[2], czr[2];          // column sums of dy * xhat, dy and dz over this wave's rows
    cgr[0] = cgr[1] = cbr[0] = cbr[1] = czr[0] = czr[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      g[c] = *reinterpret_cast<const f32x4*>(a.gamma[tw] + 4 * (lane + 64 * c));
      b[c] = *reinterpret_cast<const f32x4*>(a.beta[tw] + 4 * (lane + 64 * c));
    }
    const int cbk = t % B_CPR, krb = t / B_CPR;        // this thread's chunk of the step's [16][IN] input tile (threads below 16 IN / 4)
    const bool has_b = wave < BK * B_CPR / 64, in_b = 4 * cbk < a.in;   // in_b: the chunk exists in the input rows
    const bool norm = a.mean != nullptr;
    f32x4 pv0 = {0.f, 0.f, 0.f, 0.f}, pv1 = {1.f, 1.f, 1.f, 1.f};
    if (norm && in_b) {
      pv0 = *reinterpret_cast<const f32x4*>(a.mean + 4 * cbk);
      pv1 = *reinterpret_cast<const f32x4*>(a.std + 4 * cbk);
    }
    const float invn = 1.0f / (float)N;
    f32x4 x[RPW][2], d[RPW][2], ob;                        // [row][chunk]; a row is refilled with the next step's row as soon as it is done
    float rr[RPW];
    auto gload_row = [&](int st, int i) __attribute__((always_inline)) {
      const int64_t row = min(r0 + st * BK + RPW * wave + i, a.M - 1);   // rows past the range are loaded clamped and zeroed below
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        x[i][c] = *reinterpret_cast<const f32x4*>(xh + row * N + 4 * (lane + 64 * c));
        // dH is read here for the last time: a non-temporal load leaves its 134 MB first in line for eviction (update
        // 215.5 -> 212.5 ms on one box, 209.1 -> 207.0 on another; the same hint on the x-hat loads as well: 213.9)
        if constexpr (LNWG0_D_NT) d[i][c] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dh + row * N + 4 * (lane + 64 * c)));
        else d[i][c] = *reinterpret_cast<const f32x4*>(dh + row * N + 4 * (lane + 64 * c));
      }
      rr[i] = rs[row];
    };
    auto gload_in = [&](int st) __attribute__((always_inline)) {
      if (has_b && in_b) ob = *reinterpret_cast<const f32x4*>(xin + (int64_t)min(r0 + st * BK + krb, a.M - 1) * a.in + 4 * cbk);
    };
    auto step = [&](int st, auto PAR) __attribute__((always_inline)) {
      constexpr int S = decltype(PAR)::value;
      char* buf = lds + S * WSTAGE;
      const int base = r0 + st * BK;
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int row = base + RPW * wave + i;
        if (row >= r1) {                                // a row past the range contributes nothing anywhere (rare: a real branch)
          asm volatile("" ::: "memory");
          d[i][0] = d[i][1] = zero;
        }
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xv = x[i][c][e];
            const float y = xv * g[c][e] + b[c][e];
            float dact;
            if constexpr (LNWG0_ABL & 8) dact = y;
            else if constexpr (ELU) dact = y > 0.f ? 1.0f : __builtin_amdgcn_exp2f(1.44269504088896340736f * y);
            else {
              const float sg = sigmoid_fast(y);
              dact = sg * (1.0f + y * (1.0f - sg));
            }
            const float dy = d[i][c][e] * dact;
            cgr[c][e] += dy * xv;
            cbr[c][e] += dy;
            const float dxh = dy * g[c][e];
            d[i][c][e] = dxh;
            t1 += dxh;
            t2 += dxh * xv;
          }
        }
        float m1, m2;                                   // the wave holds the whole row: its sums are the row's
        if constexpr (LNWG0_ABL & 2) {
          m1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t1))) * invn;
          m2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t2))) * invn;
        } else {
          m1 = wave_total(t1) * invn;
          m2 = wave_total(t2) * invn;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          f32x4 z;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            z[e] = rr[i] * (d[i][c][e] - m1 - x[i][c][e] * m2);
            czr[c][e] += z[e];
          }
          if (dzo && row < r1) *reinterpret_cast<f32x4*>(dzo + (int64_t)row * N + 4 * (lane + 64 * c)) = z;
          u32x2 h, m, l;
          if constexpr (LNWG0_ABL & 16) { h = u32x2{__float_as_uint(z[0]), __float_as_uint(z[1])}; m = u32x2{__float_as_uint(z[2]), __float_as_uint(z[3])}; l = h; }
          else split4(z, h, m, l);
          char* qa = buf + (RPW * wave + i) * AROW + (lane + 64 * c) * 8;
          *reinterpret_cast<u32x2*>(qa) = h;
          *reinterpret_cast<u32x2*>(qa + APLANE) = m;
          *reinterpret_cast<u32x2*>(qa + 2 * APLANE) = l;
        }
        if (!(LNWG0_ABL & 4) || a.M < 0) gload_row(st + 1, i);
        // the sums take one row at a time: left to itself the compiler postpones the additions and keeps the addends of all
        // four rows in registers (+60 registers per sum, spills)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(cgr[c][e]), "+v"(cbr[c][e]), "+v"(czr[c][e]));

      }
      if (has_b) {
        f32x4 v = ob;
        if (norm) {                                     // mineral/common/normalizers.py:65-68
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf((v[e] - pv0[e]) / pv1[e], a.clamp_lo), a.clamp_hi);
        }
        u32x2 h, m, l;
        split4((base + krb < r1 && in_b) ? v : zero, h, m, l);
        char* qb = buf + AOPER + krb * BROW + cbk * 8;
        *reinterpret_cast<u32x2*>(qb) = h;
        *reinterpret_cast<u32x2*>(qb + BPLANE) = m;
        *reinterpret_cast<u32x2*>(qb + 2 * BPLANE) = l;
      }
      gload_in(st + 1);
    };
#pragma unroll
    for (int i = 0; i < RPW; ++i) gload_row(0, i);
    gload_in(0);
    int st = 0;
    for (; st + 1 < nsteps; st += 2) {
      step(st, I0{});
      __syncthreads();
      step(st + 1, I1{});
      __syncthreads();
    }
    if (st < nsteps) {
      step(st, I0{});
      __syncthreads();
    }
    __syncthreads();                                    // the matrix waves multiply the last step
    f32x4* csum = reinterpret_cast<f32x4*>(lds) + wave * 3 * 128 + lane;   // [row wave][3][128 chunks]; the matrix waves are through with the stages
#pragma unroll
    for (int c = 0; c < 2; ++c) { csum[64 * c] = cgr[c]; csum[128 + 64 * c] = cbr[c]; csum[256 + 64 * c] = czr[c]; }
  } else {
    // ================= matrix waves: dZ columns 128 (wave - RW) .. +127 times the IN inputs, one step behind =================
    const int mw = wave - RW;
    const int gq = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int fa = (8 * (gq >> 1) + q4) * AROW + (16 * (gq & 1) + 4 * p4) * 2 + (mw * 128) * 2;
    const int fb = AOPER + (8 * (gq >> 1) + q4) * BROW + (16 * (gq & 1) + 4 * p4) * 2;
    auto tr_frag_p = [](const char* base, int pitch) __attribute__((always_inline)) {
      typedef s16x4 __attribute__((address_space(3))) * lds_p;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + 4 * pitch));
      return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    f32x16 acc[4][TN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;
    auto multiply = [&](const char* buf) __attribute__((always_inline)) {
      bf16x8 fra[4][3], frb[TN][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fra[i][pl] = tr_frag_p(buf + fa + pl * APLANE + i * 64, AROW);
#pragma unroll
        for (int i = 0; i < TN; ++i) frb[i][pl] = tr_frag_p(buf + fb + pl * BPLANE + i * 64, BROW);
      }
      if constexpr (LNWG0_ABL & 1) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { acc[0][0][pl] += (float)fra[0][pl][0] + (float)fra[3][pl][1]; acc[1][1][pl] += (float)frb[0][pl][0] + (float)frb[1][pl][1]; }
        return;
      }
      constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
#pragma unroll
      for (int term = 0; term < 6; ++term)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[i][TA[term]], frb[jn][TB[term]], acc[i][jn], 0, 0, 0);
    };
    __syncthreads();                                    // step 0 is in stage 0
    for (int st = 0; st < nsteps; ++st) {
      multiply(lds + (st & 1) * WSTAGE);
      __syncthreads();                                  // step st + 1 is staged; stage st & 1 may be written again
    }
    float* out = a.slab[tw] + (int64_t)blockIdx.x * N * a.in;
    const int col = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mw * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (jn * 32 + col < a.in) out[row * a.in + jn * 32 + col] = acc[i][jn][r];
        }
  }
  __syncthreads();
  if (t < 3 * 128) {
    const f32x4* cs = reinterpret_cast<const f32x4*>(lds);
    const int k = t >> 7, jj = t & 127;
    f32x4 v = cs[k * 128 + jj];                         // the row waves in index order
#pragma unroll
    for (int w = 1; w < RW; ++w) v += cs[(w * 3 + k) * 128 + jj];
    *reinterpret_cast<f32x4*>(a.colpart[tw] + (int64_t)blockIdx.x * 3 * N + k * N + 4 * jj) = v;
  }
}

}  // namespace mrl
