// The tail of a PPO step on batches above the chain kernels' row count, for LayerNorm towers whose last layer has 128
// outputs: heads, losses and their derivatives, the heads' input gradient and the LayerNorm + activation backward of the
// last layer, in one launch (mineral/agents/ppo/ppo.py:277-303 on mineral/agents/ppo/models.py:93-107).
//
// The per-layer path ran three launches here -- the heads GEMM with the loss epilogue (reads xhat of the last layer), the
// narrow dgrad with the fused LayerNorm backward (reads the seeds and xhat again, writes dZ) -- each far from both the
// matrix and the memory roof: the GEMMs are tiny (128 x 32) and everything else is local to a batch row.  Here a WAVE owns
// 16 batch rows at a time and never meets another wave until the end of the block:
//   lane (r = lane % 16, g = lane / 16) holds features 16 t + 4 g + i (t < 8, i < 4) of row r: eight 16-byte loads of xhat
//   h = act(gamma xhat + beta) and act'                                  (gamma / beta from LDS)
//   head pre-activations T[col][row] = Wh h on v_mfma_f32_16x16x4_f32 (the product orientation of chain.hip.h: a lane
//     holds one batch row), Wh from an LDS image in fragment order -> a wave-local LDS tile
//   losses, metrics, seeds per (row, action) as in chain_heads_train, four rows per pass -> wave-local seed tile, memory
//   dH[feature][row] = Wh^T seeds on the same MFMA: the result has the layout of xhat in the lane's registers
//   LayerNorm backward: row sums over the lane's 32 features and the row's four lanes, dZ stored with 16-byte stores,
//   column sums of dY xhat, dY, dZ kept per lane (pinned per group: see ln_wgrad0.hip.h) and reduced once at the end
// Eight waves per CU run eight such groups side by side, which is what hides the load -> MFMA -> loss -> MFMA chain.
// Outputs: dZ of the last layer, seeds [rows][32] (the heads' weight gradient still comes from its own launch), mu / sigma
// write-back, and per block colpart [3][128], headb_part [2][32], metric_part [2][MRL_NMETRIC].
#pragma once

#ifndef TAIL_ABL
#define TAIL_ABL 0   // timing-only builds: bit 0 no weight-image gather, 1 no loss passes, 2 no column sums, 3 no dZ stores
#endif

namespace mrl {

constexpr int TAIL_U = 128;          // width of the last layer
constexpr int TAIL_WAVES = 4;
constexpr int TAIL_TS = 33;          // row pitch of the head-output tile
constexpr int TAIL_SS = 36;          // row pitch of the seed tile (16-byte aligned rows)
constexpr int TAIL_XS = 132;         // row pitch of the column-sum tile
static_assert(16 * TAIL_TS + 16 * TAIL_SS <= 16 * TAIL_XS, "the head and seed tiles fit the column-sum tile");

struct PpoTailP {
  HeadsP h;
  const float* xhat[2];
  const float* rstd[2];
  const float* gamma[2];
  const float* beta[2];
  const float* Wh[2];       // [head_n][128] row-major
  const float* bh[2];
  int head_n[2];
  float* dz[2];             // [rows][128]
  float* colpart[2];        // [gridDim.x][3][128]
  int M, rows_per_block;    // rows_per_block a multiple of 16
  int tw0;                  // tower of blockIdx.y == 0 (per-tower launches)
};

template <bool ELU>
__global__ __launch_bounds__(64 * TAIL_WAVES, 2) void ppo_tail_kernel(const PpoTailP p) {
  constexpr int U = TAIL_U, ACT = ELU ? CH_ACT_ELU : CH_ACT_SILU;
  __shared__ f32x4 WfL[2][8][64];                       // Wh as the A operand of the heads product: [col tile][t][lane] = Wh[16 ct + lane % 16][16 t + 4 (lane / 16) .. + 3]
  __shared__ f32x4 WbL[8][2][64];                       // Wh^T as the A operand of dH: [feature tile][sq][lane][i] = Wh[8 (lane / 16) + 4 sq + i][16 ft + lane % 16]
  __shared__ __attribute__((aligned(16))) float gb[2][U];
  __shared__ float biasL[32];
  // per wave: the head-output tile and the seed tile of a group, and later (both read by then) the [16 rows][128 features] tile
  // through which the column sums of the group are taken
  __shared__ __attribute__((aligned(16))) float Wt[TAIL_WAVES][16 * TAIL_XS];
  __shared__ __attribute__((aligned(16))) float colx[TAIL_WAVES][3][U];
  __shared__ float hbx[TAIL_WAVES][32];
  __shared__ double mx[TAIL_WAVES][MRL_NMETRIC];
  const int t_ = threadIdx.x, lane = t_ & 63, tw = blockIdx.y + p.tw0;
  const int wave = __builtin_amdgcn_readfirstlane(t_ >> 6);
  const int r = lane & 15, g = lane >> 4;
  const HeadsP& hp = p.h;
  const int A = hp.act_dim, Nv = p.head_n[tw];
  const float* __restrict__ Wh = p.Wh[tw];
  // ---- block prologue: the weight images, gamma / beta, the head bias ----
  for (int e = t_; e < 2 * 8 * 64; e += 64 * TAIL_WAVES) {
    const int ln = e & 63, tt = (e >> 6) & 7, ct = e >> 9;
    const int col = 16 * ct + (ln & 15);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (col < Nv) v = *reinterpret_cast<const f32x4*>(Wh + (int64_t)col * U + 16 * tt + 4 * (ln >> 4));
    WfL[ct][tt][ln] = v;
  }
  for (int e = t_; e < ((TAIL_ABL & 1) ? 0 : 8 * 2 * 64); e += 64 * TAIL_WAVES) {
    const int ln = e & 63, sq = (e >> 6) & 1, ft = e >> 7;
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = 8 * (ln >> 4) + 4 * sq + i;
      v[i] = c < Nv ? Wh[(int64_t)c * U + 16 * ft + (ln & 15)] : 0.f;
    }
    WbL[ft][sq][ln] = v;
  }
  for (int e = t_; e < U; e += 64 * TAIL_WAVES) {
    gb[0][e] = p.gamma[tw][e];
    gb[1][e] = p.beta[tw][e];
  }
  if (t_ < 32) biasL[t_] = t_ < Nv ? p.bh[tw][t_] : 0.f;
  __syncthreads();

  const float* __restrict__ xh = p.xhat[tw];
  const float* __restrict__ rsd = p.rstd[tw];
  float* __restrict__ dzp = p.dz[tw];
  float* Tw = Wt[wave];
  float* Sw = Wt[wave] + 16 * TAIL_TS + 4;   // (16 * 33 + 4 floats: a multiple of 16 bytes)
  float* Xw = Wt[wave];
  const int r0 = blockIdx.x * p.rows_per_block, r1 = min(p.M, r0 + p.rows_per_block);
  const int ngroups = (r1 - r0 + 15) >> 4;
  auto wave_sync = []() {                               // orders this wave's LDS writes before its later LDS reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  float cs[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};   // column sums of dY xhat, dY, dZ over this wave's rows: features 2 lane, 2 lane + 1
  // a [16][128] tile of the group (this lane's 32 values `v`) summed over its rows: through the wave's LDS tile, rows in index order
  auto colsum = [&](const f32x4* v, float* acc2) __attribute__((always_inline)) {
    if constexpr (TAIL_ABL & 4) { acc2[0] += v[0][0]; acc2[1] += v[7][3]; return; }
#pragma unroll
    for (int t = 0; t < 8; ++t) *reinterpret_cast<f32x4*>(Xw + r * TAIL_XS + 16 * t + 4 * g) = v[t];
    wave_sync();
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float2 e = *reinterpret_cast<const float2*>(Xw + q * TAIL_XS + 2 * lane);
      a0 += e.x;
      a1 += e.y;
    }
    acc2[0] += a0;
    acc2[1] += a1;
    wave_sync();
  };
  float hb = 0.f;                                       // lanes < 32: column sum of the seeds (head bias gradient)
  // a state-independent log-std is read once: selecting between the parameter's address and the LDS tile's inside the
  // loop makes the access a flat load, which waits for every outstanding store of the wave on each of the four passes
  const float fixed_ls = (hp.fixed_logstd && tw == 0) ? hp.fixed_logstd[r < A ? r : 0] : 0.f;
  double macc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  auto fexp = [](float x) { return __builtin_amdgcn_exp2f(1.44269504088896340736f * x); };
  auto flog = [](float x) { return 0.69314718055994530942f * __builtin_amdgcn_logf(x); };
  auto frcp = [](float x) { return __builtin_amdgcn_rcpf(x); };
  auto ftanh = [&](float x) {
    const float e = __builtin_amdgcn_exp2f(2.88539008177792681472f * fabsf(x));
    return copysignf(1.0f - 2.0f * frcp(1.0f + e), x);
  };

  for (int grp = wave; grp < ngroups; grp += TAIL_WAVES) {
    const int m0 = r0 + 16 * grp;
    const int row = m0 + r;
    const bool rv = row < r1;
    const int64_t rowc = min(row, p.M - 1);
    f32x4 x[8], da[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = *reinterpret_cast<const f32x4*>(xh + rowc * U + 16 * t + 4 * g);
    const float rs = rsd[rowc];
    // the rollout columns of the group's losses, requested before the products so that they arrive under them
    float in_a[4], in_mu[4], in_sg[4], in_nl[4], in_adv[4];
    float in_ret = 0.f, in_vp = 0.f;
    if (tw == 0) {
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int lrow = m0 + 4 * pass + g;
        const int64_t lrc = lrow < r1 ? lrow : 0;
        const bool jv = r < A;
        in_a[pass] = jv ? hp.actions[lrc * A + r] : 0.f;
        in_mu[pass] = jv ? hp.mu[lrc * A + r] : 0.f;
        in_sg[pass] = jv ? hp.sigma[lrc * A + r] : 1.f;
        in_nl[pass] = hp.old_neglogp[lrc];
        in_adv[pass] = hp.adv[lrc];
      }
    } else {
      in_ret = hp.returns[rowc];
      in_vp = hp.value_preds[rowc];
    }
    // ---- heads: T[col][row] = Wh . act(gamma xhat + beta) ----
    f32x4 accT[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(&gb[0][16 * t + 4 * g]);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(&gb[1][16 * t + 4 * g]);
      f32x4 h4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float y = x[t][i] * g4[i] + b4[i];
        h4[i] = ch_act<ACT>(y);
        da[t][i] = ch_dact<ACT>(y);
      }
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const f32x4 w4 = WfL[ct][t][lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) accT[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[i], h4[i], accT[ct], 0, 0, 0);
      }
    }
    // lane (g, r) holds T[16 ct + 4 g + i][row r]
    if (tw == 0) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) Tw[r * TAIL_TS + 16 * ct + 4 * g + i] = accT[ct][i] + biasL[16 * ct + 4 * g + i];
      wave_sync();
      // ---- actor losses: four rows per pass, one per 16-lane group, lane j of the group action j (chain_heads_train) ----
      const int j = r;
      const bool jv = j < A;
#pragma unroll
      for (int pass = 0; pass < ((TAIL_ABL & 2) ? 1 : 4); ++pass) {
        const int lr = 4 * pass + g, lrow = m0 + lr;
        const bool lrv = lrow < r1;
        const float a = in_a[pass], omu = in_mu[pass], osg = in_sg[pass], oldnl = in_nl[pass], adv = in_adv[pass];
        const float o = Tw[lr * TAIL_TS + j];
        const float ls = hp.fixed_logstd ? fixed_ls : Tw[lr * TAIL_TS + ((A + j) & 31)];
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
        if (!lrv || !jv) { s_mu = 0.f; s_ls = 0.f; }
        if (jv) {                                  // (with a state-independent log-std its seed still rides in column A + j: the
          Sw[lr * TAIL_SS + j] = s_mu;             // head has no such row, and the column sum is the parameter's gradient)
          Sw[lr * TAIL_SS + A + j] = s_ls;
        }
        for (int c = 2 * A + j; c < 32; c += 16) Sw[lr * TAIL_SS + c] = 0.f;
        if (lrv) {
          float* sd = hp.seeds[0] + (int64_t)lrow * 32;
          if (jv) {
            sd[j] = s_mu;
            sd[A + j] = s_ls;
            hp.mu[(int64_t)lrow * A + j] = mu;                  // experience.py:83-87 write-back
            hp.sigma[(int64_t)lrow * A + j] = sigma;
          }
          for (int c = 2 * A + j; c < 32; c += 16) sd[c] = 0.f;
        }
        if (j == 0 && lrv) {
          macc[0] += (double)al; macc[1] += (double)ent; macc[2] += (double)bl; macc[3] += (double)kl;
          macc[4] += (fabsf(ratio - 1.0f) > hp.e_clip) ? 1.0 : 0.0;
        }
      }
    } else {
      // ---- critic: column 0 is the value (models.py:147-149); ppo.py:466-473.  Lanes 0 .. 15 hold it for rows 0 .. 15 ----
      const float v = accT[0][0] + biasL[0];
      const bool mine = rv && g == 0;
      const float ret = in_ret, vp = in_vp;
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
      if (g == 0) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f}, first = {s_v, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) *reinterpret_cast<f32x4*>(Sw + r * TAIL_SS + 4 * ch) = ch == 0 ? first : z;
        if (rv) {
          f32x4* sd = reinterpret_cast<f32x4*>(hp.seeds[1] + (int64_t)row * 32);
#pragma unroll
          for (int ch = 0; ch < 8; ++ch) sd[ch] = ch == 0 ? first : z;
        }
      }
      if (mine) {
        const double dr = (double)ret - (double)v;
        macc[0] += (double)cl; macc[1] += dr; macc[2] += dr * dr; macc[3] += (double)ret; macc[4] += (double)ret * ret;
      }
    }
    wave_sync();
    // ---- head bias partials, dH = Wh^T seeds ----
    if (lane < 32) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) s += Sw[q * TAIL_SS + lane];
      hb += s;
    }
    const f32x4 sd0 = *reinterpret_cast<const f32x4*>(Sw + r * TAIL_SS + 8 * g);
    const f32x4 sd1 = *reinterpret_cast<const f32x4*>(Sw + r * TAIL_SS + 8 * g + 4);
    // dY = dH act' (in place of act')
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const f32x4 w0 = WbL[ft][0][lane], w1 = WbL[ft][1][lane];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[i], sd0[i], acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[i], sd1[i], acc, 0, 0, 0);
      // lane (g, r) holds dH[16 ft + 4 g + i][row r]: the layout of x
#pragma unroll
      for (int i = 0; i < 4; ++i) da[ft][i] *= acc[i];
    }
    wave_sync();                                        // the seed tile has been read: its memory becomes the column-sum tile
    {
      f32x4 v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = da[t] * x[t];
      colsum(v, cs[0]);
    }
    colsum(da, cs[1]);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(&gb[0][16 * t + 4 * g]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = da[t][i] * g4[i];
        da[t][i] = v;                                   // d xhat
        s1 += v;
        s2 += v * x[t][i];
      }
    }
    s1 = add_xor32(add_xor16(s1));
    s2 = add_xor32(add_xor16(s2));
    const float m1 = s1 * (1.0f / U), m2 = s2 * (1.0f / U);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int i = 0; i < 4; ++i) da[t][i] = rs * (da[t][i] - m1 - x[t][i] * m2);
      if (rv && (!(TAIL_ABL & 8) || da[t][0] == 1.2345e-30f)) *reinterpret_cast<f32x4*>(dzp + (int64_t)row * U + 16 * t + 4 * g) = da[t];
    }
    colsum(da, cs[2]);                                  // ends with a wave_sync: the next group may overwrite the tiles
  }

  // ---- block epilogue: the waves' sums in index order ----
#pragma unroll
  for (int k = 0; k < 3; ++k) *reinterpret_cast<float2*>(&colx[wave][k][2 * lane]) = float2{cs[k][0], cs[k][1]};
  if (lane < 32) hbx[wave][lane] = hb;
  // metrics: the actor's sums sit in lanes 0, 16, 32, 48 (action 0 of each row group), the critic's in lanes 0 .. 15
#pragma unroll
  for (int m = 0; m < 5; ++m) {
    double v = macc[m];
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    macc[m] = v;
  }
  if (lane == 0) {
#pragma unroll
    for (int m = 0; m < MRL_NMETRIC; ++m) mx[wave][m] = 0.0;
#pragma unroll
    for (int m = 0; m < 5; ++m) mx[wave][(tw == 0 ? 0 : 5) + m] = macc[m];
  }
  __syncthreads();
  for (int e = t_; e < 3 * U; e += 64 * TAIL_WAVES) {
    const int k = e / U, f = e % U;
    p.colpart[tw][(int64_t)blockIdx.x * 3 * U + e] = (colx[0][k][f] + colx[1][k][f]) + (colx[2][k][f] + colx[3][k][f]);
  }
  if (t_ < 32) hp.headb_part[((int64_t)blockIdx.x * 2 + tw) * 32 + t_] = (hbx[0][t_] + hbx[1][t_]) + (hbx[2][t_] + hbx[3][t_]);
  if (t_ < MRL_NMETRIC) hp.metric_part[((int64_t)blockIdx.x * 2 + tw) * MRL_NMETRIC + t_] = (mx[0][t_] + mx[1][t_]) + (mx[2][t_] + mx[3][t_]);
}

}  // namespace mrl
