// C ABI of the PPO hot path: streaming kernels (rollout write, GAE scan, running moments, Adam) and the launch
// sequences that drive the fp32 MFMA GEMM family of gemm.hip.h.  gfx950 only.
#include "../../include/mineral_ppo.h"
#include "gemm.hip.h"
#include "split_bf16.hip.h"
#include "ln_wgrad0.hip.h"
#include "ln_fwd0.hip.h"
#include "chain.hip.h"
#include "ppo_tail.hip.h"
#include "wgrad_group.hip.h"

#include <hip/hip_ext.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <mutex>
#include <utility>
#include <vector>

using namespace mrl;

static thread_local char g_err[512] = "";

#define MRL_FAIL(code, ...)                      \
  do {                                           \
    snprintf(g_err, sizeof(g_err), __VA_ARGS__); \
    return (code);                               \
  } while (0)

#define MRL_CHECK_LAUNCH(what)                                                            \
  do {                                                                                    \
    hipError_t e__ = hipGetLastError();                                                   \
    if (e__ != hipSuccess) MRL_FAIL(MRL_EHIP, "%s: %s", what, hipGetErrorString(e__));    \
  } while (0)

extern "C" const char* mrl_last_error(void) { return g_err; }

// for the other translation units of the library (replay.hip)
namespace mrl {
void set_error(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }
}  // namespace mrl
extern "C" int mrl_version(void) { return 1; }

// ---------------------------------------------------------------------------------------------------------------
// parameter / workspace layouts
// ---------------------------------------------------------------------------------------------------------------
static inline int64_t align4(int64_t x) { return (x + 3) & ~int64_t(3); }
static inline int64_t align64(int64_t x) { return (x + 63) & ~int64_t(63); }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

struct ParamLayout {
  int64_t W[2][MRL_MAX_LAYERS], b[2][MRL_MAX_LAYERS], g[2][MRL_MAX_LAYERS], be[2][MRL_MAX_LAYERS];
  int64_t Wh[2], bh[2];   // heads: [head_rows][U] / [head_bias_n] per tower
  int64_t total;
  int in_dim[MRL_MAX_LAYERS];
  int n_towers, head_rows[2];
  int head_bias_n[2];     // entries of the head's bias tensor: head_rows, plus the act_dim log-std entries when sigma is fixed
  bool has_norm;          // LayerNorm gain / bias per layer (absent in the plain ELU towers)
};

static int chain_mask();
static int check_net(const mrl_net* net) {
  if (!net) MRL_FAIL(MRL_EINVAL, "net is NULL");
  if (net->n_layers < 1 || net->n_layers > MRL_MAX_LAYERS) MRL_FAIL(MRL_EINVAL, "n_layers %d not in 1..%d", net->n_layers, MRL_MAX_LAYERS);
  if (net->obs_dim < 4 || net->obs_dim % 4) MRL_FAIL(MRL_EINVAL, "obs_dim %d must be a positive multiple of 4", net->obs_dim);
  if (net->act_dim < 1 || net->act_dim > 16) MRL_FAIL(MRL_EINVAL, "act_dim %d not in 1..16", net->act_dim);
  for (int l = 0; l < net->n_layers; ++l) {
    int u = net->units[l];
    if (!(u == 32 || u == 64 || u == 128 || u == 256 || u == 512)) MRL_FAIL(MRL_EINVAL, "units[%d]=%d must be 32/64/128/256/512", l, u);
  }
  if (net->dist_type != MRL_DIST_NORMAL && net->dist_type != MRL_DIST_DREAMERV3_NORMAL) MRL_FAIL(MRL_EINVAL, "dist_type %d unsupported", net->dist_type);
  if (net->tower_kind < 0 || net->tower_kind >= CH_NKINDS) MRL_FAIL(MRL_EINVAL, "tower_kind %d unsupported", net->tower_kind);
  if (net->tower_kind > MRL_TOWER_LAYERNORM_ELU && ((chain_mask() & 7) != 7 || net->obs_dim > 512))
    MRL_FAIL(MRL_EINVAL, "tower_kind %d runs through the chain kernels only (MRL_CHAIN bits 1, 2, 4; at most 512 observation columns)", net->tower_kind);
  if (net->fixed_sigma != 0 && net->fixed_sigma != 1) MRL_FAIL(MRL_EINVAL, "fixed_sigma %d not in {0, 1}", net->fixed_sigma);
  return MRL_OK;
}

// Flat layout of `n_towers` MLP towers ([W, b, ln_gamma, ln_beta] per layer, or [W, b] without a normalisation layer)
// followed by one head per tower ([head_rows][U] weights, [head_bias_n] bias); every tensor starts at a multiple of 4 floats.
static void param_layout_ex(int in_dim0, int n_layers, const int32_t* units, int n_towers, const int* head_rows, ParamLayout* L,
                            bool has_norm = true, const int* head_bias_n = nullptr) {
  int64_t o = 0;
  memset(L, 0, sizeof(*L));
  L->has_norm = has_norm;
  for (int tw = 0; tw < n_towers; ++tw) {
    int in = in_dim0;
    for (int l = 0; l < n_layers; ++l) {
      int u = units[l];
      L->in_dim[l] = in;
      L->W[tw][l] = o; o = align4(o + (int64_t)u * in);
      L->b[tw][l] = o; o = align4(o + u);
      if (has_norm) {
        L->g[tw][l] = o; o = align4(o + u);
        L->be[tw][l] = o; o = align4(o + u);
      }
      in = u;
    }
  }
  const int U = units[n_layers - 1];
  for (int tw = 0; tw < n_towers; ++tw) {
    L->head_rows[tw] = head_rows[tw];
    L->head_bias_n[tw] = head_bias_n ? head_bias_n[tw] : head_rows[tw];
    L->Wh[tw] = o; o = align4(o + (int64_t)head_rows[tw] * U);
    L->bh[tw] = o; o = align4(o + L->head_bias_n[tw]);
  }
  L->n_towers = n_towers;
  L->total = o;
}

constexpr int TOWER_LAYERNORM_ELU = MRL_TOWER_LAYERNORM_ELU;   // LayerNorm + ELU (also the critic of DFlexAntSHAC.yaml)

// ActorCritic: actor tower with the [mu; sigma] head (or the mu head with [mu bias; log-std] when sigma is fixed), critic
// tower with the value head
static void param_layout(const mrl_net* net, ParamLayout* L) {
  const int A = net->act_dim;
  const int hr[2] = {net->fixed_sigma ? A : 2 * A, 1};
  const int hb[2] = {2 * A, 1};
  param_layout_ex(net->obs_dim, net->n_layers, net->units, 2, hr, L, ch_kind_norm(net->tower_kind), hb);
}

extern "C" int64_t mrl_param_count(const mrl_net* net) {
  if (check_net(net)) return MRL_EINVAL;
  ParamLayout L;
  param_layout(net, &L);
  return L.total;
}

extern "C" int mrl_param_offsets(const mrl_net* net, int64_t* offs, int32_t* n) {
  if (check_net(net)) return MRL_EINVAL;
  if (!offs || !n) MRL_FAIL(MRL_EINVAL, "NULL output");
  ParamLayout L;
  param_layout(net, &L);
  int k = 0;
  for (int tw = 0; tw < 2; ++tw)
    for (int l = 0; l < net->n_layers; ++l) {
      offs[k++] = L.W[tw][l]; offs[k++] = L.b[tw][l];
      if (L.has_norm) { offs[k++] = L.g[tw][l]; offs[k++] = L.be[tw][l]; }
    }
  offs[k++] = L.Wh[0]; offs[k++] = L.bh[0]; offs[k++] = L.Wh[1]; offs[k++] = L.bh[1];
  *n = k;
  return MRL_OK;
}

static inline int fwd_bm(int n) { return n == 512 ? 64 : 128; }   // rows per block of the full-row (LayerNorm) tiles
// wide layers take the streaming LayerNorm backward: rows per block of that kernel
static inline bool ln_bwd_streamed(int n) { return n >= 256; }
// rows per block of the fused dgrad + LayerNorm backward of a narrow layer: 32-row tiles for small batches of the 128-wide layer
static inline int fused_bwd_rows(int u, int64_t rows) { return (u == 128 && rows <= 8192) ? 32 : 128; }
static inline int ln_bwd_rows(int64_t rows) { int64_t r = cdiv(rows, 512); return (int)(r < 4 ? 4 : cdiv(r, 4) * 4); }

constexpr int MRL_LDS_PER_CU = 160 * 1024;   // gfx950
constexpr int MRL_WGRAD_RESIDENT = 0;        // blocks per CU the split-K plan is sized for

struct WgradPlan { int bm, bn, tiles_m, tiles_n, nsplit, ksplit; };

// Batches of at most MRL_SMALL_M rows run the row-local chain kernels and the one-launch weight gradient of chain.hip.h.
// MRL_CHAIN in the environment (read once) is a bit mask for A/B measurements: 1 forward chain, 2 backward chain + weight
// gradients, 4 (with both) forward, heads and backward of a PPO step in one launch; default 7.  The workspace layout (column-partial blocks, slab counts) follows bit 1, so it cannot change while a
// process runs.
// (MRL_SMALL_M in the environment, read once, moves the boundary for A/B runs: tools/threshold_ab.sh)
static const int MRL_SMALL_M = [] { const char* e = getenv("MRL_SMALL_M"); return e ? atoi(e) : 8192; }();
static int chain_mask() {
  static const int m = [] { const char* e = getenv("MRL_CHAIN"); return e ? atoi(e) : 7; }();
  return m;
}
// MRL_CHAIN bit 8: the chain kernels read the weights from the parameter tensors as they are (no re-blocked copies, no launch that makes them)
static bool chain_natural() { return (chain_mask() & 8) != 0; }
static bool chain_ok(const mrl_net* net, int64_t rows, int bit) {
  // (tower kinds above MRL_TOWER_LAYERNORM_ELU have no other kernels: any row count)
  return (chain_mask() & bit) == bit && (rows <= MRL_SMALL_M || net->tower_kind > MRL_TOWER_LAYERNORM_ELU) && net->n_layers <= CH_MAXL &&
         net->obs_dim <= 512;   // LDS tiles of at most 512 columns
}
// A first layer of 512 LayerNorm outputs on at most 64 inputs, on batches above MRL_SMALL_M rows: its LayerNorm backward and its weight
// gradient are one launch (ln_wgrad0.hip.h); slabs and column partials are then per row range of that launch.
// MRL_FUSED_FIRST=0 in the environment (read once) keeps the two launches, for A/B runs.
static bool first_fused(const mrl_net* net, const int* in_dim, int64_t rows) {
  static const bool on = [] { const char* e = getenv("MRL_FUSED_FIRST"); return e ? atoi(e) != 0 : true; }();
  return on && rows > MRL_SMALL_M && net->units[0] == LNWG0_N && in_dim[0] <= 64 && in_dim[0] % 4 == 0 &&
         (net->tower_kind == MRL_TOWER_LAYERNORM_SILU || net->tower_kind == MRL_TOWER_LAYERNORM_ELU) && !chain_ok(net, rows, 2);
}
// rows per block of that launch: about 128 blocks per tower (one block per CU with two towers), whole 16-row steps
static int first_fused_rows(int64_t rows) { return (int)std::max<int64_t>(64, cdiv(cdiv(rows, 128), 16) * 16); }

// LayerNorm towers whose last layer has 128 outputs, on batches above MRL_SMALL_M rows: heads, losses, the heads' input gradient and
// the last layer's LayerNorm backward are one launch (ppo_tail.hip.h), one block per 128 rows (the row blocks of the launches
// it replaces, so the partial buffers keep their shapes).  MRL_PPO_TAIL=0 in the environment (read once) keeps the two launches.
static bool ppo_tail_ok(const mrl_net* net, const ParamLayout& L, int64_t rows) {
  static const bool on = [] { const char* e = getenv("MRL_PPO_TAIL"); return e ? atoi(e) != 0 : true; }();
  const int nl = net->n_layers;
  return on && rows > MRL_SMALL_M && nl >= 2 && net->units[nl - 1] == TAIL_U && net->act_dim <= 16 && L.head_rows[0] <= 32 && L.head_rows[1] <= 32 &&
         (net->tower_kind == MRL_TOWER_LAYERNORM_SILU || net->tower_kind == MRL_TOWER_LAYERNORM_ELU) && !chain_ok(net, rows, 2) &&
         fused_bwd_rows(TAIL_U, rows) == 128;
}
// 64 x 64 weight-gradient tiles of one tower (layers + head) and the row ranges they are split into: about 1,024 wave tasks
// for two towers (one per SIMD: the launch keeps one block of four waves per CU), at least 64 rows each, whole multiples of 32 rows
static int chain_wgrad_tiles(const mrl_net* net, const int* in_dim) {
  int t = 0;
  for (int l = 0; l < net->n_layers; ++l) t += (int)(cdiv(net->units[l], 64) * cdiv(in_dim[l], 64));
  return t + (int)cdiv(net->units[net->n_layers - 1], 64);
}
static void chain_wgrad_split(const mrl_net* net, const int* in_dim, int64_t rows, int* nsplit, int* ksplit) {
  static const int target = [] { const char* e = getenv("MRL_CHAIN_TASKS"); return e ? atoi(e) : 1024; }();
  const int tiles = 2 * chain_wgrad_tiles(net, in_dim);
  int64_t want = (target + tiles / 2) / tiles;
  if (want < 1) want = 1;
  int64_t ks = cdiv(cdiv(rows, want), 32) * 32;   // whole register groups of the weight-gradient kernel (32 rows)
  if (ks < 64) ks = 64;
  *ksplit = (int)ks;
  *nsplit = (int)cdiv(rows, ks);
}

// Block-tile weight gradients (wgrad_tile_kernel; MRL_CHAIN bit 16 keeps the wave-tile kernel): column blocks of 64 for inputs of at
// most 64 columns, else 128; row ranges per gradient so that the launch has at most one block per CU (MRL_CHAIN_TASKS, default 256)
// and every block about the same number of MFMAs: gradients with 64-column tiles get half as many ranges.  Index n_layers = the heads.
struct ChainWgPlan { int bm[MRL_MAX_LAYERS + 1], bn[MRL_MAX_LAYERS + 1], nsplit[MRL_MAX_LAYERS + 1]; bool split_bf16; };
static bool chain_tile_wgrad(const mrl_net* net, int64_t rows) { return chain_ok(net, rows, 3) && (chain_mask() & 16) == 0; }
static ChainWgPlan chain_wgrad_plan(const mrl_net* net, const int* in_dim, int64_t rows) {
  static const int target = [] { const char* e = getenv("MRL_CHAIN_TASKS"); return e ? atoi(e) : 256; }();
  ChainWgPlan pl;
  const int nl = net->n_layers;
  int tiles[MRL_MAX_LAYERS + 1], cost[MRL_MAX_LAYERS + 1];   // cost of a tile in quarters of a 128 x 128 one
  // every gradient with 64 or a multiple of 128 columns: the launch on the bf16 matrix cores (wgrad_group_kernel, tiles of 128 rows;
  // MRL_CHAIN bit 32 keeps the fp32 one)
  pl.split_bf16 = (chain_mask() & 32) == 0;
  for (int l = 0; l <= nl; ++l) {
    const int in = l == nl ? net->units[nl - 1] : in_dim[l];
    if (in != 64 && in % 128 != 0) pl.split_bf16 = false;
  }
  for (int l = 0; l <= nl; ++l) {
    const int u = l == nl ? 32 : net->units[l], in = l == nl ? net->units[nl - 1] : in_dim[l];
    pl.bm[l] = (u <= 32 && in > 64 && !pl.split_bf16) ? 32 : 128;
    pl.bn[l] = in <= 64 ? 64 : 128;
    // (the split-bf16 kernel's step is bound by the operand split and the barrier, not by its MFMAs: every tile shape costs the same)
    cost[l] = pl.split_bf16 ? 4 : (pl.bm[l] == 32 ? 1 : (pl.bn[l] == 64 ? 2 : 4));
    tiles[l] = (int)(cdiv(u, pl.bm[l]) * cdiv(in, pl.bn[l]));
  }
  const int maxs = (int)std::max<int64_t>(1, cdiv(rows, 32) / 2);   // at least 64 rows per range
  // the launch also carries the blocks that sum the backward chain's column sums (WgColDesc): everything in one wave of blocks
  int col_blocks = 0;
  for (int l = 0; l < nl; ++l) col_blocks += 2 * (int)cdiv((int64_t)(3 * net->units[l] / 4) * std::min<int64_t>(8, cdiv(rows, CH_ROWS)), 1024);
  auto splits_of = [&](int l, int sp) { return std::max(1, (sp * cost[l] + 3) / 4); };
  int best = 1;
  for (int sp = 1; sp <= maxs; ++sp) {
    int total = 0;
    for (int l = 0; l <= nl; ++l) total += 2 * tiles[l] * splits_of(l, sp);
    if (total + col_blocks <= target) best = sp;
  }
  for (int l = 0; l <= nl; ++l) pl.nsplit[l] = splits_of(l, best);
  return pl;
}

static WgradPlan wgrad_plan(int gm, int gn, int64_t rows) {
  WgradPlan w;
  w.bm = (gm <= 32) ? 32 : 128;
  w.bn = (w.bm == 32) ? 128 : (gn <= 64 ? 64 : 128);
  w.tiles_m = (int)cdiv(gm, w.bm);
  w.tiles_n = (int)cdiv(gn, w.bn);
  // blocks wanted per tower: two per CU; one per CU for the gradients of one or two tiles on large batches, whose blocks
  // would otherwise store a slab after 16 steps (measured: 128 / 256 / 512 -> 259.5 / 250.8 / 253.0 ms per update)
  // (the heads' gradient, bm == 32, keeps 512: its slabs are 16 KB, and with twice the blocks it runs 18.0 instead of 21.8 us)
  const int64_t blocks = (w.tiles_m * w.tiles_n <= 2 && rows > 8192 && w.bm != 32) ? 256 : 512;
  int64_t want = cdiv(blocks, (int64_t)w.tiles_m * w.tiles_n * 2);
  int64_t ks = cdiv(cdiv(rows, want), 16) * 16;
  if (ks < 64) ks = 64;
  w.ksplit = (int)ks;
  w.nsplit = (int)cdiv(rows, ks);
  return w;
}

struct Workspace {
  // offsets in floats from the workspace base
  int64_t xhat[2][MRL_MAX_LAYERS], rstd[2][MRL_MAX_LAYERS], dz[2][MRL_MAX_LAYERS];
  int64_t dy[2][MRL_MAX_LAYERS];   // small batches: d loss / d LayerNorm output per layer (the weight-gradient launch sums its columns)
  int64_t hact[2][MRL_MAX_LAYERS]; // small batches: activated layer outputs (operands of the weight gradients)
  int64_t xnorm;                   // small batches: the normalised input rows
  int64_t seeds[2];
  int64_t colpart[2][MRL_MAX_LAYERS];
  int64_t colblk[2][MRL_MAX_LAYERS];   // small batches, block-tile weight gradients: [16-row blocks][3][u] column sums from the backward chain
  int col_groups;        // ... summed in this many groups into colpart
  int64_t colpart_h[2];  // small batches: [row ranges][3][32] column sums of the head seeds (head-bias gradients) from the weight-gradient launch
  bool chain_bwd;        // the backward pass of this row count runs the chain kernels and the one-launch weight gradient
  int64_t slab[2][MRL_MAX_LAYERS], slab_h[2];
  int64_t metric_part;   // doubles, 8-byte aligned
  int64_t headb_part;    // [row blocks][2][32] column sums of the head seeds (bias gradients); row blocks of 128 rows, or of 16 (chain kernels)
  int64_t small;         // misc: obs mean/std f32 (2*obs_dim), partial sums
  int64_t norm_part;     // up to 512 doubles: squared-norm partials of the flat gradient (reduce_flat_kernel)
  int64_t wplanes;       // the parameter vector split into three bf16 planes (wplane_stride elements each), refreshed by the launches that read it
  int64_t wplane_stride;
  int64_t wblk_f[2][MRL_MAX_LAYERS], wblk_b[2][MRL_MAX_LAYERS];   // weights re-blocked for the chain kernels (forward / backward order)
  int64_t wblk_h[2];     // head weights re-blocked in forward order ([32][U], zero rows past the head's outputs)
  int64_t total;
  WgradPlan wp[MRL_MAX_LAYERS], wph;
  int rowblocks[MRL_MAX_LAYERS];
};

static void workspace_layout(const mrl_net* net, int64_t rows, Workspace* w) {
  int64_t o = 0;
  ParamLayout L;
  param_layout(net, &L);
  int U = net->units[net->n_layers - 1];
  // activation tensors are padded to whole 128-row tiles: the LayerNorm epilogues read and write full tiles
  const int64_t rows_pad = cdiv(rows, 128) * 128;
  const bool chain_bwd = chain_ok(net, rows, 2);
  w->chain_bwd = chain_bwd;
  const bool chain_fb = chain_bwd && chain_ok(net, rows, 1);   // the forward chain stores what the weight gradients read
  int ch_nsplit = 1, ch_ksplit = 64;
  if (chain_bwd) chain_wgrad_split(net, L.in_dim, rows, &ch_nsplit, &ch_ksplit);
  const bool tile_wg = chain_tile_wgrad(net, rows);
  ChainWgPlan tp;
  if (tile_wg) tp = chain_wgrad_plan(net, L.in_dim, rows);
  for (int tw = 0; tw < 2; ++tw) {
    for (int l = 0; l < net->n_layers; ++l) {
      int u = net->units[l];
      w->xhat[tw][l] = o; o = align64(o + rows_pad * u);
      w->rstd[tw][l] = o; o = align64(o + rows_pad);
      w->dz[tw][l] = o; o = align64(o + rows_pad * u);
      if (chain_bwd) { w->dy[tw][l] = o; o = align64(o + rows_pad * u); }
      if (chain_fb) { w->hact[tw][l] = o; o = align64(o + rows_pad * u); }
      if (tile_wg) {   // column sums per 16-row block from the backward chain, summed in at most 8 groups by the weight-gradient launch
        const int64_t nblk = cdiv(rows, CH_ROWS);
        w->col_groups = (int)std::min<int64_t>(8, nblk);
        w->colblk[tw][l] = o; o = align64(o + nblk * 3 * u);
        w->rowblocks[l] = w->col_groups;
      } else if (chain_bwd) w->rowblocks[l] = ch_nsplit;                                      // column partials per row range of the weight-gradient launch
      else if (net->tower_kind == MRL_TOWER_ELU) w->rowblocks[l] = (int)cdiv(rows, fwd_bm(u));   // full-row tiles of the fused dgrad
      else w->rowblocks[l] = ln_bwd_streamed(u) ? (int)cdiv(rows, ln_bwd_rows(rows)) : (int)cdiv(rows, fused_bwd_rows(u, rows));
      const bool fused0 = l == 0 && first_fused(net, L.in_dim, rows);
      if (fused0) w->rowblocks[l] = (int)cdiv(rows, first_fused_rows(rows));
      w->colpart[tw][l] = o; o = align64(o + (int64_t)w->rowblocks[l] * 3 * u);
      w->wp[l] = wgrad_plan(u, L.in_dim[l], rows);
      if (chain_bwd) { w->wp[l].nsplit = tile_wg ? tp.nsplit[l] : ch_nsplit; w->wp[l].ksplit = ch_ksplit; }
      if (fused0) { w->wp[l].nsplit = w->rowblocks[l]; w->wp[l].ksplit = first_fused_rows(rows); }   // one slab per block of the fused launch
      w->slab[tw][l] = o; o = align64(o + (int64_t)w->wp[l].nsplit * u * L.in_dim[l]);
    }
    w->seeds[tw] = o; o = align64(o + rows * 32);
    w->wph = wgrad_plan(32, U, rows);
    if (chain_bwd) { w->wph.nsplit = tile_wg ? tp.nsplit[net->n_layers] : ch_nsplit; w->wph.ksplit = ch_ksplit; }
    w->slab_h[tw] = o; o = align64(o + (int64_t)w->wph.nsplit * 32 * U);
    w->colpart_h[tw] = o;
    if (chain_bwd) o = align64(o + (int64_t)w->wph.nsplit * 3 * 32);
  }
  w->xnorm = o;
  if (chain_fb) o = align64(o + rows_pad * net->obs_dim);
  const int64_t hb = chain_fb ? cdiv(rows, CH_ROWS) : cdiv(rows, 128);   // row blocks of the kernel that evaluates the heads
  w->metric_part = o; o = align64(o + 2 * hb * 2 * MRL_NMETRIC + 64);
  w->headb_part = o; o = align64(o + hb * 64);
  w->small = o; o = align64(o + 4096 + 2 * (int64_t)net->obs_dim + 16 * cdiv(rows, 64));
  w->norm_part = o; o = align64(o + 1024);
  w->wplane_stride = align64(L.total);
  w->wplanes = o; o = align64(o + (6 * w->wplane_stride + 1) / 2);   // natural order, then k-blocked order
  if (chain_ok(net, rows, 1) || chain_ok(net, rows, 2)) {
    for (int tw = 0; tw < 2; ++tw)
      for (int l = 0; l < net->n_layers; ++l) {
        const int u = net->units[l], kn = (l + 1 < net->n_layers) ? net->units[l + 1] : 32;
        w->wblk_f[tw][l] = o; o = align64(o + (int64_t)u * ((L.in_dim[l] + 31) & ~31));
        w->wblk_b[tw][l] = o; o = align64(o + (int64_t)((kn + 15) & ~15) * u);
      }
    for (int tw = 0; tw < 2; ++tw) { w->wblk_h[tw] = o; o = align64(o + 32 * (int64_t)U); }
  }
  w->total = o;
}

extern "C" int64_t mrl_workspace_bytes(const mrl_net* net, int64_t rows) {
  if (check_net(net)) return MRL_EINVAL;
  if (rows < 1) MRL_FAIL(MRL_EINVAL, "rows %lld < 1", (long long)rows);
  Workspace w;
  workspace_layout(net, rows, &w);
  return w.total * 4;
}

// ---------------------------------------------------------------------------------------------------------------
// GEMM launch helpers
// ---------------------------------------------------------------------------------------------------------------
// ---- optional per-kernel timing with HIP events (benchmarks only; off by default, not thread-safe) ---------------
// mode 1 times every GEMM instantiation, mode 2 only the one that has accumulated the most time so far.
struct ProfClass {
  char name[96];
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  double flops = 0;
};
static std::vector<ProfClass> g_prof;
static int g_prof_mode = 0, g_prof_only = -1;

static int prof_register(const char* name) {
  g_prof.emplace_back();
  snprintf(g_prof.back().name, sizeof(g_prof.back().name), "%s", name);
  return (int)g_prof.size() - 1;
}

static double prof_total_ms(ProfClass& c) {
  double t = 0;
  for (auto& e : c.ev) {
    float ms = 0;
    if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) t += ms;
  }
  return t;
}

static void prof_clear() {
  for (auto& c : g_prof) {
    for (auto& e : c.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    c.ev.clear();
    c.flops = 0;
  }
}

// Times the launches enqueued on `s` during its lifetime as one sample of class `cls` (when that class is being timed).
struct ProfScope {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int cls;
  hipStream_t s;
  double flops;
  ProfScope(int cls_, hipStream_t s_, double flops_ = 0) : cls(cls_), s(s_), flops(flops_) {
    if (g_prof_mode == 1 || (g_prof_mode == 2 && g_prof_only == cls)) {
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0, s);
    }
  }
  ~ProfScope() {
    if (!e0) return;
    (void)hipEventRecord(e1, s);
    g_prof[cls].ev.emplace_back(e0, e1);
    g_prof[cls].flops += flops;
  }
};
// streaming (HBM-bound) kernels worth a line in the per-kernel table
#define MRL_PROF_STREAM(name, stream) static const int prof_cls__ = prof_register(name); ProfScope prof_scope__(prof_cls__, stream)

extern "C" int mrl_profile_enable(int mode) {
  if (mode < 0 || mode > 2) MRL_FAIL(MRL_EINVAL, "profile mode %d", mode);
  if (mode == 2) {
    int best = -1;
    double bt = 0;
    for (size_t i = 0; i < g_prof.size(); ++i) {
      double t = prof_total_ms(g_prof[i]);
      if (t > bt) { bt = t; best = (int)i; }
    }
    if (best >= 0) g_prof_only = best;   // nothing timed since the last call: keep the instantiation chosen then
    if (g_prof_only < 0) mode = 1;
  }
  prof_clear();
  g_prof_mode = mode;
  return MRL_OK;
}

extern "C" int mrl_profile_read(double* total_ms, int64_t* launches, double* flops, char* name, int32_t name_len) {
  if (!total_ms || !launches || !flops) MRL_FAIL(MRL_EINVAL, "NULL output");
  int best = -1;
  double bt = -1;
  for (size_t i = 0; i < g_prof.size(); ++i) {
    if (g_prof[i].ev.empty()) continue;
    double t = prof_total_ms(g_prof[i]);
    if (t > bt) { bt = t; best = (int)i; }
  }
  *total_ms = 0; *launches = 0; *flops = 0;
  if (best < 0) return MRL_OK;
  *total_ms = bt;
  *launches = (int64_t)g_prof[best].ev.size();
  *flops = g_prof[best].flops;
  if (name && name_len > 0) snprintf(name, name_len, "%s", g_prof[best].name);
  return MRL_OK;
}

// Every timed instantiation as one line of text each: "<total ms>\t<launches>\t<algorithmic flops>\t<name>\n", slowest first.
// Returns the number of bytes written (without the terminating NUL) or a negative status.
extern "C" int mrl_profile_read_all(char* text_host, int32_t text_len) {
  if (!text_host || text_len < 1) MRL_FAIL(MRL_EINVAL, "NULL / empty output");
  std::vector<std::pair<double, int>> order;
  for (size_t i = 0; i < g_prof.size(); ++i)
    if (!g_prof[i].ev.empty()) order.emplace_back(-prof_total_ms(g_prof[i]), (int)i);
  std::sort(order.begin(), order.end());
  int n = 0;
  text_host[0] = 0;
  for (auto& o : order) {
    const ProfClass& c = g_prof[o.second];
    const int w = snprintf(text_host + n, (size_t)(text_len - n), "%.6f\t%lld\t%.6e\t%s\n", -o.first, (long long)c.ev.size(), c.flops, c.name);
    if (w < 0 || w >= text_len - n) break;
    n += w;
  }
  return n;
}

// `resident` > 0 caps the blocks of this launch that share a CU: unused dynamic LDS is requested so that exactly
// that many fit (a grid of resident x CUs blocks then spreads evenly instead of stacking up on part of the CUs).
#ifdef MRL_KTRACE
// instrumented diagnostic build only: copy the cycle stamps of the traced block to host memory
extern "C" int mrl_debug_ktrace(unsigned long long* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ktrace), sizeof(unsigned long long) * 8 * KTRACE_SLOTS) == hipSuccess ? 0 : -3;
}
extern "C" int mrl_debug_kblocks(unsigned long long* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_kblock), sizeof(unsigned long long) * 2 * KTRACE_BLOCKS) == hipSuccess ? 0 : -3;
}
#endif

// A launch that does not wait for the launches before it on its stream (hipExtAnyOrderLaunch: the dispatch packet goes out
// without the barrier bit): set around a launch whose inputs are complete once the launch before it has STARTED, so that
// the two run side by side; the next ordinary launch waits for both.  Used for the weight-gradient launches of the backward
// chain (each goes out right behind the dgrad launch that reads the same dZ) and the reduction of the heads' metric partials
// (behind the first of those dgrad launches): the same 2 % per update as separate streams with events (234.7 against 238.8 ms,
// streams 234.9) without either.  An any-order launch is never the last launch of an entry point: what the caller puts on the
// stream next (an event record, a collective) must find an ordinary launch, which completes after everything before it, at the
// end.  MRL_ANY_ORDER=0 in the environment switches it off; it is off while kernels are being timed.
static thread_local bool g_any_order = false;
static bool any_order_enabled() {
  static const int on = [] { const char* e = getenv("MRL_ANY_ORDER"); return e ? atoi(e) : 1; }();
  return on != 0 && g_prof_mode == 0;
}
#define MRL_LAUNCH(kernel, grid, block, lds, stream, arg)                                                                  \
  do {                                                                                                                    \
    if (g_any_order) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, nullptr, hipExtAnyOrderLaunch, arg); \
    else hipLaunchKernelGGL(kernel, grid, block, lds, stream, arg);                                                       \
  } while (0)

template <int BM, int BN, int BK, int WGM, int WGN, int LA, int LB, int EPI, int PRO, int NBUF = 2, int MM = 0>
static void launch_gemm(const GemmP& p, dim3 grid, hipStream_t s, int resident = 0) {
  using Cfg = GemmCfg<BM, BN, BK, WGM, WGN, LA, LB, EPI>;
  constexpr int main_lds = MM ? NBUF * 3 * (BM + BN) * 8 : NBUF * Cfg::LDS_TILES;
  constexpr int static_lds = 4 * ((main_lds > Cfg::LDS_EPI) ? main_lds : Cfg::LDS_EPI);
  unsigned dyn_lds = 0;
  if (resident > 0) {
    const int per_block = (MRL_LDS_PER_CU / resident) & ~1023;
    if (per_block > static_lds) dyn_lds = (unsigned)(per_block - static_lds);
  }
  static int cls = [] {
    char nm[96];
    snprintf(nm, sizeof(nm), "mrl::gemm_kernel<%d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d>", BM, BN, BK, WGM, WGN, LA, LB, EPI, PRO, NBUF, MM);
    return prof_register(nm);
  }();
  const bool timed = g_prof_mode == 1 || (g_prof_mode == 2 && g_prof_only == cls);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timed) {
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
  }
#ifdef MRL_KTRACE
  GemmP pt = p;
  {
    static const char* want = getenv("MRL_KTRACE");
    static const char* want_m = getenv("MRL_KTRACE_M");
    char nm[96];
    snprintf(nm, sizeof(nm), "<%d, %d, %d, %d, %d, %d, %d, %d, %d, %d>", BM, BN, BK, WGM, WGN, LA, LB, EPI, PRO, NBUF);
    static const char* want_z = getenv("MRL_KTRACE_Z");
    static const char* abl = getenv("MRL_KTRACE_ABLATE");
    pt.ktrace = (want && strcmp(want, nm) == 0 && (!want_m || atoi(want_m) == p.M)) ? 1 + (want_z ? atoi(want_z) : 0) : 0;
    pt.ablate = (pt.ktrace && abl) ? atoi(abl) : 0;
  }
  hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, WGM, WGN, LA, LB, EPI, PRO, NBUF, MM>), grid, dim3(64 * WGM * WGN), dyn_lds, s, pt);
#else
  MRL_LAUNCH((gemm_kernel<BM, BN, BK, WGM, WGN, LA, LB, EPI, PRO, NBUF, MM>), grid, dim3(64 * WGM * WGN), dyn_lds, s, p);
#endif
  if (timed) {
    (void)hipEventRecord(e1, s);
    ProfClass& c = g_prof[cls];
    c.ev.emplace_back(e0, e1);
    for (unsigned z = 0; z < grid.z; ++z)   // algorithmic FLOPs of the launch: 2*M*N*K with the true (unpadded) sizes
      c.flops += 2.0 * p.M * (p.Nt[z] ? p.Nt[z] : p.N) * (double)((EPI != EPI_SLAB && p.Kt[z]) ? p.Kt[z] : p.K);
  }
}

// full-row tiles, selected by the output width; batches of at most MRL_SMALL_M rows take 32-row tiles so that a
// 2048-row minibatch still spreads over 64 blocks per tower instead of 16
// waves per 32-row forward tile of the wide layers on small batches: 8 waves halve the staging and MFMA work of each wave
// in a regime where a block's serial reduction loop, not the chip's throughput, sets the time
constexpr int MRL_SMALL_WGN = 8;
#define MRL_DISPATCH_ROWTILE(N, LA, LB, BKBIG, EPI, PRO, p, towers, s)                                                     \
  do {                                                                                                                    \
    if (p.M <= MRL_SMALL_M) {                                                                                             \
      const dim3 g_((unsigned)cdiv(p.M, 32), 1, towers);                                                                  \
      switch (N) {                                                                                                        \
        case 512: launch_gemm<32, 512, 16, 1, MRL_SMALL_WGN, LA, LB, EPI, PRO, 1>(p, g_, s); break;                       \
        case 256: launch_gemm<32, 256, 32, 1, MRL_SMALL_WGN, LA, LB, EPI, PRO>(p, g_, s); break;                          \
        case 128: launch_gemm<32, 128, BKBIG, 1, 4, LA, LB, EPI, PRO>(p, g_, s); break;                                   \
        case 64: launch_gemm<128, 64, BKBIG, 4, 1, LA, LB, EPI, PRO>(p, dim3((unsigned)cdiv(p.M, 128), 1, towers), s); break;   \
        default: launch_gemm<128, 32, BKBIG, 4, 1, LA, LB, EPI, PRO>(p, dim3((unsigned)cdiv(p.M, 128), 1, towers), s); break;   \
      }                                                                                                                   \
      break;                                                                                                              \
    }                                                                                                                     \
    switch (N) {                                                                                                          \
      case 512: launch_gemm<64, 512, 16, 1, 4, LA, LB, EPI, PRO, 1>(p, dim3((unsigned)cdiv(p.M, 64), 1, towers), s); break;        \
      case 256: launch_gemm<128, 256, 16, 2, 2, LA, LB, EPI, PRO>(p, dim3((unsigned)cdiv(p.M, 128), 1, towers), s); break;   \
      case 128: launch_gemm<128, 128, BKBIG, 2, 2, LA, LB, EPI, PRO>(p, dim3((unsigned)cdiv(p.M, 128), 1, towers), s); break;   \
      case 64: launch_gemm<128, 64, BKBIG, 4, 1, LA, LB, EPI, PRO>(p, dim3((unsigned)cdiv(p.M, 128), 1, towers), s); break;     \
      default: launch_gemm<128, 32, BKBIG, 4, 1, LA, LB, EPI, PRO>(p, dim3((unsigned)cdiv(p.M, 128), 1, towers), s); break;     \
    }                                                                                                                     \
  } while (0)

static inline unsigned slab_grid(int n_groups, int group_size) { return (unsigned)(cdiv(n_groups, 8) * 8 * group_size); }

// The 128 x 128-tile weight gradients whose sizes are whole tiles run on the bf16 matrix cores from exact three-term
// splits of the fp32 operands (split_bf16.hip.h): same grid, splits and slabs as the fp32 kernel it replaces.
constexpr int64_t MRL_WGRAD_SPLIT_MIN_ROWS = 8192;   // below this the launch is latency-bound and the fp32 kernel's shorter prologue wins

static bool wgrad_split_eligible(const GemmP& p, const WgradPlan& w, int towers) {
  if (w.bm != 128 || p.M % 128 || p.K < MRL_WGRAD_SPLIT_MIN_ROWS) return false;
  if (!(w.bn == 128 && p.N % 128 == 0) && !(w.bn == 64 && p.N == 64)) return false;
  if (p.lda % 4 || p.ldb % 4) return false;
  for (int z = 0; z < towers; ++z)
    if (p.Nt[z] && p.Nt[z] != p.N) return false;
  return true;
}

template <int PRO, int BNT, int BMT>
static void launch_wgrad_split(const GemmP& p, dim3 grid, hipStream_t s) {
  static int cls = [] {
    char nm[96];
    snprintf(nm, sizeof(nm), "mrl::wgrad_split_kernel<%d, %d, %d>", PRO, BNT, BMT);
    return prof_register(nm);
  }();
  const bool timed = g_prof_mode == 1 || (g_prof_mode == 2 && g_prof_only == cls);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timed) {
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
  }
  MRL_LAUNCH((wgrad_split_kernel<PRO, BNT, BMT>), grid, dim3(BMT * 2), 0, s, p);
  if (timed) {
    (void)hipEventRecord(e1, s);
    ProfClass& c = g_prof[cls];
    c.ev.emplace_back(e0, e1);
    c.flops += 2.0 * p.M * (double)p.N * (double)p.K * grid.z;   // algorithmic FLOPs: one multiply-add per (m, n, row)
  }
}

// input gradients of the wide layers (N a multiple of 128, K of 16) through dgrad_split_kernel; same grid as the fp32 launch
static void launch_dgrad_split(const GemmP& p, dim3 grid, hipStream_t s) {
  static int cls_planes = prof_register("mrl::dgrad_split_kernel<true>"), cls_split = prof_register("mrl::dgrad_split_kernel<false>");
  bool planes = p.Bpl[0] != nullptr;                  // 16-byte plane chunks need weight matrices that start on a multiple of 8 elements
  for (unsigned z = 0; z < grid.z; ++z) planes = planes && p.Bpl[z] && ((uintptr_t)p.Bpl[z] % 16 == 0);
  const int cls = planes ? cls_planes : cls_split;
  const bool timed = g_prof_mode == 1 || (g_prof_mode == 2 && g_prof_only == cls);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timed) {
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
  }
  // MRL_NT in the environment (read once; A/B runs): bit 0 non-temporal output stores when the streaming LayerNorm backward reads
  // the output next (default on), bit 1 also when the fused first-layer kernel does (default off: slower)
  static const int nt_mask = [] { const char* e = getenv("MRL_NT"); return e ? atoi(e) : 1; }();
  const bool nt = p.nt_hint == 1 ? (nt_mask & 1) != 0 : p.nt_hint == 2 ? (nt_mask & 2) != 0 : false;
  if (planes && nt) hipLaunchKernelGGL((dgrad_split_kernel<true, true>), grid, dim3(256), 0, s, p);
  else if (planes) hipLaunchKernelGGL(dgrad_split_kernel<true>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(dgrad_split_kernel<false>, grid, dim3(256), 0, s, p);
  if (timed) {
    (void)hipEventRecord(e1, s);
    ProfClass& c = g_prof[cls];
    c.ev.emplace_back(e0, e1);
    c.flops += 2.0 * p.M * (double)p.N * (double)p.K * grid.z;
  }
}

template <int PRO>
static void launch_wgrad_pro(const GemmP& p, const WgradPlan& w, int towers, hipStream_t s) {
  dim3 grid(slab_grid(w.nsplit, w.tiles_m * w.tiles_n), 1, towers);
  if (wgrad_split_eligible(p, w, towers)) {
    if (w.bn == 128 && w.tiles_m % 2 == 0) {           // 256-row tiles: same splits and slabs, half the tiles, 8 waves each
      GemmP q = p;
      q.m_tiles = w.tiles_m / 2;
      launch_wgrad_split<PRO, 128, 256>(q, dim3(slab_grid(w.nsplit, q.m_tiles * w.tiles_n), 1, towers), s);
    } else if (w.bn == 128) launch_wgrad_split<PRO, 128, 128>(p, grid, s);
    else launch_wgrad_split<PRO, 64, 128>(p, grid, s);
  }
  else if (w.bm == 32)
    launch_gemm<32, 128, 16, 1, 4, LAY_R, LAY_R, EPI_SLAB, PRO>(p, grid, s, MRL_WGRAD_RESIDENT);
  else if (w.bn == 64)
    launch_gemm<128, 64, 16, 4, 1, LAY_R, LAY_R, EPI_SLAB, PRO>(p, grid, s, MRL_WGRAD_RESIDENT);
  else
    launch_gemm<128, 128, 16, 2, 2, LAY_R, LAY_R, EPI_SLAB, PRO>(p, grid, s, MRL_WGRAD_RESIDENT);
}

static void launch_wgrad(const GemmP& p, const WgradPlan& w, int pro, int towers, hipStream_t s) {
  if (pro == PRO_NORM) launch_wgrad_pro<PRO_NORM>(p, w, towers, s);
  else if (pro == PRO_LNSILU) launch_wgrad_pro<PRO_LNSILU>(p, w, towers, s);
  else if (pro == PRO_ELU) launch_wgrad_pro<PRO_ELU>(p, w, towers, s);
  else if (pro == PRO_LNELU) launch_wgrad_pro<PRO_LNELU>(p, w, towers, s);
  else launch_wgrad_pro<PRO_NONE>(p, w, towers, s);
}

// The whole parameter vector as three bf16 planes with x = h + m + l exactly (the split of split_bf16.hip.h), twice:
//   natural order (element i of the flat vector at i of each plane): dgrad_split_kernel<true> reads its [K][N] weight tiles
//   from it instead of splitting them in every block;
//   k-blocked order for the forward kernels, per weight matrix W[N][K] with K a multiple of 16: element (n, k) at
//   (k / 16) * N * 16 + n * 16 + k % 16 from the matrix's start, so that the [N rows][16 k] tile of one reduction step is one
//   contiguous run (the forward main loop stages exactly that tile).  Tensors that are not such matrices keep the natural order.
struct SplitDesc { int64_t off; int rows, cols; };          // weight matrices to re-block (rows = N, cols = K)
#define MRL_MAX_SPLITDESC 16
struct SplitTable { SplitDesc d[MRL_MAX_SPLITDESC]; int n; };

__global__ void split_params_kernel(const float* __restrict__ params, int64_t n, unsigned short* __restrict__ planes, int64_t stride,
                                    unsigned short* __restrict__ kplanes, const SplitTable tab) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = params[i];
  const unsigned hp = split_pack2(x, 0.f);            // the split of split_rne4, one value at a time
  const float r1 = x - split_lo(hp);
  const unsigned mp = split_pack2(r1, 0.f);
  const float r2 = r1 - split_lo(mp);
  const unsigned short h = (unsigned short)hp, m = (unsigned short)mp, l = (unsigned short)split_pack2(r2, 0.f);
  planes[i] = h;
  planes[stride + i] = m;
  planes[2 * stride + i] = l;
  int64_t j = i;
  for (int q = 0; q < tab.n; ++q) {
    const int64_t e = i - tab.d[q].off;
    if (e >= 0 && e < (int64_t)tab.d[q].rows * tab.d[q].cols) {
      const int row = (int)(e / tab.d[q].cols), k = (int)(e % tab.d[q].cols);
      j = tab.d[q].off + (int64_t)(k >> 4) * tab.d[q].rows * 16 + (int64_t)row * 16 + (k & 15);
    }
  }
  kplanes[j] = h;
  kplanes[stride + j] = m;
  kplanes[2 * stride + j] = l;
}

static SplitTable split_table(const ParamLayout& L) {
  SplitTable tab;
  memset(&tab, 0, sizeof(tab));
  int nl = 0;
  while (nl < MRL_MAX_LAYERS && L.in_dim[nl] > 0) ++nl;
  for (int tw = 0; tw < L.n_towers; ++tw)
    for (int l = 0; l < nl; ++l) {
      const int rows = (int)((L.b[tw][l] - L.W[tw][l]) / L.in_dim[l]);   // at least the layer's outputs (the offset of b is padded to 4)
      static_assert(MRL_MAX_SPLITDESC >= 2 * MRL_MAX_LAYERS, "one descriptor per weight matrix of two towers");
      if (L.in_dim[l] % 16 == 0) tab.d[tab.n++] = SplitDesc{L.W[tw][l], rows, L.in_dim[l]};
    }
  return tab;
}

static unsigned short* launch_split_params(const ParamLayout& L, const Workspace& W, float* ws, const float* params, hipStream_t s) {
  unsigned short* planes = reinterpret_cast<unsigned short*>(ws + W.wplanes);
  const SplitTable tab = split_table(L);
  hipLaunchKernelGGL(split_params_kernel, dim3((unsigned)cdiv(L.total, 256)), dim3(256), 0, s, params, L.total, planes, W.wplane_stride,
                     planes + 3 * W.wplane_stride, tab);
  return planes;
}

// Hidden LayerNorm layers of 128 or 256 outputs on batches above MRL_SMALL_M rows run the split-bf16 main loop of gemm_kernel
// (fp32 accuracy on the bf16 matrix cores); false = not eligible, the caller launches the fp32 tile.
static bool lnfwd_split_eligible(int u, int64_t rows, int in) { return rows > MRL_SMALL_M && in % 16 == 0 && in >= 16 && (u == 256 || u == 128); }

// weight planes usable by the 16-byte chunk loads of the split main loop: given for every tower, starting on a multiple of 8
// elements, rows a multiple of 8 elements long
static bool planes_usable(const GemmP& p, int towers) {
  if (p.ldb % 8) return false;
  for (int z = 0; z < towers; ++z)
    if (!p.Bpl[z] || (uintptr_t)p.Bpl[z] % 16) return false;
  return true;
}

template <int PRO>
static void launch_lnfwd_split(int u, const GemmP& p, int towers, hipStream_t s) {
  const dim3 grid((unsigned)cdiv(p.M, 128), 1, towers);
  const bool wpl = planes_usable(p, towers);
  if (u == 256 && wpl) launch_gemm<128, 256, 16, 2, 2, LAY_K, LAY_K, EPI_LNFWD, PRO, 2, 2>(p, grid, s);
  else if (u == 256) launch_gemm<128, 256, 16, 2, 2, LAY_K, LAY_K, EPI_LNFWD, PRO, 2, 1>(p, grid, s);
  else if (wpl) launch_gemm<128, 128, 16, 2, 2, LAY_K, LAY_K, EPI_LNFWD, PRO, 2, 2>(p, grid, s);
  else launch_gemm<128, 128, 16, 2, 2, LAY_K, LAY_K, EPI_LNFWD, PRO, 2, 1>(p, grid, s);
}

// Forward of the towers on `rows` samples: fills xhat/rstd of every layer.  tw0/towers select the towers.
// LDS row stride of the chain kernels: every layer width and the (32-padded) input width, rounded up to 64 floats
// one instantiation of a chain kernel per tower kind (MRL_TOWER_* = the CH_* kinds of chain.hip.h)
#define MRL_CHAIN_LAUNCH(KERNEL, kind, grid, lds, stream, arg)                                                      \
  do {                                                                                                              \
    switch (kind) {                                                                                                 \
      case CH_PLAIN_ELU: hipLaunchKernelGGL(KERNEL<CH_PLAIN_ELU>, grid, dim3(64 * CH_WAVES), lds, stream, arg); break;   \
      case CH_LN_ELU: hipLaunchKernelGGL(KERNEL<CH_LN_ELU>, grid, dim3(64 * CH_WAVES), lds, stream, arg); break;         \
      case CH_PLAIN_SILU: hipLaunchKernelGGL(KERNEL<CH_PLAIN_SILU>, grid, dim3(64 * CH_WAVES), lds, stream, arg); break; \
      case CH_LN_RELU: hipLaunchKernelGGL(KERNEL<CH_LN_RELU>, grid, dim3(64 * CH_WAVES), lds, stream, arg); break;       \
      case CH_PLAIN_RELU: hipLaunchKernelGGL(KERNEL<CH_PLAIN_RELU>, grid, dim3(64 * CH_WAVES), lds, stream, arg); break; \
      case CH_LN_TANH: hipLaunchKernelGGL(KERNEL<CH_LN_TANH>, grid, dim3(64 * CH_WAVES), lds, stream, arg); break;       \
      case CH_PLAIN_TANH: hipLaunchKernelGGL(KERNEL<CH_PLAIN_TANH>, grid, dim3(64 * CH_WAVES), lds, stream, arg); break; \
      default: hipLaunchKernelGGL(KERNEL<CH_LN_SILU>, grid, dim3(64 * CH_WAVES), lds, stream, arg); break;               \
    }                                                                                                               \
  } while (0)
template <int KIND>
static void chain_allow_lds_kind() {
  const void* fns[] = {reinterpret_cast<const void*>(chain_fwd_kernel<KIND>), reinterpret_cast<const void*>(chain_bwd_kernel<KIND>),
                       reinterpret_cast<const void*>(ppo_chain_kernel<KIND>)};
  for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
static int chain_ld(const mrl_net* net, int in_dim) {
  int m = (in_dim + 31) & ~31;
  for (int l = 0; l < net->n_layers; ++l) m = std::max(m, (int)net->units[l]);
  return (m + 63) & ~63;
}
static size_t chain_lds_bytes(int ld) { return (size_t)(CH_RED + 2 * CH_ROWS * ld) * sizeof(float); }
// the backward chain adds a third tile and the exchange of its column sums
static size_t chain_bwd_extra_lds(int ld) { return (size_t)(CH_ROWS * ld + CH_COLRED_F) * sizeof(float); }
// the widest networks need a little more dynamic LDS than the 64 KB a kernel may ask for by default
static void chain_allow_lds() {
  static const bool once = [] {
    chain_allow_lds_kind<CH_LN_SILU>(); chain_allow_lds_kind<CH_PLAIN_ELU>(); chain_allow_lds_kind<CH_LN_ELU>(); chain_allow_lds_kind<CH_PLAIN_SILU>();
    chain_allow_lds_kind<CH_LN_RELU>(); chain_allow_lds_kind<CH_PLAIN_RELU>(); chain_allow_lds_kind<CH_LN_TANH>(); chain_allow_lds_kind<CH_PLAIN_TANH>();
    const void* fns[] = {reinterpret_cast<const void*>(wgrad_small_kernel), reinterpret_cast<const void*>(wgrad_tile_kernel),
                         reinterpret_cast<const void*>(wgrad_group_kernel)};
    for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)once;
}
__global__ void rms_merge_kernel(double* mean, double* var, double* count, const float* __restrict__ bmean, const float* __restrict__ bvar,
                                 double bcount, int n, float eps, float* mean_f, float* std_f);
// Weights of towers tw0 .. tw0+towers-1 into the workspace's re-blocked copies: forward order (W of every layer) and / or
// backward order (W of the layer above, the head for the last layer).  One launch.
static void launch_reblock(const mrl_net* net, const ParamLayout& L, const Workspace& W, float* ws, const float* params, int tw0, int towers, bool fwd,
                           bool bwd, hipStream_t s, bool heads = false, const RmsMergeArgs* merge = nullptr) {
  if (chain_natural()) {
    if (merge)
      hipLaunchKernelGGL(rms_merge_kernel, dim3(1), dim3(1024), 0, s, merge->mean, merge->var, merge->count, merge->bmean, merge->bvar, merge->bcount,
                         merge->n, merge->eps, merge->mean_f, merge->std_f);
    return;
  }
  ReblockP p;
  memset(&p, 0, sizeof(p));
  if (merge) { p.merge = *merge; p.has_merge = 1; }
  int piece = 0;
  const int nl = net->n_layers;
  for (int t = 0; t < towers; ++t) {
    const int tw = tw0 + t;
    for (int l = 0; l < nl; ++l) {
      const int u = net->units[l];
      if (fwd) {
        ReblockDesc& d = p.d[p.nd++];
        d.kind = 0; d.src = params + L.W[tw][l]; d.dst = ws + W.wblk_f[tw][l]; d.N = u; d.K = L.in_dim[l]; d.rows = u;
        d.piece0 = piece;
        piece += u * ((L.in_dim[l] + 31) & ~31) / 4;
      }
      if (bwd) {
        ReblockDesc& d = p.d[p.nd++];
        const bool last = l == nl - 1;
        d.kind = 1; d.src = last ? params + L.Wh[tw] : params + L.W[tw][l + 1]; d.dst = ws + W.wblk_b[tw][l]; d.N = u;
        d.K = last ? 32 : net->units[l + 1];
        d.rows = last ? L.head_rows[tw] : net->units[l + 1];
        d.piece0 = piece;
        piece += ((d.K + 15) & ~15) * u / 4;
      }
    }
    if (heads) {   // the head as one more forward-order matrix of 32 rows
      const int U = net->units[nl - 1];
      ReblockDesc& d = p.d[p.nd++];
      d.kind = 0; d.src = params + L.Wh[tw]; d.dst = ws + W.wblk_h[tw]; d.N = 32; d.K = U; d.rows = L.head_rows[tw];
      d.piece0 = piece;
      piece += 32 * U / 4;
    }
  }
  p.npieces = piece;
  MRL_PROF_STREAM("mrl::reblock_kernel", s);
  hipLaunchKernelGGL(reblock_kernel, dim3((unsigned)cdiv(piece, 256) + (merge ? 1 : 0)), dim3(256), 0, s, p);
}

static void fill_chain_fwd(const mrl_net* net, const ParamLayout& L, const Workspace& W, float* ws, const float* params, const float* x, int64_t rows,
                           const float* obs_mean, const float* obs_std, float lo, float hi, int tw0, int towers, ChainFwdP* out) {
  ChainFwdP& p = *out;
  memset(&p, 0, sizeof(p));
  p.x = x; p.mean = obs_mean; p.stdv = obs_std; p.lo = lo; p.hi = hi;
  p.M = (int)rows; p.in_dim = L.in_dim[0]; p.n_layers = net->n_layers;
  const bool stores_act = chain_ok(net, rows, 2);   // the weight-gradient launch of the backward pass reads the activations from memory
  p.ld = chain_ld(net, L.in_dim[0]);
  p.rows_written = (int)(cdiv(rows, 128) * 128);   // the activation tensors are padded to whole 128-row tiles; all of it ends up finite
  p.ln_eps = net->ln_eps;
  for (int l = 0; l < net->n_layers; ++l) {
    p.units[l] = net->units[l];
    for (int t = 0; t < towers; ++t) {
      const int tw = tw0 + t;
      p.W[t][l] = chain_natural() ? params + L.W[tw][l] : ws + W.wblk_f[tw][l];
      p.b[t][l] = params + L.b[tw][l];
      p.g[t][l] = L.has_norm ? params + L.g[tw][l] : nullptr;
      p.be[t][l] = L.has_norm ? params + L.be[tw][l] : nullptr;
      p.xhat[t][l] = ws + W.xhat[tw][l];
      p.rstd[t][l] = ws + W.rstd[tw][l];
      p.hact[t][l] = stores_act ? ws + W.hact[tw][l] : nullptr;
    }
  }
  p.xnorm = (stores_act && obs_mean) ? ws + W.xnorm : nullptr;
  p.natural = chain_natural();
}

static void launch_chain_fwd(const mrl_net* net, const ParamLayout& L, const Workspace& W, float* ws, const float* params, const float* x, int64_t rows,
                             const float* obs_mean, const float* obs_std, float lo, float hi, int tw0, int towers, hipStream_t s, bool blocked_ready) {
  if (!blocked_ready) launch_reblock(net, L, W, ws, params, tw0, towers, true, false, s);
  ChainFwdP p;
  fill_chain_fwd(net, L, W, ws, params, x, rows, obs_mean, obs_std, lo, hi, tw0, towers, &p);
  chain_allow_lds();
  MRL_PROF_STREAM("mrl::chain_fwd_kernel", s);
  const dim3 grid((unsigned)(p.rows_written / CH_ROWS), towers);
  const size_t lds = chain_lds_bytes(p.ld);
  MRL_CHAIN_LAUNCH(chain_fwd_kernel, net->tower_kind, grid, lds, s, p);
}

// MRL_FWD0=0 in the environment (read once): the first layer's forward through gemm_kernel<64,512> (A/B runs)
static bool fwd0_on() { static const bool on = [] { const char* e = getenv("MRL_FWD0"); return e ? atoi(e) != 0 : true; }(); return on; }
static void launch_towers_fwd(const mrl_net* net, const ParamLayout& L, const Workspace& W, float* ws, const float* params,
                              const float* x, int64_t rows, const float* obs_mean, const float* obs_std, float lo, float hi,
                              int tw0, int towers, hipStream_t s, const unsigned short* planes = nullptr, bool blocked_ready = false) {
  if (chain_ok(net, rows, 1)) {
    launch_chain_fwd(net, L, W, ws, params, x, rows, obs_mean, obs_std, lo, hi, tw0, towers, s, blocked_ready);
    return;
  }
  for (int l = 0; l < net->n_layers; ++l) {
    GemmP p;
    memset(&p, 0, sizeof(p));
    int u = net->units[l], in = L.in_dim[l];
    for (int t = 0; t < towers; ++t) {
      int tw = tw0 + t;
      p.A[t] = (l == 0) ? x : ws + W.xhat[tw][l - 1];
      p.B[t] = params + L.W[tw][l];
      p.Bpl[t] = (planes && in % 16 == 0) ? planes + 3 * W.wplane_stride + L.W[tw][l] : nullptr;   // the k-blocked planes
      p.pro0[t] = (l == 0) ? obs_mean : (L.has_norm ? params + L.g[tw][l - 1] : nullptr);
      p.pro1[t] = (l == 0) ? obs_std : (L.has_norm ? params + L.be[tw][l - 1] : nullptr);
      p.out[t] = ws + W.xhat[tw][l];   // plain towers keep the pre-activation z in the same slot
      p.rstd[t] = ws + W.rstd[tw][l];
      p.bias[t] = params + L.b[tw][l];
    }
    p.lda = in; p.ldb = in; p.ldo = u;
    p.pl_stride = W.wplane_stride;
    p.M = (int)rows; p.N = u; p.K = in;
    p.Nt[0] = p.Nt[1] = u; p.Kt[0] = p.Kt[1] = in;
    p.clamp_lo = lo; p.clamp_hi = hi;
    p.ln_eps = net->ln_eps;
    if (net->tower_kind == MRL_TOWER_ELU) {
      if (l > 0) MRL_DISPATCH_ROWTILE(u, LAY_K, LAY_K, 32, EPI_BIASFWD, PRO_ELU, p, towers, s);
      else if (obs_mean) MRL_DISPATCH_ROWTILE(u, LAY_K, LAY_K, 32, EPI_BIASFWD, PRO_NORM, p, towers, s);
      else MRL_DISPATCH_ROWTILE(u, LAY_K, LAY_K, 32, EPI_BIASFWD, PRO_NONE, p, towers, s);
    } else if (l > 0 && net->tower_kind == TOWER_LAYERNORM_ELU) {
      if (lnfwd_split_eligible(u, rows, in)) launch_lnfwd_split<PRO_LNELU>(u, p, towers, s);
      else MRL_DISPATCH_ROWTILE(u, LAY_K, LAY_K, 32, EPI_LNFWD, PRO_LNELU, p, towers, s);
    } else if (l > 0) {
      if (lnfwd_split_eligible(u, rows, in)) launch_lnfwd_split<PRO_LNSILU>(u, p, towers, s);
      else MRL_DISPATCH_ROWTILE(u, LAY_K, LAY_K, 32, EPI_LNFWD, PRO_LNSILU, p, towers, s);
    }
    else if (u == LNF0_N && rows > MRL_SMALL_M && in <= LNF0_K && in % 4 == 0 && fwd0_on()) {
      // first layer of 512 outputs on at most 64 inputs: the weights stay in registers, a block walks a row range (ln_fwd0.hip.h)
      LnFwd0Args f;
      memset(&f, 0, sizeof(f));
      f.x = x; f.in = in; f.mean = obs_mean; f.std = obs_std; f.clamp_lo = lo; f.clamp_hi = hi;
      for (int t = 0; t < towers; ++t) {
        const int tw = tw0 + t;
        f.W[tw] = params + L.W[tw][0]; f.bias[tw] = params + L.b[tw][0];
        f.xhat[tw] = ws + W.xhat[tw][0]; f.rstd[tw] = ws + W.rstd[tw][0];
      }
      f.M = (int)rows; f.Mpad = (int)(cdiv(rows, 128) * 128);
      f.rows_per_block = (int)std::max<int64_t>(LNF0_ROWS, cdiv(cdiv(f.Mpad, 256 / towers), LNF0_ROWS) * LNF0_ROWS);
      f.tw0 = tw0; f.eps = net->ln_eps;
      static const bool lds_ok = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lnfwd0_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
      }();
      (void)lds_ok;
      MRL_PROF_STREAM("lnfwd0_kernel", s);
      hipLaunchKernelGGL(lnfwd0_kernel, dim3((unsigned)cdiv(f.Mpad, f.rows_per_block), towers), dim3(LNF0_THREADS), LNF0_LDS, s, f);
    }
    else if (u == 512 && rows > MRL_SMALL_M && in % 16 == 0) {     // first layer of 512 outputs: split-bf16 main loop, single stage
      const dim3 grid((unsigned)cdiv(p.M, 64), 1, towers);
      const bool wpl = planes_usable(p, towers);
      if (obs_mean && wpl) launch_gemm<64, 512, 16, 1, 4, LAY_K, LAY_K, EPI_LNFWD, PRO_NORM, 1, 2>(p, grid, s);
      else if (obs_mean) launch_gemm<64, 512, 16, 1, 4, LAY_K, LAY_K, EPI_LNFWD, PRO_NORM, 1, 1>(p, grid, s);
      else launch_gemm<64, 512, 16, 1, 4, LAY_K, LAY_K, EPI_LNFWD, PRO_NONE, 1, 1>(p, grid, s);
    }
    else if (obs_mean) MRL_DISPATCH_ROWTILE(u, LAY_K, LAY_K, 32, EPI_LNFWD, PRO_NORM, p, towers, s);
    else MRL_DISPATCH_ROWTILE(u, LAY_K, LAY_K, 32, EPI_LNFWD, PRO_NONE, p, towers, s);
  }
}

static void fill_heads_common(const mrl_net* net, HeadsP* h) {
  h->act_dim = net->act_dim;
  h->dist_type = net->dist_type == MRL_DIST_DREAMERV3_NORMAL ? DIST_DREAMERV3 : DIST_NORMAL;
  h->minstd = net->minstd;
  h->maxstd = net->maxstd;
}

// ---------------------------------------------------------------------------------------------------------------
// streaming kernels.  Floating-point contraction is switched off for them: each mul / add rounds once, like the
// element-wise torch ops they restate, so GAE, reward shaping and the Adam update agree with the reference bit for
// bit (hipcc would otherwise fuse a*b+c into one fma).
// ---------------------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)

// single-rounding helpers compiled without contraction (the __f*_rn wrappers of the HIP headers still fuse)
static __device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
static __device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
static __device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }
static __device__ __forceinline__ float div_rn(float a, float b) { return a / b; }

__global__ void copy_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t rows, int cols4,
                                 int64_t dst_row_stride4, int64_t dst_off4) {
  // src [rows][cols] contiguous -> dst[row*stride + off + c]; all in float4 units
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = rows * cols4;
  for (; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / cols4;
    int c = (int)(i % cols4);
    reinterpret_cast<f32x4*>(dst)[r * dst_row_stride4 + dst_off4 + c] = reinterpret_cast<const f32x4*>(src)[i];
  }
}

__global__ void rms_to_f32_kernel(const double* mean, const double* var, float eps, int n, float* mean_f, float* std_f) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    mean_f[i] = (float)mean[i];
    std_f[i] = sqrtf((float)var[i] + eps);
  }
}

// Episode bookkeeping of one env step.  Block b owns envs [b*1024, (b+1)*1024): it counts the finished episodes of the
// whole step and of the envs before its own (the done flags are one byte per env, so every block can afford to read
// them all), ranks its own finished episodes in env order with ballots, and places them in the rings at
// (head + kept - 1 - rank_from_end) mod window exactly like appends to a deque(maxlen=window).  head / num_episodes are
// only read here; episode_commit_kernel advances them afterwards.
__device__ __forceinline__ int nonzero_bytes(unsigned w) {
  unsigned t = w | (w >> 4);
  t |= t >> 2;
  t |= t >> 1;
  return __popc(t & 0x01010101u);
}

// number of non-zero bytes in dones[0, n) and in dones[0, before): block-wide, returned to every thread
__device__ __forceinline__ void count_done(const uint8_t* __restrict__ dones, int64_t n, int64_t before, int64_t* total_out, int64_t* before_out) {
  __shared__ int red[2][4];
  int tot = 0, bef = 0;
  if ((reinterpret_cast<uintptr_t>(dones) & 3) == 0) {
    const unsigned* w = reinterpret_cast<const unsigned*>(dones);
    const int64_t nw = n >> 2;
    for (int64_t i = threadIdx.x; i < nw; i += 256) {
      const int c = nonzero_bytes(w[i]);
      tot += c;
      if (4 * i + 4 <= before) bef += c;
      else if (4 * i < before)
        for (int k = 0; k < (int)(before - 4 * i); ++k) bef += dones[4 * i + k] != 0;
    }
    for (int64_t i = 4 * nw + threadIdx.x; i < n; i += 256) {
      tot += dones[i] != 0;
      bef += (i < before) && dones[i] != 0;
    }
  } else {
    for (int64_t i = threadIdx.x; i < n; i += 256) {
      tot += dones[i] != 0;
      bef += (i < before) && dones[i] != 0;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    tot += __shfl_xor(tot, o);
    bef += __shfl_xor(bef, o);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = tot; red[1][wave] = bef; }
  __syncthreads();
  *total_out = (int64_t)red[0][0] + red[0][1] + red[0][2] + red[0][3];
  *before_out = (int64_t)red[1][0] + red[1][1] + red[1][2] + red[1][3];
}

__global__ __launch_bounds__(256) void episode_update_kernel(const float* __restrict__ rewards, const uint8_t* __restrict__ dones, int64_t n,
                                                             float* __restrict__ cur_r, int64_t* __restrict__ cur_l, double* __restrict__ ring_r,
                                                             double* __restrict__ ring_l, int window, const int64_t* __restrict__ head) {
  constexpr int U = 4;                         // envs per thread: index u*256 + tid of the block's 1024
  __shared__ int cnt[U][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t start = (int64_t)blockIdx.x * (256 * U);
  int64_t total, before;
  count_done(dones, n, start, &total, &before);
  float val[U];
  int64_t len[U];
  bool d[U];
  unsigned long long m[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = start + u * 256 + threadIdx.x;
    const bool valid = i < n;
    d[u] = valid && dones[i] != 0;
    val[u] = valid ? cur_r[i] + rewards[i] : 0.f;
    len[u] = valid ? cur_l[i] + 1 : 0;
    m[u] = __ballot(d[u]);
    if (lane == 0) cnt[u][wave] = __popcll(m[u]);
  }
  __syncthreads();
  const int64_t L = window, h0 = head[0];
  const int64_t kept = total < L ? total : L;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = start + u * 256 + threadIdx.x;
    if (i >= n) continue;
    if (d[u]) {
      int64_t k = before + __popcll(m[u] & ((1ull << lane) - 1ull));
      for (int q = 0; q < u * 4 + wave; ++q) k += cnt[q >> 2][q & 3];   // finished episodes of the block before this wave's slice
      const int64_t rank_from_end = total - 1 - k;
      if (rank_from_end < L) {
        const int64_t pos = (h0 + kept - 1 - rank_from_end) % L;
        ring_r[pos] = (double)val[u];
        ring_l[pos] = (double)len[u];
      }
      cur_r[i] = 0.f;
      cur_l[i] = 0;
    } else {
      cur_r[i] = val[u];
      cur_l[i] = len[u];
    }
  }
}

// One regex-selected per-env scalar of Metrics._update_infos (metrics.py:76-113) for one aggregation: mode 0 running sum,
// 1 running minimum, 2 running maximum (NaN = nothing seen yet; fmin / fmax skip it like torch.fmin / torch.fmax), 3 the
// value at the episode's last step.  Finished envs append their value to the ring in env order, exactly where
// episode_update_kernel puts the episode's reward (same head, same ranks), and restart at 0 or NaN.  Reads head only.
__global__ __launch_bounds__(256) void info_track_kernel(const float* __restrict__ x, const uint8_t* __restrict__ dones, int64_t n, int mode,
                                                         int reset_nan, float* __restrict__ cur, double* __restrict__ ring, int window,
                                                         const int64_t* __restrict__ head) {
  constexpr int U = 4;
  __shared__ int cnt[U][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t start = (int64_t)blockIdx.x * (256 * U);
  int64_t total, before;
  count_done(dones, n, start, &total, &before);
  float val[U];
  bool d[U];
  unsigned long long m[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = start + u * 256 + threadIdx.x;
    const bool valid = i < n;
    d[u] = valid && dones[i] != 0;
    float v = 0.f;
    if (valid) {
      const float c = cur[i], xi = x[i];
      v = mode == 0 ? add_rn(c, xi) : mode == 1 ? fminf(c, xi) : mode == 2 ? fmaxf(c, xi) : (d[u] ? xi : c);
    }
    val[u] = v;
    m[u] = __ballot(d[u]);
    if (lane == 0) cnt[u][wave] = __popcll(m[u]);
  }
  __syncthreads();
  const int64_t L = window, h0 = head[0];
  const int64_t kept = total < L ? total : L;
  const float reset = reset_nan ? __builtin_nanf("") : 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = start + u * 256 + threadIdx.x;
    if (i >= n) continue;
    if (d[u]) {
      int64_t k = before + __popcll(m[u] & ((1ull << lane) - 1ull));
      for (int q = 0; q < u * 4 + wave; ++q) k += cnt[q >> 2][q & 3];
      const int64_t rank_from_end = total - 1 - k;
      if (rank_from_end < L) ring[(h0 + kept - 1 - rank_from_end) % L] = (double)val[u];
      cur[i] = reset;
    } else {
      cur[i] = val[u];
    }
  }
}

__global__ __launch_bounds__(256) void episode_commit_kernel(const uint8_t* __restrict__ dones, int64_t n, int window, int64_t* __restrict__ head,
                                                             int64_t* __restrict__ num_episodes) {
  int64_t total, before;
  count_done(dones, n, 0, &total, &before);
  if (threadIdx.x == 0) {
    const int64_t L = window;
    head[0] = (head[0] + (total < L ? total : L)) % L;
    num_episodes[0] += total;
  }
}

__global__ void rollout_write_kernel(float* __restrict__ rewards_t, uint8_t* __restrict__ dones_t,
                                     const float* __restrict__ values_t, const float* __restrict__ r,
                                     const uint8_t* __restrict__ dones, const uint8_t* __restrict__ time_outs, int64_t n,
                                     float scale, float gamma) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = mul_rn(r[i], scale);                                   // reward_shaper.py:10-12
  if (time_outs) {                                                    // ppo.py:159-162
    float b = mul_rn(mul_rn(gamma, values_t[i]), time_outs[i] ? 1.0f : 0.0f);
    s = add_rn(s, b);
  }
  rewards_t[i] = s;
  dones_t[i] = dones[i] ? 1 : 0;
}

// GAE(lambda) reverse scan for 64 consecutive envs per block (one env per lane), then a coalesced write of the
// env-major flat arrays through an LDS transpose.  experience.py:96-108 (+ flatten of :110-118).
__global__ __launch_bounds__(64) void gae_kernel(const float* __restrict__ rewards, const float* __restrict__ values,
                                                 const uint8_t* __restrict__ dones, const float* __restrict__ neglogp_tn,
                                                 const float* __restrict__ last_values, int64_t N, int T, float gamma,
                                                 float gamma_tau, float* __restrict__ returns_f, float* __restrict__ values_f,
                                                 float* __restrict__ adv_f, float* __restrict__ neglogp_f,
                                                 double* __restrict__ part) {
  extern __shared__ float sm[];   // 4 tiles [64][T+1]
  const int S = T + 1;
  float* s_ret = sm;
  float* s_val = sm + 64 * S;
  float* s_adv = sm + 2 * 64 * S;
  float* s_nlp = sm + 3 * 64 * S;
  const int lane = threadIdx.x;
  const int64_t n0 = (int64_t)blockIdx.x * 64;
  const int64_t n = n0 + lane;
  double sa = 0, sa2 = 0, sv = 0, sv2 = 0, sr = 0, sr2 = 0;
  if (n < N) {
    float next_v = last_values[n];
    float last = 0.0f;
    auto step = [&](int t, float v, float r, float dn, float nl) {
      float nnt = 1.0f - dn;
      // delta = r + gamma*next_values*nnt - v ; adv = delta + (gamma*tau)*nnt*last   (unfused, as torch evaluates it)
      float delta = sub_rn(add_rn(r, mul_rn(mul_rn(gamma, next_v), nnt)), v);
      last = add_rn(delta, mul_rn(mul_rn(gamma_tau, nnt), last));
      float ret = add_rn(last, v);
      s_ret[lane * S + t] = ret;
      s_val[lane * S + t] = v;
      float adv = sub_rn(ret, v);      // experience.py:119: advantages = returns - values
      s_adv[lane * S + t] = adv;
      s_nlp[lane * S + t] = nl;
      sa += adv; sa2 += (double)adv * adv;
      sv += v; sv2 += (double)v * v;
      sr += ret; sr2 += (double)ret * ret;
      next_v = v;
    };
    // the recurrence is sequential in t, its operands are not: 8 time steps at a time, their 32 loads in flight together
    int t = T - 1;
    for (; t >= 7; t -= 8) {
      float v8[8], r8[8], d8[8], n8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int64_t i = (int64_t)(t - k) * N + n;
        v8[k] = values[i]; r8[k] = rewards[i]; d8[k] = (float)dones[i]; n8[k] = neglogp_tn[i];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) step(t - k, v8[k], r8[k], d8[k], n8[k]);
    }
    for (; t >= 0; --t) {
      const int64_t i = (int64_t)t * N + n;
      step(t, values[i], rewards[i], (float)dones[i], neglogp_tn[i]);
    }
  }
  double vals[6] = {sa, sa2, sv, sv2, sr, sr2};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    double x = vals[k];
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    vals[k] = x;
  }
  if (lane == 0)
    for (int k = 0; k < 6; ++k) part[(int64_t)blockIdx.x * 6 + k] = vals[k];
  __syncthreads();
  int64_t nvalid = (N - n0 < 64) ? (N - n0) : 64;
  int64_t cnt = nvalid * T;
  for (int64_t j = lane; j < cnt; j += 64) {
    int e = (int)(j / T), t = (int)(j % T);
    int64_t o = n0 * T + j;
    returns_f[o] = s_ret[e * S + t];
    values_f[o] = s_val[e * S + t];
    adv_f[o] = s_adv[e * S + t];
    neglogp_f[o] = s_nlp[e * S + t];
  }
}

// One block: fixed-order sum of the per-block partials -> sums[6] = sum adv, adv^2, value, value^2, return, return^2.
__global__ __launch_bounds__(256) void gae_sums_kernel(const double* __restrict__ part, int nblocks, double* __restrict__ sums) {
  __shared__ double red[6][256];
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int b = threadIdx.x; b < nblocks; b += 256)
    for (int k = 0; k < 6; ++k) acc[k] += part[(int64_t)b * 6 + k];
  for (int k = 0; k < 6; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  if (threadIdx.x < 6) {
    double x = 0;
    for (int i = 0; i < 256; ++i) x += red[threadIdx.x][i];
    sums[threadIdx.x] = x;
  }
}

// Whole-batch advantage statistics and the two-stage value_rms update (ppo.py:175-183, experience.py:119-120) from the
// six sums.  stats: [0] adv mean [1] adv std [2,3] value mean/std after update 1 [4,5] after update 2.
__global__ void gae_stats_kernel(const double* __restrict__ s, double count, double* rmean, double* rvar, double* rcount,
                                 float eps, float* __restrict__ stats) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double n = count;
  const double am = s[0] / n;
  const double avar = (s[1] - s[0] * s[0] / n) / (n - 1.0);
  stats[0] = (float)am;
  stats[1] = (float)sqrt(avar > 0 ? avar : 0.0);
  if (rmean) {
    double mean = rmean[0], var = rvar[0], cnt = rcount[0];
    for (int stage = 0; stage < 2; ++stage) {
      const double sx = s[2 + 2 * stage], sx2 = s[3 + 2 * stage];
      // batch moments are fp32 tensors in the reference (x.mean(0), x.var(0, correction=1))
      const double bm = (double)(float)(sx / n);
      const double bv = (double)(float)((sx2 - sx * sx / n) / (n - 1.0));
      const double delta = bm - mean, tot = cnt + n;
      const double new_mean = mean + delta * n / tot;
      const double m2 = var * cnt + bv * n + delta * delta * cnt * n / tot;
      mean = new_mean; var = m2 / tot; cnt = tot;
      stats[2 + 2 * stage] = (float)mean;
      stats[3 + 2 * stage] = sqrtf((float)var + eps);
    }
    rmean[0] = mean; rvar[0] = var; rcount[0] = cnt;
  }
}

__global__ void gae_normalize_kernel(float* __restrict__ adv, float* __restrict__ values, float* __restrict__ returns,
                                     const float* __restrict__ stats, int64_t n, int norm_value, float lo, float hi) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  adv[i] = div_rn(sub_rn(adv[i], stats[0]), add_rn(stats[1], 1e-8f));   // experience.py:120
  if (norm_value) {
    values[i] = fminf(fmaxf(div_rn(sub_rn(values[i], stats[2]), stats[3]), lo), hi);
    returns[i] = fminf(fmaxf(div_rn(sub_rn(returns[i], stats[4]), stats[5]), lo), hi);
  }
}

// Column sums / sums of squares (fp64) of a chunk of rows of one group.  grid = (chunks, groups).
__global__ __launch_bounds__(256) void moments_partial_kernel(const float* __restrict__ x, int cols, int64_t group_rows,
                                                              int64_t chunk_rows, double* __restrict__ part) {
  extern __shared__ double smd[];   // [2][rowlanes][cols]
  const int rowlanes = 256 / cols > 0 ? 256 / cols : 1;
  const int c = threadIdx.x % cols, rl = threadIdx.x / cols;
  const int64_t g = blockIdx.y, ch = blockIdx.x;
  int64_t r0 = ch * chunk_rows, r1 = r0 + chunk_rows;
  if (r1 > group_rows) r1 = group_rows;
  double s = 0, s2 = 0;
  if (rl < rowlanes) {
    const float* base = x + (g * group_rows) * cols;
    for (int64_t r = r0 + rl; r < r1; r += rowlanes) {
      float v = base[r * cols + c];
      s += v;
      s2 += (double)v * v;
    }
    smd[rl * cols + c] = s;
    smd[(rowlanes + rl) * cols + c] = s2;
  }
  __syncthreads();
  if (threadIdx.x < cols) {
    double a = 0, b = 0;
    for (int i = 0; i < rowlanes; ++i) {
      a += smd[i * cols + threadIdx.x];
      b += smd[(rowlanes + i) * cols + threadIdx.x];
    }
    double* o = part + ((g * gridDim.x + ch) * 2) * cols;
    o[threadIdx.x] = a;
    o[cols + threadIdx.x] = b;
  }
}

// The same for column counts that are multiples of 4: a thread takes 4 columns (16-byte loads) and every (256 / (cols / 4))-th row,
// four rows per pass so that four loads are in flight; fp64 accumulators per column, combined over the thread rows in a fixed order.
__global__ __launch_bounds__(256) void moments_partial4_kernel(const float* __restrict__ x, int cols, int64_t group_rows,
                                                               int64_t chunk_rows, double* __restrict__ part) {
  extern __shared__ double smd[];   // [2][rowlanes][cols]
  const int c4n = cols >> 2, rowlanes = 256 / c4n;
  const int c4 = threadIdx.x % c4n, rl = threadIdx.x / c4n;
  const int64_t g = blockIdx.y, ch = blockIdx.x;
  int64_t r0 = ch * chunk_rows, r1 = r0 + chunk_rows;
  if (r1 > group_rows) r1 = group_rows;
  double s[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (rl < rowlanes) {
    const float* base = x + (g * group_rows) * cols + 4 * c4;
    int64_t r = r0 + rl;
    for (; r + 3 * rowlanes < r1; r += 4 * rowlanes) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(base + (r + u * rowlanes) * cols);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += v[u][e]; s2[e] += (double)v[u][e] * v[u][e]; }
    }
    for (; r < r1; r += rowlanes) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + r * cols);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[e] += v[e]; s2[e] += (double)v[e] * v[e]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      smd[rl * cols + 4 * c4 + e] = s[e];
      smd[(rowlanes + rl) * cols + 4 * c4 + e] = s2[e];
    }
  }
  __syncthreads();
  if (threadIdx.x < cols) {
    double a = 0, b = 0;
    for (int i = 0; i < rowlanes; ++i) {
      a += smd[i * cols + threadIdx.x];
      b += smd[(rowlanes + i) * cols + threadIdx.x];
    }
    double* o = part + ((g * gridDim.x + ch) * 2) * cols;
    o[threadIdx.x] = a;
    o[cols + threadIdx.x] = b;
  }
}

// One block per group and per 256 columns: 256 / cols thread rows walk the chunk partials side by side and are combined
// in a fixed order through LDS (one thread per column walking all the chunks costs a dependent load per chunk).
__global__ __launch_bounds__(256) void moments_final_kernel(const double* __restrict__ part, int chunks, int cols, double n, int correction,
                                                            float* __restrict__ mean_out, float* __restrict__ var_out) {
  __shared__ double smd[2][256];
  const int bc = cols < 256 ? cols : 256;          // columns of this block
  const int lanes = 256 / bc;                      // thread rows over the chunks
  const int cl = threadIdx.x % bc, rl = threadIdx.x / bc;
  const int c = blockIdx.x * 256 + cl;
  const int64_t g = blockIdx.y;
  double s = 0, s2 = 0;
  if (rl < lanes && c < cols) {
    for (int ch = rl; ch < chunks; ch += lanes) {
      const double* o = part + ((g * chunks + ch) * 2) * cols;
      s += o[c];
      s2 += o[cols + c];
    }
  }
  smd[0][threadIdx.x] = s;
  smd[1][threadIdx.x] = s2;
  __syncthreads();
  if (rl == 0 && c < cols) {
    for (int i = 1; i < lanes; ++i) {
      s += smd[0][i * bc + cl];
      s2 += smd[1][i * bc + cl];
    }
    double m = s / n;
    double v = (s2 - s * s / n) / (n - (double)correction);
    mean_out[g * cols + c] = (float)m;
    var_out[g * cols + c] = (float)(v > 0 ? v : 0.0);
  }
}

__global__ void rms_merge_kernel(double* mean, double* var, double* count, const float* __restrict__ bmean,
                                 const float* __restrict__ bvar, double bcount, int n, float eps, float* mean_f,
                                 float* std_f) {
  RmsMergeArgs a;
  a.mean = mean; a.var = var; a.count = count; a.bmean = bmean; a.bvar = bvar; a.bcount = bcount; a.n = n; a.eps = eps; a.mean_f = mean_f; a.std_f = std_f;
  rms_merge_body(a);
}

__global__ void rms_normalize_kernel(const double* __restrict__ mean, const double* __restrict__ var, float eps, float lo,
                                     float hi, const float* __restrict__ x, float* __restrict__ y, int64_t total, int cols,
                                     int unnorm) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int c = (int)(i % cols);
  float m = (float)mean[c];
  float s = sqrtf((float)var[c] + eps);
  float v = x[i];
  if (unnorm) {
    v = fminf(fmaxf(v, lo), hi);                       // normalizers.py:70-73
    y[i] = add_rn(mul_rn(s, v), m);
  } else {
    y[i] = fminf(fmaxf(div_rn(sub_rn(v, m), s), lo), hi);   // normalizers.py:65-68
  }
}

// LayerNorm + SiLU backward as a streaming pass for the wide layers (N = 256 / 512), one wave per row:
//   dh (in `dz`, overwritten) -> dy = dh*silu'(y), y = xhat*gamma+beta -> dxhat = dy*gamma ->
//   dz = rstd*(dxhat - mean(dxhat) - xhat*mean(dxhat*xhat)); column partials dgamma, dbeta, dbias per block.
// Reads 2N floats and writes N floats per row, fully coalesced (16 B per lane).  grid = (blocks, towers).
struct LnBwdArgs {
  float* dz[2];
  const float* xhat[2];
  const float* rstd[2];
  const float* gamma[2];
  const float* beta[2];
  float* colpart[2];   // [gridDim.x][3][N]
  int M, rows_per_block;
  int act_elu;         // the activation after the LayerNorm is ELU (else SiLU)
};

template <int N>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdArgs a) {
  constexpr int C = N / 256;                 // float4 chunks per lane
  __shared__ f32x4 red[3][4][N / 4];
  const int tw = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* __restrict__ dz = a.dz[tw];
  const float* __restrict__ xh = a.xhat[tw];
  const float* __restrict__ rs = a.rstd[tw];
  f32x4 g[C], b[C], cg[C], cb[C], cz[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    g[c] = *reinterpret_cast<const f32x4*>(a.gamma[tw] + 4 * (lane + 64 * c));
    b[c] = *reinterpret_cast<const f32x4*>(a.beta[tw] + 4 * (lane + 64 * c));
    cg[c] = cb[c] = cz[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int r0 = blockIdx.x * a.rows_per_block;
  const int r1 = min(a.M, r0 + a.rows_per_block);
  const float invn = 1.0f / (float)N;
  for (int row = r0 + wave; row < r1; row += 4) {
    f32x4 x[C], d[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      x[c] = *reinterpret_cast<const f32x4*>(xh + (int64_t)row * N + 4 * (lane + 64 * c));
      d[c] = *reinterpret_cast<const f32x4*>(dz + (int64_t)row * N + 4 * (lane + 64 * c));
    }
    const float rstd = rs[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y = x[c][e] * g[c][e] + b[c][e];
        float dact;
        if (a.act_elu) dact = y > 0.f ? 1.0f : __builtin_amdgcn_exp2f(1.44269504088896340736f * y);
        else {
          const float sg = sigmoid_fast(y);
          dact = sg * (1.0f + y * (1.0f - sg));
        }
        const float dy = d[c][e] * dact;
        cg[c][e] += dy * x[c][e];
        cb[c][e] += dy;
        const float dxh = dy * g[c][e];
        d[c][e] = dxh;
        s1 += dxh;
        s2 += dxh * x[c][e];
      }
    s1 = mrl::wave_sum_desc(s1);
    s2 = mrl::wave_sum_desc(s2);
    const float m1 = s1 * invn, m2 = s2 * invn;
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = rstd * (d[c][e] - m1 - x[c][e] * m2);
        d[c][e] = z;
        cz[c][e] += z;
      }
      *reinterpret_cast<f32x4*>(dz + (int64_t)row * N + 4 * (lane + 64 * c)) = d[c];
    }
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    red[0][wave][lane + 64 * c] = cg[c];
    red[1][wave][lane + 64 * c] = cb[c];
    red[2][wave][lane + 64 * c] = cz[c];
  }
  __syncthreads();
  float* cp = a.colpart[tw] + (int64_t)blockIdx.x * 3 * N;
  for (int i = threadIdx.x; i < 3 * N / 4; i += 256) {
    const int qn = i / (N / 4), j = i % (N / 4);
    const f32x4 t = (red[qn][0][j] + red[qn][1][j]) + (red[qn][2][j] + red[qn][3][j]);
    *reinterpret_cast<f32x4*>(cp + 4 * i) = t;
  }
}

// Sum `nslabs` partial slabs into the flat gradient.  One descriptor per gradient tensor; grid.y = descriptor.
struct ReduceDesc { const float* src; float* dst; int64_t count; int64_t stride; int nslabs; int block0; };
#define MRL_MAX_RDESC 48
struct ReduceTable { ReduceDesc d[MRL_MAX_RDESC]; int n; };

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const ReduceTable tab) {
  // blockIdx.x counts the 16-chunk pieces of all descriptors (ReduceDesc::block0 = first piece of a descriptor)
  // block = 16 column chunks (of 4 floats) x 16 slab lanes: lane q sums slabs q, q+16, ... and the 16 partial sums are
  // combined in a fixed order through LDS, so the result does not depend on scheduling.  Many short dependent chains
  // beat long coalesced ones here: 64 chunks x 4 slab lanes measured 49 us against 28 us for this mapping.
  constexpr int CH = 16, QL = 16;
  __shared__ f32x4 red[QL][CH];
  int di = 0;
  for (int k = 1; k < tab.n; ++k)
    if ((int)blockIdx.x >= tab.d[k].block0) di = k;
  const ReduceDesc d = tab.d[di];
  const int lb = blockIdx.x - d.block0, nb = (di + 1 < tab.n ? tab.d[di + 1].block0 : (int)gridDim.x) - d.block0;
  const int cx = threadIdx.x & (CH - 1), q = threadIdx.x / CH;
  if ((d.count & 3) == 0 && (d.stride & 3) == 0) {
    const int64_t n4 = d.count >> 2;
    for (int64_t c0 = (int64_t)lb * CH; c0 < n4; c0 += (int64_t)nb * CH) {
      const int64_t i = c0 + cx;
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
      if (i < n4) {
        const float* src = d.src + 4 * i;
        int k = q;
        for (; k + 3 * QL < d.nslabs; k += 4 * QL) {
          s0 += *reinterpret_cast<const f32x4*>(src + (int64_t)k * d.stride);
          s1 += *reinterpret_cast<const f32x4*>(src + (int64_t)(k + QL) * d.stride);
          s2 += *reinterpret_cast<const f32x4*>(src + (int64_t)(k + 2 * QL) * d.stride);
          s3 += *reinterpret_cast<const f32x4*>(src + (int64_t)(k + 3 * QL) * d.stride);
        }
        for (; k < d.nslabs; k += QL) s0 += *reinterpret_cast<const f32x4*>(src + (int64_t)k * d.stride);
      }
      red[q][cx] = (s0 + s1) + (s2 + s3);
      __syncthreads();
      if (q == 0 && i < n4) {
        f32x4 t = red[0][cx];
#pragma unroll
        for (int w = 1; w < QL; ++w) t += red[w][cx];
        *reinterpret_cast<f32x4*>(d.dst + 4 * i) = t;
      }
      __syncthreads();
    }
  } else {
    for (int64_t i = (int64_t)lb * 256 + threadIdx.x; i < d.count; i += (int64_t)nb * 256) {
      float s = 0.f;
      for (int k = 0; k < d.nslabs; ++k) s += d.src[k * d.stride + i];
      d.dst[i] = s;
    }
  }
}

struct MetricsArgs {
  const double* part;   // [nblocks][2][MRL_NMETRIC] partial sums of the heads' row blocks
  int nblocks;
  double mb;
  float critic_coef, entropy_coef, bounds_coef;
  double* out;          // [MRL_NMETRIC]
};

// the first 192 threads of a block: 16 threads per metric, fixed-order sums so results are reproducible
__device__ __forceinline__ void metrics_final_body(const double* __restrict__ part, int nblocks, double mb, float critic_coef, float entropy_coef,
                                                   float bounds_coef, double* __restrict__ out, double (*red)[MRL_NMETRIC], double* s) {
  const int m = threadIdx.x % MRL_NMETRIC, j = threadIdx.x / MRL_NMETRIC;
  if (j < 16) {
    double a = 0;
#pragma unroll 8   // the loads of eight blocks in flight at once; the adds keep their order
    for (int b = j; b < nblocks; b += 16)
      a += part[((int64_t)b * 2 + 0) * MRL_NMETRIC + m] + part[((int64_t)b * 2 + 1) * MRL_NMETRIC + m];
    red[j][m] = a;
  }
  __syncthreads();
  if (threadIdx.x < MRL_NMETRIC) {
    double t = 0;
    for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
    s[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a_loss = s[0] / mb, ent = s[1] / mb, b_loss = s[2] / mb, kl = s[3] / mb, cf = s[4] / mb, c_loss = s[5] / mb;
    double var_d = (s[7] - s[6] * s[6] / mb) / (mb - 1.0);
    double var_r = (s[9] - s[8] * s[8] / mb) / (mb - 1.0);
    double ev = 1.0 - var_d / (var_r + 1e-8);          // ppo.py:475
    ev = ev < 0 ? 0 : (ev > 1 ? 1 : ev);
    out[0] = a_loss; out[1] = ent; out[2] = b_loss; out[3] = kl; out[4] = cf; out[5] = c_loss; out[6] = ev;
    out[7] = a_loss + 0.5 * c_loss * critic_coef - ent * entropy_coef + b_loss * bounds_coef;   // ppo.py:285
    out[8] = out[9] = out[10] = out[11] = 0;
  }
}

__global__ __launch_bounds__(192) void metrics_final_kernel(const double* __restrict__ part, int nblocks, double mb,
                                                            float critic_coef, float entropy_coef, float bounds_coef,
                                                            double* __restrict__ out) {
  __shared__ double red[16][MRL_NMETRIC];
  __shared__ double s[MRL_NMETRIC];
  metrics_final_body(part, nblocks, mb, critic_coef, entropy_coef, bounds_coef, out, red, s);
}

// ---- the squared-norm partials of a flat gradient vector --------------------------------------------------------------
// The vector is cut into 16-byte chunks; partial b covers chunks [b * 256 * cpt, (b + 1) * 256 * cpt), thread t of its block the
// chunks (b * cpt + i) * 256 + t, i = 0 .. cpt-1, in that order; the 256 thread sums are combined by a halving tree.  Every
// kernel that produces these partials (here from the finished vector, reduce_flat_kernel while it writes the vector) follows this
// one order, so the clip coefficient does not depend on which of them ran.  cpt is chosen so that there are at most 512 partials.
static inline int norm_cpt(int64_t count) { return (int)std::max<int64_t>(1, cdiv(cdiv(count, 4), 256 * 512)); }
static inline int norm_parts(int64_t count) { return (int)cdiv(cdiv(count, 4), 256 * (int64_t)norm_cpt(count)); }

__device__ __forceinline__ double norm_tree(double sq, double* red) {
  red[threadIdx.x] = sq;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  return red[0];
}
__device__ __forceinline__ double norm_chunk(double sq, f32x4 v) {
#pragma unroll
  for (int e = 0; e < 4; ++e) sq += (double)v[e] * (double)v[e];
  return sq;
}

// Sum the partial slabs into the flat gradient, one thread per 16-byte chunk of the vector (the descriptors, sorted by their
// place in the vector, are searched per chunk; chunks that belong to no tensor become zero), slabs added in index order.  The
// blocks are those of the norm partials above and write them; one more block finishes the metrics of the step.
struct FlatDesc { const float* src; int64_t stride; int nslabs; int c0, c1; };   // destination chunks [c0, c1)
struct FlatTable { FlatDesc d[MRL_MAX_RDESC]; int n, cpt, nchunks; };

__global__ __launch_bounds__(256) void reduce_flat_kernel(const FlatTable tab, float* __restrict__ grads, double* __restrict__ norm_part,
                                                          const MetricsArgs ma) {
  __shared__ double red[256];
  if (ma.part && blockIdx.x == gridDim.x - 1) {   // the extra block of launches that finish the metrics
    double (*mred)[MRL_NMETRIC] = reinterpret_cast<double (*)[MRL_NMETRIC]>(red);
    __shared__ double ms[MRL_NMETRIC];
    metrics_final_body(ma.part, ma.nblocks, ma.mb, ma.critic_coef, ma.entropy_coef, ma.bounds_coef, ma.out, mred, ms);
    return;
  }
  __shared__ int s_c0[MRL_MAX_RDESC];
  __shared__ FlatDesc s_d[MRL_MAX_RDESC];
  if ((int)threadIdx.x < tab.n) {
    s_d[threadIdx.x] = tab.d[threadIdx.x];
    s_c0[threadIdx.x] = tab.d[threadIdx.x].c0;
  }
  __syncthreads();
  double sq = 0.0;
  for (int i = 0; i < tab.cpt; ++i) {
    const int c = (blockIdx.x * tab.cpt + i) * 256 + threadIdx.x;
    if (c >= tab.nchunks) break;
    int lo = 0, hi = tab.n - 1;   // the last descriptor that starts at or before c
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (s_c0[mid] <= c) lo = mid; else hi = mid - 1;
    }
    const FlatDesc d = s_d[lo];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c >= d.c0 && c < d.c1) {
      // the pointer comes out of the LDS copy of the table: say that it points to global memory, or every load is a flat one
      typedef const __attribute__((address_space(1))) f32x4* gptr;
      const float* src = d.src + 4 * (int64_t)(c - d.c0);
      const int ns = d.nslabs;
      int k = 0;
      for (; k + 8 <= ns; k += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(gptr)(src + (int64_t)(k + u) * d.stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      }
      for (; k < ns; ++k) acc += *(gptr)(src + (int64_t)k * d.stride);
    }
    *reinterpret_cast<f32x4*>(grads + 4 * (int64_t)c) = acc;
    sq = norm_chunk(sq, acc);
  }
  const double t = norm_tree(sq, red);
  if (threadIdx.x == 0) norm_part[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void gradnorm_partial_kernel(const float* __restrict__ g, int64_t n, int cpt, float scale,
                                                               double* __restrict__ part) {
  __shared__ double red[256];
  const int64_t nchunks = (n + 3) >> 2;
  double sq = 0.0;
  for (int i = 0; i < cpt; ++i) {
    const int64_t c = ((int64_t)blockIdx.x * cpt + i) * 256 + threadIdx.x;
    if (c >= nchunks) break;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (4 * c + 3 < n) v = *reinterpret_cast<const f32x4*>(g + 4 * c);
    else
      for (int e = 0; 4 * c + e < n; ++e) v[e] = g[4 * c + e];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = mul_rn(v[e], scale);
    sq = norm_chunk(sq, v);
  }
  const double t = norm_tree(sq, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

struct AdamArgs {
  float lr_wd_mul;      // 1 - lr*wd           (AdamW)
  float l2;             // weight decay folded into the gradient (Adam)
  float one_m_b1, b2, one_m_b2;
  float bc2_sqrt, eps, neg_step_size;
  float max_norm, grad_scale;
  int adamw, nparts;
  int vec4;             // the four vectors are 16-byte aligned
};

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, const double* __restrict__ norm_part,
                                                   const AdamArgs a, float* __restrict__ gnorm_out) {
  // every block re-reduces the (<= 512) norm partials in the same fixed order: two partials per thread, then a tree
  // through LDS (a single thread walking the list costs a dependent global load per partial)
  __shared__ double s_red[256];
  __shared__ float s_coef;
  s_red[threadIdx.x] = (((int)threadIdx.x < a.nparts) ? norm_part[threadIdx.x] : 0.0) + (((int)threadIdx.x + 256 < a.nparts) ? norm_part[threadIdx.x + 256] : 0.0);
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float norm = (float)sqrt(s_red[0]);
    float coef = 1.0f;
    if (a.max_norm > 0.f) coef = fminf(a.max_norm / (norm + 1e-6f), 1.0f);   // torch clip_grad_norm_
    s_coef = coef;
    if (blockIdx.x == 0 && gnorm_out) gnorm_out[0] = norm;
  }
  __syncthreads();
  const float coef = s_coef;
  auto update = [&](float gi, float& pi, float& mi, float& vi) {
    gi = mul_rn(mul_rn(gi, a.grad_scale), coef);
    if (a.adamw) pi = mul_rn(pi, a.lr_wd_mul);
    else if (a.l2 != 0.f) gi = add_rn(gi, mul_rn(a.l2, pi));
    mi = add_rn(mi, mul_rn(a.one_m_b1, sub_rn(gi, mi)));                       // exp_avg.lerp_(g, 1-b1)
    vi = add_rn(mul_rn(vi, a.b2), mul_rn(mul_rn(a.one_m_b2, gi), gi));      // mul_(b2).addcmul_(g,g,1-b2)
    const float denom = add_rn(div_rn(sqrtf(vi), a.bc2_sqrt), a.eps);
    pi = add_rn(pi, mul_rn(a.neg_step_size, div_rn(mi, denom)));                // addcdiv_(m, denom, -step)
  };
  // 16 bytes per thread and pass when the vectors allow it (all four share the alignment of the pointers checked by the host)
  const int64_t n4 = a.vec4 ? (n >> 2) : 0;
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n4; c += (int64_t)gridDim.x * 256) {
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(g + 4 * c);
    f32x4 p4 = *reinterpret_cast<f32x4*>(p + 4 * c), m4 = *reinterpret_cast<f32x4*>(m + 4 * c), v4 = *reinterpret_cast<f32x4*>(v + 4 * c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pi = p4[e], mi = m4[e], vi = v4[e];
      update(g4[e], pi, mi, vi);
      p4[e] = pi; m4[e] = mi; v4[e] = vi;
    }
    *reinterpret_cast<f32x4*>(p + 4 * c) = p4;
    *reinterpret_cast<f32x4*>(m + 4 * c) = m4;
    *reinterpret_cast<f32x4*>(v + 4 * c) = v4;
  }
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float pi = p[i], mi = m[i], vi = v[i];
    update(g[i], pi, mi, vi);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// entry points
// ---------------------------------------------------------------------------------------------------------------
static int check_buf(const mrl_rollout* b) {
  if (!b) MRL_FAIL(MRL_EINVAL, "rollout buffer is NULL");
  if (b->num_envs < 1 || b->horizon < 1) MRL_FAIL(MRL_EINVAL, "bad rollout shape N=%lld T=%d", (long long)b->num_envs, b->horizon);
  if (b->num_envs * (int64_t)b->horizon >= (int64_t)1 << 31) MRL_FAIL(MRL_EINVAL, "N*T must be < 2^31");
  return MRL_OK;
}

extern "C" int mrl_policy_act(const mrl_net* net, const float* params, const mrl_rms* obs_rms, const mrl_rms* value_rms,
                              const float* obs, const float* noise, int32_t t, const mrl_rollout* buf,
                              float* actions_clamped, float* last_values, void* workspace, int64_t workspace_bytes,
                              void* stream) {
  if (check_net(net) || check_buf(buf)) return MRL_EINVAL;
  if (!params || !obs || !workspace) MRL_FAIL(MRL_EINVAL, "NULL params/obs/workspace");
  const int T = buf->horizon;
  const int64_t N = buf->num_envs;
  if (t < 0 || t > T) MRL_FAIL(MRL_EINVAL, "step %d outside 0..%d", t, T);
  const bool boot = (t == T);
  if (boot && !last_values) MRL_FAIL(MRL_EINVAL, "last_values required for the bootstrap step");
  if (!boot && !noise) MRL_FAIL(MRL_EINVAL, "noise required");
  if (obs_rms && obs_rms->size != net->obs_dim) MRL_FAIL(MRL_EINVAL, "obs_rms size %d != obs_dim %d", obs_rms->size, net->obs_dim);
  Workspace W;
  workspace_layout(net, N, &W);
  if (workspace_bytes < W.total * 4) MRL_FAIL(MRL_ENOSPC, "workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)W.total * 4);
  if (reinterpret_cast<uintptr_t>(workspace) & 255) MRL_FAIL(MRL_EINVAL, "workspace %p is not 256-byte aligned", workspace);
  ParamLayout L;
  param_layout(net, &L);
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const int O = net->obs_dim, A = net->act_dim;

  if (!boot) {
    int64_t total4 = N * (O / 4);
    int blocks = (int)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(copy_rows_kernel, dim3(blocks), dim3(256), 0, s, obs, buf->obs, N, O / 4, (int64_t)T * (O / 4), (int64_t)t * (O / 4));
  }
  float *mean_f = nullptr, *std_f = nullptr;
  float lo = -5.f, hi = 5.f;
  if (obs_rms) {
    mean_f = ws + W.small;
    std_f = mean_f + O;
    lo = obs_rms->clamp_lo; hi = obs_rms->clamp_hi;
    hipLaunchKernelGGL(rms_to_f32_kernel, dim3((O + 63) / 64), dim3(64), 0, s, obs_rms->mean, obs_rms->var, obs_rms->eps, O, mean_f, std_f);
  }
  const int tw0 = boot ? 1 : 0, towers = boot ? 1 : 2;
  const bool hact_ready = net->tower_kind > MRL_TOWER_LAYERNORM_ELU;   // these kinds read the activated output the chain kernel stored
  const unsigned short* wpl = (L.has_norm && N > MRL_SMALL_M && !chain_ok(net, N, 1)) ? launch_split_params(L, W, ws, params, s) : nullptr;
  launch_towers_fwd(net, L, W, ws, params, obs, N, mean_f, std_f, lo, hi, tw0, towers, s, wpl);

  GemmP p;
  memset(&p, 0, sizeof(p));
  const int last = net->n_layers - 1, U = net->units[last];
  for (int k = 0; k < towers; ++k) {
    int tw = tw0 + k;
    p.A[k] = hact_ready ? ws + W.hact[tw][last] : ws + W.xhat[tw][last];
    p.B[k] = params + L.Wh[tw];
    p.pro0[k] = L.has_norm ? params + L.g[tw][last] : nullptr;
    p.pro1[k] = L.has_norm ? params + L.be[tw][last] : nullptr;
    p.bias[k] = params + L.bh[tw];
    p.Nt[k] = L.head_rows[tw];
    p.Kt[k] = U;
  }
  if (net->fixed_sigma) p.h.fixed_logstd = params + L.bh[0] + A;
  p.tw0 = tw0;
  p.lda = U; p.ldb = U;
  p.M = (int)N; p.N = 32; p.K = U;
  fill_heads_common(net, &p.h);
  p.h.noise = noise;
  p.h.act_out = buf->actions; p.h.mu = buf->mu; p.h.sigma = buf->sigma;
  p.h.out_rs = (int64_t)T * A; p.h.out_off = (int64_t)t * A;
  p.h.act_clamped = actions_clamped;
  p.h.neglogp_out = boot ? nullptr : buf->neglogp_tn + (int64_t)t * N;
  p.h.value_out = boot ? last_values : buf->values_tn + (int64_t)t * N;
  if (value_rms) {
    p.h.vrms_mean = value_rms->mean; p.h.vrms_var = value_rms->var; p.h.vrms_eps = value_rms->eps;
    p.clamp_lo = value_rms->clamp_lo; p.clamp_hi = value_rms->clamp_hi;
  }
  if (hact_ready) launch_gemm<128, 32, 32, 4, 1, LAY_K, LAY_K, EPI_HEADS_ACT, PRO_NONE>(p, dim3((unsigned)cdiv(N, 128), 1, towers), s);
  else if (net->tower_kind == MRL_TOWER_ELU) launch_gemm<128, 32, 32, 4, 1, LAY_K, LAY_K, EPI_HEADS_ACT, PRO_ELU>(p, dim3((unsigned)cdiv(N, 128), 1, towers), s);
  else if (net->tower_kind == TOWER_LAYERNORM_ELU) launch_gemm<128, 32, 32, 4, 1, LAY_K, LAY_K, EPI_HEADS_ACT, PRO_LNELU>(p, dim3((unsigned)cdiv(N, 128), 1, towers), s);
  else launch_gemm<128, 32, 32, 4, 1, LAY_K, LAY_K, EPI_HEADS_ACT, PRO_LNSILU>(p, dim3((unsigned)cdiv(N, 128), 1, towers), s);
  MRL_CHECK_LAUNCH("mrl_policy_act");
  return MRL_OK;
}

extern "C" int mrl_rollout_write(const mrl_rollout* buf, int32_t t, const float* rewards, const uint8_t* dones,
                                 const uint8_t* time_outs, double reward_scale, double gamma, void* stream) {
  if (check_buf(buf)) return MRL_EINVAL;
  if (t < 0 || t >= buf->horizon) MRL_FAIL(MRL_EINVAL, "step %d outside 0..%d", t, buf->horizon - 1);
  if (!rewards || !dones) MRL_FAIL(MRL_EINVAL, "NULL rewards/dones");
  const int64_t N = buf->num_envs;
  {
    MRL_PROF_STREAM("rollout_write_kernel", (hipStream_t)stream);
    hipLaunchKernelGGL(rollout_write_kernel, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream,
                       buf->rewards_tn + (int64_t)t * N, buf->dones_tn + (int64_t)t * N, buf->values_tn + (int64_t)t * N, rewards,
                       dones, time_outs, N, (float)reward_scale, (float)gamma);
  }
  MRL_CHECK_LAUNCH("mrl_rollout_write");
  return MRL_OK;
}

extern "C" int mrl_gae_scan(const mrl_rollout* buf, const float* last_values, double gamma, double tau, double* sums_out,
                            void* workspace, int64_t workspace_bytes, void* stream) {
  if (check_buf(buf)) return MRL_EINVAL;
  if (!last_values || !workspace || !sums_out) MRL_FAIL(MRL_EINVAL, "NULL last_values/workspace/sums_out");
  const int64_t N = buf->num_envs;
  const int T = buf->horizon;
  const int nblocks = (int)cdiv(N, 64);
  const int64_t need = (int64_t)nblocks * 6 * 8;
  if (workspace_bytes < need) MRL_FAIL(MRL_ENOSPC, "workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
  if (reinterpret_cast<uintptr_t>(workspace) & 255) MRL_FAIL(MRL_EINVAL, "workspace %p is not 256-byte aligned", workspace);
  size_t smem = (size_t)4 * 64 * (T + 1) * sizeof(float);
  if (smem > 64 * 1024) MRL_FAIL(MRL_EINVAL, "horizon %d too long for the scan tile", T);
  hipStream_t s = (hipStream_t)stream;
  double* part = (double*)workspace;
  const float gamma_tau = (float)(gamma * tau);   // the reference multiplies the two Python floats first
  {
    MRL_PROF_STREAM("gae_kernel", s);
    hipLaunchKernelGGL(gae_kernel, dim3(nblocks), dim3(64), smem, s, buf->rewards_tn, buf->values_tn, buf->dones_tn,
                       buf->neglogp_tn, last_values, N, T, (float)gamma, gamma_tau, buf->returns, buf->values, buf->advantages,
                       buf->neglogp, part);
  }
  hipLaunchKernelGGL(gae_sums_kernel, dim3(1), dim3(256), 0, s, part, nblocks, sums_out);
  MRL_CHECK_LAUNCH("mrl_gae_scan");
  return MRL_OK;
}

extern "C" int mrl_gae_normalize(const mrl_rollout* buf, const double* sums, double count, const mrl_rms* value_rms,
                                 float* stats_out, void* stream) {
  if (check_buf(buf)) return MRL_EINVAL;
  if (!sums || !stats_out) MRL_FAIL(MRL_EINVAL, "NULL sums/stats_out");
  if (count < 2) MRL_FAIL(MRL_EINVAL, "count %g < 2", count);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gae_stats_kernel, dim3(1), dim3(64), 0, s, sums, count, value_rms ? value_rms->mean : nullptr,
                     value_rms ? value_rms->var : nullptr, value_rms ? value_rms->count : nullptr,
                     value_rms ? value_rms->eps : 0.f, stats_out);
  const int64_t B = buf->num_envs * (int64_t)buf->horizon;
  {
    MRL_PROF_STREAM("gae_normalize_kernel", s);
    hipLaunchKernelGGL(gae_normalize_kernel, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, s, buf->advantages, buf->values,
                       buf->returns, stats_out, B, value_rms ? 1 : 0, value_rms ? value_rms->clamp_lo : 0.f,
                       value_rms ? value_rms->clamp_hi : 0.f);
  }
  MRL_CHECK_LAUNCH("mrl_gae_normalize");
  return MRL_OK;
}

extern "C" int mrl_episode_update(const float* rewards, const uint8_t* dones, int64_t num_envs, float* cur_rewards, int64_t* cur_lengths,
                                  double* ring_rewards, double* ring_lengths, int32_t window, int64_t* head, int64_t* num_episodes, void* stream) {
  if (!rewards || !dones || !cur_rewards || !cur_lengths || !ring_rewards || !ring_lengths || !head || !num_episodes) MRL_FAIL(MRL_EINVAL, "NULL pointer");
  if (num_envs < 1 || window < 1) MRL_FAIL(MRL_EINVAL, "num_envs %lld, window %d", (long long)num_envs, window);
  hipLaunchKernelGGL(episode_update_kernel, dim3((unsigned)cdiv(num_envs, 1024)), dim3(256), 0, (hipStream_t)stream, rewards, dones, num_envs,
                     cur_rewards, cur_lengths, ring_rewards, ring_lengths, window, (const int64_t*)head);
  hipLaunchKernelGGL(episode_commit_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, dones, num_envs, window, head, num_episodes);
  MRL_CHECK_LAUNCH("mrl_episode_update");
  return MRL_OK;
}

extern "C" int mrl_info_track(const float* x, const uint8_t* dones, int64_t num_envs, int32_t mode, int32_t reset_nan, float* cur, double* ring,
                              int32_t window, const int64_t* head, void* stream) {
  if (!x || !dones || !cur || !ring || !head) MRL_FAIL(MRL_EINVAL, "NULL pointer");
  if (num_envs < 1 || window < 1) MRL_FAIL(MRL_EINVAL, "num_envs %lld, window %d", (long long)num_envs, window);
  if (mode < 0 || mode > 3) MRL_FAIL(MRL_EINVAL, "mode %d not in 0..3 (sum, min, max, final)", mode);
  hipLaunchKernelGGL(info_track_kernel, dim3((unsigned)cdiv(num_envs, 1024)), dim3(256), 0, (hipStream_t)stream, x, dones, num_envs, mode, reset_nan,
                     cur, ring, window, head);
  MRL_CHECK_LAUNCH("mrl_info_track");
  return MRL_OK;
}

extern "C" int mrl_policy_mean(const mrl_net* net, const float* params, const mrl_rms* obs_rms, const float* obs, int64_t rows, float* mu_out,
                               void* workspace, int64_t workspace_bytes, void* stream) {
  if (check_net(net)) return MRL_EINVAL;
  if (!params || !obs || !mu_out || !workspace) MRL_FAIL(MRL_EINVAL, "NULL params/obs/mu_out/workspace");
  if (rows < 1 || rows >= ((int64_t)1 << 31)) MRL_FAIL(MRL_EINVAL, "rows %lld", (long long)rows);
  if (obs_rms && obs_rms->size != net->obs_dim) MRL_FAIL(MRL_EINVAL, "obs_rms size %d != obs_dim %d", obs_rms->size, net->obs_dim);
  Workspace W;
  workspace_layout(net, rows, &W);
  if (workspace_bytes < W.total * 4) MRL_FAIL(MRL_ENOSPC, "workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)W.total * 4);
  if (reinterpret_cast<uintptr_t>(workspace) & 255) MRL_FAIL(MRL_EINVAL, "workspace %p is not 256-byte aligned", workspace);
  ParamLayout L;
  param_layout(net, &L);
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const int O = net->obs_dim, A = net->act_dim, last = net->n_layers - 1, U = net->units[last];
  float *mean_f = nullptr, *std_f = nullptr;
  float lo = -5.f, hi = 5.f;
  if (obs_rms) {
    mean_f = ws + W.small;
    std_f = mean_f + O;
    lo = obs_rms->clamp_lo; hi = obs_rms->clamp_hi;
    hipLaunchKernelGGL(rms_to_f32_kernel, dim3((O + 63) / 64), dim3(64), 0, s, obs_rms->mean, obs_rms->var, obs_rms->eps, O, mean_f, std_f);
  }
  const bool hact_ready = net->tower_kind > MRL_TOWER_LAYERNORM_ELU;
  const unsigned short* wpl = (L.has_norm && rows > MRL_SMALL_M && !chain_ok(net, rows, 1)) ? launch_split_params(L, W, ws, params, s) : nullptr;
  launch_towers_fwd(net, L, W, ws, params, obs, rows, mean_f, std_f, lo, hi, 0, 1, s, wpl);
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A[0] = hact_ready ? ws + W.hact[0][last] : ws + W.xhat[0][last];
  p.B[0] = params + L.Wh[0];
  p.pro0[0] = L.has_norm ? params + L.g[0][last] : nullptr;
  p.pro1[0] = L.has_norm ? params + L.be[0][last] : nullptr;
  p.bias[0] = params + L.bh[0];
  p.Nt[0] = L.head_rows[0];
  p.Kt[0] = U;
  if (net->fixed_sigma) p.h.fixed_logstd = params + L.bh[0] + A;
  p.lda = U; p.ldb = U;
  p.M = (int)rows; p.N = 32; p.K = U;
  fill_heads_common(net, &p.h);
  p.h.raw_mu = 1;
  p.h.mu = mu_out;
  p.h.out_rs = A; p.h.out_off = 0;
  if (hact_ready) launch_gemm<128, 32, 32, 4, 1, LAY_K, LAY_K, EPI_HEADS_ACT, PRO_NONE>(p, dim3((unsigned)cdiv(rows, 128), 1, 1), s);
  else if (net->tower_kind == MRL_TOWER_ELU) launch_gemm<128, 32, 32, 4, 1, LAY_K, LAY_K, EPI_HEADS_ACT, PRO_ELU>(p, dim3((unsigned)cdiv(rows, 128), 1, 1), s);
  else if (net->tower_kind == TOWER_LAYERNORM_ELU) launch_gemm<128, 32, 32, 4, 1, LAY_K, LAY_K, EPI_HEADS_ACT, PRO_LNELU>(p, dim3((unsigned)cdiv(rows, 128), 1, 1), s);
  else launch_gemm<128, 32, 32, 4, 1, LAY_K, LAY_K, EPI_HEADS_ACT, PRO_LNSILU>(p, dim3((unsigned)cdiv(rows, 128), 1, 1), s);
  MRL_CHECK_LAUNCH("mrl_policy_mean");
  return MRL_OK;
}

extern "C" int mrl_gae_finish(const mrl_rollout* buf, const float* last_values, double gamma, double tau,
                              const mrl_rms* value_rms, float* stats_out, void* workspace, int64_t workspace_bytes,
                              void* stream) {
  if (check_buf(buf)) return MRL_EINVAL;
  if (!workspace) MRL_FAIL(MRL_EINVAL, "NULL workspace");
  const int nblocks = (int)cdiv(buf->num_envs, 64);
  const int64_t need = (int64_t)nblocks * 6 * 8 + 6 * 8 + 8 * 4;
  if (workspace_bytes < need) MRL_FAIL(MRL_ENOSPC, "workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
  if (reinterpret_cast<uintptr_t>(workspace) & 255) MRL_FAIL(MRL_EINVAL, "workspace %p is not 256-byte aligned", workspace);
  double* sums = (double*)workspace + (int64_t)nblocks * 6;
  float* stats = stats_out ? stats_out : (float*)(sums + 6);
  int rc = mrl_gae_scan(buf, last_values, gamma, tau, sums, workspace, workspace_bytes, stream);
  if (rc) return rc;
  return mrl_gae_normalize(buf, sums, (double)(buf->num_envs * (int64_t)buf->horizon), value_rms, stats, stream);
}

// ---- the entry point that allocates nothing ----------------------------------------------------------------------
// mrl_gae_finish with its scratch (per-block partial sums, the six sums, the statistics) in a file-scope device array
// instead of a caller workspace.  Every GPU the library runs on holds its own copy of the array at its own address;
// the address is looked up once per device, with that device current, and kept.  Nothing in the scratch has to start
// at zero: gae_kernel writes every partial that gae_sums_kernel reads, and the sums / statistics are written before
// they are read, so the contents left by an earlier (larger) call are never consumed.
constexpr int64_t MRL_STATIC_GAE_MAX_ENVS = (int64_t)1 << 20;
constexpr int64_t MRL_STATIC_SCRATCH_BYTES = (MRL_STATIC_GAE_MAX_ENVS / 64) * 6 * 8 + 256;   // 786,688 bytes per GPU
static __device__ __align__(256) unsigned char scratch[MRL_STATIC_SCRATCH_BYTES];
constexpr int MRL_MAX_DEVICES = 64;
static std::atomic<void*> g_scratch_ptr[MRL_MAX_DEVICES];

static int static_scratch(void** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) MRL_FAIL(MRL_EHIP, "hipGetDevice: %s", hipGetErrorString(e));
  if (dev < 0 || dev >= MRL_MAX_DEVICES) MRL_FAIL(MRL_EINVAL, "device %d outside 0..%d", dev, MRL_MAX_DEVICES - 1);
  void* ptr = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (!ptr) {
    e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));   // the address of THIS device's copy (the current device)
    if (e != hipSuccess || !ptr) MRL_FAIL(MRL_EHIP, "hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
    g_scratch_ptr[dev].store(ptr, std::memory_order_release);   // concurrent first calls look up the same address
  }
  *out = ptr;
  return MRL_OK;
}

extern "C" int mrl_gae_finish_static(const mrl_rollout* buf, const float* last_values, double gamma, double tau,
                                     const mrl_rms* value_rms, float* stats_out, void* stream) {
  if (check_buf(buf)) return MRL_EINVAL;
  if (buf->num_envs > MRL_STATIC_GAE_MAX_ENVS)
    MRL_FAIL(MRL_EINVAL, "mrl_gae_finish_static: %lld envs exceed the built-in scratch (at most %lld envs; use mrl_gae_finish with a workspace)",
             (long long)buf->num_envs, (long long)MRL_STATIC_GAE_MAX_ENVS);
  void* ptr = nullptr;
  if (int rc = static_scratch(&ptr)) return rc;
  return mrl_gae_finish(buf, last_values, gamma, tau, value_rms, stats_out, ptr, MRL_STATIC_SCRATCH_BYTES, stream);
}

extern "C" int mrl_batch_moments(const float* x, int64_t rows, int32_t cols, int64_t group_rows, int32_t correction,
                                 float* mean_out, float* var_out, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!x || !mean_out || !var_out || !workspace) MRL_FAIL(MRL_EINVAL, "NULL pointer");
  if (cols < 1 || cols > 256) MRL_FAIL(MRL_EINVAL, "cols %d not in 1..256", cols);
  if (rows < 1 || group_rows < 1 || rows % group_rows) MRL_FAIL(MRL_EINVAL, "rows %lld not a multiple of group_rows %lld", (long long)rows, (long long)group_rows);
  const int64_t groups = rows / group_rows;
  int64_t chunks = cdiv(1024, groups);
  int64_t chunk_rows = cdiv(group_rows, chunks);
  if (chunk_rows < 64) chunk_rows = 64;
  chunks = cdiv(group_rows, chunk_rows);
  if (groups > 65535) MRL_FAIL(MRL_EINVAL, "too many groups");
  const int64_t need = groups * chunks * 2 * cols * 8;
  if (workspace_bytes < need) MRL_FAIL(MRL_ENOSPC, "workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
  if (reinterpret_cast<uintptr_t>(workspace) & 255) MRL_FAIL(MRL_EINVAL, "workspace %p is not 256-byte aligned", workspace);
  hipStream_t s = (hipStream_t)stream;
  MRL_PROF_STREAM("moments_partial_kernel", s);
  if (cols % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const int rowlanes = 256 / (cols / 4);
    hipLaunchKernelGGL(moments_partial4_kernel, dim3((unsigned)chunks, (unsigned)groups), dim3(256), (size_t)2 * rowlanes * cols * sizeof(double), s, x, cols,
                       group_rows, chunk_rows, (double*)workspace);
  } else {
    const int rowlanes = 256 / cols > 0 ? 256 / cols : 1;
    hipLaunchKernelGGL(moments_partial_kernel, dim3((unsigned)chunks, (unsigned)groups), dim3(256), (size_t)2 * rowlanes * cols * sizeof(double), s, x, cols,
                       group_rows, chunk_rows, (double*)workspace);
  }
  hipLaunchKernelGGL(moments_final_kernel, dim3((cols + 255) / 256, (unsigned)groups), dim3(256), 0, s, (const double*)workspace,
                     (int)chunks, cols, (double)group_rows, correction, mean_out, var_out);
  MRL_CHECK_LAUNCH("mrl_batch_moments");
  return MRL_OK;
}

extern "C" int mrl_rms_merge(const mrl_rms* rms, const float* batch_mean, const float* batch_var, double batch_count,
                             float* mean_f32_out, float* std_f32_out, void* stream) {
  if (!rms || !rms->mean || !rms->var || !rms->count || !batch_mean || !batch_var) MRL_FAIL(MRL_EINVAL, "NULL pointer");
  if (rms->size < 1 || rms->size > 1024) MRL_FAIL(MRL_EINVAL, "rms size %d not in 1..1024", rms->size);
  hipLaunchKernelGGL(rms_merge_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rms->mean, rms->var, rms->count, batch_mean,
                     batch_var, batch_count, rms->size, rms->eps, mean_f32_out, std_f32_out);
  MRL_CHECK_LAUNCH("mrl_rms_merge");
  return MRL_OK;
}

extern "C" int mrl_rms_normalize(const mrl_rms* rms, const float* x, float* y, int64_t rows, int32_t unnormalize, void* stream) {
  if (!rms || !x || !y) MRL_FAIL(MRL_EINVAL, "NULL pointer");
  if (rows < 0) MRL_FAIL(MRL_EINVAL, "rows < 0");
  int64_t total = rows * rms->size;
  if (total == 0) return MRL_OK;
  hipLaunchKernelGGL(rms_normalize_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, rms->mean, rms->var,
                     rms->eps, rms->clamp_lo, rms->clamp_hi, x, y, total, rms->size, unnormalize);
  MRL_CHECK_LAUNCH("mrl_rms_normalize");
  return MRL_OK;
}

// gradient with respect to the input of the first layer, one [mb][in_dim] matrix per tower: dX = dZ0 * W0
static void launch_input_grad(const mrl_net* net, const ParamLayout& L, const Workspace& W, float* ws, const float* params, int64_t mb, int tw0, int towers,
                              float* const* input_grad, hipStream_t s) {
  GemmP p;
  memset(&p, 0, sizeof(p));
  const int u0 = net->units[0], in0 = L.in_dim[0];
  for (int k = 0; k < towers; ++k) {
    const int tw = tw0 + k;
    p.A[k] = ws + W.dz[tw][0];
    p.B[k] = params + L.W[tw][0];
    p.Kt[k] = u0; p.Nt[k] = in0;
    p.out[k] = input_grad[k];
  }
  p.lda = u0; p.ldb = in0; p.ldo = in0;
  p.M = (int)mb; p.N = in0; p.K = u0;
  p.ksplit = u0; p.m_tiles = (int)cdiv(mb, 128); p.n_tiles = (int)cdiv(in0, 128);
  p.n_groups = p.m_tiles; p.group_mode = 0;
  launch_gemm<128, 128, 16, 2, 2, LAY_K, LAY_R, EPI_SLAB, PRO_NONE>(p, dim3(slab_grid(p.n_groups, p.n_tiles), 1, towers), s);
}

static void fill_chain_bwd(const mrl_net* net, const ParamLayout& L, const Workspace& W, float* ws, const float* params, int64_t mb, int tw0, int towers,
                           ChainBwdP* out) {
  ChainBwdP& p = *out;
  memset(&p, 0, sizeof(p));
  const int nl = net->n_layers;
  p.M = (int)mb; p.n_layers = nl; p.ld = chain_ld(net, 32);
  for (int l = 0; l < nl; ++l) {
    p.units[l] = net->units[l];
    for (int t = 0; t < towers; ++t) {
      const int tw = tw0 + t;
      const bool lastl = l == nl - 1;
      p.Wn[t][l] = chain_natural() ? (lastl ? params + L.Wh[tw] : params + L.W[tw][l + 1]) : ws + W.wblk_b[tw][l];
      p.wn_rows[t][l] = lastl ? L.head_rows[tw] : net->units[l + 1];
      p.xhat[t][l] = ws + W.xhat[tw][l];
      p.rstd[t][l] = ws + W.rstd[tw][l];
      p.g[t][l] = L.has_norm ? params + L.g[tw][l] : nullptr;
      p.be[t][l] = L.has_norm ? params + L.be[tw][l] : nullptr;
      p.dz[t][l] = ws + W.dz[tw][l];
      p.dy[t][l] = L.has_norm ? ws + W.dy[tw][l] : nullptr;
      p.colblk[t][l] = chain_tile_wgrad(net, mb) ? ws + W.colblk[tw][l] : nullptr;
    }
  }
  for (int t = 0; t < towers; ++t) p.seeds[t] = ws + W.seeds[tw0 + t];
  p.natural = chain_natural();
}

static void launch_chain_wgrad(const mrl_net* net, const ParamLayout& L, const Workspace& W, float* ws, const float* params, const float* x, int64_t mb,
                               const float* obs_mean, const float* obs_std, float clamp_lo, float clamp_hi, int tw0, int towers, hipStream_t s);

// Backward of a small batch (chain.hip.h): the dgrad chain of all layers in one launch, then (wgrad) every weight gradient of
// the towers in one launch.  Same buffers as the per-layer launches: dZ per layer, column partials per row range, slabs.
static void launch_chain_bwd(const mrl_net* net, const ParamLayout& L, const Workspace& W, float* ws, const float* params, const float* x, int64_t mb,
                             const float* obs_mean, const float* obs_std, float clamp_lo, float clamp_hi, int tw0, int towers, bool wgrad, hipStream_t s,
                             bool blocked_ready) {
  if (!blocked_ready) launch_reblock(net, L, W, ws, params, tw0, towers, false, true, s);
  {
    ChainBwdP p;
    fill_chain_bwd(net, L, W, ws, params, mb, tw0, towers, &p);
    chain_allow_lds();
    MRL_PROF_STREAM("mrl::chain_bwd_kernel", s);
    const dim3 grid((unsigned)cdiv(mb, CH_ROWS), towers);
    const size_t lds = chain_lds_bytes(p.ld) + chain_bwd_extra_lds(p.ld);
    MRL_CHAIN_LAUNCH(chain_bwd_kernel, net->tower_kind, grid, lds, s, p);
  }
  if (wgrad) launch_chain_wgrad(net, L, W, ws, params, x, mb, obs_mean, obs_std, clamp_lo, clamp_hi, tw0, towers, s);
}

// Every weight gradient of the towers (layers and heads) in one launch, from the dZ / dY / activations the chain kernels stored.
static void launch_chain_wgrad(const mrl_net* net, const ParamLayout& L, const Workspace& W, float* ws, const float* params, const float* x, int64_t mb,
                               const float* obs_mean, const float* obs_std, float clamp_lo, float clamp_hi, int tw0, int towers, hipStream_t s) {
  const int nl = net->n_layers, last = nl - 1, U = net->units[last];
  if (chain_tile_wgrad(net, mb)) {
    const ChainWgPlan pl = chain_wgrad_plan(net, L.in_dim, mb);
    WgTileP q;
    memset(&q, 0, sizeof(q));
    q.M = (int)mb;
    int task = 0;
    for (int t = 0; t < towers; ++t) {
      const int tw = tw0 + t;
      for (int l = last + 1; l >= 0; --l) {
        WgTileDesc& d = q.d[q.nd++];
        const bool head = l == last + 1;
        d.u = head ? 32 : net->units[l];
        d.in = head ? U : L.in_dim[l];
        d.dz = head ? ws + W.seeds[tw] : ws + W.dz[tw][l];
        d.xin = (l == 0) ? (obs_mean ? ws + W.xnorm : x) : ws + W.hact[tw][l - 1];
        d.slab = head ? ws + W.slab_h[tw] : ws + W.slab[tw][l];
        d.colpart = head ? ws + W.colpart_h[tw] : nullptr;   // the layers' column sums come from the backward chain (colblk)
        d.bm = pl.bm[l]; d.bn = pl.bn[l];
        d.nsplit = pl.nsplit[l];
        d.tiles_n = (int)cdiv(d.in, d.bn);
        d.tiles = (int)cdiv(d.u, d.bm) * d.tiles_n;
        d.task0 = task;
        task += d.tiles * d.nsplit;
      }
    }
    q.ntasks = task;
    const int nblk = (int)cdiv(mb, CH_ROWS);
    for (int t = 0; t < towers; ++t)
      for (int l = 0; l < nl; ++l) {
        WgColDesc& cd = q.cd[q.ncd++];
        cd.src = ws + W.colblk[tw0 + t][l]; cd.dst = ws + W.colpart[tw0 + t][l];
        cd.n4 = 3 * net->units[l] / 4; cd.nblk = nblk; cd.groups = W.col_groups;
        cd.task0 = task;
        task += (int)cdiv((int64_t)cd.n4 * cd.groups, 1024);
      }
    if (pl.split_bf16) {
      MRL_PROF_STREAM("mrl::wgrad_group_kernel", s);
      hipLaunchKernelGGL(wgrad_group_kernel, dim3((unsigned)task), dim3(512), WGG_LDS, s, q);
      return;
    }
    MRL_PROF_STREAM("mrl::wgrad_tile_kernel", s);
    hipLaunchKernelGGL(wgrad_tile_kernel, dim3((unsigned)task), dim3(512), WT_LDS_F * sizeof(float), s, q);
    return;
  }
  WgradSmallP q;
  memset(&q, 0, sizeof(q));
  q.M = (int)mb; q.nsplit = W.wph.nsplit; q.ksplit = W.wph.ksplit; q.lo = clamp_lo; q.hi = clamp_hi;
  // with the forward chain the operands were stored ready-made (activated outputs, normalised input); otherwise they are
  // rebuilt from x-hat / the raw input by the operand prologue
  const bool ready = chain_ok(net, mb, 1);
  const int pro_hidden = ready ? PRO_NONE : (L.has_norm ? (net->tower_kind == TOWER_LAYERNORM_ELU ? PRO_LNELU : PRO_LNSILU) : PRO_ELU);
  int task = 0;
  for (int t = 0; t < towers; ++t) {
    const int tw = tw0 + t;
    for (int l = last + 1; l >= 0; --l) {
      WgDesc& d = q.d[q.nd++];
      const bool head = l == last + 1;
      d.u = head ? 32 : net->units[l];
      d.in = head ? U : L.in_dim[l];
      d.ldz = d.u; d.slab_rows = d.u;
      d.dz = head ? ws + W.seeds[tw] : ws + W.dz[tw][l];
      d.xin = (l == 0) ? ((ready && obs_mean) ? ws + W.xnorm : x) : (ready ? ws + W.hact[tw][l - 1] : ws + W.xhat[tw][l - 1]);
      d.pro = (l == 0) ? ((obs_mean && !ready) ? PRO_NORM : PRO_NONE) : pro_hidden;
      d.pro0 = (l == 0) ? obs_mean : (L.has_norm ? params + L.g[tw][l - 1] : nullptr);
      d.pro1 = (l == 0) ? obs_std : (L.has_norm ? params + L.be[tw][l - 1] : nullptr);
      d.slab = head ? ws + W.slab_h[tw] : ws + W.slab[tw][l];
      if (!head) {   // column sums of this layer: d bias from dZ, d gamma / d beta from dY and x-hat
        d.colpart = ws + W.colpart[tw][l];
        d.dy = L.has_norm ? ws + W.dy[tw][l] : nullptr;
        d.xout = L.has_norm ? ws + W.xhat[tw][l] : nullptr;
      } else {       // the head's bias gradient: the column sums of the seeds
        d.colpart = ws + W.colpart_h[tw];
      }
      d.tiles_n = (int)cdiv(d.in, 64);
      d.tiles = (int)cdiv(d.u, 64) * d.tiles_n;
      d.task0 = task;
      task += d.tiles * q.nsplit;
    }
  }
  q.ntasks = task;
  MRL_PROF_STREAM("mrl::wgrad_small_kernel", s);
  // 84 KB of (unused) dynamic LDS keep the blocks (8 waves: 4 tasks, two waves each) at one per CU: a second block would share
  // the matrix cores and take twice as long while other CUs sit idle
  static const int wg_lds = [] { const char* e = getenv("MRL_CHAIN_WGLDS"); return e ? atoi(e) : 84 * 1024; }();
  hipLaunchKernelGGL(wgrad_small_kernel, dim3((unsigned)cdiv(task, 4)), dim3(512), wg_lds, s, q);
}

// Backward of `towers` towers (tower indices tw0 .. tw0+towers-1 of the workspace / layout) from the head seeds
// ws[W.seeds[tw]] ([rows][32], zero padded): dgrad chain with the LayerNorm/SiLU backward of every layer, then (wgrad)
// the partial weight-gradient slabs of every layer and head, and (input_grad) the gradient with respect to the first
// layer's input, one [rows][in_dim] matrix per tower.
// true when the backward of this network on `mb` rows reads its weights from planes split beforehand
static bool bwd_uses_planes(const mrl_net* net, const ParamLayout& L, int64_t mb) {
  if (!L.has_norm || mb <= MRL_SMALL_M) return false;
  for (int l = 0; l + 1 < net->n_layers; ++l)
    if (ln_bwd_streamed(net->units[l]) && net->units[l + 1] % 16 == 0) return true;
  return false;
}

// planes_ready: the caller has already refreshed the weight planes of the workspace on a stream this one is ordered behind
static void launch_towers_bwd(const mrl_net* net, const ParamLayout& L, const Workspace& W, float* ws, const float* params, const float* x, int64_t mb,
                              const float* obs_mean, const float* obs_std, float clamp_lo, float clamp_hi, int tw0, int towers, bool wgrad,
                              float* const* input_grad, hipStream_t s, bool planes_ready = false, hipStream_t wgrad_stream = nullptr,
                              hipEvent_t* wgrad_events = nullptr, const std::function<void(hipStream_t)>* beside_first = nullptr,
                              bool* beside_first_done = nullptr, bool last_done = false) {
  const int nl = net->n_layers, last = nl - 1, U = net->units[last];
  if (chain_ok(net, mb, 2)) {
    // small batches: the whole dgrad chain in one launch, every weight gradient in a second one
    launch_chain_bwd(net, L, W, ws, params, x, mb, obs_mean, obs_std, clamp_lo, clamp_hi, tw0, towers, wgrad, s, planes_ready);
    if (input_grad) launch_input_grad(net, L, W, ws, params, mb, tw0, towers, input_grad, s);
    return;
  }
  // weight-gradient slabs of layer l (l == last + 1: the heads) on stream st
  auto wgrad_of = [&](int l, hipStream_t st) {
    GemmP p;
    memset(&p, 0, sizeof(p));
    const bool head = (l == last + 1);
    const int gm = head ? 32 : net->units[l];
    const int gn = head ? U : L.in_dim[l];
    const WgradPlan& wp = head ? W.wph : W.wp[l];
    for (int k = 0; k < towers; ++k) {
      const int tw = tw0 + k;
      p.A[k] = head ? ws + W.seeds[tw] : ws + W.dz[tw][l];
      p.B[k] = (l == 0) ? x : ws + W.xhat[tw][l - 1];
      p.pro0[k] = (l == 0) ? obs_mean : (L.has_norm ? params + L.g[tw][l - 1] : nullptr);
      p.pro1[k] = (l == 0) ? obs_std : (L.has_norm ? params + L.be[tw][l - 1] : nullptr);
      p.out[k] = head ? ws + W.slab_h[tw] : ws + W.slab[tw][l];
      p.Nt[k] = gn; p.Kt[k] = (int)mb;
    }
    p.lda = gm; p.ldb = gn; p.ldo = gn;
    p.M = gm; p.N = gn; p.K = (int)mb;
    p.clamp_lo = clamp_lo; p.clamp_hi = clamp_hi;
    p.ksplit = wp.ksplit; p.m_tiles = wp.tiles_m; p.n_tiles = wp.tiles_n;
    p.n_groups = wp.nsplit; p.group_mode = 1;
    launch_wgrad(p, wp, (l == 0) ? (obs_mean ? PRO_NORM : PRO_NONE) : (L.has_norm ? (net->tower_kind == TOWER_LAYERNORM_ELU ? PRO_LNELU : PRO_LNSILU) : PRO_ELU), towers, st);
  };
  // With a weight-gradient stream, the slabs of layer l are launched there as soon as dZ of layer l is final on s, beside
  // the rest of the dgrad chain (they only read dZ[l] and xhat[l-1]; every layer has buffers of its own).  The caller joins
  // that stream before it reduces the slabs.
  const bool pair_wgrad = wgrad && !(wgrad_stream && wgrad_events) && any_order_enabled();
  const bool side_wgrad = wgrad && ((wgrad_stream && wgrad_events) || pair_wgrad);
  bool first_done = false;   // the first layer's weight gradient came out of the fused LayerNorm-backward launch
  int pending = -1;   // pair_wgrad: the layer whose weight gradient goes out right behind the next dgrad launch, beside it
  auto wgrad_beside = [&](int l) {
    if (pair_wgrad) { pending = l; return; }
    hipEvent_t e = wgrad_events[l];
    if (hipEventRecord(e, s) == hipSuccess && hipStreamWaitEvent(wgrad_stream, e, 0) == hipSuccess) wgrad_of(l, wgrad_stream);
    else wgrad_of(l, s);   // ordering on s is always sufficient
  };
  // beside_first: one more launch of the caller's that only needs what was complete before this call; it goes out any-order
  // behind the first dgrad launch (never as the last launch of the chain: whatever follows an entry point on the stream --
  // an event record, a collective -- must find an ordinary launch, which completes after everything before it, at the end)
  bool hook_pending = beside_first != nullptr;
  auto flush_pending = [&](bool beside) {
    if (pending < 0) return;
    g_any_order = beside;
    wgrad_of(pending, s);
    if (beside && hook_pending) {
      (*beside_first)(s);
      hook_pending = false;
      if (beside_first_done) *beside_first_done = true;
    }
    g_any_order = false;
    pending = -1;
  };
  if (side_wgrad) wgrad_beside(last + 1);
  const unsigned short* planes = nullptr;             // refreshed here: the caller may have changed the parameters since the forward
  if (bwd_uses_planes(net, L, mb))
    planes = planes_ready ? reinterpret_cast<const unsigned short*>(ws + W.wplanes) : launch_split_params(L, W, ws, params, s);
  // ---- dgrad chain (each launch ends in the LayerNorm/SiLU backward of the layer below) --------------------------
  for (int l = last; l >= 0; --l) {
    if (l == last && last_done) {   // dZ of the last layer came out of the caller's tail launch
      flush_pending(false);
      if (side_wgrad) wgrad_beside(l);
      continue;
    }
    GemmP p;
    memset(&p, 0, sizeof(p));
    const int u = net->units[l];
    p.pl_stride = W.wplane_stride;
    for (int k = 0; k < towers; ++k) {
      const int tw = tw0 + k;
      if (l == last) {
        p.A[k] = ws + W.seeds[tw];
        p.B[k] = params + L.Wh[tw];
        p.Kt[k] = L.head_rows[tw];
      } else {
        p.A[k] = ws + W.dz[tw][l + 1];
        p.B[k] = params + L.W[tw][l + 1];
        p.Bpl[k] = planes ? planes + L.W[tw][l + 1] : nullptr;
        p.Kt[k] = net->units[l + 1];
      }
      p.Nt[k] = u;
      p.out[k] = ws + W.dz[tw][l];
      p.xhat[k] = ws + W.xhat[tw][l];
      p.rstd[k] = ws + W.rstd[tw][l];
      p.gamma[k] = L.has_norm ? params + L.g[tw][l] : nullptr;
      p.beta[k] = L.has_norm ? params + L.be[tw][l] : nullptr;
      p.colpart[k] = ws + W.colpart[tw][l];
    }
    const int kdim = (l == last) ? 32 : net->units[l + 1];
    p.lda = kdim; p.ldb = u; p.ldo = u;
    p.M = (int)mb; p.N = u; p.K = kdim;
    p.act_elu = net->tower_kind == TOWER_LAYERNORM_ELU;
    if (net->tower_kind == MRL_TOWER_ELU) {
      // plain towers: dgrad fused with the ELU derivative of the layer below, full-row tiles (bias-gradient partials per row block)
      const dim3 grid((unsigned)cdiv(p.M, fwd_bm(u)), 1, towers);
      if (u == 512) launch_gemm<64, 512, 16, 1, 4, LAY_K, LAY_R, EPI_ACTBWD, PRO_NONE>(p, grid, s);
      else if (u == 256) launch_gemm<128, 256, 16, 2, 2, LAY_K, LAY_R, EPI_ACTBWD, PRO_NONE>(p, grid, s);
      else if (u == 128) launch_gemm<128, 128, 16, 2, 2, LAY_K, LAY_R, EPI_ACTBWD, PRO_NONE>(p, grid, s);
      else if (u == 64) launch_gemm<128, 64, 16, 4, 1, LAY_K, LAY_R, EPI_ACTBWD, PRO_NONE>(p, grid, s);
      else launch_gemm<128, 32, 16, 4, 1, LAY_K, LAY_R, EPI_ACTBWD, PRO_NONE>(p, grid, s);
      flush_pending(true);
    } else if (ln_bwd_streamed(u)) {
      // plain dgrad tile grid (no full-row constraint), then the streaming LayerNorm/SiLU backward in place
      p.ksplit = kdim; p.n_tiles = u / 128; p.group_mode = 0;
      if (mb <= MRL_SMALL_M) {   // 32-row tiles: four times the blocks for a batch that would leave most CUs idle
        p.m_tiles = (int)cdiv(mb, 32); p.n_groups = p.m_tiles;
        launch_gemm<32, 128, 16, 1, 4, LAY_K, LAY_R, EPI_SLAB, PRO_NONE>(p, dim3(slab_grid(p.n_groups, p.n_tiles), 1, towers), s);
      } else {
        p.m_tiles = (int)cdiv(mb, 128); p.n_groups = p.m_tiles;
        const dim3 grid(slab_grid(p.n_groups, p.n_tiles), 1, towers);
        p.nt_hint = (l == 0 && wgrad && first_fused(net, L.in_dim, mb)) ? 2 : 1;   // who reads the output next (launch_dgrad_split)
        if (l != last && kdim % 16 == 0) launch_dgrad_split(p, grid, s);   // A = dZ of the layer above, all K columns valid
        else launch_gemm<128, 128, 16, 2, 2, LAY_K, LAY_R, EPI_SLAB, PRO_NONE>(p, grid, s);
      }
      flush_pending(true);
      if (l == 0 && wgrad && first_fused(net, L.in_dim, mb)) {
        // first layer: LayerNorm backward and weight gradient in one launch; dZ is only stored for an input-gradient launch
        LnWg0Args f;
        memset(&f, 0, sizeof(f));
        for (int k = 0; k < towers; ++k) {
          const int tw = tw0 + k;
          f.dh[k] = ws + W.dz[tw][0]; f.dz_out[k] = input_grad ? ws + W.dz[tw][0] : nullptr;
          f.xhat[k] = ws + W.xhat[tw][0]; f.rstd[k] = ws + W.rstd[tw][0];
          f.gamma[k] = params + L.g[tw][0]; f.beta[k] = params + L.be[tw][0];
          f.slab[k] = ws + W.slab[tw][0]; f.colpart[k] = ws + W.colpart[tw][0];
        }
        f.x = x; f.in = L.in_dim[0]; f.mean = obs_mean; f.std = obs_std; f.clamp_lo = clamp_lo; f.clamp_hi = clamp_hi;
        f.M = (int)mb; f.rows_per_block = first_fused_rows(mb);
        static const bool lds_ok = [] {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lnbwd_wgrad0_kernel<64, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lnbwd_wgrad0_kernel<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          return true;
        }();
        (void)lds_ok;
        const dim3 fgrid(W.rowblocks[0], towers);
        MRL_PROF_STREAM("lnbwd_wgrad0_kernel", s);
        if (net->tower_kind == TOWER_LAYERNORM_ELU) hipLaunchKernelGGL((lnbwd_wgrad0_kernel<64, true>), fgrid, dim3(LNWG0_THREADS), LnWg0Lds<64>::BYTES, s, f);
        else hipLaunchKernelGGL((lnbwd_wgrad0_kernel<64, false>), fgrid, dim3(LNWG0_THREADS), LnWg0Lds<64>::BYTES, s, f);
        first_done = true;
        continue;
      }
      LnBwdArgs a;
      memset(&a, 0, sizeof(a));
      for (int k = 0; k < towers; ++k) {
        const int tw = tw0 + k;
        a.dz[k] = ws + W.dz[tw][l]; a.xhat[k] = ws + W.xhat[tw][l]; a.rstd[k] = ws + W.rstd[tw][l];
        a.gamma[k] = params + L.g[tw][l]; a.beta[k] = params + L.be[tw][l]; a.colpart[k] = ws + W.colpart[tw][l];
      }
      a.M = (int)mb; a.rows_per_block = (l == 0 && first_fused(net, L.in_dim, mb)) ? first_fused_rows(mb) : ln_bwd_rows(mb);   // = the row ranges of W.rowblocks
      a.act_elu = net->tower_kind == TOWER_LAYERNORM_ELU;
      dim3 grid(W.rowblocks[l], towers);
      if (u == 512) { MRL_PROF_STREAM("ln_bwd_kernel<512>", s); hipLaunchKernelGGL(ln_bwd_kernel<512>, grid, dim3(256), 0, s, a); }
      else { MRL_PROF_STREAM("ln_bwd_kernel<256>", s); hipLaunchKernelGGL(ln_bwd_kernel<256>, grid, dim3(256), 0, s, a); }
    } else {
      // narrow layers (<= 128 wide): the whole row fits one block's accumulators, LayerNorm backward fused
      const dim3 grid((unsigned)cdiv(p.M, 128), 1, towers);
      if (u == 128 && fused_bwd_rows(u, mb) == 32)
        launch_gemm<32, 128, 16, 1, 4, LAY_K, LAY_R, EPI_LNBWD, PRO_NONE>(p, dim3((unsigned)cdiv(p.M, 32), 1, towers), s);
      else if (u == 128) launch_gemm<128, 128, 16, 2, 2, LAY_K, LAY_R, EPI_LNBWD, PRO_NONE>(p, grid, s);
      else if (u == 64) launch_gemm<128, 64, 16, 4, 1, LAY_K, LAY_R, EPI_LNBWD, PRO_NONE>(p, grid, s);
      else launch_gemm<128, 32, 16, 4, 1, LAY_K, LAY_R, EPI_LNBWD, PRO_NONE>(p, grid, s);
      flush_pending(true);
    }
    if (side_wgrad) wgrad_beside(l);
  }

  // ---- gradient with respect to the input of the first layer: dX = dZ0 * W0 ----------------------------------------
  if (input_grad) {
    launch_input_grad(net, L, W, ws, params, mb, tw0, towers, input_grad, s);
    flush_pending(true);
  }
  flush_pending(false);   // the first layer's weight gradient when no launch is left to run beside
  if (!wgrad || side_wgrad) return;
  for (int l = last + 1; l >= (first_done ? 1 : 0); --l) wgrad_of(l, s);
}

// Fixed-order reduction of the partial slabs, column partials and head-bias partials of towers tw0 .. tw0+towers-1
// into the flat gradient vector (layout L).  hblocks = 128-row blocks that produced head-bias partials.
// whole: the call covers every tensor of the layout, i.e. the whole vector `grads` (then small slab counts take the one-pass kernel
// that also leaves the squared-norm partials of the vector in the workspace, and finishes the step's metrics when `ma` is given).
static void launch_grad_reduce(const mrl_net* net, const ParamLayout& L, const Workspace& W, float* ws, float* grads, int tw0, int towers, int hblocks,
                               hipStream_t s, bool whole = false, const MetricsArgs* ma = nullptr, bool* flat_done = nullptr) {
  const int nl = net->n_layers, U = net->units[nl - 1];
  // Every float of the flat gradient is written below: tensor sizes are multiples of 4 floats except the head biases, whose
  // descriptors are rounded up to whole 4-float chunks (zero columns of the partials), and the layout pads offsets to 4.
  ReduceTable tab;
  tab.n = 0;
  auto add = [&](const float* src, int64_t off, int64_t count, int64_t stride, int nslabs) {
    ReduceDesc& d = tab.d[tab.n++];
    d.src = src; d.dst = grads + off; d.count = count; d.stride = stride; d.nslabs = nslabs;
  };
  int64_t maxcount = 0;
  for (int k = 0; k < towers; ++k) {
    const int tw = tw0 + k;
    for (int l = 0; l < nl; ++l) {
      const int u = net->units[l];
      add(ws + W.slab[tw][l], L.W[tw][l], (int64_t)u * L.in_dim[l], (int64_t)u * L.in_dim[l], W.wp[l].nsplit);
      const float* cp = ws + W.colpart[tw][l];
      add(cp + 2 * u, L.b[tw][l], u, 3 * u, W.rowblocks[l]);
      if (L.has_norm) {
        add(cp, L.g[tw][l], u, 3 * u, W.rowblocks[l]);
        add(cp + u, L.be[tw][l], u, 3 * u, W.rowblocks[l]);
      }
      if ((int64_t)u * L.in_dim[l] > maxcount) maxcount = (int64_t)u * L.in_dim[l];
    }
    const int hn = L.head_rows[tw];
    add(ws + W.slab_h[tw], L.Wh[tw], (int64_t)hn * U, (int64_t)32 * U, W.wph.nsplit);
    // counts are rounded up to whole 4-float chunks (the extra columns are zero and land in the layout's padding)
    if (W.chain_bwd) add(ws + W.colpart_h[tw] + 64, L.bh[tw], align4(L.head_bias_n[tw]), 96, W.wph.nsplit);
    else add(ws + W.headb_part + tw * 32, L.bh[tw], align4(L.head_bias_n[tw]), 64, hblocks);
  }
  int max_slabs = 0;
  bool aligned = (L.total & 3) == 0;
  for (int i = 0; i < tab.n; ++i) {
    max_slabs = std::max(max_slabs, tab.d[i].nslabs);
    aligned = aligned && ((tab.d[i].dst - grads) & 3) == 0 && (tab.d[i].count & 3) == 0 && (tab.d[i].stride & 3) == 0;
  }
  if (flat_done) *flat_done = false;
  if (whole && aligned && max_slabs <= 32) {
    FlatTable ft;
    ft.n = tab.n;
    std::sort(tab.d, tab.d + tab.n, [](const ReduceDesc& a, const ReduceDesc& b) { return a.dst < b.dst; });
    for (int i = 0; i < tab.n; ++i) {
      const ReduceDesc& d = tab.d[i];
      ft.d[i].src = d.src; ft.d[i].stride = d.stride; ft.d[i].nslabs = d.nslabs;
      ft.d[i].c0 = (int)((d.dst - grads) >> 2);
      ft.d[i].c1 = ft.d[i].c0 + (int)(d.count >> 2);
    }
    ft.cpt = norm_cpt(L.total);
    ft.nchunks = (int)(L.total >> 2);
    MetricsArgs m;
    memset(&m, 0, sizeof(m));
    if (ma) m = *ma;
    const int nb = norm_parts(L.total);
    MRL_PROF_STREAM("reduce_flat_kernel", s);
    hipLaunchKernelGGL(reduce_flat_kernel, dim3(nb + (ma ? 1 : 0)), dim3(256), 0, s, ft, grads, (double*)(ws + W.norm_part), m);
    if (flat_done) *flat_done = true;
    return;
  }
  (void)maxcount;
  int nblocks = 0;   // one block per 16 chunks of 4 floats, at most 1024 per descriptor (measured 28.3 us against 31.0 with 2048 and 28.5 with 512)
  for (int i = 0; i < tab.n; ++i) {
    tab.d[i].block0 = nblocks;
    nblocks += (int)std::min<int64_t>(1024, std::max<int64_t>(1, cdiv(tab.d[i].count, 64)));
  }
  MRL_PROF_STREAM("reduce_slabs_kernel", s);
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(nblocks), dim3(256), 0, s, tab);
}

// A second stream per device for the per-tower launch chains of mrl_ppo_grad (actor on the caller's stream, critic on this
// one; both at the default priority), with the two events of the fork and the join.  Created on first use, never destroyed.
struct SideStream {
  hipStream_t s = nullptr, w[2] = {nullptr, nullptr};   // critic chain; weight-gradient launches of each tower chain
  hipEvent_t fork = nullptr, join = nullptr, wjoin[2] = {nullptr, nullptr};
  hipEvent_t wev[2][MRL_MAX_LAYERS + 2] = {};
};
static SideStream g_side[MRL_MAX_DEVICES];
static int side_stream(SideStream** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= MRL_MAX_DEVICES) MRL_FAIL(MRL_EHIP, "hipGetDevice: %s (device %d)", hipGetErrorString(e), dev);
  SideStream& ss = g_side[dev];
  {
    static std::mutex mu;   // taken on every call: this path is an A/B variant, not the default launch structure
    std::lock_guard<std::mutex> lock(mu);
    if (!ss.s) {
      hipStream_t st = nullptr;
      if ((e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) MRL_FAIL(MRL_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
      if ((e = hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming)) != hipSuccess || (e = hipEventCreateWithFlags(&ss.join, hipEventDisableTiming)) != hipSuccess)
        MRL_FAIL(MRL_EHIP, "hipEventCreate: %s", hipGetErrorString(e));
      for (int k = 0; k < 2; ++k) {
        if ((e = hipStreamCreateWithFlags(&ss.w[k], hipStreamNonBlocking)) != hipSuccess) MRL_FAIL(MRL_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
        if ((e = hipEventCreateWithFlags(&ss.wjoin[k], hipEventDisableTiming)) != hipSuccess) MRL_FAIL(MRL_EHIP, "hipEventCreate: %s", hipGetErrorString(e));
        for (int l = 0; l < MRL_MAX_LAYERS + 2; ++l)
          if ((e = hipEventCreateWithFlags(&ss.wev[k][l], hipEventDisableTiming)) != hipSuccess) MRL_FAIL(MRL_EHIP, "hipEventCreate: %s", hipGetErrorString(e));
      }
      ss.s = st;
    }
  }
  *out = &ss;
  return MRL_OK;
}
// Alternative launch structure of mrl_ppo_grad, kept as a measured variant (MRL_TOWER_STREAMS=1; same results, same speed as
// the default of one stream with any-order weight-gradient launches).  false: both towers in every launch (blockIdx.z).  true: one launch chain per tower on
// two streams, so that the store bursts and epilogues of one tower's kernels run beside the main loops of the other's
// (bit-identical results; measured 248.7 -> 243.0 ms per full-size update), and the weight-gradient launches of each
// chain on a third and fourth stream beside the rest of its dgrad chain (another 1.3 %: 249.6 / 248.3 -> 246.0 / 245.4 ms
// in two A/B pairs on one box).  Only for batches above MRL_SMALL_M rows, where every launch fills the chip several times
// over.  While kernel timing is on (mrl_profile_enable) the single-stream order is used.
static bool tower_streams(int64_t mb) {
  static const int forced = [] { const char* e = getenv("MRL_TOWER_STREAMS"); return e ? atoi(e) : -1; }();
  if (g_prof_mode != 0) return false;   // per-kernel timing wants one launch per kernel and nothing running beside it
  return forced > 0 && mb > MRL_SMALL_M;
}

// norm_ready (optional): set to true when the squared-norm partials of `grads` were left in the workspace (Workspace::norm_part).
// merge (optional): the running-statistics merge of the step, folded into the first launch.
static int ppo_grad_impl(const mrl_net* net, const float* params, const mrl_ppo_hp* hp, const mrl_rollout* buf,
                         int64_t row0, int64_t mb, const float* obs_mean, const float* obs_std, float clamp_lo,
                         float clamp_hi, float* grads, double* metrics_out, void* workspace, int64_t workspace_bytes,
                         void* stream, bool* norm_ready, const RmsMergeArgs* merge, bool blocked_valid = false) {
  if (norm_ready) *norm_ready = false;
  if (check_net(net) || check_buf(buf)) return MRL_EINVAL;
  if (!params || !hp || !grads || !metrics_out || !workspace) MRL_FAIL(MRL_EINVAL, "NULL pointer");
  const int64_t B = buf->num_envs * (int64_t)buf->horizon;
  if (mb < 2 || row0 < 0 || row0 + mb > B) MRL_FAIL(MRL_EINVAL, "minibatch rows [%lld,%lld) outside the batch of %lld", (long long)row0, (long long)(row0 + mb), (long long)B);
  if ((obs_mean == nullptr) != (obs_std == nullptr)) MRL_FAIL(MRL_EINVAL, "obs_mean and obs_std must both be given or both be NULL");
  if (hp->bounds_type != 0 && hp->bounds_type != 1) MRL_FAIL(MRL_EINVAL, "bounds_type %d", hp->bounds_type);
  Workspace W;
  workspace_layout(net, mb, &W);
  if (workspace_bytes < W.total * 4) MRL_FAIL(MRL_ENOSPC, "workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)W.total * 4);
  if (reinterpret_cast<uintptr_t>(workspace) & 255) MRL_FAIL(MRL_EINVAL, "workspace %p is not 256-byte aligned", workspace);
  ParamLayout L;
  param_layout(net, &L);
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const int O = net->obs_dim, A = net->act_dim, nl = net->n_layers, last = nl - 1, U = net->units[last];
  const float* x = buf->obs + row0 * O;

  double* metric_part = (double*)(ws + W.metric_part);
  // small batches: the weights re-blocked for the chain kernels, once per call for the forward and the backward pass; with both
  // chains on (MRL_CHAIN bit 4: in one launch with the heads between them)
  const bool ch_f = chain_ok(net, mb, 1), ch_b = chain_ok(net, mb, 2), fused = ch_f && ch_b && (chain_mask() & 4) != 0;
  if (net->tower_kind > MRL_TOWER_LAYERNORM_ELU && !fused) MRL_FAIL(MRL_EINVAL, "tower_kind %d needs the one-launch chain (at most %d layers)", net->tower_kind, CH_MAXL);
  const int hblocks = (int)cdiv(mb, fused ? CH_ROWS : 128);   // row blocks of the kernel that evaluates the heads
  auto fill_heads = [&](HeadsP& h) {
    h.seeds[0] = ws + W.seeds[0];
    h.seeds[1] = ws + W.seeds[1];
    if (net->fixed_sigma) h.fixed_logstd = params + L.bh[0] + A;
    fill_heads_common(net, &h);
    h.actions = buf->actions + row0 * A;
    h.mu = buf->mu + row0 * A;
    h.sigma = buf->sigma + row0 * A;
    h.old_neglogp = buf->neglogp + row0;
    h.adv = buf->advantages + row0;
    h.returns = buf->returns + row0;
    h.value_preds = buf->values + row0;
    h.metric_part = metric_part;
    h.headb_part = ws + W.headb_part;
    h.inv_mb = 1.0f / (float)mb;
    h.e_clip = hp->e_clip; h.critic_coef = hp->critic_coef; h.entropy_coef = hp->entropy_coef;
    h.bounds_coef = hp->bounds_loss_coef; h.bounds_type = hp->bounds_type; h.clip_value_loss = hp->clip_value_loss;
    h.smooth_clamp = hp->use_smooth_clamp != 0;
  };
  // heads of towers [tw0, tw0 + towers): losses, KL, seeds d loss / d head, mu / sigma write-back, head-bias partials
  auto launch_heads = [&](int tw0, int towers, hipStream_t st) {
    GemmP p;
    memset(&p, 0, sizeof(p));
    for (int k = 0; k < towers; ++k) {
      const int tw = tw0 + k;
      p.A[k] = ws + W.xhat[tw][last];
      p.B[k] = params + L.Wh[tw];
      p.pro0[k] = L.has_norm ? params + L.g[tw][last] : nullptr;
      p.pro1[k] = L.has_norm ? params + L.be[tw][last] : nullptr;
      p.bias[k] = params + L.bh[tw];
      p.Nt[k] = L.head_rows[tw];
      p.Kt[k] = U;
    }
    p.tw0 = tw0;
    p.lda = U; p.ldb = U;
    p.M = (int)mb; p.N = 32; p.K = U;
    fill_heads(p.h);
    if (net->tower_kind == MRL_TOWER_ELU) launch_gemm<128, 32, 32, 4, 1, LAY_K, LAY_K, EPI_HEADS_TRAIN, PRO_ELU>(p, dim3(hblocks, 1, towers), st);
    else if (net->tower_kind == TOWER_LAYERNORM_ELU) launch_gemm<128, 32, 32, 4, 1, LAY_K, LAY_K, EPI_HEADS_TRAIN, PRO_LNELU>(p, dim3(hblocks, 1, towers), st);
    else launch_gemm<128, 32, 32, 4, 1, LAY_K, LAY_K, EPI_HEADS_TRAIN, PRO_LNSILU>(p, dim3(hblocks, 1, towers), st);
  };
  // heads, losses, seeds and the last layer's dZ of towers [tw0, tw0 + towers) in one launch (ppo_tail_ok)
  auto launch_tail = [&](int tw0, int towers, hipStream_t st) {
    PpoTailP tp;
    memset(&tp, 0, sizeof(tp));
    fill_heads(tp.h);
    for (int tw = 0; tw < 2; ++tw) {
      tp.xhat[tw] = ws + W.xhat[tw][last]; tp.rstd[tw] = ws + W.rstd[tw][last];
      tp.gamma[tw] = params + L.g[tw][last]; tp.beta[tw] = params + L.be[tw][last];
      tp.Wh[tw] = params + L.Wh[tw]; tp.bh[tw] = params + L.bh[tw]; tp.head_n[tw] = L.head_rows[tw];
      tp.dz[tw] = ws + W.dz[tw][last]; tp.colpart[tw] = ws + W.colpart[tw][last];
    }
    tp.M = (int)mb; tp.rows_per_block = 128; tp.tw0 = tw0;
    MRL_PROF_STREAM("ppo_tail_kernel", st);
    if (net->tower_kind == TOWER_LAYERNORM_ELU) hipLaunchKernelGGL(ppo_tail_kernel<true>, dim3(hblocks, towers), dim3(64 * TAIL_WAVES), 0, st, tp);
    else hipLaunchKernelGGL(ppo_tail_kernel<false>, dim3(hblocks, towers), dim3(64 * TAIL_WAVES), 0, st, tp);
  };
  // the reduction of the heads' metric partials: any-order behind the first dgrad launch of the backward chain when that is
  // on (launch_towers_bwd), otherwise an ordinary launch at the end
  auto launch_metrics = [&](hipStream_t st) {
    if (g_any_order)
      hipExtLaunchKernelGGL(metrics_final_kernel, dim3(1), dim3(192), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, metric_part, hblocks, (double)mb,
                            hp->critic_coef, hp->entropy_coef, hp->bounds_loss_coef, metrics_out);
    else
      hipLaunchKernelGGL(metrics_final_kernel, dim3(1), dim3(192), 0, st, metric_part, hblocks, (double)mb, hp->critic_coef,
                         hp->entropy_coef, hp->bounds_loss_coef, metrics_out);
  };
  const std::function<void(hipStream_t)> metrics_hook = launch_metrics;
  bool metrics_done = false;
  // the parameter vector as three bf16 planes, once per call: read by the split-bf16 forward and input-gradient kernels
  const bool planes = L.has_norm && mb > MRL_SMALL_M && !chain_ok(net, mb, 1);
  const unsigned short* wpl = !planes ? nullptr
                              : (blocked_valid ? reinterpret_cast<const unsigned short*>(ws + W.wplanes) : launch_split_params(L, W, ws, params, s));
  if (blocked_valid) {
    if (!(fused || planes) || merge) MRL_FAIL(MRL_EINVAL, "internal: derived weight copies can only be carried between the steps of one call");
  } else if (ch_f || ch_b) launch_reblock(net, L, W, ws, params, 0, 2, ch_f, ch_b, s, fused, merge);
  else if (merge)
    hipLaunchKernelGGL(rms_merge_kernel, dim3(1), dim3(1024), 0, s, merge->mean, merge->var, merge->count, merge->bmean, merge->bvar, merge->bcount,
                       merge->n, merge->eps, merge->mean_f, merge->std_f);
  if (fused) {
    // forward, heads and backward of both towers in one launch; every weight gradient in a second one
    ChainPpoP p;
    fill_chain_fwd(net, L, W, ws, params, x, mb, obs_mean, obs_std, clamp_lo, clamp_hi, 0, 2, &p.f);
    fill_chain_bwd(net, L, W, ws, params, mb, 0, 2, &p.b);
    p.b.ld = p.f.ld;
    memset(&p.h, 0, sizeof(p.h));
    fill_heads(p.h);
    for (int tw = 0; tw < 2; ++tw) {
      p.Wh[tw] = chain_natural() ? params + L.Wh[tw] : ws + W.wblk_h[tw];
      p.bh[tw] = params + L.bh[tw];
      p.head_n[tw] = L.head_rows[tw];
    }
    chain_allow_lds();
    {
      MRL_PROF_STREAM("mrl::ppo_chain_kernel", s);
      const dim3 grid((unsigned)(p.f.rows_written / CH_ROWS), 2);
      const size_t lds = chain_lds_bytes(p.f.ld) + CH_HEADS_F * sizeof(float) + chain_bwd_extra_lds(p.f.ld);
      MRL_CHAIN_LAUNCH(ppo_chain_kernel, net->tower_kind, grid, lds, s, p);
    }
    launch_chain_wgrad(net, L, W, ws, params, x, mb, obs_mean, obs_std, clamp_lo, clamp_hi, 0, 2, s);
    MetricsArgs ma;
    ma.part = metric_part; ma.nblocks = hblocks; ma.mb = (double)mb;
    ma.critic_coef = hp->critic_coef; ma.entropy_coef = hp->entropy_coef; ma.bounds_coef = hp->bounds_loss_coef;
    ma.out = metrics_out;
    launch_grad_reduce(net, L, W, ws, grads, 0, 2, hblocks, s, true, &ma, &metrics_done);
    if (norm_ready) *norm_ready = metrics_done;
  } else if (tower_streams(mb)) {
    SideStream* ss = nullptr;
    if (int rc = side_stream(&ss)) return rc;
    if (hipEventRecord(ss->fork, s) != hipSuccess || hipStreamWaitEvent(ss->s, ss->fork, 0) != hipSuccess) MRL_FAIL(MRL_EHIP, "fork of the tower streams failed");
    for (int tw = 0; tw < 2; ++tw) {
      hipStream_t st = tw == 0 ? s : ss->s;
      launch_towers_fwd(net, L, W, ws, params, x, mb, obs_mean, obs_std, clamp_lo, clamp_hi, tw, 1, st, wpl);
      const bool tail = ppo_tail_ok(net, L, mb);
      if (tail) launch_tail(tw, 1, st); else launch_heads(tw, 1, st);
      // weight-gradient launches of each chain on a stream of their own (MRL_WGRAD_STREAMS=0 keeps them in the chain)
      static const int wside = [] { const char* e = getenv("MRL_WGRAD_STREAMS"); return e ? atoi(e) : 1; }();
      launch_towers_bwd(net, L, W, ws, params, x, mb, obs_mean, obs_std, clamp_lo, clamp_hi, tw, 1, true, nullptr, st, planes, wside ? ss->w[tw] : nullptr, ss->wev[tw],
                        nullptr, nullptr, tail);
      if (wside && (hipEventRecord(ss->wjoin[tw], ss->w[tw]) != hipSuccess || hipStreamWaitEvent(st, ss->wjoin[tw], 0) != hipSuccess))
        MRL_FAIL(MRL_EHIP, "join of the weight-gradient stream failed");
      launch_grad_reduce(net, L, W, ws, grads, tw, 1, hblocks, st);
    }
    if (hipEventRecord(ss->join, ss->s) != hipSuccess || hipStreamWaitEvent(s, ss->join, 0) != hipSuccess) MRL_FAIL(MRL_EHIP, "join of the tower streams failed");
  } else {
    launch_towers_fwd(net, L, W, ws, params, x, mb, obs_mean, obs_std, clamp_lo, clamp_hi, 0, 2, s, wpl, true);
    const bool tail = ppo_tail_ok(net, L, mb);
    if (tail) launch_tail(0, 2, s); else launch_heads(0, 2, s);
    launch_towers_bwd(net, L, W, ws, params, x, mb, obs_mean, obs_std, clamp_lo, clamp_hi, 0, 2, true, nullptr, s, planes || ch_b, nullptr, nullptr, &metrics_hook,
                      &metrics_done, tail);
    launch_grad_reduce(net, L, W, ws, grads, 0, 2, hblocks, s, true, nullptr, norm_ready);
  }
  if (!metrics_done) launch_metrics(s);
  MRL_CHECK_LAUNCH("mrl_ppo_grad");
  return MRL_OK;
}

extern "C" int mrl_ppo_grad(const mrl_net* net, const float* params, const mrl_ppo_hp* hp, const mrl_rollout* buf,
                            int64_t row0, int64_t mb, const float* obs_mean, const float* obs_std, float clamp_lo,
                            float clamp_hi, float* grads, double* metrics_out, void* workspace, int64_t workspace_bytes,
                            void* stream) {
  return ppo_grad_impl(net, params, hp, buf, row0, mb, obs_mean, obs_std, clamp_lo, clamp_hi, grads, metrics_out, workspace, workspace_bytes, stream,
                       nullptr, nullptr);
}

// adam_kernel that also keeps the re-blocked weight copies of the chain kernels current (reblock_kernel's layouts), for callers that
// run several optimizer steps back to back: chunks of the parameter vector that belong to a weight matrix go to their places in the
// forward-order and backward-order copies as they are written.  One more block merges the next step's batch moments into the
// running statistics.  Zero padding of the copies is never touched: a reblock_kernel launch must have written them once.
// For the per-layer kernels of large batches the derived copies are the three bf16 planes of split_params_kernel, in both of
// its orders: every chunk of four updated parameters is split (the same split, so the planes are bit-identical to a
// split_params_kernel launch after the step) and stored as 8 bytes per plane and order.
struct ArDesc { int c0, c1, K; float* fdst; int fsteps; float* bdst; int bJ, bsteps; };   // chunks [c0, c1) are a row-major [rows][K] matrix
constexpr int AR_MAXDESC = 2 * (MRL_MAX_LAYERS + 1);
struct ArTable {
  ArDesc d[AR_MAXDESC];
  int n, has_merge;
  RmsMergeArgs merge;
  // batches above MRL_SMALL_M rows: the bf16 planes of the parameter vector instead (split_params_kernel's two orders), or null
  unsigned short* planes;
  unsigned short* kplanes;
  int64_t plane_stride;
  SplitTable split;
};

__global__ __launch_bounds__(256) void adam_reblock_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, int64_t n, const double* __restrict__ norm_part,
                                                           const AdamArgs a, float* __restrict__ gnorm_out, const ArTable tab) {
  if (tab.has_merge && blockIdx.x == gridDim.x - 1) {
    rms_merge_body(tab.merge);
    return;
  }
  __shared__ double s_red[256];
  __shared__ float s_coef;
  const int nblk = gridDim.x - (tab.has_merge ? 1 : 0);
  const int64_t n4 = n >> 2;   // the host checks n % 4 == 0 and the alignment
  // the operands of the thread's first chunk are requested before the norm partials are reduced: one memory round trip less
  const int64_t cfirst = (int64_t)blockIdx.x * 256 + threadIdx.x;
  f32x4 g4 = {0.f, 0.f, 0.f, 0.f}, p4 = g4, m4 = g4, v4 = g4;
  if (cfirst < n4) {
    g4 = *reinterpret_cast<const f32x4*>(g + 4 * cfirst);
    p4 = *reinterpret_cast<f32x4*>(p + 4 * cfirst); m4 = *reinterpret_cast<f32x4*>(m + 4 * cfirst); v4 = *reinterpret_cast<f32x4*>(v + 4 * cfirst);
  }
  s_red[threadIdx.x] = (((int)threadIdx.x < a.nparts) ? norm_part[threadIdx.x] : 0.0) + (((int)threadIdx.x + 256 < a.nparts) ? norm_part[threadIdx.x + 256] : 0.0);
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float norm = (float)sqrt(s_red[0]);
    float coef = 1.0f;
    if (a.max_norm > 0.f) coef = fminf(a.max_norm / (norm + 1e-6f), 1.0f);   // torch clip_grad_norm_
    s_coef = coef;
    if (blockIdx.x == 0 && gnorm_out) gnorm_out[0] = norm;
  }
  __syncthreads();
  const float coef = s_coef;
  for (int64_t c = cfirst; c < n4; c += (int64_t)nblk * 256) {
    if (c != cfirst) {
      g4 = *reinterpret_cast<const f32x4*>(g + 4 * c);
      p4 = *reinterpret_cast<f32x4*>(p + 4 * c); m4 = *reinterpret_cast<f32x4*>(m + 4 * c); v4 = *reinterpret_cast<f32x4*>(v + 4 * c);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gi = mul_rn(mul_rn(g4[e], a.grad_scale), coef);
      float pi = p4[e], mi = m4[e], vi = v4[e];
      if (a.adamw) pi = mul_rn(pi, a.lr_wd_mul);
      else if (a.l2 != 0.f) gi = add_rn(gi, mul_rn(a.l2, pi));
      mi = add_rn(mi, mul_rn(a.one_m_b1, sub_rn(gi, mi)));
      vi = add_rn(mul_rn(vi, a.b2), mul_rn(mul_rn(a.one_m_b2, gi), gi));
      const float denom = add_rn(div_rn(sqrtf(vi), a.bc2_sqrt), a.eps);
      pi = add_rn(pi, mul_rn(a.neg_step_size, div_rn(mi, denom)));
      p4[e] = pi; m4[e] = mi; v4[e] = vi;
    }
    *reinterpret_cast<f32x4*>(p + 4 * c) = p4;
    *reinterpret_cast<f32x4*>(m + 4 * c) = m4;
    *reinterpret_cast<f32x4*>(v + 4 * c) = v4;
    if (tab.planes) {
      u32x2 h, mm, l;
      split_rne4(p4, h, mm, l);
      *reinterpret_cast<u32x2*>(tab.planes + 4 * c) = h;
      *reinterpret_cast<u32x2*>(tab.planes + tab.plane_stride + 4 * c) = mm;
      *reinterpret_cast<u32x2*>(tab.planes + 2 * tab.plane_stride + 4 * c) = l;
      int64_t j = 4 * c;                               // a chunk lies inside one 16-k run of a matrix row (K a multiple of 16, offsets of 4)
      for (int q = 0; q < tab.split.n; ++q) {
        const int64_t e = 4 * c - tab.split.d[q].off;
        if (e >= 0 && e < (int64_t)tab.split.d[q].rows * tab.split.d[q].cols) {
          const int row = (int)(e / tab.split.d[q].cols), k = (int)(e % tab.split.d[q].cols);
          j = tab.split.d[q].off + (int64_t)(k >> 4) * tab.split.d[q].rows * 16 + (int64_t)row * 16 + (k & 15);
        }
      }
      *reinterpret_cast<u32x2*>(tab.kplanes + j) = h;
      *reinterpret_cast<u32x2*>(tab.kplanes + tab.plane_stride + j) = mm;
      *reinterpret_cast<u32x2*>(tab.kplanes + 2 * tab.plane_stride + j) = l;
    }
    int hit = -1;
    for (int q = 0; q < tab.n; ++q)
      if (c >= tab.d[q].c0 && c < tab.d[q].c1) hit = q;
    if (hit >= 0) {
      int c0 = 0, K = 4, fsteps = 0, bJ = 1, bsteps = 0;
      float* fdst = nullptr;
      float* bdst = nullptr;
      for (int q = 0; q < tab.n; ++q)   // the descriptors stay in scalar registers; the lane picks its own by comparison
        if (q == hit) { c0 = tab.d[q].c0; K = tab.d[q].K; fdst = tab.d[q].fdst; fsteps = tab.d[q].fsteps; bdst = tab.d[q].bdst; bJ = tab.d[q].bJ; bsteps = tab.d[q].bsteps; }
      const int e = 4 * ((int)c - c0);
      const int row = e / K, k = e - row * K;
      if (fdst) {
        const int piece = (((row >> 4) * fsteps + (k >> 5)) * 2 + ((k >> 4) & 1)) * 64 + ((k >> 2) & 3) * 16 + (row & 15);
        *reinterpret_cast<f32x4*>(fdst + (int64_t)piece * 4) = p4;
      }
      if (bdst) {
        const int w16 = 16 * bJ, nw = k / w16, rem = k - nw * w16;
        const int idx = ((((nw * bsteps + (row >> 4)) * 4 + (row & 3)) * 4 + ((row >> 2) & 3)) * 16) * bJ + rem;
        *reinterpret_cast<f32x4*>(bdst + idx) = p4;
      }
    }
  }
}

// prenorm: the squared-norm partials of `grads` (norm_parts(count) doubles in the order of gradnorm_partial_kernel) when some
// earlier launch has already produced them, else null
static int adam_impl(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t count, double lr,
                     double beta1, double beta2, double eps, double weight_decay, int32_t adamw, int64_t step,
                     float max_norm, float grad_scale, float* gnorm_out, void* scratch, void* stream, const double* prenorm,
                     const ArTable* reblock = nullptr) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !scratch) MRL_FAIL(MRL_EINVAL, "NULL pointer");
  if (count < 1 || step < 1) MRL_FAIL(MRL_EINVAL, "count %lld / step %lld must be >= 1", (long long)count, (long long)step);
  hipStream_t s = (hipStream_t)stream;
  AdamArgs a;
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  a.lr_wd_mul = (float)(1.0 - lr * weight_decay);
  a.l2 = adamw ? 0.f : (float)weight_decay;
  a.one_m_b1 = (float)(1.0 - beta1);
  a.b2 = (float)beta2;
  a.one_m_b2 = (float)(1.0 - beta2);
  a.bc2_sqrt = (float)sqrt(bc2);
  a.eps = (float)eps;
  a.neg_step_size = (float)(-(lr / bc1));
  a.max_norm = max_norm;
  a.grad_scale = grad_scale;
  a.adamw = adamw;
  a.nparts = 0;
  MRL_PROF_STREAM("gradnorm_partial_kernel + adam_kernel", s);
  const double* norm_part = (const double*)scratch;
  if (max_norm > 0.f) {
    a.nparts = norm_parts(count);
    if (prenorm && grad_scale == 1.0f) norm_part = prenorm;
    else hipLaunchKernelGGL(gradnorm_partial_kernel, dim3(a.nparts), dim3(256), 0, s, grads, count, norm_cpt(count), grad_scale, (double*)scratch);
  }
  a.vec4 = ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(exp_avg) |
             reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0;
  int64_t blocks = cdiv(a.vec4 ? cdiv(count, 4) : count, 256);
  if (blocks > 1024) blocks = 1024;
  if (reblock) {
    if (!a.vec4 || (count & 3)) MRL_FAIL(MRL_EINVAL, "the multi-step update needs 16-byte aligned vectors of a multiple of 4 floats");
    hipLaunchKernelGGL(adam_reblock_kernel, dim3((unsigned)blocks + (reblock->has_merge ? 1 : 0)), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, count,
                       norm_part, a, gnorm_out, *reblock);
    MRL_CHECK_LAUNCH("mrl_ppo_mini_epoch");
    return MRL_OK;
  }
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, count, norm_part, a, gnorm_out);
  MRL_CHECK_LAUNCH("mrl_adam_step");
  return MRL_OK;
}

extern "C" int mrl_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t count, double lr,
                             double beta1, double beta2, double eps, double weight_decay, int32_t adamw, int64_t step,
                             float max_norm, float grad_scale, float* gnorm_out, void* scratch, void* stream) {
  return adam_impl(params, grads, exp_avg, exp_avg_sq, count, lr, beta1, beta2, eps, weight_decay, adamw, step, max_norm, grad_scale, gnorm_out, scratch,
                   stream, nullptr);
}

extern "C" int mrl_ppo_step(const mrl_net* net, float* params, const mrl_ppo_hp* hp, const mrl_rollout* buf, int64_t row0, int64_t mb,
                            const mrl_rms* obs_rms, const float* batch_mean, const float* batch_var, double batch_count, float* obs_mean_f32,
                            float* obs_std_f32, float clamp_lo, float clamp_hi, float* grads, double* metrics_out, void* workspace,
                            int64_t workspace_bytes, float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2, double eps,
                            double weight_decay, int32_t adamw, int64_t step, float max_norm, float* gnorm_out, void* scratch, void* stream) {
  RmsMergeArgs m;
  memset(&m, 0, sizeof(m));
  if (obs_rms) {
    if (!obs_rms->mean || !obs_rms->var || !obs_rms->count || !batch_mean || !batch_var || !obs_mean_f32 || !obs_std_f32) MRL_FAIL(MRL_EINVAL, "NULL pointer");
    if (check_net(net)) return MRL_EINVAL;
    if (obs_rms->size != net->obs_dim) MRL_FAIL(MRL_EINVAL, "rms size %d != obs_dim %d", obs_rms->size, net->obs_dim);
    m.mean = obs_rms->mean; m.var = obs_rms->var; m.count = obs_rms->count; m.bmean = batch_mean; m.bvar = batch_var; m.bcount = batch_count;
    m.n = obs_rms->size; m.eps = obs_rms->eps; m.mean_f = obs_mean_f32; m.std_f = obs_std_f32;
  }
  bool norm_ready = false;
  if (int rc = ppo_grad_impl(net, params, hp, buf, row0, mb, obs_mean_f32, obs_std_f32, clamp_lo, clamp_hi, grads, metrics_out, workspace,
                             workspace_bytes, stream, &norm_ready, obs_rms ? &m : nullptr))
    return rc;
  const double* prenorm = nullptr;
  if (norm_ready) {
    Workspace W;
    workspace_layout(net, mb, &W);
    prenorm = (const double*)((const float*)workspace + W.norm_part);
  }
  ParamLayout L;
  param_layout(net, &L);
  return adam_impl(params, grads, exp_avg, exp_avg_sq, L.total, lr, beta1, beta2, eps, weight_decay, adamw, step, max_norm, 1.0f, gnorm_out, scratch,
                   stream, prenorm);
}

static int mini_epoch_impl(const mrl_net* net, float* params, const mrl_ppo_hp* hp, const mrl_rollout* buf, int64_t mb, int64_t n_minibatches,
                           const mrl_rms* obs_rms, const float* batch_means, const float* batch_vars, double batch_count, float* obs_mean_f32,
                           float* obs_std_f32, float clamp_lo, float clamp_hi, float* grads, double* metrics_out, void* workspace,
                           int64_t workspace_bytes, float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2, double eps,
                           double weight_decay, int32_t adamw, int64_t first_step, float max_norm, float* gnorm_out, void* scratch, void* stream,
                           float grad_scale, mrl_reduce_fn reduce, void* ctx) {
  if (check_net(net) || check_buf(buf)) return MRL_EINVAL;
  if (n_minibatches < 1 || mb < 2 || mb * n_minibatches > buf->num_envs * (int64_t)buf->horizon)
    MRL_FAIL(MRL_EINVAL, "%lld minibatches of %lld rows do not fit the batch", (long long)n_minibatches, (long long)mb);
  if (!metrics_out) MRL_FAIL(MRL_EINVAL, "NULL pointer");
  if (obs_rms && (!batch_means || !batch_vars || !obs_mean_f32 || !obs_std_f32 || !obs_rms->mean || !obs_rms->var || !obs_rms->count)) MRL_FAIL(MRL_EINVAL, "NULL pointer");
  if (obs_rms && obs_rms->size != net->obs_dim) MRL_FAIL(MRL_EINVAL, "rms size %d != obs_dim %d", obs_rms->size, net->obs_dim);
  const int O = net->obs_dim;
  ParamLayout L;
  param_layout(net, &L);
  Workspace W;
  workspace_layout(net, mb, &W);
  auto merge_of = [&](int64_t i) {
    RmsMergeArgs m;
    memset(&m, 0, sizeof(m));
    m.mean = obs_rms->mean; m.var = obs_rms->var; m.count = obs_rms->count; m.bmean = batch_means + i * O; m.bvar = batch_vars + i * O; m.bcount = batch_count;
    m.n = obs_rms->size; m.eps = obs_rms->eps; m.mean_f = obs_mean_f32; m.std_f = obs_std_f32;
    return m;
  };
  // With the one-launch chain the optimizer keeps the re-blocked weight copies current from the second step on (and merges the next
  // step's moments); otherwise this is a plain loop over mrl_ppo_step.
  const bool carry = chain_ok(net, mb, 3) && (chain_mask() & 4) != 0 && !chain_natural() && (L.total & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(exp_avg) |
                       reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0;
  // The per-layer kernels of large batches read the weights from bf16 planes: the optimizer keeps those current likewise.
  const bool vec_ok = (L.total & 3) == 0 && ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(exp_avg) |
                                              reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0;
  static const bool carry_planes_on = [] { const char* e = getenv("MRL_CARRY_PLANES"); return e ? atoi(e) != 0 : true; }();
  const bool carry_planes = !carry && carry_planes_on && vec_ok && L.has_norm && mb > MRL_SMALL_M && !chain_ok(net, mb, 1) && !chain_ok(net, mb, 2) && !tower_streams(mb);
  ArTable tab;
  memset(&tab, 0, sizeof(tab));
  if (carry_planes) {
    tab.planes = reinterpret_cast<unsigned short*>((float*)workspace + W.wplanes);
    tab.plane_stride = W.wplane_stride;
    tab.kplanes = tab.planes + 3 * W.wplane_stride;
    tab.split = split_table(L);
  }
  if (carry) {
    float* ws = (float*)workspace;
    const int nl = net->n_layers, U = net->units[nl - 1];
    auto jof = [](int n) { return n >= 512 ? 4 : (n >= 256 ? 2 : 1); };
    for (int tw = 0; tw < 2; ++tw) {
      for (int l = 0; l < nl; ++l) {
        ArDesc& d = tab.d[tab.n++];
        const int u = net->units[l], K = L.in_dim[l];
        d.c0 = (int)(L.W[tw][l] >> 2); d.c1 = d.c0 + u * K / 4; d.K = K;
        d.fdst = ws + W.wblk_f[tw][l]; d.fsteps = (K + 31) >> 5;
        if (l > 0) { d.bdst = ws + W.wblk_b[tw][l - 1]; d.bJ = jof(K); d.bsteps = (u + 15) >> 4; }
      }
      ArDesc& d = tab.d[tab.n++];
      d.c0 = (int)(L.Wh[tw] >> 2); d.c1 = d.c0 + L.head_rows[tw] * U / 4; d.K = U;
      d.fdst = ws + W.wblk_h[tw]; d.fsteps = U >> 5;
      d.bdst = ws + W.wblk_b[tw][nl - 1]; d.bJ = jof(U); d.bsteps = 2;
    }
  }
  for (int64_t i = 0; i < n_minibatches; ++i) {
    RmsMergeArgs m0;
    const bool first = i == 0 || !(carry || carry_planes);
    if (obs_rms && first) m0 = merge_of(i);
    bool norm_ready = false;
    if (int rc = ppo_grad_impl(net, params, hp, buf, i * mb, mb, obs_mean_f32, obs_std_f32, clamp_lo, clamp_hi, grads, metrics_out + i * MRL_NMETRIC,
                               workspace, workspace_bytes, stream, &norm_ready, (obs_rms && first) ? &m0 : nullptr, !first))
      return rc;
    // the caller's all-reduce of the gradient (its norm partials, taken while this rank's gradient was written, no longer hold)
    if (reduce) {
      if (const int rc = reduce(ctx, i, stream)) MRL_FAIL(MRL_EHIP, "the gradient reduction callback failed at step %lld (%d)", (long long)i, rc);
    }
    const double* prenorm = (norm_ready && !reduce) ? (const double*)((const float*)workspace + W.norm_part) : nullptr;
    const bool more = (carry || carry_planes) && i + 1 < n_minibatches;
    tab.has_merge = (more && obs_rms) ? 1 : 0;
    if (tab.has_merge) tab.merge = merge_of(i + 1);
    if (int rc = adam_impl(params, grads, exp_avg, exp_avg_sq, L.total, lr, beta1, beta2, eps, weight_decay, adamw, first_step + i, max_norm, grad_scale,
                           gnorm_out ? gnorm_out + i : nullptr, scratch, stream, prenorm, more ? &tab : nullptr))
      return rc;
  }
  return MRL_OK;
}

extern "C" int mrl_ppo_mini_epoch(const mrl_net* net, float* params, const mrl_ppo_hp* hp, const mrl_rollout* buf, int64_t mb, int64_t n_minibatches,
                                  const mrl_rms* obs_rms, const float* batch_means, const float* batch_vars, double batch_count, float* obs_mean_f32,
                                  float* obs_std_f32, float clamp_lo, float clamp_hi, float* grads, double* metrics_out, void* workspace,
                                  int64_t workspace_bytes, float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2, double eps,
                                  double weight_decay, int32_t adamw, int64_t first_step, float max_norm, float* gnorm_out, void* scratch, void* stream) {
  return mini_epoch_impl(net, params, hp, buf, mb, n_minibatches, obs_rms, batch_means, batch_vars, batch_count, obs_mean_f32, obs_std_f32, clamp_lo,
                         clamp_hi, grads, metrics_out, workspace, workspace_bytes, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, adamw,
                         first_step, max_norm, gnorm_out, scratch, stream, 1.0f, nullptr, nullptr);
}

extern "C" int mrl_ppo_mini_epoch_reduce(const mrl_net* net, float* params, const mrl_ppo_hp* hp, const mrl_rollout* buf, int64_t mb, int64_t n_minibatches,
                                         const mrl_rms* obs_rms, const float* batch_means, const float* batch_vars, double batch_count,
                                         float* obs_mean_f32, float* obs_std_f32, float clamp_lo, float clamp_hi, float* grads, double* metrics_out,
                                         void* workspace, int64_t workspace_bytes, float* exp_avg, float* exp_avg_sq, double lr, double beta1,
                                         double beta2, double eps, double weight_decay, int32_t adamw, int64_t first_step, float max_norm,
                                         float* gnorm_out, void* scratch, void* stream, float grad_scale, mrl_reduce_fn reduce, void* ctx) {
  if (!(grad_scale > 0.f)) MRL_FAIL(MRL_EINVAL, "grad_scale %g must be positive", (double)grad_scale);
  return mini_epoch_impl(net, params, hp, buf, mb, n_minibatches, obs_rms, batch_means, batch_vars, batch_count, obs_mean_f32, obs_std_f32, clamp_lo,
                         clamp_hi, grads, metrics_out, workspace, workspace_bytes, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, adamw,
                         first_step, max_norm, gnorm_out, scratch, stream, grad_scale, reduce, ctx);
}

// ---------------------------------------------------------------------------------------------------------------
// SAC update loop (include/mineral_sac.h) on the same helpers
// ---------------------------------------------------------------------------------------------------------------
#include "sac_capi.hip.inc"
#include "shac_capi.hip.inc"
