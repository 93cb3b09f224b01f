// Device-resident replay path for gfx950: n-step FIFO + masked discounted return, ring-buffer write, batch gather.
// All three are copy kernels bound by HBM; one wave moves one transition with 16-byte lanes.
#include "../../include/mineral_ppo.h"
#include "../../include/mineral_replay.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace mrl {
void set_error(const char* msg);   // defined next to mrl_last_error()
}

#define RPL_FAIL(code, ...)                       \
  do {                                            \
    char b_[256];                                 \
    snprintf(b_, sizeof(b_), __VA_ARGS__);        \
    mrl::set_error(b_);                           \
    return (code);                                \
  } while (0)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Soa {
  float* obs;
  float* action;
  float* reward;
  float* next_obs;
  void* done;
};

__device__ __forceinline__ void copy_row(float* __restrict__ dst, const float* __restrict__ src, int n4, int lane) {
  for (int c = lane; c < n4; c += 64)
    reinterpret_cast<f32x4*>(dst)[c] = reinterpret_cast<const f32x4*>(src)[c];
}

// One wave per emitted transition.  Window position j of emitted step i is call step i - (nstep-1) + j; negative call
// steps live in the history (slot H + step, H = nstep - 1).
__global__ __launch_bounds__(256) void nstep_emit_kernel(Soa traj, Soa hist, Soa out, int64_t num_envs, int horizon, int nstep, int first_step,
                                                         int64_t out_rows, int obs_dim, int act_dim, const float* __restrict__ gamma_pow) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= out_rows) return;
  const int64_t e = row / num_envs, n = row % num_envs;
  const int i = first_step + (int)e;   // call step whose push completes this window
  const int H = nstep - 1;
  const float* tdone = static_cast<const float*>(traj.done);
  const float* hdone = static_cast<const float*>(hist.done);
  // first done step of the window (all lanes compute the same few scalars)
  int last = nstep - 1;
  bool any = false;
  float term[MRL_NSTEP_MAX];
#pragma unroll
  for (int j = 0; j < MRL_NSTEP_MAX; ++j) {
    term[j] = 0.0f;
    if (j < nstep) {
      const int s = i - H + j;
      const float d = s >= 0 ? tdone[n * horizon + s] : hdone[n * H + (H + s)];
      const float r = s >= 0 ? traj.reward[n * horizon + s] : hist.reward[n * H + (H + s)];
      if (d != 0.0f && !any) { any = true; last = j; }
      term[j] = r * gamma_pow[j];
    }
  }
  // masked products, summed in the order of torch's reduction over the short step dimension: four accumulators seeded
  // by the first four values, every further value into the first, then combined left to right
#pragma unroll
  for (int j = 0; j < MRL_NSTEP_MAX; ++j) term[j] = term[j] * ((j <= last) ? 1.0f : 0.0f);
  float total;
  if (nstep < 4) {
    total = term[0];
    for (int j = 1; j < nstep; ++j) total = total + term[j];
  } else {
    float a0 = term[0];
    for (int j = 4; j < nstep; ++j) a0 = a0 + term[j];
    total = ((a0 + term[1]) + term[2]) + term[3];
  }
  const int s_old = i - H;           // oldest window step: source of obs / action
  const int s_nxt = i - H + last;    // source of next_obs
  const float* src_obs = s_old >= 0 ? traj.obs + (n * horizon + s_old) * obs_dim : hist.obs + (n * H + (H + s_old)) * obs_dim;
  const float* src_act = s_old >= 0 ? traj.action + (n * horizon + s_old) * act_dim : hist.action + (n * H + (H + s_old)) * act_dim;
  const float* src_nxt = s_nxt >= 0 ? traj.next_obs + (n * horizon + s_nxt) * obs_dim : hist.next_obs + (n * H + (H + s_nxt)) * obs_dim;
  copy_row(out.obs + row * obs_dim, src_obs, obs_dim / 4, lane);
  copy_row(out.next_obs + row * obs_dim, src_nxt, obs_dim / 4, lane);
  if (lane < act_dim) out.action[row * act_dim + lane] = src_act[lane];
  if (lane == 0) {
    out.reward[row] = total;
    const float dlast = tdone[n * horizon + i];   // the newest window step is always a step of this call
    static_cast<float*>(out.done)[row] = any ? 1.0f : dlast;
  }
}

// Advance the per-env history to the end of this call: slot h takes call step horizon - H + h, or an older history
// slot when that step precedes the call.  One wave per env walks the slots oldest first, so every slot is read before
// it is overwritten.
__global__ __launch_bounds__(256) void nstep_history_kernel(Soa traj, Soa hist, int64_t num_envs, int horizon, int H, int obs_dim, int act_dim) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= num_envs) return;
  float* hdone = static_cast<float*>(hist.done);
  const float* tdone = static_cast<const float*>(traj.done);
  for (int h = 0; h < H; ++h) {
    const int s = horizon - H + h;
    const int64_t d = n * H + h;
    if (s >= 0) {
      const int64_t r = n * horizon + s;
      copy_row(hist.obs + d * obs_dim, traj.obs + r * obs_dim, obs_dim / 4, lane);
      copy_row(hist.next_obs + d * obs_dim, traj.next_obs + r * obs_dim, obs_dim / 4, lane);
      if (lane < act_dim) hist.action[d * act_dim + lane] = traj.action[r * act_dim + lane];
      if (lane == 0) { hist.reward[d] = traj.reward[r]; hdone[d] = tdone[r]; }
    } else {
      const int64_t r = n * H + (H + s);   // = slot h + horizon, a later slot of the old history
      // each lane reads its own elements before writing them (same lane, earlier slot), so the shift is safe in place
      for (int c = lane; c < obs_dim / 4; c += 64) {
        reinterpret_cast<f32x4*>(hist.obs + d * obs_dim)[c] = reinterpret_cast<const f32x4*>(hist.obs + r * obs_dim)[c];
        reinterpret_cast<f32x4*>(hist.next_obs + d * obs_dim)[c] = reinterpret_cast<const f32x4*>(hist.next_obs + r * obs_dim)[c];
      }
      if (lane < act_dim) hist.action[d * act_dim + lane] = hist.action[r * act_dim + lane];
      if (lane == 0) { hist.reward[d] = hist.reward[r]; hdone[d] = hdone[r]; }
    }
  }
}

// Work items of one transition: 16-byte chunks of obs, then of next_obs, then the action scalars, reward, done.
// A lane owns item `lane` (+64, +128 ... for wide observations) of every row its wave moves, and a wave keeps ROWS_PER_WAVE
// rows in flight: all their loads are issued before the first store.
constexpr int ROWS_PER_WAVE = 4;

template <bool RING_SRC>   // RING_SRC: source `done` is uint8 (ring), destination float; otherwise float -> uint8
__device__ __forceinline__ void move_rows(const Soa& dst, const Soa& src, const int64_t (&drow)[ROWS_PER_WAVE], const int64_t (&srow)[ROWS_PER_WAVE],
                                          const bool (&live)[ROWS_PER_WAVE], int obs_dim, int act_dim, int lane) {
  const int C = obs_dim / 4, items = 2 * C + act_dim + 2;
  for (int k0 = 0; k0 < items; k0 += 64) {
    const int k = k0 + lane;
    f32x4 v[ROWS_PER_WAVE];
#pragma unroll
    for (int q = 0; q < ROWS_PER_WAVE; ++q) {
      v[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!live[q] || k >= items) continue;
      const int64_t r = srow[q];
      if (k < C) v[q] = reinterpret_cast<const f32x4*>(src.obs + r * obs_dim)[k];
      else if (k < 2 * C) v[q] = reinterpret_cast<const f32x4*>(src.next_obs + r * obs_dim)[k - C];
      else if (k < 2 * C + act_dim) v[q][0] = src.action[r * act_dim + (k - 2 * C)];
      else if (k == 2 * C + act_dim) v[q][0] = src.reward[r];
      else v[q][0] = RING_SRC ? (static_cast<const uint8_t*>(src.done)[r] ? 1.0f : 0.0f) : static_cast<const float*>(src.done)[r];
    }
#pragma unroll
    for (int q = 0; q < ROWS_PER_WAVE; ++q) {
      if (!live[q] || k >= items) continue;
      const int64_t r = drow[q];
      if (k < C) reinterpret_cast<f32x4*>(dst.obs + r * obs_dim)[k] = v[q];
      else if (k < 2 * C) reinterpret_cast<f32x4*>(dst.next_obs + r * obs_dim)[k - C] = v[q];
      else if (k < 2 * C + act_dim) dst.action[r * act_dim + (k - 2 * C)] = v[q][0];
      else if (k == 2 * C + act_dim) dst.reward[r] = v[q][0];
      else if (RING_SRC) static_cast<float*>(dst.done)[r] = v[q][0];                       // .float()
      else static_cast<uint8_t*>(dst.done)[r] = v[q][0] != 0.0f ? 1 : 0;                     // .bool()
    }
  }
}

__global__ __launch_bounds__(256) void replay_write_kernel(Soa ring, Soa src, int64_t capacity, int64_t next_p, int64_t rows, int obs_dim, int act_dim) {
  const int lane = threadIdx.x & 63;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS_PER_WAVE;
  int64_t drow[ROWS_PER_WAVE], srow[ROWS_PER_WAVE];
  bool live[ROWS_PER_WAVE];
#pragma unroll
  for (int q = 0; q < ROWS_PER_WAVE; ++q) {
    srow[q] = r0 + q;
    live[q] = srow[q] < rows;
    drow[q] = (next_p + srow[q]) % capacity;
  }
  move_rows<false>(ring, src, drow, srow, live, obs_dim, act_dim, lane);
}

__global__ __launch_bounds__(256) void replay_gather_kernel(Soa ring, Soa out, const int64_t* __restrict__ indices, int64_t batch, int obs_dim, int act_dim) {
  const int lane = threadIdx.x & 63;
  const int64_t b0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS_PER_WAVE;
  int64_t drow[ROWS_PER_WAVE], srow[ROWS_PER_WAVE];
  bool live[ROWS_PER_WAVE];
#pragma unroll
  for (int q = 0; q < ROWS_PER_WAVE; ++q) {
    drow[q] = b0 + q;
    live[q] = drow[q] < batch;
    srow[q] = live[q] ? indices[drow[q]] : 0;
  }
  move_rows<true>(out, ring, drow, srow, live, obs_dim, act_dim, lane);
}

Soa soa(const mrl_transitions* t) { return Soa{t->obs, t->action, t->reward, t->next_obs, t->done}; }

int check_tr(const mrl_transitions* t, const char* what) {
  if (!t) RPL_FAIL(MRL_EINVAL, "%s is NULL", what);
  if (!t->obs || !t->action || !t->reward || !t->next_obs || !t->done) RPL_FAIL(MRL_EINVAL, "%s has a NULL array", what);
  if (t->obs_dim < 4 || t->obs_dim % 4) RPL_FAIL(MRL_EINVAL, "%s: obs_dim %d must be a positive multiple of 4", what, t->obs_dim);
  if (t->act_dim < 1 || t->act_dim > 64) RPL_FAIL(MRL_EINVAL, "%s: act_dim %d not in 1..64", what, t->act_dim);
  return MRL_OK;
}

int same_dims(const mrl_transitions* a, const mrl_transitions* b) {
  if (a->obs_dim != b->obs_dim || a->act_dim != b->act_dim)
    RPL_FAIL(MRL_EINVAL, "transition sets disagree on obs_dim / act_dim (%d,%d vs %d,%d)", a->obs_dim, a->act_dim, b->obs_dim, b->act_dim);
  return MRL_OK;
}

int launch_ok(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) RPL_FAIL(MRL_EHIP, "%s: %s", what, hipGetErrorString(e));
  return MRL_OK;
}

}  // namespace

extern "C" int mrl_nstep_emit(const mrl_transitions* traj, int64_t num_envs, int32_t horizon, mrl_transitions* window, int64_t count_before,
                              int32_t nstep, const float* gamma_pow, mrl_transitions* out, void* stream) {
  if (check_tr(traj, "traj") || check_tr(window, "window") || check_tr(out, "out")) return MRL_EINVAL;
  if (same_dims(traj, window) || same_dims(traj, out)) return MRL_EINVAL;
  if (nstep < 2 || nstep > MRL_NSTEP_MAX) RPL_FAIL(MRL_EINVAL, "nstep %d not in 2..%d (nstep 1 passes trajectories through unchanged)", nstep, MRL_NSTEP_MAX);
  if (num_envs < 1 || horizon < 1 || count_before < 0) RPL_FAIL(MRL_EINVAL, "num_envs %lld, horizon %d, count_before %lld", (long long)num_envs, horizon, (long long)count_before);
  if (!gamma_pow) RPL_FAIL(MRL_EINVAL, "gamma_pow is NULL");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // step i is emitted when count_before + i + 1 >= nstep
  int64_t first = (int64_t)nstep - 1 - count_before;
  if (first < 0) first = 0;
  const int64_t emitted = horizon > first ? horizon - first : 0;
  const int H = nstep - 1;
  if (emitted > 0) {
    const int64_t rows = emitted * num_envs;
    hipLaunchKernelGGL(nstep_emit_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, soa(traj), soa(window), soa(out), num_envs, horizon, nstep,
                       (int)first, rows, traj->obs_dim, traj->act_dim, gamma_pow);
    if (launch_ok("nstep_emit_kernel")) return MRL_EHIP;
  }
  hipLaunchKernelGGL(nstep_history_kernel, dim3((unsigned)((num_envs + 3) / 4)), dim3(256), 0, s, soa(traj), soa(window), num_envs, horizon, H,
                     traj->obs_dim, traj->act_dim);
  if (launch_ok("nstep_history_kernel")) return MRL_EHIP;
  return (int)emitted;
}

extern "C" int mrl_replay_write(mrl_transitions* ring, int64_t capacity, int64_t next_p, const mrl_transitions* src, int64_t rows, void* stream) {
  if (check_tr(ring, "ring") || check_tr(src, "src") || same_dims(ring, src)) return MRL_EINVAL;
  if (capacity < 1 || next_p < 0 || next_p > capacity) RPL_FAIL(MRL_EINVAL, "capacity %lld, next_p %lld", (long long)capacity, (long long)next_p);
  if (rows < 0 || rows > capacity) RPL_FAIL(MRL_EINVAL, "rows %lld not in 0..capacity %lld", (long long)rows, (long long)capacity);
  if (rows == 0) return MRL_OK;
  hipLaunchKernelGGL(replay_write_kernel, dim3((unsigned)((rows + 4 * ROWS_PER_WAVE - 1) / (4 * ROWS_PER_WAVE))), dim3(256), 0, static_cast<hipStream_t>(stream), soa(ring), soa(src), capacity, next_p,
                     rows, ring->obs_dim, ring->act_dim);
  return launch_ok("replay_write_kernel");
}

extern "C" int mrl_replay_gather(const mrl_transitions* ring, int64_t capacity, const int64_t* indices, int64_t batch, mrl_transitions* out, void* stream) {
  if (check_tr(ring, "ring") || check_tr(out, "out") || same_dims(ring, out)) return MRL_EINVAL;
  if (!indices) RPL_FAIL(MRL_EINVAL, "indices is NULL");
  if (capacity < 1 || batch < 0) RPL_FAIL(MRL_EINVAL, "capacity %lld, batch %lld", (long long)capacity, (long long)batch);
  if (batch == 0) return MRL_OK;
  hipLaunchKernelGGL(replay_gather_kernel, dim3((unsigned)((batch + 4 * ROWS_PER_WAVE - 1) / (4 * ROWS_PER_WAVE))), dim3(256), 0, static_cast<hipStream_t>(stream), soa(ring), soa(out), indices, batch,
                     ring->obs_dim, ring->act_dim);
  return launch_ok("replay_gather_kernel");
}
