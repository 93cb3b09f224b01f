/* mineral_replay.h — C ABI of the device-resident replay path (gfx950): n-step FIFO, ring buffer, batch gather.
 *
 * Same conventions as mineral_ppo.h: device pointers, fp32 unless typed otherwise, `stream` is a hipStream_t passed
 * as void*, 0 / negative MRL_E* status with mrl_last_error(), nothing allocates or synchronises.
 * Citations are `mineral/<path>:<lines>` in the reference checkout (jacr13/mineral).
 */
#ifndef MINERAL_REPLAY_H
#define MINERAL_REPLAY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRL_NSTEP_MAX 7   /* the reference's fp32 summation order is pinned up to this n */

/* A set of transitions as a structure of arrays — the five tensors of mineral/buffers/simple_replay.py:4-16 for a
 * single flat observation key.  `done` is float where the reference hands floats around (trajectories, n-step
 * output, sampled batches) and uint8 where it stores bool (the ring): each entry point says which. */
typedef struct mrl_transitions {
  float* obs;       /* [rows][obs_dim] */
  float* action;    /* [rows][act_dim] */
  float* reward;    /* [rows]          */
  float* next_obs;  /* [rows][obs_dim] */
  void* done;       /* [rows]          */
  int32_t obs_dim;  /* multiple of 4   */
  int32_t act_dim;
} mrl_transitions;

/* NStepReplay.add_to_buffer for nstep > 1 (mineral/buffers/nstep_replay.py:31-62, compute_nstep_return :71-94).
 *   traj    : this call's trajectory, ENV-MAJOR rows n*horizon + t (the (N, T, ...) tensors of
 *             mineral/agents/sac/sac.py:140-186), done float
 *   window  : the last nstep-1 transitions of every env from earlier calls, rows n*(nstep-1) + h, oldest first, done
 *             float; read, then advanced to the end of this call (the FIFO of nstep_replay.py:66-68 without the
 *             torch.cat reallocation)
 *   count_before : transitions pushed per env before this call (nstep_count); step i of this call is emitted when
 *             count_before + i + 1 >= nstep
 *   gamma_pow : [nstep] fp32 powers gamma**j, rounded once from double like nstep_replay.py:28
 *   out     : emitted transitions, STEP-MAJOR rows e*num_envs + n like the torch.cat of nstep_replay.py:59-62,
 *             done float; capacity >= emitted * num_envs rows
 * Per emitted row: obs/action of the oldest window step; reward = sum of gamma_pow[j] * reward[j] up to and
 * including the first done step (fp32, in the association torch uses for that reduction); next_obs of that step (the
 * newest when no step is done); done = any done in the window.
 * Returns the number of emitted steps (>= 0) or a negative status. */
int mrl_nstep_emit(const mrl_transitions* traj, int64_t num_envs, int32_t horizon, mrl_transitions* window, int64_t count_before,
                   int32_t nstep, const float* gamma_pow, mrl_transitions* out, void* stream);

/* ReplayBuffer.add_to_buffer (mineral/buffers/simple_replay.py:33-71): copy `rows` transitions (done float) into the
 * ring (done uint8) at positions (next_p + r) mod capacity; rows <= capacity. */
int mrl_replay_write(mrl_transitions* ring, int64_t capacity, int64_t next_p, const mrl_transitions* src, int64_t rows, void* stream);

/* ReplayBuffer.sample_batch with the index draw supplied by the caller (simple_replay.py:73-85): out[b] =
 * ring[indices[b]], done converted to float.  Indices must lie in [0, capacity). */
int mrl_replay_gather(const mrl_transitions* ring, int64_t capacity, const int64_t* indices, int64_t batch, mrl_transitions* out,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif
