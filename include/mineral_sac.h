/* mineral_sac.h — C ABI of the MI355X (gfx950) SAC update loop: policy sampling, critic and actor gradients, Polyak.
 *
 * Same conventions as mineral_ppo.h (device pointers, fp32, `stream` = hipStream_t as void*, 0 / negative MRL_E*
 * status, nothing allocates: callers pass a workspace sized by mrl_sac_workspace_bytes).  Replay storage and
 * sampling are in mineral_replay.h; clip + Adam/AdamW of each parameter group is mrl_adam_step of mineral_ppo.h.
 * Citations are `mineral/<path>:<lines>` in the reference checkout (jacr13/mineral).
 */
#ifndef MINERAL_SAC_H
#define MINERAL_SAC_H

#include <stdint.h>

#include "mineral_ppo.h"
#include "mineral_replay.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MRL_SAC_NMETRIC 8   /* doubles written by the gradient entry points, see each */

/* Actor (mineral/agents/ddpg/models.py:8-69: MLP tower, `mu` and `sigma` heads, squashed_normal with the log-std clamp
 * of mineral/nets/dist.py:39-43) and EnsembleQ with two critics on cat(obs, action) (models.py:72-118).  Towers are
 * [Linear -> LayerNorm -> SiLU] x n_layers with widths in {32,64,128,256,512}; obs_dim and obs_dim + act_dim must be
 * multiples of 4; act_dim <= 16. */
typedef struct mrl_sac_net {
  int32_t obs_dim;
  int32_t act_dim;
  int32_t actor_layers;
  int32_t actor_units[MRL_MAX_LAYERS];
  int32_t critic_layers;
  int32_t critic_units[MRL_MAX_LAYERS];
  float minlogstd, maxlogstd;
  float ln_eps;
  int32_t actor_kind;    /* MRL_TOWER_* (mineral_ppo.h) of the actor's mlp_kwargs; 0 = LayerNorm + SiLU (DFlexAntSAC.yaml) */
  int32_t critic_kind;   /* the same for the critics */
} mrl_sac_net;

/* Flat parameter vectors.  which = 0: the actor (tower, then the [mu; sigma] head of 2*act_dim rows);
 * which = 1: the critic ensemble (tower 0, tower 1, head 0, head 1) — also the layout of the target critics.
 * mrl_sac_param_offsets lists, per tower, per layer W, b, ln_gamma, ln_beta (W, b for kinds without a normalisation layer), then
 * per head W, b (float offsets). */
int64_t mrl_sac_param_count(const mrl_sac_net* net, int32_t which);
int mrl_sac_param_offsets(const mrl_sac_net* net, int32_t which, int64_t* offsets, int32_t* n_offsets);
int64_t mrl_sac_workspace_bytes(const mrl_sac_net* net, int64_t rows);

/* SAC.get_actions(sample=True, logprob=True) (mineral/agents/sac/sac.py:118-136): normalise obs with (obs_mean, obs_std,
 * clamp) when given (normalizers.py:65-68), actor forward, a = tanh(mu + sigma * noise), log-prob of a.
 * actions [rows][act_dim]; logp [rows] or NULL. */
int mrl_sac_actor_sample(const mrl_sac_net* net, const float* actor_params, const float* obs, int64_t rows, const float* obs_mean,
                         const float* obs_std, float clamp_lo, float clamp_hi, const float* noise, float* actions, float* logp,
                         void* workspace, int64_t workspace_bytes, void* stream);

/* SAC.update_critic up to the gradient (sac.py:287-313): target action and log-prob from next_obs with `noise_next`,
 * target = reward + (1 - done) * gamma_n * (min_i Qtarget_i - alpha * logp) (alpha = exp(*log_alpha), dropped when
 * backup_entropy is 0), loss = sum_i mean((Q_i(obs, action) - target)^2), gradient with respect to the critic
 * parameters.  batch->done is float.  metrics: [0] loss, [1] mean target, [2] sum of squared targets / rows, [3] max
 * target, [4] min target. */
int mrl_sac_critic_grad(const mrl_sac_net* net, const float* actor_params, const float* critic_params, const float* target_params,
                        const float* log_alpha, const float* obs_mean, const float* obs_std, float clamp_lo, float clamp_hi,
                        const mrl_transitions* batch, int64_t rows, const float* noise_next, float gamma_n, int32_t backup_entropy,
                        float* critic_grads, double* metrics, void* workspace, int64_t workspace_bytes, void* stream);

/* SAC.update_actor up to the gradients (sac.py:315-334): a, logp from obs with `noise`; actor loss
 * mean(alpha * logp - min_i Q_i(obs, a)) differentiated through the critics into the actor parameters; alpha loss
 * mean(exp(log_alpha) * (-logp - target_entropy)) differentiated with respect to log_alpha (alpha_grad [1]).
 * metrics: [0] actor loss, [1] alpha loss, [2] mean entropy (-logp). */
int mrl_sac_actor_grad(const mrl_sac_net* net, const float* actor_params, const float* critic_params, const float* log_alpha,
                       const float* obs_mean, const float* obs_std, float clamp_lo, float clamp_hi, const float* obs, int64_t rows,
                       const float* noise, float target_entropy, float* actor_grads, float* alpha_grad, double* metrics, void* workspace,
                       int64_t workspace_bytes, void* stream);

/* soft_update (mineral/agents/ddpg/utils.py:7-11): target = target * (1 - tau) + current * tau, element-wise. */
int mrl_polyak(float* target, const float* current, int64_t count, float tau, void* stream);

#ifdef __cplusplus
}
#endif
#endif
