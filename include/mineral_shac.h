/* mineral_shac.h — C ABI of the critic side of SHAC on the MI355X (gfx950): TD(lambda) target scan, ensemble-critic
 * regression gradient, target-critic blend.  Same conventions as mineral_ppo.h (device pointers, fp32, `stream` =
 * hipStream_t as void*, 0 / negative MRL_E* status, nothing allocates).  Clip + Adam/AdamW is mrl_adam_step.
 * Citations are `mineral/<path>:<lines>` in the reference checkout (jacr13/mineral).  The actor side of SHAC
 * back-propagates through a differentiable simulator and stays outside this library.
 */
#ifndef MINERAL_SHAC_H
#define MINERAL_SHAC_H

#include <stdint.h>

#include "mineral_ppo.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MRL_CRITIC_ACT_SILU 0
#define MRL_CRITIC_ACT_ELU 1

/* EnsembleCritic / Critic (mineral/agents/diffrl/models.py:149-256): n_critics (1 or 2) towers of
 * [Linear -> LayerNorm -> activation] x n_layers on the observation, each ending in a plain Linear to 1.  act_type: SiLU
 * (DFlexAntSHAC2.yaml, DFlexAntSAPO.yaml) or ELU (DFlexAntSHAC.yaml).  Widths in {32,64,128,256,512}; obs_dim a multiple
 * of 4. */
typedef struct mrl_critic_net {
  int32_t obs_dim;
  int32_t n_critics;
  int32_t n_layers;
  int32_t units[MRL_MAX_LAYERS];
  float ln_eps;
  int32_t act_type;
} mrl_critic_net;

/* Flat parameter vector: tower 0, (tower 1,) head 0, (head 1); offsets per tower per layer W, b, ln_gamma, ln_beta, then per
 * head W, b. */
int64_t mrl_critic_param_count(const mrl_critic_net* net);
int mrl_critic_param_offsets(const mrl_critic_net* net, int64_t* offsets, int32_t* n_offsets);
int64_t mrl_critic_workspace_bytes(const mrl_critic_net* net, int64_t rows);

/* SHAC.compute_target_values (mineral/agents/diffrl/shac.py:876-889).  rew, done_mask, next_values, target_values are
 * TIME-MAJOR [horizon][num_envs] like the reference's buffers (shac.py:210-216).  one_step != 0: rew + gamma * next_values;
 * otherwise the TD(lambda) reverse scan that carries (Ai, Bi, lam) per env.  gamma and lam are doubles because the reference
 * combines them as Python floats before rounding to fp32 (lam * gamma, 1 - lam). */
int mrl_tdlambda_targets(const float* rew, const float* done_mask, const float* next_values, int64_t num_envs, int32_t horizon, double gamma,
                         double lam, int32_t one_step, float* target_values, void* stream);

/* SHAC.compute_target_values_with_entropy after its entropy transforms (mineral/agents/diffrl/shac.py:900-916): the same
 * targets with rew + alpha * entropy in place of rew.  entropy: [horizon][num_envs], already offset / scaled by the target
 * entropy where the config asks for it (shac.py:894-898 are element-wise and stay with the caller); alpha: the entropy
 * coefficient or the current temperature. */
int mrl_tdlambda_targets_entropy(const float* rew, const float* done_mask, const float* next_values, const float* entropy, double alpha,
                                 int64_t num_envs, int32_t horizon, double gamma, double lam, int32_t one_step, float* target_values,
                                 void* stream);

/* EnsembleCritic.forward(return_type='all') (models.py:236-243): values [n_critics][rows].  obs is normalised with
 * (obs_mean, obs_std, clamp) when given. */
int mrl_critic_values(const mrl_critic_net* net, const float* params, const float* obs, int64_t rows, const float* obs_mean, const float* obs_std,
                      float clamp_lo, float clamp_hi, float* values, void* workspace, int64_t workspace_bytes, void* stream);

/* One batch of SHAC.update_critic up to the optimizer step (shac.py:836-842, 868-874): loss = mean over critics of
 * mean((V_i(obs) - target)^2), its gradient with non-finite entries replaced by 0 (nan_to_num_(0, 0, 0)).
 * metrics[0] = loss. */
int mrl_critic_mse_grad(const mrl_critic_net* net, const float* params, const float* obs, const float* target, int64_t rows, const float* obs_mean,
                        const float* obs_std, float clamp_lo, float clamp_hi, float* grads, double* metrics, void* workspace,
                        int64_t workspace_bytes, void* stream);

/* soft_update of mineral/agents/diffrl/utils.py:25-28: target = w_target * target + w_current * current, element-wise
 * (the reference passes alpha and 1 - alpha). */
int mrl_lerp_params(float* target, const float* current, int64_t count, float w_target, float w_current, void* stream);

#ifdef __cplusplus
}
#endif
#endif
