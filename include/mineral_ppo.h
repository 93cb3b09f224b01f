/* mineral_ppo.h — C ABI of the MI355X (gfx950) PPO actor-learner hot path.
 *
 * The reference (jacr13/mineral) is pure Python on torch and has no FFI of its own; its plugin boundary is the
 * Python classes named below.  Each entry point here is what a binding for that reference call would use.
 * Citations are `mineral/<path>:<lines>` in the reference checkout.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`; fp32 unless typed otherwise
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is enqueued, nothing syncs
 *   - every function returns 0 on success or a negative MRL_E* code; mrl_last_error() gives the message of the
 *     calling thread's last failure.  Nothing here allocates device memory: callers pass workspaces sized by
 *     mrl_workspace_bytes() and aligned to 256 bytes (hipMalloc and torch allocations are); a workspace only has to
 *     outlive the calls that use it, and calls that share one must not overlap.
 *   - layouts: the rollout buffer is ENV-MAJOR for per-sample vectors, so the flat sample index of
 *     (env n, step t) is n*T + t — exactly the index map produced by transpose(0,1)+reshape in
 *     mineral/agents/ppo/experience.py:7-14 — and minibatch i is the contiguous row range [i*mb, (i+1)*mb)
 *     (experience.py:61-64).  Per-step scalars are staged TIME-MAJOR ([T][N]) during collection and converted by
 *     mrl_gae_finish.
 *   - arithmetic: fp32 operands, fp32 accumulation, fp32 results everywhere.  On batches above 8,192 rows the GEMMs of the
 *     hidden layers multiply on the bf16 matrix cores from exact three-term splits of their fp32 operands (six bf16
 *     products per fp32 product, accumulated in fp32); the error against an exact sum is below that of an fp32 fma chain,
 *     and results differ from the fp32-MFMA path only by summation order and that rounding.
 */
#ifndef MINERAL_PPO_H
#define MINERAL_PPO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRL_OK 0
#define MRL_EINVAL (-1)   /* bad argument / unsupported shape */
#define MRL_ENOSPC (-2)   /* workspace too small */
#define MRL_EHIP (-3)     /* a HIP call or launch failed */

#define MRL_MAX_LAYERS 4
#define MRL_DIST_NORMAL 0
#define MRL_DIST_DREAMERV3_NORMAL 1
#define MRL_NMETRIC 12

#define MRL_TOWER_LAYERNORM_SILU 0   /* [Linear -> LayerNorm -> SiLU] per layer (DFlexAntPPO.yaml, AllegroHandPPO.yaml) */
#define MRL_TOWER_ELU 1              /* [Linear -> ELU] per layer, no normalisation (CartpolePPO.yaml) */
#define MRL_TOWER_LAYERNORM_ELU 2     /* [Linear -> LayerNorm -> ELU] per layer (the mlp_kwargs of DFlexAntSHAC.yaml / DFlexAntBPTT.yaml) */
/* the other norm_type / act_type combinations of mineral/nets/mlp.py:5-22 (norm None or LayerNorm; SiLU, ELU, ReLU, Tanh); these
 * five run through the row-local chain kernels at every batch size (the per-layer split-bf16 kernels of large batches are built
 * for kinds 0-2 only) */
#define MRL_TOWER_SILU 3             /* [Linear -> SiLU] */
#define MRL_TOWER_LAYERNORM_RELU 4   /* [Linear -> LayerNorm -> ReLU] */
#define MRL_TOWER_RELU 5             /* [Linear -> ReLU] (the default act_type of mlp.py) */
#define MRL_TOWER_LAYERNORM_TANH 6   /* [Linear -> LayerNorm -> Tanh] */
#define MRL_TOWER_TANH 7             /* [Linear -> Tanh] */

/* ActorCritic description: two towers of n_layers hidden layers, heads mu/sigma (actor) and value (critic).
 * mineral/agents/ppo/models.py:29-73, mineral/nets/mlp.py:25-77, mineral/nets/dist.py:31-47.
 * tower_kind: one of MRL_TOWER_*.  fixed_sigma != 0: the log-std is a state-independent parameter of act_dim entries
 * (models.py:66-67,151) stored right after the mu bias; otherwise it is a second head beside mu.
 * Limits of this build: units in {32,64,128,256,512}, act_dim <= 16, obs_dim a multiple of 4: observation rows and the rows of
 * the first layer's weights are 16-byte vectors.  Callers with another width pad both with zero columns (the host classes
 * do: 37 observations of DFlex Ant become rows of 40). */
typedef struct mrl_net {
  int32_t obs_dim;
  int32_t act_dim;
  int32_t n_layers;
  int32_t units[MRL_MAX_LAYERS];
  int32_t dist_type;
  float minstd, maxstd;
  float ln_eps;
  int32_t tower_kind;
  int32_t fixed_sigma;
} mrl_net;

/* RunningMeanStd state, mineral/common/normalizers.py:19-42 (fp64 buffers as built at ppo.py:30). */
typedef struct mrl_rms {
  double* mean;   /* [size] */
  double* var;    /* [size] */
  double* count;  /* [1]    */
  int32_t size;
  float eps;
  float clamp_lo, clamp_hi;
} mrl_rms;

/* Device-resident rollout storage, mineral/agents/ppo/experience.py:18-56. */
typedef struct mrl_rollout {
  int64_t num_envs;      /* N */
  int32_t horizon;       /* T */
  float* obs;            /* [N][T][obs_dim] */
  float* actions;        /* [N][T][act_dim]  (unclamped samples) */
  float* mu;             /* [N][T][act_dim] */
  float* sigma;          /* [N][T][act_dim] */
  float* neglogp_tn;     /* [T][N] */
  float* values_tn;      /* [T][N]  un-normalised value predictions */
  float* rewards_tn;     /* [T][N]  shaped rewards */
  uint8_t* dones_tn;     /* [T][N] */
  /* flat per-sample scalars (index n*T+t), written by mrl_gae_finish */
  float* values;         /* value_preds, normalised if normalize_value */
  float* returns;
  float* advantages;
  float* neglogp;
} mrl_rollout;

/* PPO loss hyper-parameters, mineral/agents/ppo/ppo.py:63-76. */
typedef struct mrl_ppo_hp {
  float e_clip;
  float critic_coef;
  float entropy_coef;
  float bounds_loss_coef;
  int32_t bounds_type;       /* 0 'bound', 1 'reg' (ppo.py:479-489) */
  int32_t clip_value_loss;
  int32_t use_smooth_clamp;  /* ratio clamp of the actor loss: 0 torch.clamp, 1 smooth_clamp (ppo.py:449-457) */
} mrl_ppo_hp;

const char* mrl_last_error(void);
int mrl_version(void);

/* Number of fp32 elements of the flat parameter vector and the offset of every tensor in it.
 * Order: per tower (actor, critic) per layer W[u][in], b[u], ln_gamma[u], ln_beta[u]; then the actor head
 * W[2A][U] (rows 0..A-1 = mu.weight, A..2A-1 = sigma.weight), b[2A]; then value.weight[1][U], value.bias[1].
 * Every offset is a multiple of 4.  offsets_host has room for 2*4*MRL_MAX_LAYERS + 4 entries. */
int64_t mrl_param_count(const mrl_net* net);
int mrl_param_offsets(const mrl_net* net, int64_t* offsets_host, int32_t* n_tensors_host);

/* Bytes of scratch the calls below need for `rows` simultaneous samples (minibatch rows or envs). */
int64_t mrl_workspace_bytes(const mrl_net* net, int64_t rows);

/* PPO.model_act + ExperienceBuffer.update_data for one env step (ppo.py:125-131,144-148; models.py:110-126):
 * normalise obs with obs_rms (no update), run both towers, sample a = mu + sigma*noise, neglogp, un-normalise the
 * value with value_rms, and write row t of the buffer (obs, actions, mu, sigma, neglogp_tn, values_tn).
 * actions_clamped [N][A] receives clamp(a,-1,1) for env.step (ppo.py:150).  With t == horizon only the values are
 * computed and written to last_values [N] (the bootstrap forward at ppo.py:169-170).
 * obs_rms / value_rms may be NULL (normalize_input / normalize_value off). */
int mrl_policy_act(const mrl_net* net, const float* params, const mrl_rms* obs_rms, const mrl_rms* value_rms,
                   const float* obs, const float* noise, int32_t t, const mrl_rollout* buf, float* actions_clamped,
                   float* last_values, void* workspace, int64_t workspace_bytes, void* stream);

/* Metrics.update of one env step (mineral/common/metrics.py:59-67, mineral/common/tracker.py:10-30; callers
 * ppo.py:165-167, sac.py:163-165) without the reference's host round trip (`torch.where(dones)[0].tolist()`):
 * cur_rewards += rewards, cur_lengths += 1; the finished episodes (dones != 0, in env order) are appended to the two
 * rings that hold the last `window` finished episodes (oldest entries fall out), *head advances, *num_episodes grows,
 * and the finished envs' running reward / length restart at 0.  dones: num_envs bytes (bool / uint8).
 * ring_rewards / ring_lengths: `window` doubles each (the trackers' deques); head, num_episodes: one int64 each. */
int mrl_episode_update(const float* rewards, const uint8_t* dones, int64_t num_envs, float* cur_rewards,
                       int64_t* cur_lengths, double* ring_rewards, double* ring_lengths, int32_t window, int64_t* head,
                       int64_t* num_episodes, void* stream);

/* ActorCritic.act(obs_dict, sample=False) (models.py:112-114): the raw output of the mu head, before the distribution's
 * transform (no tanh for dreamerv3_normal), mu_out [rows][act_dim].  obs_rms may be NULL (input already normalised, as
 * when the reference's model is called directly). */
int mrl_policy_mean(const mrl_net* net, const float* params, const mrl_rms* obs_rms, const float* obs, int64_t rows,
                    float* mu_out, void* workspace, int64_t workspace_bytes, void* stream);

/* Metrics._update_infos of one env step for ONE regex-selected per-env scalar and ONE aggregation (metrics.py:76-113):
 * mode 0 `sum` (cur += x), 1 `min` (cur = fmin(cur, x)), 2 `max` (cur = fmax(cur, x)), 3 `final` (cur[done] = x[done]).
 * The values of the finished envs (dones != 0, in env order) are appended to `ring` (`window` doubles, the key's Tracker) at
 * the positions mrl_episode_update gives the same episodes' rewards, and those envs restart at 0 or, with reset_nan != 0,
 * at NaN (metrics.py:110-113).  `head` is the reward ring's head: it is only read here, so call this BEFORE the
 * mrl_episode_update of the same step.  cur [num_envs] starts at 0 (sum, final) or NaN (min, max). */
int mrl_info_track(const float* x, const uint8_t* dones, int64_t num_envs, int32_t mode, int32_t reset_nan, float* cur,
                   double* ring, int32_t window, const int64_t* head, void* stream);

/* Reward shaping + time-out bootstrap + done flags of one step (ppo.py:153-163, reward_shaper.py:10-15):
 * rewards_tn[t] = r*scale (+ gamma*values_tn[t]*time_outs), dones_tn[t] = dones.  time_outs may be NULL. */
int mrl_rollout_write(const mrl_rollout* buf, int32_t t, const float* rewards, const uint8_t* dones,
                      const uint8_t* time_outs, double reward_scale, double gamma, void* stream);

/* ExperienceBuffer.compute_return + prepare_training + the value-normalisation stage
 * (experience.py:96-121, ppo.py:172-183): GAE(lambda) reverse scan, env-major flattening of the per-sample
 * scalars, whole-batch advantage normalisation (unbiased std, +1e-8), then value_rms.update(values) /
 * normalize / update(returns) / normalize.  value_rms may be NULL (normalize_value off).
 * stats_out (optional, 8 floats): adv mean, adv std, value mean/std after 1st update, after 2nd update. */
int mrl_gae_finish(const mrl_rollout* buf, const float* last_values, double gamma, double tau,
                   const mrl_rms* value_rms, float* stats_out, void* workspace, int64_t workspace_bytes,
                   void* stream);

/* The two halves of mrl_gae_finish, for runs where the envs are sharded over ranks: mrl_gae_scan does the scan and
 * the flattening and leaves sums_out[6] = sum of advantages, advantages^2, values, values^2, returns, returns^2 of
 * THIS rank's samples (fp64, device); the caller all-reduces them, and mrl_gae_normalize (count = samples of all
 * ranks) then normalises with the global statistics, so value_rms stays identical on every rank.
 * stats_out is required here (8 floats, device). */
int mrl_gae_scan(const mrl_rollout* buf, const float* last_values, double gamma, double tau, double* sums_out,
                 void* workspace, int64_t workspace_bytes, void* stream);
int mrl_gae_normalize(const mrl_rollout* buf, const double* sums, double count, const mrl_rms* value_rms,
                      float* stats_out, void* stream);

/* No counterpart in the reference: mrl_gae_finish for callers that cannot allocate device memory or pass a workspace
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same arguments, kernels and
 * results as mrl_gae_finish, but the scratch (per-block partial sums, sums, statistics) lives in a device array built
 * into the library: 786,688 bytes that every GPU the library runs on holds until the process ends.  At most 1,048,576
 * envs (MRL_EINVAL beyond that).  Allocates nothing.  The device that owns `stream` must be the current device.  All
 * calls on one GPU share that GPU's array: calls that overlap on one GPU (different streams or threads) race.  The first call
 * on a device looks up the array's address (which may load the library's code object on that device): make one call
 * outside any stream capture before capturing. */
int mrl_gae_finish_static(const mrl_rollout* buf, const float* last_values, double gamma, double tau,
                          const mrl_rms* value_rms, float* stats_out, void* stream);

/* Batch moments of x [rows][cols] per group of `group_rows` consecutive rows: mean (fp32) and variance with the
 * given correction (fp32), as x.mean(0) / x.var(0, correction) in RunningMeanStd.update
 * (normalizers.py:57-63).  Outputs are [rows/group_rows][cols]. */
int mrl_batch_moments(const float* x, int64_t rows, int32_t cols, int64_t group_rows, int32_t correction,
                      float* mean_out, float* var_out, void* workspace, int64_t workspace_bytes, void* stream);

/* RunningMeanStd._update_from_moments (normalizers.py:44-55) in the state's fp64, from fp32 batch moments;
 * also emits mean.float() and sqrt(var.float()+eps) [size each] for the normalising kernels. */
int mrl_rms_merge(const mrl_rms* rms, const float* batch_mean, const float* batch_var, double batch_count,
                  float* mean_f32_out, float* std_f32_out, void* stream);

/* RunningMeanStd.normalize / unnormalize (normalizers.py:65-78), element-wise, for callers outside the fused
 * path and for tests. */
int mrl_rms_normalize(const mrl_rms* rms, const float* x, float* y, int64_t rows, int32_t unnormalize, void* stream);

/* One minibatch of PPO.train_epoch up to (not including) the optimizer step (ppo.py:254-292,302-313):
 * forward of both towers on rows [row0, row0+mb) of the buffer, actor/critic/bounds/entropy losses, backward,
 * flat gradient (layout of mrl_param_offsets), policy_kl against the stored mu/sigma, and the write-back of the
 * new mu/sigma.  obs_mean/obs_std are the fp32 normaliser vectors of mrl_rms_merge (NULL = no input norm).
 * metrics_out: MRL_NMETRIC doubles = a_loss, entropy, b_loss, kl, clip_frac, c_loss (all means over the
 * minibatch), explained_var, total loss, then 4 reserved. */
int mrl_ppo_grad(const mrl_net* net, const float* params, const mrl_ppo_hp* hp, const mrl_rollout* buf,
                 int64_t row0, int64_t mb, const float* obs_mean, const float* obs_std, float clamp_lo,
                 float clamp_hi, float* grads, double* metrics_out, void* workspace, int64_t workspace_bytes,
                 void* stream);

/* clip_grad_norm_ + torch.optim.Adam/AdamW step on the flat vectors (ppo.py:294-300):
 * g = grads*grad_scale; norm = ||g||; g *= min(max_norm/(norm+1e-6), 1) if max_norm > 0; then Adam (adamw=0:
 * L2 weight decay folded into g) or AdamW (decoupled decay).  `step` is the 1-based step count.
 * gnorm_out (optional, 1 float) receives the pre-clip norm.  scratch: >= 4096 bytes. */
int mrl_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t count, double lr,
                  double beta1, double beta2, double eps, double weight_decay, int32_t adamw, int64_t step,
                  float max_norm, float grad_scale, float* gnorm_out, void* scratch, void* stream);

/* One whole optimizer step of PPO.train_epoch on one minibatch (ppo.py:254-313) for callers without a gradient
 * all-reduce between backward and optimizer.step(): exactly mrl_rms_merge (when obs_rms != NULL) + mrl_ppo_grad +
 * mrl_adam_step(grad_scale = 1) on the same arguments, and bit-identical to those three calls, but the library is free
 * to fold their launches into each other (the statistics merge rides in the first launch of the step, the gradient
 * norm is taken while the gradient vector is written).  obs_rms == NULL: no merge; obs_mean_f32 / obs_std_f32 are
 * then inputs (both NULL = no input normalisation).  count of the Adam vectors = mrl_param_count(net). */
int mrl_ppo_step(const mrl_net* net, float* params, const mrl_ppo_hp* hp, const mrl_rollout* buf, int64_t row0,
                 int64_t mb, const mrl_rms* obs_rms, const float* batch_mean, const float* batch_var,
                 double batch_count, float* obs_mean_f32, float* obs_std_f32, float clamp_lo, float clamp_hi,
                 float* grads, double* metrics_out, void* workspace, int64_t workspace_bytes, float* exp_avg,
                 float* exp_avg_sq, double lr, double beta1, double beta2, double eps, double weight_decay,
                 int32_t adamw, int64_t step, float max_norm, float* gnorm_out, void* scratch, void* stream);

/* One mini-epoch of PPO.train_epoch (the inner loop of ppo.py:254-313 over the minibatches of the rollout) for callers
 * without a gradient all-reduce: minibatch i = rows [i*mb, (i+1)*mb) of the buffer, i = 0 .. n_minibatches-1, in
 * order; exactly n_minibatches calls of mrl_ppo_step with batch_mean / batch_var = row i of batch_means / batch_vars
 * ([n_minibatches][obs_dim]), metrics_out + i*MRL_NMETRIC, step = first_step + i, gnorm_out + i (gnorm_out may be
 * NULL), and bit-identical to them; the learning rate is the same for all steps (the KL-adaptive schedule changes
 * it between mini-epochs, ppo.py:337-345).  Inside one call the library carries state from step to step (the
 * optimizer writes the parameters straight into the layouts the next step's kernels read). */
int mrl_ppo_mini_epoch(const mrl_net* net, float* params, const mrl_ppo_hp* hp, const mrl_rollout* buf, int64_t mb,
                       int64_t n_minibatches, const mrl_rms* obs_rms, const float* batch_means,
                       const float* batch_vars, double batch_count, float* obs_mean_f32, float* obs_std_f32,
                       float clamp_lo, float clamp_hi, float* grads, double* metrics_out, void* workspace,
                       int64_t workspace_bytes, float* exp_avg, float* exp_avg_sq, double lr, double beta1,
                       double beta2, double eps, double weight_decay, int32_t adamw, int64_t first_step,
                       float max_norm, float* gnorm_out, void* scratch, void* stream);

/* mrl_ppo_mini_epoch for callers WITH a gradient all-reduce between backward and optimizer.step() (the multi_gpu branch of
 * ppo.py:290-293): when the gradient of step i is complete in stream order, the library calls reduce(ctx, i, stream) on the
 * calling thread.  The callback enqueues -- on `stream`, or ordered behind it and in front of whatever is enqueued on
 * `stream` afterwards -- the in-place sum of `grads` over the ranks (it may also read metrics_out + i*MRL_NMETRIC there, which
 * is complete at that point) and returns 0; any other value aborts the call with MRL_EHIP.  The optimizer step then runs on
 * grads * grad_scale (1 / world size), taking the gradient norm from the reduced vector.  reduce == NULL: mrl_ppo_mini_epoch.
 * With one rank and grad_scale == 1 the results are those of mrl_ppo_mini_epoch bit for bit. */
typedef int (*mrl_reduce_fn)(void* ctx, int64_t step_index, void* stream);
int mrl_ppo_mini_epoch_reduce(const mrl_net* net, float* params, const mrl_ppo_hp* hp, const mrl_rollout* buf, int64_t mb,
                              int64_t n_minibatches, const mrl_rms* obs_rms, const float* batch_means,
                              const float* batch_vars, double batch_count, float* obs_mean_f32, float* obs_std_f32,
                              float clamp_lo, float clamp_hi, float* grads, double* metrics_out, void* workspace,
                              int64_t workspace_bytes, float* exp_avg, float* exp_avg_sq, double lr, double beta1,
                              double beta2, double eps, double weight_decay, int32_t adamw, int64_t first_step,
                              float max_norm, float* gnorm_out, void* scratch, void* stream, float grad_scale,
                              mrl_reduce_fn reduce, void* ctx);

/* Benchmark support, no counterpart in the reference: time the MFMA GEMM launches with HIP events on the stream
 * they are issued to.  mode 0 = off (default), 1 = every GEMM instantiation, 2 = only the instantiation that has
 * accumulated the most time so far (call after a mode-1 warm-up; with nothing timed since the last call the choice of the last
 * mode-2 call stands); every call drops the data gathered before.  While timing is on, mrl_ppo_grad issues its launches in
 * the single-stream order.
 * mrl_profile_read waits for the recorded events and reports the instantiation with the largest total time:
 * summed duration, launch count, summed algorithmic FLOPs (2*M*N*K per launch) and the kernel's name.
 * Not thread-safe; leave off outside benchmarks. */
int mrl_profile_enable(int mode);
int mrl_profile_read(double* total_ms_host, int64_t* launches_host, double* flops_host, char* name_host,
                     int32_t name_len);
/* The same for every timed instantiation, slowest first, as text: one line "<total ms>\t<launches>\t<flops>\t<name>\n" each.
 * Returns the number of bytes written (terminating NUL not counted) or a negative status. */
int mrl_profile_read_all(char* text_host, int32_t text_len);

#ifdef __cplusplus
}
#endif
#endif
