"""PPO agent whose two hot loops run as HIP kernels (contract of mineral/agents/ppo/ppo.py).

`play_steps` and `train_epoch` keep the reference's sequential semantics — obs statistics are merged before
every minibatch, mu/sigma are written back after every step, the KL-adaptive learning rate changes once per
mini-epoch — but every tensor operation is a call into libmineral_ppo.so.  With more than one rank the envs are
sharded; each optimizer step all-reduces the flat gradient once (RCCL via torch.distributed) and the
per-minibatch observation moments are combined across ranks so that all replicas stay bit-identical.
"""

import collections
import json
import os
import re

import torch

from ... import _hip, parallel
from ...common import normalizers
from ...common.reward_shaper import RewardShaper
from ...common.timer import Timer
from ..agent import Agent
from . import models
from .experience import ExperienceBuffer
from .utils import AdaptiveScheduler, LinearScheduler, adjust_learning_rate_cos


class PPO(Agent):
    r"""Proximal Policy Optimization (MI355X kernels)."""

    def __init__(self, full_cfg, **kwargs):
        self.network_config = full_cfg.agent.network
        self.ppo_config = full_cfg.agent.ppo
        self.num_actors = self.ppo_config.num_actors
        self.max_agent_steps = int(self.ppo_config.max_agent_steps)
        super().__init__(full_cfg, **kwargs)
        _hip.lib()   # fail now, loudly, if the HIP library is not built
        if not str(self.device).startswith('cuda'):
            raise _hip.HipLibraryError(f'the PPO hot path runs on an MI355X only (rl_device={self.device}); no CPU fallback')
        if self.ppo_config.get('dapg', None) is not None:
            raise NotImplementedError('DAPG demonstrations are outside the device PPO path')
        if 'obs' not in self.obs_space or len(self.obs_space['obs']) != 1:
            raise NotImplementedError('the device PPO path needs a flat `obs` entry (dict observations through an encoder are outside it)')
        # further keys of the observation dict are not network inputs (models.py:128-134 reads only `obs` when it is present);
        # they reach the metrics (info_keys) and are not stored in the rollout
        self._extra_obs_keys = [k for k in self.obs_space if k != 'obs']
        self.obs_space = {'obs': self.obs_space['obs']}
        dev = torch.device(self.device)
        _hip.require_current_device(dev)

        # --- Normalizers (ppo.py:29-43) ---
        rms_config = dict(eps=1e-5, with_clamp=True, initial_count=1, dtype=torch.float64, device=dev)
        if self.normalize_input:
            if not re.match(self.obs_rms_keys, 'obs'):
                raise NotImplementedError('obs_rms_keys that exclude `obs`')
            tracked = [k for k in self._extra_obs_keys if re.match(self.obs_rms_keys, k)]
            if tracked:
                raise NotImplementedError(f'running statistics of observation keys that are not network inputs ({tracked}): set '
                                          "network.obs_rms_keys to '^obs$'")
            self.obs_rms = None   # built after the model: its state is padded like the model's input rows
        else:
            self.obs_rms = None
        self.value_rms = normalizers.RunningMeanStd((1,), **rms_config)

        # --- Model (ppo.py:45-51) ---
        if self.network_config.get('encoder', None) is not None:
            raise NotImplementedError('encoders are outside the device PPO path')
        ModelCls = getattr(models, self.network_config.get('actor_critic', 'ActorCritic'))
        model_kwargs = dict(self.network_config.get('actor_critic_kwargs', {}))
        self.model = ModelCls(self.obs_space, self.action_dim, device=dev, **model_kwargs)
        print(self.model, '\n')
        if self.normalize_input:
            self.obs_rms = {'obs': normalizers.RunningMeanStd(self.obs_space['obs'], pad_to=self.model.obs_pad, **rms_config)}

        # --- Optim (ppo.py:53-60): Adam / AdamW on the flat parameter vector ---
        optim_kwargs = dict(self.ppo_config.get('optim_kwargs', {}))
        self.optim_type = self.ppo_config.optim_type
        if self.optim_type not in ('Adam', 'AdamW'):
            raise NotImplementedError(self.optim_type)
        self.init_lr = float(optim_kwargs.get('lr', 3e-4))   # ppo.py:60
        self.last_lr = self.init_lr
        # the learning rate the optimizer holds (param_group['lr']): the reference copies last_lr into it at the end of every
        # mini-epoch (ppo.py:344-345), so a 'linear' update made after the loop (ppo.py:347-348) takes effect one mini-epoch late
        self._optim_lr = self.init_lr
        self.betas = tuple(optim_kwargs.get('betas', (0.9, 0.999)))
        self.adam_eps = float(optim_kwargs.get('eps', 1e-8))
        self.weight_decay = float(optim_kwargs.get('weight_decay', 0.01 if self.optim_type == 'AdamW' else 0.0))
        if optim_kwargs.get('amsgrad', False):
            raise NotImplementedError('amsgrad')
        P = self.model.param_count
        # the flat gradient, followed by a few floats that ride in its all-reduce when the envs are sharded over ranks
        self._grads_full = torch.zeros(P + parallel.GRAD_TAIL, dtype=torch.float32, device=dev)
        self.grads = self._grads_full[:P]
        self.exp_avg = torch.zeros(P, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(P, dtype=torch.float32, device=dev)
        self.optim_step = 0

        # --- PPO params (ppo.py:62-83) ---
        c = self.ppo_config
        self.e_clip = c['e_clip']
        self.use_smooth_clamp = bool(c['use_smooth_clamp'])
        self.clip_value_loss = c['clip_value_loss']
        self.entropy_coef, self.critic_coef = c['entropy_coef'], c['critic_coef']
        self.bounds_loss_coef, self.bounds_type = c['bounds_loss_coef'], c['bounds_type']
        if self.bounds_type not in ('bound', 'reg'):
            raise NotImplementedError(self.bounds_type)
        self.gamma, self.tau = c['gamma'], c['tau']
        self.truncate_grads, self.max_grad_norm = c['truncate_grads'], c['max_grad_norm']
        self.value_bootstrap = c['value_bootstrap']
        self.normalize_advantage = c['normalize_advantage']
        self.normalize_value = c['normalize_value']
        self.horizon_len = c['horizon_len']
        self.batch_size = self.horizon_len * self.num_actors
        self.minibatch_size = c['minibatch_size']
        self.mini_epochs = c['mini_epochs']
        assert self.batch_size % self.minibatch_size == 0 or 'train' not in self.full_cfg.run

        # --- LR schedule (ppo.py:85-93) ---
        self.lr_schedule = c['lr_schedule']
        if self.lr_schedule == 'kl':
            self.kl_threshold = c['kl_threshold']
            self.scheduler = AdaptiveScheduler(self.kl_threshold, c.get('min_lr', 1e-6), c.get('max_lr', 1e-2))
        elif self.lr_schedule == 'linear':
            self.scheduler = LinearScheduler(self.init_lr, c.get('min_lr', 1e-6), c['max_agent_steps'])

        # --- Storage (ppo.py:95-107) ---
        self.obs, self.dones = None, None
        self.storage = ExperienceBuffer(self.num_actors, self.horizon_len, self.batch_size, self.minibatch_size,
                                        self.obs_space, self.action_dim, dev, self.cpu_obs_keys)
        self.reward_shaper = RewardShaper(**c['reward_shaper'])
        self.hp = _hip.PpoHp(self.e_clip, self.critic_coef, self.entropy_coef, self.bounds_loss_coef,
                             0 if self.bounds_type == 'bound' else 1, int(bool(self.clip_value_loss)), int(self.use_smooth_clamp))

        # --- Scratch: sized once for max(num_envs, minibatch) rows; nothing is allocated inside the loops ---
        L = _hip.lib()
        # the layout is not monotone in the row count (tile heights and split plans change with it): size for both uses
        ws_bytes = max(int(L.mrl_workspace_bytes(self.model.net, self.num_actors)), int(L.mrl_workspace_bytes(self.model.net, self.minibatch_size)))
        n_mb = max(1, self.batch_size // self.minibatch_size)
        mom_bytes = n_mb * 1024 * 2 * self.model.obs_pad * 8 + 65536
        self._ws = torch.empty(max(ws_bytes, mom_bytes), dtype=torch.uint8, device=dev)
        A, O = self.action_dim, self.model.obs_pad
        self._actions_clamped = torch.empty(self.num_actors, A, dtype=torch.float32, device=dev)
        self._noise = torch.empty(self.num_actors, A, dtype=torch.float32, device=dev)
        self._obs_padded = torch.zeros(self.num_actors, O, dtype=torch.float32, device=dev) if O != self.model.obs_dim else None
        self._mb_mean = torch.empty(n_mb, O, dtype=torch.float32, device=dev)
        self._mb_var = torch.empty(n_mb, O, dtype=torch.float32, device=dev)
        self._obs_mean_f = torch.empty(O, dtype=torch.float32, device=dev)
        self._obs_std_f = torch.empty(O, dtype=torch.float32, device=dev)
        self._metrics = torch.zeros(self.mini_epochs * n_mb, _hip.NMETRIC, dtype=torch.float64, device=dev)
        self._gnorm = torch.zeros(self.mini_epochs * n_mb, dtype=torch.float32, device=dev)
        self._adam_scratch = torch.empty(4096, dtype=torch.uint8, device=dev)
        self._moment_count = None
        self.noise_fn = None   # tests inject `noise_fn(step_index, shape) -> tensor`; default torch.randn on device
        self._noise_calls = 0

        # --- Timing (ppo.py:109-113), with a device sync at scope ends ---
        self.timer = Timer(sync=lambda: torch.cuda.synchronize(dev))
        self.timer.wrap('agent', self, ['play_steps', 'train_epoch'])
        self.timer.wrap('env', self.env, ['step'], sync=False)   # per env step: host time only
        self.timer_total_names = ('agent.play_steps', 'agent.train_epoch')

        # --- Multi-GPU (ppo.py:115-123): no wrapper objects; see _sync_* below ---
        self._dist = None
        # ranked runs take one library call per mini-epoch with a hook for the all-reduce (MRL_RANKED_CALLS=1: the four calls per
        # minibatch instead -- merge, gradient, all-reduce, optimizer -- with the same results bit for bit)
        self._one_call_ranked = os.environ.get('MRL_RANKED_CALLS', '0') != '1'
        if self.multi_gpu:   # also with a group of one rank: the collectives then run (through RCCL) on this rank's own data
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError('multi_gpu=True needs an initialised torch.distributed process group (backend nccl)')
            self._dist = parallel.collectives(dist)
            flat = self.model.flat
            self._dist.broadcast(flat, src=0)   # DDP's initial parameter broadcast

    # ---------------------------------------------------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(torch.device(self.device)).cuda_stream

    def _draw_noise(self):
        if self.noise_fn is not None:
            self._noise.copy_(self.noise_fn(self._noise_calls, tuple(self._noise.shape)))
        else:
            self._noise.normal_()
        self._noise_calls += 1
        return self._noise

    def model_act(self, obs_dict, t, last_values=None):
        """ppo.py:125-131 + the buffer writes of ppo.py:146-148, in one launch sequence."""
        obs = obs_dict['obs'] if isinstance(obs_dict, dict) else obs_dict
        if self.model.obs_pad != self.model.obs_dim:
            self._obs_padded[:, : self.model.obs_dim].copy_(obs)    # the padding columns stay zero
            obs = self._obs_padded
        obs = obs.contiguous()
        boot = t == self.horizon_len
        noise = self._draw_noise()    # the reference also draws (and discards) a sample for the bootstrap forward
        obs_rms = self.obs_rms['obs'].c_struct() if self.normalize_input else None
        value_rms = self.value_rms.c_struct() if self.normalize_value else None
        _hip.check(_hip.lib().mrl_policy_act(self.model.net, _hip.ptr(self.model.flat), obs_rms, value_rms, _hip.ptr(obs),
                                             None if boot else _hip.ptr(noise), t, self.storage.c,
                                             None if boot else _hip.ptr(self._actions_clamped),
                                             _hip.ptr(last_values) if boot else None, _hip.ptr(self._ws), self._ws.numel(),
                                             self._stream()), 'mrl_policy_act')
        return self._actions_clamped

    @torch.no_grad()
    def play_steps(self):
        for n in range(self.horizon_len):
            if not self.env_autoresets:
                # ppo.py:136-142: envs that do not reset themselves are reset here, one host round trip per step like there
                done_ids = torch.where(torch.as_tensor(self.dones, device=self.device).reshape(-1) != 0)[0]
                if done_ids.numel() > 0:
                    obs_reset = self._convert_obs(self.env.reset_idx(done_ids.tolist()))
                    for k, v in obs_reset.items():
                        self.obs[k][done_ids] = v
            actions = self.model_act(self.obs, n)
            obs, r, self.dones, infos = self.env.step(actions)           # ppo.py:150-151
            self.obs = self._convert_obs(obs)
            r = torch.as_tensor(r, device=self.device, dtype=torch.float32)
            self.dones = torch.as_tensor(self.dones, device=self.device)
            time_outs = infos.get('time_outs', None) if self.value_bootstrap else None
            if time_outs is not None:
                time_outs = torch.as_tensor(time_outs, device=self.device)
            self.storage.write_step(n, r, self.dones, time_outs, self.reward_shaper.scale, self.gamma)
            self.metrics.update(self.epoch, self.env, self.obs, r, self.dones, infos)
        self.metrics.flush_video(self.epoch)
        self.model_act(self.obs, self.horizon_len, last_values=self.storage.last_values)   # ppo.py:169-170
        exchange = None
        self._moment_count = None
        if self._dist is not None:
            # sharded envs: the rollout is complete, so the batch moments of this rank's rows of every minibatch can be taken
            # now and travel together with the GAE sums -- one exchange per update besides the gradient all-reduces
            if self.normalize_input:
                self._batch_moments()

            def exchange(sums):
                self._moment_count = parallel.exchange_update_statistics(
                    sums, self._mb_mean if self.normalize_input else None, self._mb_var if self.normalize_input else None,
                    self.minibatch_size, self._dist, self.rank_size)
        self.storage.compute_return(self.storage.last_values, self.gamma, self.tau,
                                    self.value_rms if self.normalize_value else None,     # ppo.py:172-183
                                    all_reduce=exchange, world_size=self.rank_size)
        self.storage.prepare_training()

    def train(self):
        obs = self.env.reset()
        self.obs = self._convert_obs(obs)
        self.dones = torch.zeros((self.num_actors,), dtype=torch.bool, device=self.device)
        while self.agent_steps < self.max_agent_steps:
            self.epoch += 1
            self.set_eval()
            self.play_steps()
            self.agent_steps += self.batch_size * (self.rank_size if self.multi_gpu else 1)   # ppo.py:195
            self.set_train()
            results = self.train_epoch()
            self.storage.data_dict = None
            if not self.multi_gpu or self.rank <= 0:
                self._log_epoch(results)
        timings = self.timer.stats(step=self.agent_steps)
        print(timings)
        self.save(os.path.join(self.ckpt_dir, 'final.pth'))

    def _log_epoch(self, results):
        metrics = {k: float(torch.mean(v).item()) for k, v in results.items() if v.dim() == 1}
        metrics.update({k: results[k].cpu().numpy() for k in ('mu', 'sigma')})
        metrics.update({'epoch': self.epoch, 'mini_epoch': self.mini_epoch, 'last_lr': self.last_lr, 'e_clip': self.e_clip})
        metrics = {f'train_stats/{k}': v for k, v in metrics.items()}
        timings = self.timer.stats(step=self.agent_steps, total_names=self.timer_total_names, reset=False)
        metrics.update({f'train_timings/{k}': v for k, v in timings.items()})
        episode_metrics = {
            'train_scores/episode_rewards': self.metrics.episode_trackers['rewards'].mean(),
            'train_scores/episode_lengths': self.metrics.episode_trackers['lengths'].mean(),
            'train_scores/num_episodes': self.metrics.num_episodes,
            **self.metrics.result(prefix='train'),
        }
        metrics.update(episode_metrics)
        self.writer.add(self.agent_steps, metrics)
        self.writer.write()
        self._checkpoint_save(metrics['train_scores/episode_rewards'])
        if self.print_every > 0 and (self.epoch + 1) % self.print_every == 0:
            print(f'Epochs: {self.epoch + 1} |', f'Agent Steps: {int(self.agent_steps):,} |',
                  f'Best: {self.best_stat if self.best_stat is not None else -float("inf"):.2f} |', 'Stats:',
                  f'ep_rewards {episode_metrics["train_scores/episode_rewards"]:.2f},',
                  f'ep_lengths {episode_metrics["train_scores/episode_lengths"]:.2f},',
                  f'last_sps {timings["lastrate"]:.2f},', f'SPS {timings["totalrate"]:.2f} |')

    # ---------------------------------------------------------------------------------------------------------------
    def _batch_moments(self):
        """Batch mean / unbiased variance of the observations of every minibatch of the finished rollout (ppo.py:258-262 takes
        them per minibatch; the rollout does not change during the update, so they are computed once)."""
        _hip.check(_hip.lib().mrl_batch_moments(_hip.ptr(self.storage.obs_store), self.batch_size, self.model.obs_pad, self.minibatch_size, 1,
                                                _hip.ptr(self._mb_mean), _hip.ptr(self._mb_var), _hip.ptr(self._ws), self._ws.numel(),
                                                self._stream()), 'mrl_batch_moments')

    def train_epoch(self):
        L = _hip.lib()
        s = self._stream()
        n_mb = len(self.storage)
        mb = self.minibatch_size
        buf = self.storage
        net, flat = self.model.net, self.model.flat
        P = flat.numel()
        count = float(mb)
        if self.normalize_input:
            # only the (sequential) Welford merge happens per step; with sharded envs the moments were taken and merged over the
            # ranks at the end of play_steps
            if self._dist is None or self._moment_count is None:
                self._batch_moments()
            else:
                count = self._moment_count
            obs_rms = self.obs_rms['obs'].c_struct()
            lo, hi = obs_rms.clamp_lo, obs_rms.clamp_hi
        else:
            obs_rms = None
            lo, hi = 0.0, 0.0
        adamw = 1 if self.optim_type == 'AdamW' else 0
        max_norm = float(self.max_grad_norm) if self.truncate_grads else 0.0
        grad_scale = 1.0 / self.rank_size if self._dist is not None else 1.0
        k = 0
        avg_kls = []
        for mini_ep in range(self.mini_epochs):
            self.mini_epoch += 1
            k0 = k
            if self._dist is None:
                # no collective between backward and optimizer.step(): the whole mini-epoch in one call (bit-identical to the
                # per-minibatch calls of the other branch)
                _hip.check(L.mrl_ppo_mini_epoch(net, _hip.ptr(flat), self.hp, buf.c, mb, n_mb, obs_rms,
                                                _hip.ptr(self._mb_mean) if obs_rms is not None else None,
                                                _hip.ptr(self._mb_var) if obs_rms is not None else None, count,
                                                _hip.ptr(self._obs_mean_f) if obs_rms is not None else None,
                                                _hip.ptr(self._obs_std_f) if obs_rms is not None else None, lo, hi,
                                                _hip.ptr(self.grads), _hip.ptr(self._metrics[k]), _hip.ptr(self._ws), self._ws.numel(),
                                                _hip.ptr(self.exp_avg), _hip.ptr(self.exp_avg_sq), self._optim_lr, self.betas[0], self.betas[1],
                                                self.adam_eps, self.weight_decay, adamw, self.optim_step + 1, max_norm,
                                                _hip.ptr(self._gnorm[k:]), _hip.ptr(self._adam_scratch), s), 'mrl_ppo_mini_epoch')
                self.optim_step += n_mb
                k += n_mb
            if self._dist is not None and self._one_call_ranked:
                # ranked: the same call with a hook between backward and optimizer.step() for the one collective per step
                # (ppo.py:292); on the last step of a mini-epoch the sum of the mini-epoch's KL values rides in the tail
                # (the reference gathers its metrics every step, ppo.py:305-311)
                failure = []

                def _reduce(_ctx, i, _stream, k0=k0):
                    try:
                        if i == n_mb - 1:
                            parallel.pack_scalar(self._grads_full[P:], self._metrics[k0:k0 + n_mb, 3].sum())
                        parallel.average_gradients(self._grads_full, self._dist)
                        return 0
                    except BaseException as e:   # never unwind through the C frames: report and let the call fail
                        failure.append(e)
                        return 1
                cb = _hip.REDUCE_FN(_reduce)
                status = L.mrl_ppo_mini_epoch_reduce(net, _hip.ptr(flat), self.hp, buf.c, mb, n_mb, obs_rms,
                                                     _hip.ptr(self._mb_mean) if obs_rms is not None else None,
                                                     _hip.ptr(self._mb_var) if obs_rms is not None else None, count,
                                                     _hip.ptr(self._obs_mean_f) if obs_rms is not None else None,
                                                     _hip.ptr(self._obs_std_f) if obs_rms is not None else None, lo, hi,
                                                     _hip.ptr(self.grads), _hip.ptr(self._metrics[k]), _hip.ptr(self._ws), self._ws.numel(),
                                                     _hip.ptr(self.exp_avg), _hip.ptr(self.exp_avg_sq), self._optim_lr, self.betas[0],
                                                     self.betas[1], self.adam_eps, self.weight_decay, adamw, self.optim_step + 1, max_norm,
                                                     _hip.ptr(self._gnorm[k:]), _hip.ptr(self._adam_scratch), s, grad_scale, cb, None)
                if failure:
                    raise failure[0]
                _hip.check(status, 'mrl_ppo_mini_epoch_reduce')
                self.optim_step += n_mb
                k += n_mb
            for i in range(n_mb if (self._dist is not None and not self._one_call_ranked) else 0):
                if self.normalize_input:
                    _hip.check(L.mrl_rms_merge(obs_rms, _hip.ptr(self._mb_mean[i]), _hip.ptr(self._mb_var[i]), count,
                                               _hip.ptr(self._obs_mean_f), _hip.ptr(self._obs_std_f), s), 'mrl_rms_merge')
                _hip.check(L.mrl_ppo_grad(net, _hip.ptr(flat), self.hp, buf.c, i * mb, mb,
                                          _hip.ptr(self._obs_mean_f) if self.normalize_input else None,
                                          _hip.ptr(self._obs_std_f) if self.normalize_input else None, lo, hi,
                                          _hip.ptr(self.grads), _hip.ptr(self._metrics[k]), _hip.ptr(self._ws), self._ws.numel(), s),
                           'mrl_ppo_grad')
                if self._dist is not None:
                    # the one collective per optimizer step (ppo.py:292); on the last step of a mini-epoch the sum of the
                    # mini-epoch's KL values rides in the tail (the reference gathers its metrics every step, ppo.py:305-311)
                    if i == n_mb - 1:
                        parallel.pack_scalar(self._grads_full[P:], self._metrics[k0:k + 1, 3].sum())
                    parallel.average_gradients(self._grads_full, self._dist)
                self.optim_step += 1
                _hip.check(L.mrl_adam_step(_hip.ptr(flat), _hip.ptr(self.grads), _hip.ptr(self.exp_avg), _hip.ptr(self.exp_avg_sq),
                                           flat.numel(), self._optim_lr, self.betas[0], self.betas[1], self.adam_eps,
                                           self.weight_decay, adamw, self.optim_step, max_norm, grad_scale,
                                           _hip.ptr(self._gnorm[k:]), _hip.ptr(self._adam_scratch), s), 'mrl_adam_step')
                k += 1
            if self._dist is not None:
                kl = parallel.unpack_scalar(self._grads_full[P:]) / (n_mb * self.rank_size)
                self._grads_full[P:].zero_()   # the tail rides along in every all-reduce: zeros stay zeros until the next mini-epoch packs
            else:
                kl = self._metrics[k0:k, 3].mean()
            avg_kls.append(kl)
            if self.lr_schedule == 'kl':                                   # ppo.py:337-338 (one host sync per mini-epoch)
                self.last_lr = self.scheduler.update(self.last_lr, kl.item())
            elif self.lr_schedule == 'cos':
                self.last_lr = adjust_learning_rate_cos(self.init_lr, mini_ep, self.mini_epochs, self.agent_steps,
                                                        self.max_agent_steps)
            self._optim_lr = self.last_lr                                  # ppo.py:344-345
        if self.lr_schedule == 'linear':
            self.last_lr = self.scheduler.update(self.agent_steps)
        m = self._metrics[:k]
        results = collections.OrderedDict()
        results['loss/total'], results['loss/actor'], results['loss/critic'] = m[:, 7], m[:, 0], m[:, 5]
        results['loss/bounds'], results['loss/entropy'] = m[:, 2], m[:, 1]
        results['clip_frac'], results['explained_var'], results['kl'] = m[:, 4], m[:, 6], m[:, 3]
        if self.truncate_grads:
            results['grad_norm/all'] = self._gnorm[:k]
        results['avg_kl'] = torch.stack(avg_kls)
        d = self.storage.data_dict
        results['mu'], results['sigma'] = d['mu'].mean(0), d['sigma'].mean(0)
        return results

    # ---------------------------------------------------------------------------------------------------------------
    def eval(self):
        """ppo.py:352-408: sample actions until 2*N episodes have finished, dump scores.json.  The policy steps write a one-step
        buffer of their own, not the training rollout."""
        self.set_eval()
        obs = self._convert_obs(self.env.reset())
        total = self.num_actors * 2
        eval_metrics = self._create_metrics(total, self.metrics_kwargs)
        old, self.metrics = self.metrics, eval_metrics
        old_storage = self.storage
        self.storage = ExperienceBuffer(self.num_actors, 1, self.num_actors, self.num_actors, self.obs_space, self.action_dim,
                                        torch.device(self.device), self.cpu_obs_keys)
        try:
            with torch.no_grad():
                while self.metrics.num_episodes < total:
                    if not self.env_autoresets:
                        raise NotImplementedError      # ppo.py:364-365
                    actions = self.model_act(obs, 0)
                    obs, r, dones, infos = self.env.step(actions)
                    obs = self._convert_obs(obs)
                    r = torch.as_tensor(r, device=self.device, dtype=torch.float32)
                    self.metrics.update(self.epoch, self.env, obs, r, torch.as_tensor(dones, device=self.device), infos)
            self.metrics.flush_video(self.epoch)
            metrics = {
                'eval_scores/num_episodes': self.metrics.num_episodes,
                'eval_scores/episode_rewards': self.metrics.episode_trackers['rewards'].mean(),
                'eval_scores/episode_lengths': self.metrics.episode_trackers['lengths'].mean(),
                **self.metrics.result(prefix='eval'),
            }
            print(metrics)
            self.writer.add(self.agent_steps, metrics)
            self.writer.write()
            scores = {'epoch': self.epoch, 'mini_epoch': self.mini_epoch, 'agent_steps': self.agent_steps,
                      'eval_scores/num_episodes': self.metrics.num_episodes,
                      'eval_scores/episode_rewards': list(self.metrics.episode_trackers['rewards'].window),
                      'eval_scores/episode_lengths': list(self.metrics.episode_trackers['lengths'].window)}
            json.dump(scores, open(os.path.join(self.logdir, 'scores.json'), 'w'), indent=4)
        finally:
            self.metrics = old
            self.storage = old_storage

    def set_train(self):
        self.model.train()

    def set_eval(self):
        self.model.eval()

    # ---- checkpoints: same top-level keys and tensor names as ppo.py:416-446 -------------------------------------------
    def _optim_state_dict(self):
        names = list(self.model._views.keys())
        m, v = self.model.views_of(self.exp_avg), self.model.views_of(self.exp_avg_sq)
        state = {}
        if self.optim_step > 0:
            for i, n in enumerate(names):
                state[i] = {'step': torch.tensor(float(self.optim_step)), 'exp_avg': m[n].clone(), 'exp_avg_sq': v[n].clone()}
        group = {'lr': self._optim_lr, 'betas': self.betas, 'eps': self.adam_eps, 'weight_decay': self.weight_decay,
                 'amsgrad': False, 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False,
                 'fused': None, 'params': list(range(len(names)))}
        return {'state': state, 'param_groups': [group]}

    def _load_optim_state_dict(self, sd):
        names = list(self.model._views.keys())
        m, v = self.model.views_of(self.exp_avg), self.model.views_of(self.exp_avg_sq)
        step = 0
        for i, n in enumerate(names):
            st = sd['state'].get(i, None)
            if st is None:
                continue
            m[n].copy_(st['exp_avg'])
            v[n].copy_(st['exp_avg_sq'])
            step = int(float(st['step']))
        self.optim_step = step
        if sd.get('param_groups'):
            self._optim_lr = float(sd['param_groups'][0]['lr'])   # last_lr itself is not checkpointed (ppo.py:416-426)

    def save(self, f):
        ckpt = {
            'epoch': self.epoch, 'mini_epoch': self.mini_epoch, 'agent_steps': self.agent_steps,
            'model': self.model.state_dict(),
            'optim': self._optim_state_dict(),
            'obs_rms': {f'obs.{k}': v for k, v in self.obs_rms['obs'].state_dict().items()} if self.normalize_input else None,
            'value_rms': self.value_rms.state_dict() if self.normalize_value else None,
        }
        torch.save(ckpt, f)

    def load(self, f, ckpt_keys=''):
        ckpt = torch.load(f, map_location=self.device, weights_only=True)
        for k in ('epoch', 'mini_epoch', 'agent_steps', 'model', 'optim', 'obs_rms', 'value_rms'):
            if not re.match(ckpt_keys, k):
                print(f'Warning: ckpt skipped loading `{k}`')
                continue
            if k == 'obs_rms':
                if self.normalize_input:
                    self.obs_rms['obs'].load_state_dict({kk.split('.', 1)[1]: vv for kk, vv in ckpt[k].items()})
            elif k == 'value_rms':
                if self.normalize_value:
                    self.value_rms.load_state_dict(ckpt[k])
            elif k == 'model':
                self.model.load_state_dict(ckpt[k])
            elif k == 'optim':
                self._load_optim_state_dict(ckpt[k])
            else:
                setattr(self, k, ckpt[k])
