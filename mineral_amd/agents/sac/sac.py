"""Soft Actor-Critic on the MI355X kernels: same constructor, methods, config keys and checkpoint keys as
mineral/agents/sac/sac.py; the update loop (policy sampling, twin-Q target, critic / actor / alpha gradients, clip + AdamW,
Polyak) and the replay path (n-step FIFO, ring buffer, gather) run as HIP kernels through the C ABI of
include/mineral_sac.h and include/mineral_replay.h.  No CPU fallback.

Covered configuration (cfgs/agent/DFlexAntSAC.yaml): flat `obs`, no encoder, LayerNorm+SiLU towers, squashed_normal policy
with a state-dependent sigma, EnsembleQ with two critics, no_tgt_actor, learned alpha.  Anything else raises
NotImplementedError at construction.
"""

import collections
import json
import math
import os
import re

import torch

from ... import _hip
from ...buffers import NStepReplay, ReplayBuffer
from ...common import normalizers
from ...common.reward_shaper import RewardShaper
from ...common.timer import Timer
from ..agent import Agent
from . import models


class _AdamGroup:
    """One optimizer of sac.py:71-82 (Adam / AdamW on a flat vector, clip_grad_norm_ over the group, sac.py:336-347)."""

    def __init__(self, params, optim_type, kwargs, device):
        self.params = params
        self.adamw = 1 if optim_type == 'AdamW' else 0
        self.lr = float(kwargs.get('lr', 1e-3))
        self.betas = tuple(kwargs.get('betas', (0.9, 0.999)))
        self.eps = float(kwargs.get('eps', 1e-8))
        self.weight_decay = float(kwargs.get('weight_decay', 0.01 if self.adamw else 0.0))
        n = params.numel()
        self.grads = torch.zeros(n, dtype=torch.float32, device=device)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=device)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=device)
        self.scratch = torch.empty(4096, dtype=torch.uint8, device=device)
        self.step = 0

    def update(self, max_norm, stream):
        self.step += 1
        p = self.params
        _hip.check(_hip.lib().mrl_adam_step(_hip.ptr(p), _hip.ptr(self.grads), _hip.ptr(self.exp_avg), _hip.ptr(self.exp_avg_sq), p.numel(), self.lr,
                                            self.betas[0], self.betas[1], self.eps, self.weight_decay, self.adamw, self.step,
                                            float(max_norm) if max_norm is not None else 0.0, 1.0, _hip.ptr(self.gnorm), _hip.ptr(self.scratch), stream),
                   'mrl_adam_step')
        return self.gnorm


class SAC(Agent):
    r"""Soft Actor-Critic (MI355X kernels)."""

    def __init__(self, full_cfg, **kwargs):
        self.network_config = full_cfg.agent.network
        self.sac_config = full_cfg.agent.sac
        self.num_actors = self.sac_config.num_actors
        self.max_agent_steps = int(self.sac_config.max_agent_steps)
        super().__init__(full_cfg, **kwargs)
        _hip.lib()
        if not str(self.device).startswith('cuda'):
            raise _hip.HipLibraryError(f'the SAC update loop runs on an MI355X only (rl_device={self.device}); no CPU fallback')
        _hip.require_current_device(self.device)
        if list(self.obs_space.keys()) != ['obs']:
            raise NotImplementedError('dict observations are outside the device SAC path')
        if self.network_config.get('encoder', None) is not None:
            raise NotImplementedError('encoders are outside the device SAC path')
        if self.multi_gpu:
            raise NotImplementedError('multi_gpu SAC')
        c = self.sac_config
        if not c.no_tgt_actor or c.alpha is not None:
            raise NotImplementedError('the device SAC path has no_tgt_actor=True and a learned alpha')
        if c.get('actor_detach_encoder', False):
            raise NotImplementedError('actor_detach_encoder')
        dev = torch.device(self.device)

        # --- Normalizers (sac.py:33-43): eps 1e-4, clamp, initial count = eps.  The state is kept in fp64 on the device
        # (the reference keeps fp32): the statistics agree to fp32 rounding.
        if self.normalize_input:
            if not re.match(self.obs_rms_keys, 'obs'):
                raise NotImplementedError('obs_rms_keys that exclude `obs`')
            self.obs_rms = {'obs': normalizers.RunningMeanStd(self.obs_space['obs'], eps=1e-4, with_clamp=True, initial_count='eps',
                                                              dtype=torch.float64, device=dev)}
        else:
            self.obs_rms = None

        # --- Model (sac.py:59-69) ---
        O, A = int(self.obs_space['obs'][0]), self.action_dim
        akw = dict(self.network_config.get('actor_kwargs', {}))
        ckw = dict(self.network_config.get('critic_kwargs', {}))
        if self.network_config.get('actor', 'Actor') != 'Actor' or self.network_config.get('critic', 'EnsembleQ') != 'EnsembleQ':
            raise NotImplementedError('the device SAC path implements Actor + EnsembleQ')
        dist = dict(akw.get('dist_kwargs', {}))
        a_units = [int(u) for u in dict(akw.get('mlp_kwargs', {})).get('units', [512, 256, 128])]
        c_units = [int(u) for u in dict(ckw.get('mlp_kwargs', {})).get('units', [512, 256, 128])]
        minls, maxls = dist.get('minlogstd', None), dist.get('maxlogstd', None)
        if minls is None or maxls is None:
            raise NotImplementedError('squashed_normal needs minlogstd and maxlogstd on the device SAC path')
        self.net = _hip.make_sac_net(O, A, a_units, c_units, float(minls), float(maxls),
                                     actor_kind=models.tower_kind_of(dict(akw.get('mlp_kwargs', {})), 'Actor'),
                                     critic_kind=models.tower_kind_of(dict(ckw.get('mlp_kwargs', {})), 'EnsembleQ'))
        self.obs_dim = O
        self.actor = models.Actor(self.net, O, A, device=dev, **akw)
        self.critic = models.EnsembleQ(self.net, O, A, device=dev, **ckw)
        print('Actor:', self.actor)
        print('Critic:', self.critic, '\n')

        # --- Optim (sac.py:71-82) ---
        if c.optim_type not in ('Adam', 'AdamW'):
            raise NotImplementedError(c.optim_type)
        self.actor_optim = _AdamGroup(self.actor.flat, c.optim_type, dict(c.get('actor_optim_kwargs', {})), dev)
        self.critic_optim = _AdamGroup(self.critic.flat, c.optim_type, dict(c.get('critic_optim_kwargs', {})), dev)

        # --- Target networks (sac.py:86-89) ---
        self.critic_target = self.critic.clone()
        self.actor_target = self.actor

        # --- Buffers (sac.py:91-98) ---
        self.memory = ReplayBuffer(self.obs_space, A, capacity=int(c.memory_size), device=dev)
        self.n_step_buffer = NStepReplay(self.obs_space, A, self.num_actors, c.nstep, device=dev)
        self.reward_shaper = RewardShaper(**c.reward_shaper)

        # --- Entropy (sac.py:100-108) ---
        self.target_entropy = -A * c.get('target_entropy_scalar', 1.0)
        self.log_alpha = torch.full((1,), math.log(c.init_alpha), dtype=torch.float32, device=dev)
        self.alpha_optim = _AdamGroup(self.log_alpha, c.optim_type, dict(c.get('alpha_optim_kwargs', {})), dev)

        # --- Scratch, sized once ---
        L = _hip.lib()
        # the layout is not monotone in the row count (tile heights and split plans change with it): size for both uses
        ws_bytes = max(int(L.mrl_sac_workspace_bytes(self.net, self.num_actors)), int(L.mrl_sac_workspace_bytes(self.net, int(c.batch_size))))
        self._ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        self._mean_f = torch.zeros(O, dtype=torch.float32, device=dev)
        self._std_f = torch.ones(O, dtype=torch.float32, device=dev)
        self._metrics_c = torch.zeros(int(c.mini_epochs), _hip.SAC_NMETRIC, dtype=torch.float64, device=dev)
        self._metrics_a = torch.zeros(int(c.mini_epochs), _hip.SAC_NMETRIC, dtype=torch.float64, device=dev)
        # tests inject the random draws: noise_fn(shape) ~ N(0,1), rand_fn(shape) ~ U[0,1), index_fn(high, n) -> int64 indices
        self.noise_fn = None
        self.rand_fn = None
        self.index_fn = None

        # --- Timing (sac.py:110-115), with a device sync at scope ends ---
        self.timer = Timer(sync=lambda: torch.cuda.synchronize(dev))
        self.timer.wrap('agent', self, ['explore_env', 'update_net'])
        self.timer.wrap('memory', self.memory, ['add_to_buffer'])
        self.timer.wrap('env', self.env, ['step'], sync=False)   # per env step: host time only
        self.timer_total_names = ('agent.explore_env', 'memory.add_to_buffer', 'agent.update_net')

    # ---------------------------------------------------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(torch.device(self.device)).cuda_stream

    def _noise(self, shape):
        if self.noise_fn is not None:
            return self.noise_fn(tuple(shape)).to(self.device, torch.float32).contiguous()
        return torch.randn(*shape, device=self.device)

    def _norm_ptrs(self):
        """fp32 mean / std of the current observation statistics (normalizers.py:65-68), or NULLs."""
        if not self.normalize_input:
            return None, None, 0.0, 0.0
        rms = self.obs_rms['obs']
        self._mean_f.copy_(rms.running_mean.float())
        self._std_f.copy_(torch.sqrt(rms.running_var.float() + rms.eps))
        return _hip.ptr(self._mean_f), _hip.ptr(self._std_f), rms.clamp_range[0], rms.clamp_range[1]

    def get_alpha(self, detach=True, scalar=False):
        alpha = self.log_alpha.exp()
        return float(alpha.item()) if scalar else alpha

    def get_actions(self, obs=None, z=None, sample=True, logprob=False):
        """sac.py:118-136 for sample=True: a = tanh(mu + sigma * eps) and optionally its log-prob (rows, 1)."""
        if not sample:
            raise NotImplementedError('sample=False returns the pre-tanh mean in the reference; it is not used by train / eval')
        x = (obs['obs'] if isinstance(obs, dict) else obs) if z is None else z
        x = x.contiguous()
        rows = x.shape[0]
        mean, std, lo, hi = self._norm_ptrs() if z is None else (None, None, 0.0, 0.0)   # z is already normalised (sac.py:121-124)
        actions = torch.empty(rows, self.action_dim, dtype=torch.float32, device=self.device)
        logp = torch.empty(rows, dtype=torch.float32, device=self.device) if logprob else None
        noise = self._noise((rows, self.action_dim))
        _hip.check(_hip.lib().mrl_sac_actor_sample(self.net, _hip.ptr(self.actor.flat), _hip.ptr(x), rows, mean, std, lo, hi, _hip.ptr(noise), _hip.ptr(actions),
                                                   _hip.ptr(logp), _hip.ptr(self._ws), self._ws.numel(), self._stream()), 'mrl_sac_actor_sample')
        if logprob:
            return actions, None, logp.view(-1, 1)
        return actions

    @torch.no_grad()
    def explore_env(self, env, timesteps: int, random: bool = False, sample: bool = False):
        """sac.py:138-186."""
        N, A, O, dev = self.num_actors, self.action_dim, self.obs_dim, self.device
        traj_obs = torch.empty(N, timesteps, O, dtype=torch.float32, device=dev)
        traj_next = torch.empty(N, timesteps, O, dtype=torch.float32, device=dev)
        traj_actions = torch.empty(N, timesteps, A, dtype=torch.float32, device=dev)
        traj_rewards = torch.empty(N, timesteps, dtype=torch.float32, device=dev)
        traj_dones = torch.empty(N, timesteps, dtype=torch.float32, device=dev)
        for i in range(timesteps):
            if not self.env_autoresets:
                raise NotImplementedError
            if self.normalize_input:
                self.obs_rms['obs'].update(self.obs['obs'])
            if random:
                u = self.rand_fn((N, A)).to(dev, torch.float32) if self.rand_fn is not None else torch.rand((N, A), device=dev)
                actions = u * 2.0 - 1.0
            else:
                actions = self.get_actions(obs=self.obs, sample=True)
            next_obs, rewards, dones, infos = env.step(actions)
            next_obs = self._convert_obs(next_obs)
            rewards = torch.as_tensor(rewards, device=dev, dtype=torch.float32)
            dones = torch.as_tensor(dones, device=dev)
            self.metrics.update(self.epoch, self.env, self.obs, rewards, dones, infos)
            if self.sac_config.handle_timeout:
                dones = self._handle_timeout(dones, infos)
            traj_obs[:, i] = self.obs['obs']
            traj_actions[:, i] = actions
            traj_dones[:, i] = dones
            traj_rewards[:, i] = rewards
            traj_next[:, i] = next_obs['obs']
            self.obs = next_obs
        self.metrics.flush_video(self.epoch)
        traj_rewards = self.reward_shaper(traj_rewards.reshape(N, timesteps, 1))
        traj_dones = traj_dones.reshape(N, timesteps, 1)
        data = self.n_step_buffer.add_to_buffer({'obs': traj_obs}, traj_actions, traj_rewards, {'obs': traj_next}, traj_dones)
        return data, timesteps * N

    @staticmethod
    def _handle_timeout(dones, info, timeout_keys=('time_outs', 'TimeLimit.truncated', 'truncated')):
        """agent.py:158-167: a time-out is not a termination."""
        for key in timeout_keys:
            if key in info:
                return dones * (~torch.as_tensor(info[key], device=dones.device))
        return dones

    def train(self):
        """sac.py:190-249."""
        obs = self.env.reset()
        self.obs = self._convert_obs(obs)
        self.dones = torch.ones((self.num_actors,), dtype=torch.bool, device=self.device)
        self.set_eval()
        trajectory, steps = self.explore_env(self.env, self.sac_config.warm_up, random=True)
        self.memory.add_to_buffer(trajectory)
        self.agent_steps += steps
        while self.agent_steps < self.max_agent_steps:
            self.epoch += 1
            self.set_eval()
            trajectory, steps = self.explore_env(self.env, self.sac_config.horizon_len, sample=True)
            self.agent_steps += steps
            self.memory.add_to_buffer(trajectory)
            self.set_train()
            results = self.update_net(self.memory)
            self._log_epoch(results)
        timings = self.timer.stats(step=self.agent_steps)
        print(timings)
        self.save(os.path.join(self.ckpt_dir, 'final.pth'))

    def _log_epoch(self, results):
        metrics = {k: float(torch.mean(v.float()).item()) for k, v in results.items()}
        metrics.update({'epoch': self.epoch, 'mini_epoch': self.mini_epoch, 'alpha': self.get_alpha(scalar=True)})
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

    def update_net(self, memory):
        """sac.py:251-275: per mini-epoch one sampled batch, critic update, actor + alpha update, Polyak.  Everything is
        enqueued on the stream; the per-mini-epoch results stay on the device."""
        c = self.sac_config
        L = _hip.lib()
        s = self._stream()
        B, A = int(c.batch_size), self.action_dim
        mean, std, lo, hi = self._norm_ptrs()
        gamma_n = float(c.gamma ** c.nstep)
        res = collections.defaultdict(list)
        ka = 0
        for i in range(c.mini_epochs):
            self.mini_epoch += 1
            idx = self.index_fn(memory.cur_capacity, B).to(self.device) if self.index_fn is not None else None
            obs, action, reward, next_obs, done = memory.sample_batch(B, device=self.device, indices=idx)
            batch = _hip.Transitions(obs['obs'].data_ptr(), action.data_ptr(), reward.data_ptr(), next_obs['obs'].data_ptr(), done.data_ptr(), self.obs_dim, A)
            noise_next = self._noise((B, A))
            _hip.check(L.mrl_sac_critic_grad(self.net, _hip.ptr(self.actor.flat), _hip.ptr(self.critic.flat), _hip.ptr(self.critic_target.flat),
                                             _hip.ptr(self.log_alpha), mean, std, lo, hi, batch, B, _hip.ptr(noise_next), gamma_n, int(bool(c.backup_entropy)),
                                             _hip.ptr(self.critic_optim.grads), _hip.ptr(self._metrics_c[i]), _hip.ptr(self._ws), self._ws.numel(), s),
                       'mrl_sac_critic_grad')
            gn = self.critic_optim.update(c.max_grad_norm, s)
            res['grad_norm/critic'].append(gn.clone()[0])
            if self.mini_epoch % c.update_actor_interval == 0:
                noise = self._noise((B, A))
                _hip.check(L.mrl_sac_actor_grad(self.net, _hip.ptr(self.actor.flat), _hip.ptr(self.critic.flat), _hip.ptr(self.log_alpha), mean, std, lo, hi,
                                                _hip.ptr(obs['obs']), B, _hip.ptr(noise), float(self.target_entropy), _hip.ptr(self.actor_optim.grads),
                                                _hip.ptr(self.alpha_optim.grads), _hip.ptr(self._metrics_a[ka]), _hip.ptr(self._ws), self._ws.numel(), s),
                           'mrl_sac_actor_grad')
                gn = self.actor_optim.update(c.max_grad_norm, s)
                res['grad_norm/actor'].append(gn.clone()[0])
                self.alpha_optim.update(c.max_grad_norm, s)
                ka += 1
            if self.mini_epoch % c.update_targets_interval == 0:
                _hip.check(L.mrl_polyak(_hip.ptr(self.critic_target.flat), _hip.ptr(self.critic.flat), self.critic.flat.numel(), float(c.tau), s), 'mrl_polyak')
        mc, ma = self._metrics_c[: c.mini_epochs], self._metrics_a[:ka]
        results = collections.OrderedDict()
        results['loss/critic'] = mc[:, 0].clone()
        results['grad_norm/critic'] = torch.stack(res['grad_norm/critic'])
        n = float(B)
        results['target_values/mean'] = mc[:, 1].clone()
        results['target_values/std'] = torch.sqrt(torch.clamp(mc[:, 2] - mc[:, 1] ** 2, min=0.0) * (n / (n - 1.0)))
        results['target_values/max'], results['target_values/min'] = mc[:, 3].clone(), mc[:, 4].clone()
        if ka:
            results['loss/actor'], results['loss/alpha'], results['entropy'] = ma[:, 0].clone(), ma[:, 1].clone(), ma[:, 2].clone()
            results['grad_norm/actor'] = torch.stack(res['grad_norm/actor'])
        return results

    # ---------------------------------------------------------------------------------------------------------------
    def eval(self):
        """sac.py:349-398: sample actions until 2*N episodes have finished, dump scores.json."""
        self.set_eval()
        obs = self._convert_obs(self.env.reset())
        total = self.num_actors * 2
        eval_metrics = self._create_metrics(total, self.metrics_kwargs)
        old, self.metrics = self.metrics, eval_metrics
        try:
            with torch.no_grad():
                while self.metrics.num_episodes < total:
                    if not self.env_autoresets:
                        raise NotImplementedError
                    actions = self.get_actions(obs=obs, sample=True)
                    next_obs, rewards, dones, infos = self.env.step(actions)
                    next_obs = self._convert_obs(next_obs)
                    rewards = torch.as_tensor(rewards, device=self.device, dtype=torch.float32)
                    dones = torch.as_tensor(dones, device=self.device)
                    self.metrics.update(self.epoch, self.env, obs, rewards, dones, infos)
                    obs = next_obs
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
                scores = {
                    'epoch': self.epoch, 'mini_epoch': self.mini_epoch, 'agent_steps': self.agent_steps,
                    'eval_scores/num_episodes': self.metrics.num_episodes,
                    'eval_scores/episode_rewards': list(self.metrics.episode_trackers['rewards'].window),
                    'eval_scores/episode_lengths': list(self.metrics.episode_trackers['lengths'].window),
                }
                json.dump(scores, open(os.path.join(self.logdir, 'scores.json'), 'w'), indent=4)
        finally:
            self.metrics = old

    def set_train(self):
        self.actor.train()
        self.critic.train()
        self.critic_target.train()

    def set_eval(self):
        self.actor.eval()
        self.critic.eval()
        self.critic_target.eval()

    def save(self, f):
        """sac.py:418-432: same keys; `encoder` / `encoder_target` are the empty state of the identity encoder."""
        ckpt = {
            'epoch': self.epoch, 'mini_epoch': self.mini_epoch, 'agent_steps': self.agent_steps,
            'obs_rms': {f'obs.{k}': v for k, v in self.obs_rms['obs'].state_dict().items()} if self.normalize_input else None,
            'encoder': {}, 'actor': self.actor.state_dict(), 'critic': self.critic.state_dict(), 'encoder_target': {},
            'actor_target': None, 'critic_target': self.critic_target.state_dict(), 'log_alpha': self.log_alpha[0].detach().clone(),
        }
        torch.save(ckpt, f)

    def load(self, f, ckpt_keys=''):
        """sac.py:434-453."""
        all_keys = ('epoch', 'mini_epoch', 'agent_steps', 'obs_rms', 'encoder', 'actor', 'critic', 'encoder_target', 'actor_target', 'critic_target',
                    'log_alpha')
        ckpt = torch.load(f, map_location=self.device, weights_only=True)
        for k in all_keys:
            if not re.match(ckpt_keys, k):
                print(f'Warning: ckpt skipped loading `{k}`')
                continue
            if k in ('encoder', 'encoder_target', 'actor_target') or (k == 'obs_rms' and not self.normalize_input):
                continue
            if k == 'log_alpha':
                self.log_alpha.fill_(float(ckpt[k]))
            elif k == 'obs_rms':
                self.obs_rms['obs'].load_state_dict({kk.split('.', 1)[1]: v for kk, v in ckpt[k].items()})
            elif k in ('actor', 'critic', 'critic_target'):
                getattr(self, k).load_state_dict(ckpt[k])
            else:
                setattr(self, k, ckpt[k])
