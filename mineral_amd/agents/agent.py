"""Agent base class: the constructor / method contract of mineral/agents/agent.py:13-167."""

import os
import re
from contextlib import contextmanager

import numpy as np
import torch

from ..common.metrics import Metrics
from ..common.writer import JsonlWriter, TensorboardWriter, WandbWriter, Writer


class Agent:
    def __init__(self, full_cfg, logdir=None, accelerator=None, datasets=None, env=None):
        # subclasses set these before calling up (agent.py:15-16)
        assert getattr(self, 'network_config', False)
        assert getattr(self, 'num_actors', False)
        self.full_cfg = full_cfg
        self.logdir = logdir or '.'

        # --- Device (agent.py:21-31).  One process per GPU; rank / world come from the launcher's env. ---
        self.rank = -1
        self.device = full_cfg.rl_device
        self.multi_gpu = bool(full_cfg.multi_gpu)
        self.rank_size = 1
        self.accelerator = accelerator
        if self.multi_gpu:
            self.rank = int(os.getenv('LOCAL_RANK', '0'))
            self.rank_size = int(os.getenv('WORLD_SIZE', '1'))

        self.datasets = datasets
        self.env = env
        self.action_dim = self.env.action_space.shape[0]
        self.env_autoresets = full_cfg.task.get('env_autoresets', True)

        self.normalize_input = self.network_config.get('normalize_input', False)
        self.obs_rms_keys = re.compile(self.network_config.get('obs_rms_keys', ''))
        self.cpu_obs_keys = re.compile(self.network_config.get('cpu_obs_keys', '$^'))
        self.observation_space = self.env.observation_space
        try:
            obs_space = {k: v.shape for k, v in self.observation_space.spaces.items()}
        except AttributeError:
            obs_space = {'obs': self.observation_space.shape}
        self.obs_space = obs_space

        self.tracker_len = full_cfg.agent.get('tracker_len', 100)
        self.metrics_kwargs = full_cfg.agent.get('metrics_kwargs', {})
        self.env_render = full_cfg.env_render
        self.metrics = self._create_metrics(self.tracker_len, self.metrics_kwargs)

        self.ckpt_dir = os.path.join(self.logdir, 'ckpt')
        os.makedirs(self.ckpt_dir, exist_ok=True)
        self.tb_dir = os.path.join(self.logdir, 'tb')
        os.makedirs(self.tb_dir, exist_ok=True)
        writers = [WandbWriter(), TensorboardWriter(self.tb_dir, self.full_cfg.to_container() if hasattr(self.full_cfg, 'to_container') else None), JsonlWriter(os.path.join(self.logdir, 'metrics.jsonl'))]
        self.writer = Writer(writers)

        self.print_every = full_cfg.agent.get('print_every', -1)
        self.ckpt_every = full_cfg.agent.get('ckpt_every', -1)
        self.eval_every = full_cfg.agent.get('eval_every', -1)
        self.best_stat = None

        self.epoch = -1
        self.mini_epoch = -1
        self.agent_steps = 0

    def get_actions(self, obs, sample: bool = True):
        raise NotImplementedError          # agent.py:83-84

    def explore_env(self, env, timesteps: int, random: bool = False, sample: bool = False):
        raise NotImplementedError          # agent.py:86-87

    def train(self):
        raise NotImplementedError

    def eval(self):
        raise NotImplementedError

    def set_train(self):
        raise NotImplementedError

    def set_eval(self):
        raise NotImplementedError

    def save(self, f):
        raise NotImplementedError

    def load(self, f, ckpt_keys=''):
        raise NotImplementedError

    def _create_metrics(self, tracker_len, metrics_kwargs):
        scores = {'rewards': torch.zeros(self.num_actors, dtype=torch.float32, device=self.device),
                  'lengths': torch.zeros(self.num_actors, dtype=torch.int64, device=self.device)}
        return Metrics(scores, tracker_len, env_render=self.env_render, **{k: v for k, v in dict(metrics_kwargs).items()
                                                                           if k in ('info_keys', 'save_video_every',
                                                                                    'save_video_consecutive')})

    @contextmanager
    def _as_metrics(self, new_metrics):
        """agent.py:115-122: swap `self.metrics` for the duration of a block (evaluation)."""
        old = self.metrics
        self.metrics = new_metrics
        try:
            yield
        finally:
            self.metrics = old

    def _checkpoint_save(self, stat, stat_name='rewards', higher_better=True, sep=''):
        """agent.py:124-142: periodic + `latest.pth` symlink + one rolling best checkpoint."""
        if self.ckpt_every > 0 and (self.epoch + 1) % self.ckpt_every == 0:
            name = f'epochs{sep}{self.epoch + 1}_steps{sep}{int(self.agent_steps / 1000)}k_{stat_name}{sep}{stat:.2f}'
            self.save(os.path.join(self.ckpt_dir, name + '.pth'))
            latest = os.path.join(self.ckpt_dir, 'latest.pth')
            if os.path.lexists(latest):
                os.unlink(latest)
            os.symlink(name + '.pth', latest)
        if self.best_stat is None:
            better = True
        else:
            better = stat > self.best_stat if higher_better else stat < self.best_stat
        if better:
            print(f'saving current best_{stat_name}{sep}{stat:.2f}')
            if self.best_stat is not None:
                prev = os.path.join(self.ckpt_dir, f'best_{stat_name}{sep}{self.best_stat:.2f}.pth')
                if os.path.exists(prev):
                    os.remove(prev)
            self.best_stat = stat
            self.save(os.path.join(self.ckpt_dir, f'best_{stat_name}{sep}{self.best_stat:.2f}.pth'))

    def _convert_obs(self, obs):
        """agent.py:144-156: wrap in a dict, move ndarrays to the device, never alias the env's dict."""
        if not isinstance(obs, dict):
            obs = {'obs': obs}
        out = {}
        for k, v in obs.items():
            if isinstance(v, np.ndarray):   # keys matching `cpu_obs_keys` stay on the host (agent.py:152)
                out[k] = torch.tensor(v, device='cpu' if self.cpu_obs_keys.match(k) else self.device)
            else:
                out[k] = v
        return out

    @staticmethod
    def _handle_timeout(dones, info, timeout_keys=('time_outs', 'TimeLimit.truncated', 'truncated')):
        """agent.py:158-167: an env that stopped on its time limit did not reach a terminal state (the first key found decides)."""
        timeout_envs = None
        for key in timeout_keys:
            if key in info:
                timeout_envs = info[key]
                break
        if timeout_envs is not None:
            dones = dones * (~timeout_envs)
        return dones
