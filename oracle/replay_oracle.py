"""CPU restatement of the reference's replay path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(mineral_amd/) never does.  Pinned by tests/golden/g11_replay.npz, which tests/golden/make_golden_replay.py produced
by running the reference's own buffers here.

Follows mineral/buffers/simple_replay.py:19-85 (ring buffer) and mineral/buffers/nstep_replay.py:10-94 (n-step FIFO
and masked discounted return), written index-wise with numpy so that the device kernels can be checked element by
element: no torch.cat queue shifting, no boolean-mask gymnastics.
"""

import numpy as np


def torch_sum_order(terms):
    """fp32 sum of up to 7 values in the association torch's CPU reduction over a short non-contiguous dimension uses
    (observed on the reference's `.sum(1)`, nstep_replay.py:93): the first four values seed four accumulators, every
    further value is added to the first accumulator, and the accumulators are combined left to right.  For fewer than
    five values this is the plain left-to-right sum."""
    f = np.float32
    n = len(terms)
    if n > 7:
        raise NotImplementedError('summation order of the reference is pinned for nstep <= 7 only')
    if n < 4:
        t = f(terms[0])
        for j in range(1, n):
            t = f(t + terms[j])
        return t
    acc = [f(terms[k]) for k in range(4)]
    for j in range(4, n):
        acc[0] = f(acc[0] + terms[j])
    return f(f(f(acc[0] + acc[1]) + acc[2]) + acc[3])


class NStepOracle:
    """nstep_replay.py:10-62.  State: the last `nstep` transitions of every env (a sliding window over all calls)."""

    def __init__(self, num_envs, obs_dim, act_dim, nstep=3, gamma=0.99):
        self.N, self.O, self.A, self.nstep = num_envs, obs_dim, act_dim, nstep
        # gamma ** i is evaluated in Python floats and rounded to fp32 once (nstep_replay.py:28)
        self.gamma_array = np.array([gamma ** i for i in range(nstep)], dtype=np.float32)
        self.count = 0
        self.win_obs = np.zeros((num_envs, nstep, obs_dim), np.float32)
        self.win_act = np.zeros((num_envs, nstep, act_dim), np.float32)
        self.win_rew = np.zeros((num_envs, nstep), np.float32)
        self.win_nxt = np.zeros((num_envs, nstep, obs_dim), np.float32)
        self.win_done = np.zeros((num_envs, nstep), np.float32)

    def add_to_buffer(self, obs, actions, rewards, next_obs, dones):
        """obs/next_obs (N,T,O), actions (N,T,A), rewards/dones (N,T,1).  Returns flat arrays, step-major (the emitted
        steps concatenated, N rows each) like the torch.cat of nstep_replay.py:59-62; nstep == 1 passes through."""
        if self.nstep == 1:
            return obs, actions, rewards, next_obs, dones
        T = dones.shape[1]
        o_l, a_l, r_l, n_l, d_l = [], [], [], [], []
        for i in range(T):
            for win, new in ((self.win_obs, obs[:, i]), (self.win_act, actions[:, i]), (self.win_rew, rewards[:, i, 0]),
                             (self.win_nxt, next_obs[:, i]), (self.win_done, dones[:, i, 0])):
                win[:, :-1] = win[:, 1:].copy()          # fifo_shift (nstep_replay.py:66-68)
                win[:, -1] = new
            self.count += 1
            if self.count < self.nstep:
                continue
            rew = np.zeros(self.N, np.float32)
            nxt = np.empty((self.N, self.O), np.float32)
            done = np.empty(self.N, np.float32)
            for n in range(self.N):
                # compute_nstep_return (nstep_replay.py:71-94): steps up to and including the first done count
                hits = np.nonzero(self.win_done[n])[0]
                last = int(hits[0]) if len(hits) else self.nstep - 1
                terms = [np.float32(np.float32(self.win_rew[n, j] * self.gamma_array[j]) * np.float32(1.0 if j <= last else 0.0))
                         for j in range(self.nstep)]
                rew[n] = torch_sum_order(terms)
                nxt[n] = self.win_nxt[n, last]
                done[n] = 1.0 if len(hits) else self.win_done[n, -1]
            o_l.append(self.win_obs[:, 0].copy())
            a_l.append(self.win_act[:, 0].copy())
            r_l.append(rew[:, None])
            n_l.append(nxt)
            d_l.append(done[:, None])
        if not o_l:
            raise ValueError('no transition emitted: the reference fails here too (torch.cat of an empty list)')
        return (np.concatenate(o_l), np.concatenate(a_l), np.concatenate(r_l), np.concatenate(n_l), np.concatenate(d_l))


class RingOracle:
    """simple_replay.py:19-85: ring buffer with the reference's lazy wrap of `next_p`."""

    def __init__(self, obs_dim, act_dim, capacity):
        self.capacity = int(capacity)
        self.next_p, self.cur_capacity, self.if_full = 0, 0, False
        self.obs = np.zeros((capacity, obs_dim), np.float32)
        self.action = np.zeros((capacity, act_dim), np.float32)
        self.reward = np.zeros((capacity, 1), np.float32)
        self.next_obs = np.zeros((capacity, obs_dim), np.float32)
        self.done = np.zeros((capacity, 1), np.uint8)

    def add_to_buffer(self, obs, actions, rewards, next_obs, dones):
        src = (obs.reshape(-1, self.obs.shape[1]), actions.reshape(-1, self.action.shape[1]), rewards.reshape(-1, 1),
               next_obs.reshape(-1, self.obs.shape[1]), (dones.reshape(-1, 1) != 0).astype(np.uint8))
        dst = (self.obs, self.action, self.reward, self.next_obs, self.done)
        n = src[2].shape[0]
        p = self.next_p + n
        if p > self.capacity:
            self.if_full = True
            head = self.capacity - self.next_p
            p -= self.capacity
            for d, s in zip(dst, src):
                d[self.next_p:self.capacity] = s[:head]
                d[0:p] = s[n - p:]                       # the LAST p rows (simple_replay.py:54-61)
        else:
            for d, s in zip(dst, src):
                d[self.next_p:p] = s
        self.next_p = p
        self.cur_capacity = self.capacity if self.if_full else self.next_p

    def gather(self, indices):
        """sample_batch with the index draw injected (simple_replay.py:73-85); done comes back as float."""
        return (self.obs[indices], self.action[indices], self.reward[indices], self.next_obs[indices],
                self.done[indices].astype(np.float32))
