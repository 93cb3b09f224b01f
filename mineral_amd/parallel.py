"""Cross-rank reductions of the sharded PPO path (one process per GPU, torch.distributed; backend nccl = RCCL on
the GPUs, gloo in the CPU tests).  Envs are sharded over ranks; these helpers are the only exchange steps."""

import torch


class HostStaged:
    """torch.distributed look-alike for process groups that only take CPU tensors (gloo): device tensors go through
    host memory with synchronous copies, in and out.  Used to rehearse the multi-rank path where RCCL cannot run
    (several ranks sharing one GPU); the product path hands torch.distributed itself (backend nccl) to the agent."""

    def __init__(self, dist):
        self._d = dist

    def is_initialized(self):
        return self._d.is_initialized()

    def all_reduce(self, t, op=None):
        h = t.detach().cpu()
        if op is None:
            self._d.all_reduce(h)
        else:
            self._d.all_reduce(h, op=op)
        t.copy_(h)
        return t

    def all_gather(self, outs, t):
        hs = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        self._d.all_gather(hs, t.detach().cpu())
        for o, h in zip(outs, hs):
            o.copy_(h)

    def broadcast(self, t, src=0):
        h = t.detach().cpu()
        self._d.broadcast(h, src=src)
        t.copy_(h)

    def barrier(self):
        self._d.barrier()


class Counted:
    """The collectives object handed to the agent, counting what goes through it (`calls`: name -> number of calls, `elements`:
    name -> tensor elements sent) so that benchmarks can report collectives per update."""

    def __init__(self, inner):
        self._inner = inner
        self.calls, self.elements = {}, {}

    def _note(self, name, t):
        self.calls[name] = self.calls.get(name, 0) + 1
        self.elements[name] = self.elements.get(name, 0) + int(t.numel())

    def is_initialized(self):
        return self._inner.is_initialized()

    def all_reduce(self, t, op=None):
        self._note('all_reduce', t)
        return self._inner.all_reduce(t) if op is None else self._inner.all_reduce(t, op=op)

    def all_gather(self, outs, t):
        self._note('all_gather', t)
        return self._inner.all_gather(outs, t)

    def broadcast(self, t, src=0):
        self._note('broadcast', t)
        return self._inner.broadcast(t, src=src)

    def barrier(self):
        return self._inner.barrier()


def collectives(dist):
    """The object the agent calls its collectives on: torch.distributed for nccl, the host-staged wrapper for gloo, behind a
    call counter."""
    return Counted(HostStaged(dist) if dist.get_backend() == 'gloo' else dist)


def combine_moments(mean, var, count, dist, world_size):
    """Merge per-rank batch moments of equally sized shards into the moments of their union.

    mean / var: fp32 tensors (..., features) holding this rank's batch mean and UNBIASED variance over `count`
    rows.  They are overwritten in place with the statistics of the world_size*count rows of all ranks; every
    rank computes the same bits (all_gather, then a fixed-order merge in fp64).  Returns the merged row count.
    """
    n = float(count)
    m64, v64 = mean.double(), var.double()
    gm = [torch.empty_like(m64) for _ in range(world_size)]
    gv = [torch.empty_like(v64) for _ in range(world_size)]
    dist.all_gather(gm, m64)
    dist.all_gather(gv, v64)
    gmean = torch.stack(gm).mean(0)
    m2 = torch.zeros_like(gmean)
    for m, v in zip(gm, gv):
        m2 += v * (n - 1.0) + n * (m - gmean) ** 2
    mean.copy_(gmean.float())
    var.copy_((m2 / (n * world_size - 1.0)).float())
    return n * world_size


def average_gradients(flat_grads, dist):
    """The one collective of an optimizer step: sum the flat gradient over ranks (the 1/W factor is applied inside
    the clip+Adam kernel as `grad_scale`)."""
    dist.all_reduce(flat_grads)
    return flat_grads


def exchange_update_statistics(sums, mean, var, count, dist, world_size):
    """The ONE statistics exchange of a PPO update (besides the gradient all-reduces): every rank contributes the six fp64
    sums of its GAE scan (advantage / value / return sums and sums of squares) and, when observations are normalised, the
    batch moments of each of its minibatches, in one all_gather; each rank then combines the W contributions in rank order,
    so all ranks hold the same bits.  `sums` (6,) fp64 is overwritten with the totals; `mean` / `var` (fp32, (n_mb, features),
    unbiased variance over `count` rows, or None) with the moments of the union of the ranks' rows.  Returns the merged row
    count per minibatch."""
    parts = [sums.reshape(-1)]
    if mean is not None:
        parts += [mean.double().reshape(-1), var.double().reshape(-1)]
    pack = torch.cat(parts)
    got = [torch.empty_like(pack) for _ in range(world_size)]
    dist.all_gather(got, pack)
    total = got[0][:6].clone()
    for g in got[1:]:
        total += g[:6]
    sums.copy_(total.view_as(sums))
    if mean is None:
        return float(count)
    n, k = float(count), mean.numel()
    gm = [g[6:6 + k].view_as(mean) for g in got]
    gv = [g[6 + k:6 + 2 * k].view_as(var) for g in got]
    gmean = torch.stack(gm).mean(0)
    m2 = torch.zeros_like(gmean)
    for m, v in zip(gm, gv):
        m2 += v * (n - 1.0) + n * (m - gmean) ** 2
    mean.copy_(gmean.float())
    var.copy_((m2 / (n * world_size - 1.0)).float())
    return n * world_size


GRAD_TAIL = 4   # floats behind the flat gradient that ride in its all-reduce


def pack_scalar(tail, value64):
    """Store an fp64 device scalar in two fp32 slots of the gradient's tail (value = hi + lo), so that it is summed over ranks
    by the gradient's own fp32 all-reduce: exact to 48 bits with one rank, to fp32 rounding of the sum with several."""
    hi = value64.float()
    tail[0] = hi
    tail[1] = (value64 - hi.double()).float()


def unpack_scalar(tail):
    return tail[0].double() + tail[1].double()
