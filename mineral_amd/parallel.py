"""Cross-rank reductions of the sharded PPO path (one process per GPU, torch.distributed; backend nccl = RCCL on
the GPUs, gloo in the CPU tests).  Envs are sharded over ranks; these helpers are the only exchange steps."""

import torch


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
