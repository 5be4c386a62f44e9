"""Multi-GPU farming of independent pair DPs: one process per GPU, pair DPs
dealt to ranks, no data-path collective.  The only collective is the one-off broadcast of
the rate-model constant block (log-sum-exp table + per-branch substitution matrices) from
rank 0 -- RCCL over xGMI on the GPU box ("nccl" backend), gloo in the CPU tests."""
import os

import numpy as np

from . import capi, hostmodel


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def constant_block(model, t_l, t_r):
    """[lse table | subMat(t_l) | subMat(t_r)] as one flat fp64 array (built on rank 0 only)."""
    return np.concatenate([hostmodel.lse_table(), np.stack(model.sub_prob(t_l)).ravel(),
                           np.stack(model.sub_prob(t_r)).ravel()])


def block_len(model):
    a, c = len(model.alphabet), model.components()
    return capi.HX_LSE_TABLE_ENTRIES + 2 * c * a * a


def split_block(model, block):
    a, c = len(model.alphabet), model.components()
    n = capi.HX_LSE_TABLE_ENTRIES
    return (block[:n], list(block[n:n + c * a * a].reshape(c, a, a)), list(block[n + c * a * a:].reshape(c, a, a)))


def broadcast_block(block, n, rank, world, device=None):
    """Broadcast rank 0's block to every rank (torch.distributed; tensor on `device` for nccl)."""
    if world == 1:
        return block
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(block if rank == 0 else np.zeros(n))
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=0)
    return t.cpu().numpy()


def pair_seed(rank, pairs_per_rank, k):
    """Seed of the k-th pair of a rank: global pair index + 1000, so the union
    over ranks is the same set of pairs a single process with world*pairs_per_rank pairs builds."""
    return 1000 + rank * pairs_per_rank + k


def max_over_ranks(seconds, world, device=None):
    if world == 1:
        return seconds
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, world, device=None):
    if world == 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(float(t.item())))


def deal(pairs, world, rank, strong):
    """(number of pair DPs of this rank, global index of its first one).  Weak scaling: `pairs` per rank.  Strong scaling
    (BASELINE configs[3] as written): `pairs` in total, contiguous slices, the first `pairs % world` ranks get one more."""
    if not strong:
        return pairs, rank * pairs
    base, extra = divmod(pairs, world)
    return base + (1 if rank < extra else 0), rank * base + min(rank, extra)
