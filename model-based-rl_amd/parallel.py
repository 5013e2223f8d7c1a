"""Sharding of the hot path over the GPUs of one node.

The FD path has no exchange step: every (episode, t) unit depends only on its own (s_t, a_t)
(mujoco/utils/backward.py:343-480 takes one snapshot), and the backward recurrence is serial in t but
independent across episodes (model/engine/trainer.py:57).  So rank r owns a contiguous block of episodes for
the whole horizon, model tables are replicated, and the only collective is ONE flattened all-reduce per
optimisation step on the gradient of the shared policy / action-sequence parameters (a few KB: latency
bound, so a single fused call -- `torch.distributed` backend "nccl" is RCCL over xGMI on the GPU box, "gloo"
in the CPU tests).  The reference has no counterpart (single process, trainer.py:77).
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous [lo, hi) block of n items owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def active():
    """A process group exists -- also one of a single rank (MBRL_FORCE_DIST=1 runs the collectives of the multi-GPU path
    on a one-GPU box; summing over one rank changes no bit)."""
    return dist.is_available() and dist.is_initialized()


def allreduce_sum_(tensors):
    """In-place sum over ranks of a list of tensors with ONE collective (flatten -> all_reduce -> unflatten)."""
    if not active():
        return tensors
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    o = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[o:o + n].view_as(t))
        o += n
    return tensors


def max_over_ranks(x):
    """Max of a python float over ranks (bench timing)."""
    if not active():
        return float(x)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
