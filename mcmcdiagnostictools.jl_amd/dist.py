"""Multi-GPU: the params axis shards embarrassingly.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU for tests).
Every parameter is independent (the reference loops over `_eachparam`, src/ess_rhat.jl:517), so a
rank computes its contiguous block of parameters with no exchange; the only collective is the
final all-gather of the small per-parameter outputs (P/N elements per rank per output), which is
latency-bound over xGMI.
"""
try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None


def shard_bounds(nparams, rank, world):
    """contiguous block [lo, hi) of the params axis owned by `rank`; sizes differ by at most 1"""
    q, r = divmod(int(nparams), int(world))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_params(local, nparams, group=None):
    """all-gather per-parameter outputs of unequal shard sizes into the full [nparams] vector"""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    q, r = divmod(int(nparams), world)
    cap = q + (1 if r else 0)
    pad = torch.zeros(cap, dtype=local.dtype, device=local.device)
    lo, hi = shard_bounds(nparams, rank, world)
    pad[: hi - lo] = local.reshape(-1)
    buf = torch.empty(world * cap, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    parts = []
    for rk in range(world):
        l2, h2 = shard_bounds(nparams, rk, world)
        parts.append(buf[rk * cap: rk * cap + (h2 - l2)])
    return torch.cat(parts)


def gather_params_packed(locals_, nparams, group=None):
    """the same for SEVERAL per-parameter outputs at once: ONE all-gather of a packed [K, cap] block per rank instead of K
    latency-bound collectives (K = 4 for bulk+tail); returns the K full [nparams] vectors"""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    q, r = divmod(int(nparams), world)
    cap = q + (1 if r else 0)
    K = len(locals_)
    first = torch.as_tensor(locals_[0])
    pad = torch.zeros((K, cap), dtype=first.dtype, device=first.device)
    lo, hi = shard_bounds(nparams, rank, world)
    for k, o in enumerate(locals_):
        pad[k, : hi - lo] = torch.as_tensor(o).reshape(-1)
    buf = torch.empty((world, K, cap), dtype=first.dtype, device=first.device)
    dist.all_gather_into_tensor(buf.view(-1), pad.view(-1), group=group)
    outs = []
    for k in range(K):
        parts = []
        for rk in range(world):
            l2, h2 = shard_bounds(nparams, rk, world)
            parts.append(buf[rk, k, : h2 - l2])
        outs.append(torch.cat(parts))
    return tuple(outs)


def ess_rhat_sharded(local_samples, nparams_total, *, compute=None, group=None, **kwargs):
    """bulk+tail diagnostics of a params-sharded array.

    local_samples: this rank's block, shape (draws, chains, hi - lo) with (lo, hi) =
    shard_bounds(nparams_total, rank, world).  Returns the four full-length vectors
    (ess_bulk, ess_tail, rhat_bulk, rhat_tail) on every rank: one packed all-gather.
    `compute` defaults to the HIP path (api.bulk_tail); tests inject a checker.
    """
    if compute is None:
        from .api import bulk_tail as compute
    outs = compute(local_samples, **kwargs)
    return gather_params_packed([torch.as_tensor(o) for o in outs], nparams_total, group)
