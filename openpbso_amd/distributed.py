"""Multi-GPU plumbing: objects are independent, so each rank owns a contiguous
block of objects and runs its own engine; the only collective is an optional
gather of the finished audio buffers (RCCL on GPUs, gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(n_objects, world_size, rank):
    """Contiguous block [lo, hi) of `n_objects` for `rank`, sizes differ by at most one."""
    base, rem = divmod(n_objects, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_audio(local_audio, counts=None, group=None):
    """All-gather per-rank audio [n_local][samples] into [sum n_local][samples] in rank
    (= object) order.  `counts` lists every rank's n_local when the shards are ragged."""
    world = dist.get_world_size(group)
    if world == 1:
        return local_audio
    if counts is None or len(set(counts)) == 1:
        out = torch.empty((world * local_audio.shape[0],) + tuple(local_audio.shape[1:]),
                          dtype=local_audio.dtype, device=local_audio.device)
        dist.all_gather_into_tensor(out, local_audio.contiguous(), group=group)
        return out
    # ragged shards: pad every rank's block to the largest one, gather, drop the padding
    cmax = max(counts)
    padded = torch.zeros((cmax,) + tuple(local_audio.shape[1:]), dtype=local_audio.dtype, device=local_audio.device)
    padded[: local_audio.shape[0]] = local_audio
    out = torch.empty((world * cmax,) + tuple(local_audio.shape[1:]), dtype=local_audio.dtype,
                      device=local_audio.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * cmax: r * cmax + c] for r, c in enumerate(counts)], dim=0)
