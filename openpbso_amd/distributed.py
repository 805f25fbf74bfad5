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


def shard_by_modes(modes_per_object, world_size, rank):
    """Contiguous block [lo, hi) of the objects for `rank`, balanced by the SUM OF MODES:
    the work of an object is proportional to its mode count.  Cut points are the object boundaries
    nearest to the ideal prefix sums k / world_size of the total; with equal objects this is
    shard_range.  Every rank computes the same cuts from the same list (no communication)."""
    n = len(modes_per_object)
    prefix = [0]
    for m in modes_per_object:
        prefix.append(prefix[-1] + int(m))
    total = prefix[-1]
    if total == 0:
        return shard_range(n, world_size, rank)
    cuts = [0]
    for k in range(1, world_size):
        target = total * k / world_size
        # first boundary whose prefix sum is >= target, or the one before it if that is nearer
        lo, hi = cuts[-1], n
        while lo < hi:
            mid = (lo + hi) // 2
            if prefix[mid] < target:
                lo = mid + 1
            else:
                hi = mid
        c = lo
        if c > cuts[-1] and target - prefix[c - 1] < prefix[c] - target:
            c -= 1
        cuts.append(max(c, cuts[-1]))
    cuts.append(n)
    return cuts[rank], cuts[rank + 1]


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
