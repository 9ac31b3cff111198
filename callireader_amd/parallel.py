"""Multi-GPU sharding of the tile path: one process per GPU, `torch.distributed` (backend 'nccl' = RCCL on ROCm,
'gloo' in CPU tests).

The reference has no inference parallelism; this is the new exchange step north_star asks for:
tiles of a batch of pages are independent through ViT -> projector -> resampler -> VQ -> de-norm.  The character
tiles (90 % of the visual work, and what varies most from page to page) form one flat list that is split contiguously
and evenly over ranks (reading order preserved); each rank runs the visual stage on its shard and ONE all-gather hands
every rank all pseudo-token embeddings (24 576 B per tile).  Pages are owned round-robin (page p -> rank p % world)
for the LLM; a page's own tiles (2.1 MB of embeddings each, needed by nobody else) are encoded by its owner.
`all_gather_rows` is generic: it also carries the 2.1 MB page-tile rows when a caller shards those too.
"""
import torch
import torch.distributed as dist


def shard_range(total, world, rank):
    """Contiguous, even split: the first `total % world` ranks get one extra element."""
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_counts(total, world):
    return [shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)]


def all_gather_rows(local, total, group=None):
    """local: (n_local, ...) rows of this rank's contiguous shard of a `total`-row tensor -> (total, ...) on every rank.
    Ragged shards are padded to the largest one so a single dist.all_gather_into_tensor moves everything."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        assert local.shape[0] == total
        return local
    counts = shard_counts(total, world)
    assert local.shape[0] == counts[dist.get_rank(group)], (local.shape, counts)
    mx = max(counts)
    tail = local.shape[1:]
    send = local
    if local.shape[0] != mx:
        send = torch.zeros((mx,) + tuple(tail), dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    if send.is_cuda and dist.get_backend(group) == 'gloo':
        # functional fallback for single-GPU test boxes (ranks sharing one device): gloo moves host buffers
        host = torch.empty((world * mx,) + tuple(tail), dtype=local.dtype)
        dist.all_gather_into_tensor(host, send.contiguous().cpu(), group=group)
        recv = host.to(local.device)
    else:
        recv = torch.empty((world * mx,) + tuple(tail), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    if all(c == mx for c in counts):
        return recv
    return torch.cat([recv[r * mx:r * mx + c] for r, c in enumerate(counts)], dim=0)


def all_gather_rows_async(local, total, group=None):
    """Start the gather and return `finish() -> (total, ...) tensor`: on RCCL the collective runs on the communicator's
    own stream, so kernels enqueued before finish() (the page tiles' ViT, in bench.py) overlap the transfer over xGMI.
    World size 1 and the gloo host fallback complete immediately."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 or (local.is_cuda and dist.get_backend(group) == 'gloo'):
        out = all_gather_rows(local, total, group)
        return lambda: out
    counts = shard_counts(total, world)
    assert local.shape[0] == counts[dist.get_rank(group)], (local.shape, counts)
    mx = max(counts)
    tail = local.shape[1:]
    send = local
    if local.shape[0] != mx:
        send = torch.zeros((mx,) + tuple(tail), dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    recv = torch.empty((world * mx,) + tuple(tail), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(recv, send.contiguous(), group=group, async_op=True)

    def finish():
        work.wait()                                   # the current stream waits for the collective; the host does not block
        if all(c == mx for c in counts):
            return recv
        return torch.cat([recv[r * mx:r * mx + c] for r, c in enumerate(counts)], dim=0)
    return finish


def owned_pages(n_pages, world, rank):
    return list(range(rank, n_pages, world))
