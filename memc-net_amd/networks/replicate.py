"""Replicating a network's weights across the ranks of one node.

Frame pairs are independent, so multi-GPU inference is pure data parallelism: rank 0 owns the checkpoint, every
other rank receives the weights ONCE, and after that no rank talks to another on the data path.  The only
collective is this broadcast (RCCL over xGMI on GPUs -- backend "nccl" on ROCm; gloo in the CPU tests).

xGMI is point-to-point, so a broadcast of many small tensors is latency bound per link; parameters and buffers
are therefore packed into few large flat buckets (default 256 MiB: MEMC_Net_star's 281 MB go out in 2
messages instead of 192).
"""
import torch
import torch.distributed as dist


def _flat_buckets(tensors, bucket_bytes):
    bucket, size = [], 0
    for t in tensors:
        n = t.numel() * t.element_size()
        if bucket and size + n > bucket_bytes:
            yield bucket
            bucket, size = [], 0
        bucket.append(t)
        size += n
    if bucket:
        yield bucket


def broadcast_module_state(module, src=0, bucket_bytes=256 << 20, group=None):
    """In-place: every rank ends with rank `src`'s parameters and buffers.  Returns (messages, bytes) sent.
    Tensors are grouped by dtype (a flat bucket has one dtype); order is state_dict order on every rank, which is
    identical because every rank constructs the same class."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0, 0
    by_dtype = {}
    for _name, t in module.state_dict().items():          # state_dict tensors alias the live storage
        by_dtype.setdefault(t.dtype, []).append(t)
    messages = total = 0
    for dtype in sorted(by_dtype, key=str):
        for bucket in _flat_buckets(by_dtype[dtype], bucket_bytes):
            flat = torch.cat([t.reshape(-1) for t in bucket])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            with torch.no_grad():
                for t in bucket:
                    n = t.numel()
                    t.copy_(flat[off:off + n].view_as(t))
                    off += n
            messages += 1
            total += flat.numel() * flat.element_size()
    return messages, total


def shard_pairs(rank, world, pairs_per_rank):
    """Weak scaling: pair g of the global batch lives on rank g // pairs_per_rank."""
    return range(rank * pairs_per_rank, (rank + 1) * pairs_per_rank)
