"""Data-parallel sharding of frame-pair batches (one process per GPU).

Frame pairs are independent at inference (BatchNorm in eval mode, attention per sample, ``pos_embed``
a replicated parameter), so the global batch is cut into contiguous per-rank shards with weights
replicated.  The path has exactly ONE exchange step: the all-gather of the final ``[B_local, 7, 3]``
(x, y, score) keypoint tensor -- a few KB per rank, latency-bound, one hop on the fully connected
xGMI node.  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_bounds(global_batch, rank, world):
    """Contiguous shard [lo, hi) of rank ``rank``; the first ``global_batch % world`` ranks get one extra."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, extra = divmod(global_batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_inputs(inputs, rank, world):
    """Slice every [B_global, ...] tensor of ``inputs`` (dict or sequence) to this rank's shard."""
    def cut(t):
        lo, hi = shard_bounds(t.shape[0], rank, world)
        return t[lo:hi]
    if isinstance(inputs, dict):
        return {k: cut(v) for k, v in inputs.items()}
    return [cut(v) for v in inputs]


def gather_keypoints(kp_local, global_batch=None, group=None):
    """All-gather [B_local, K, 3] keypoints into [B_global, K, 3] in rank order (ragged shards allowed)."""
    if not (dist.is_available() and dist.is_initialized()):
        return kp_local                    # no process group: single-GPU run (a world of 1 still runs the collective)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    gb = global_batch if global_batch is not None else kp_local.shape[0] * world
    sizes = [shard_bounds(gb, r, world) for r in range(world)]
    if sizes[rank][1] - sizes[rank][0] != kp_local.shape[0]:
        raise ValueError("local shard has %d samples, expected %d" % (kp_local.shape[0], sizes[rank][1] - sizes[rank][0]))
    if gb % world == 0:
        out = torch.empty((gb,) + tuple(kp_local.shape[1:]), dtype=kp_local.dtype, device=kp_local.device)
        dist.all_gather_into_tensor(out, kp_local.contiguous(), group=group)
        return out
    bmax = max(hi - lo for lo, hi in sizes)                      # ragged: pad to the largest shard
    pad = torch.zeros((bmax,) + tuple(kp_local.shape[1:]), dtype=kp_local.dtype, device=kp_local.device)
    pad[:kp_local.shape[0]] = kp_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:hi - lo] for p, (lo, hi) in zip(parts, sizes)], 0)
