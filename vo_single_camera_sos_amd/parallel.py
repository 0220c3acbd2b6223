"""Multi-GPU layout of the hot path: frame pairs are independent, so ranks shard them in
contiguous blocks with no data-path collective; the only exchange is one flat all-gather of the per-pair
result records (16 float64 each: 3x4 pose, inliers, correspondences, status, best iteration) -- a few KB,
latency-bound, so a single collective and no bucketing.  Backend "nccl" is RCCL on ROCm; the same code runs
on gloo with CPU tensors (used by the world_size-2 tests)."""
import torch

RECORD_WIDTH = 16


def shard_range(n_total, rank, world_size):
    """Contiguous block of pair indices for `rank`: sizes differ by at most one, order preserved."""
    base, extra = divmod(int(n_total), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_records(records, group=None, out=None):
    """records [B_local, 16] float64 on every rank (equal B_local) -> [world * B_local, 16] in rank order."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return records
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * records.shape[0], records.shape[1]), dtype=records.dtype, device=records.device)
    dist.all_gather_into_tensor(out, records.contiguous(), group=group)
    return out


def max_over_ranks(value, device, group=None):
    """Scalar max over ranks (bench timing contract)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
