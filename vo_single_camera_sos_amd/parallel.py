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


def shard_rows_max(n_total, world_size):
    """Rows of the largest shard of shard_range(): what every rank pads its block to for the flat gather."""
    return -(-int(n_total) // int(world_size))


def gather_records(records, group=None, out=None, n_total=None, pad=None):
    """records [B_local, 16] float64 on every rank -> all records in rank (= global pair) order.

    Equal shards (n_total None): one all_gather_into_tensor into [world * B_local, 16]; the ranks' B_local are
    checked against each other first when `out` is not given (a mismatched count is a hang on RCCL), so the
    steady-state caller passes a preallocated `out` and pays for no extra collective.

    Uneven shards (n_total = global number of pairs, blocks by shard_range()): every rank pads its block to
    ceil(n_total / world) rows (`pad`: optional preallocated [ceil, 16] buffer), the same single collective runs
    on equal counts, and the padding rows are stripped -> [n_total, 16]."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return records
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    width = records.shape[1]
    if n_total is None:
        if out is None:
            lohi = torch.tensor([records.shape[0], -records.shape[0]], dtype=torch.int64, device=records.device)
            dist.all_reduce(lohi, op=dist.ReduceOp.MAX, group=group)
            lo, hi = -int(lohi[1].item()), int(lohi[0].item())
            if lo != hi:
                raise ValueError("gather_records: ranks hold %d..%d records; pass n_total for uneven shards" % (lo, hi))
            out = torch.empty((world * records.shape[0], width), dtype=records.dtype, device=records.device)
        elif out.shape[0] != world * records.shape[0]:
            raise ValueError("gather_records: out has %d rows, expected world * B_local = %d"
                             % (out.shape[0], world * records.shape[0]))
        dist.all_gather_into_tensor(out, records.contiguous(), group=group)
        return out
    n_total = int(n_total)
    lo, hi = shard_range(n_total, rank, world)
    if records.shape[0] != hi - lo:
        raise ValueError("gather_records: rank %d holds %d records, its shard of %d pairs over %d ranks has %d"
                         % (rank, records.shape[0], n_total, world, hi - lo))
    rows = shard_rows_max(n_total, world)
    if pad is None:
        pad = torch.zeros((rows, width), dtype=records.dtype, device=records.device)
    pad[: hi - lo].copy_(records)
    flat = torch.empty((world * rows, width), dtype=records.dtype, device=records.device)
    dist.all_gather_into_tensor(flat, pad, group=group)
    if n_total == world * rows:
        res = flat
    else:
        res = torch.cat([flat[r * rows: r * rows + (b - a)] for r in range(world)
                         for a, b in [shard_range(n_total, r, world)]])
    if out is not None:
        out.copy_(res)
        return out
    return res


def max_over_ranks(value, device, group=None):
    """Scalar max over ranks (bench timing contract)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
