"""Multi-GPU plumbing of the path: one process per GPU, samples sharded
contiguously by global sample index, per-rank fp32 radiance SUMS combined by ONE reduce to rank 0
(RCCL over xGMI with backend "nccl"; "gloo" on CPU in tests). Closest-hit batches shard by ray and
need no collective."""
import torch
import torch.distributed as dist

from . import shard_samples  # noqa: F401  (re-exported)


def reduce_radiance_sums(acc, dst=0):
    """acc: [H, W, 3] float32 sums of this rank's samples; after the call rank `dst` holds the sum
    over all ranks. 1024^2 x 3 x 4 B = 12 MiB: a single ring reduce, no bucketing or overlap needed."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM)
    return acc


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
