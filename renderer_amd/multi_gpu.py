"""Multi-GPU plumbing of the path, one process per GPU over torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU node, "gloo" on CPU in tests). The scene is replicated; two sharding axes:

  samples   rank r renders the contiguous sample-index shard shard_samples(spp, world, r) of the WHOLE image into its
            own fp32 sum buffer; ONE reduce (sum) lands the total on rank 0: reduce_radiance_sums / render_sharded
  tiles     rank r renders ALL samples of its own image tile mirt_shard_tile(W, H, world, r); ONE gather puts the tiles
            side by side on rank 0, no arithmetic: gather_tiles / render_sharded(shard="tiles"). Bit-equal to the
            one-GPU image (streams are keyed by the pixel of the full image and the global sample index).

Closest-hit batches shard by ray and need no collective."""
import torch
import torch.distributed as dist

from . import Tile, shard_samples, shard_tile  # noqa: F401  (re-exported)


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def reduce_radiance_sums(acc, dst=0):
    """acc: [H, W, 3] float32 sums of this rank's samples; after the call rank `dst` holds the sum
    over all ranks. 1024^2 x 3 x 4 B = 12 MiB: a single ring reduce, no bucketing or overlap needed."""
    if _world()[0] > 1:
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM)
    return acc


def gather_tiles(tile_acc, width, height, dst=0):
    """tile_acc: this rank's [th, tw, 3] float32 tile (th, tw of shard_tile(width, height, world, rank)). Rank `dst`
    gets the full [height, width, 3] image, every tile copied to its place; the other ranks get None. One gather of
    equally sized messages (tiles differ by at most one 16-pixel step per axis: padded to the largest)."""
    world, rank = _world()
    if world == 1:
        return tile_acc
    tiles = [shard_tile(width, height, world, r) for r in range(world)]
    mh, mw = max(t.height for t in tiles), max(t.width for t in tiles)
    msg = torch.zeros((mh, mw, 3), dtype=tile_acc.dtype, device=tile_acc.device)
    mine = tiles[rank]
    msg[:mine.height, :mine.width] = tile_acc.reshape(mine.height, mine.width, 3)
    staged = msg.is_cuda and dist.get_backend() == "gloo"  # (gloo gathers host tensors only: the CPU rehearsal route)
    if staged:
        msg = msg.cpu()
    parts = [torch.empty_like(msg) for _ in range(world)] if rank == dst else None
    dist.gather(msg, parts, dst=dst)
    if rank != dst:
        return None
    if staged:
        parts = [p.to(tile_acc.device) for p in parts]
    img = torch.empty((height, width, 3), dtype=tile_acc.dtype, device=tile_acc.device)
    for t, part in zip(tiles, parts):
        img[t.y0:t.y0 + t.height, t.x0:t.x0 + t.width] = part[:t.height, :t.width]
    return img


def render_sharded(dev, cam, spp, shard="samples", seed=1, max_depth=20, pttest=False, dst=0):
    """This rank's share of a spp-sample render of `cam` on `dev` (a renderer_amd.Device holding the scene), combined
    on rank `dst`. Returns (image SUM [H, W, 3] on rank dst, else None; this rank's RenderStats)."""
    world, rank = _world()
    H, W = cam.height, cam.width
    if shard == "tiles":
        t = shard_tile(W, H, world, rank)
        acc = torch.zeros((max(t.height, 0), max(t.width, 0), 3), dtype=torch.float32, device="cuda")
        st = None
        if t.width > 0 and t.height > 0:
            st = dev.render_tile_dev(cam, t, acc.data_ptr(), spp, seed=seed, max_depth=max_depth, pttest=pttest)
        torch.cuda.synchronize()
        return gather_tiles(acc, W, H, dst), st
    if shard != "samples":
        raise ValueError("shard must be 'samples' or 'tiles'")
    begin, count = shard_samples(spp, world, rank)
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    st = None
    if count > 0:
        st = dev.render_tile_dev(cam, None, acc.data_ptr(), count, spp_begin=begin, seed=seed, max_depth=max_depth,
                                 pttest=pttest)
    torch.cuda.synchronize()
    reduce_radiance_sums(acc, dst)
    return (acc if rank == dst else None), st


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if _world()[0] > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
