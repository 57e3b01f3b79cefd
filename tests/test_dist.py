"""world_size-2 test (gloo, CPU) of the multi-GPU plumbing: contiguous sample shards + one reduce of
the per-rank radiance sums reproduce the single-process image. The per-rank renderer here is the CPU
oracle (this environment has no GPU); on the GPU box bench.py runs the same plumbing over RCCL."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, spp, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import orc
    import renderer_amd as ra
    from renderer_amd.multi_gpu import max_over_ranks, reduce_radiance_sums

    scene = ra.HostScene.load_config(os.path.join(ROOT, "Data", "Config_Glossy128.xml"))
    oracle = orc.OracleScene(scene.raw())
    cam = ra.make_camera(48, 40, 33)
    begin, count = ra.shard_samples(spp, world, rank)
    img, _ = oracle.render(cam, count, spp_begin=begin, seed=5, threads=2)
    acc = torch.from_numpy(img)
    reduce_radiance_sums(acc, dst=0)
    worst = max_over_ranks(rank + 1.5, "cpu")
    if rank == 0:
        np.save(out_path, acc.numpy())
        assert worst == world + 0.5
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sample_sharding_and_reduce(tmp_path, cornell, cornell_oracle, ra):
    spp, world = 7, 2
    out = str(tmp_path / "sum.npy")
    mp.spawn(_worker, args=(world, 29731 + os.getpid() % 200, spp, out), nprocs=world, join=True)
    got = np.load(out)
    ref, _ = cornell_oracle.render(ra.make_camera(48, 40, 33), spp, seed=5, threads=4)
    # identical up to the fp32 order of the cross-rank sum
    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, ref.max())


def test_two_rank_tile_gather(tmp_path, ra):
    """image tiles: every rank holds its own tile, one gather puts them side by side on rank 0 (no arithmetic)"""
    import dist_worker

    out = str(tmp_path / "tiles.npy")
    codes = dist_worker.run_ranks(dist_worker.cpu_tiles_rank, 2, (out,))
    assert codes == [0, 0], codes
    assert np.load(out).shape == (37, 50, 3)
