"""Rank bodies of the world-size-2 tests. They run in FRESH processes that the forkserver of conftest.py forks (no
exec): the pytest process itself may already hold the GPU and must neither replace itself nor exec from that state.
Not collected by pytest (no test_ prefix)."""
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def gpu_render_rank(rank, world, port, cfg, spp, out_dir):
    """the PRODUCT with world > 1: Device.render_*_dev on this rank's shard + the collective, both sharding axes and
    both integrators; ranks share GPU 0, the collective runs over gloo"""
    _env(rank, world, port)
    import numpy as np
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    import renderer_amd as ra
    from renderer_amd.multi_gpu import render_sharded

    torch.cuda.set_device(0)
    scene = ra.HostScene.load_config(cfg)
    dev = ra.Device(device=0)
    dev.upload(scene.describe())
    cam = scene.camera()
    rays = {}
    for pttest in (False, True):
        for shard in ("samples", "tiles"):
            img, st = render_sharded(dev, cam, spp, shard=shard, seed=3, pttest=pttest)
            tot = torch.tensor([float(st.n_rays if st else 0), float(st.n_samples if st else 0)], dtype=torch.float64)
            dist.all_reduce(tot)
            name = "%s_%s" % ("pttest" if pttest else "pt", shard)
            rays[name] = [int(tot[0].item()), int(tot[1].item())]
            if rank == 0:
                np.save(os.path.join(out_dir, name + ".npy"), img.cpu().numpy())
    if rank == 0:
        json.dump(rays, open(os.path.join(out_dir, "rays.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()
    dev.close()
    scene.close()


def bench_rank(rank, world, port, argv, out_path):
    """bench.py as rank `rank` of `world` (what torch.distributed.run starts per GPU), stdout of rank 0 to out_path"""
    _env(rank, world, port)
    import runpy

    sys.argv = ["bench.py"] + list(argv)
    buf = io.StringIO()
    old = sys.stdout
    sys.stdout = buf
    try:
        runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
    finally:
        sys.stdout = old
    if rank == 0:
        open(out_path, "w").write(buf.getvalue())


def cpu_tiles_rank(rank, world, port, out_path):
    """gloo on the CPU: every rank cuts ITS tile out of the oracle's full image, gather_tiles puts them back"""
    _env(rank, world, port)
    import numpy as np
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    import orc
    import renderer_amd as ra
    from renderer_amd.multi_gpu import gather_tiles

    scene = ra.HostScene.load_config(os.path.join(ROOT, "Data", "Config_Glossy128.xml"))
    oracle = orc.OracleScene(scene.raw())
    cam = ra.make_camera(50, 37, 33)  # no multiple of 16: ragged tiles
    img, _ = oracle.render(cam, 2, seed=5, threads=2)
    t = ra.shard_tile(cam.width, cam.height, world, rank)
    mine = torch.from_numpy(np.ascontiguousarray(img[t.y0:t.y0 + t.height, t.x0:t.x0 + t.width]))
    full = gather_tiles(mine, cam.width, cam.height, dst=0)
    if rank == 0:
        assert np.array_equal(full.numpy(), img)
        np.save(out_path, full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()
    oracle.close()
    scene.close()


def run_ranks(target, world, args):
    """starts `world` fresh processes through the forkserver, waits, returns their exit codes"""
    import multiprocessing as mp

    ctx = mp.get_context("forkserver")
    port = 29000 + (os.getpid() * 7 + hash(target.__name__)) % 2000
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    codes = []
    for p in procs:
        if p.is_alive():  # a hung rank: end exactly that process
            p.terminate()
            p.join(10)
            codes.append(-999)
        else:
            codes.append(p.exitcode)
    return codes
