"""Image tiles as a sharding axis (the north star's "samples-per-pixel and image tiles shard across the GPUs"):
mirt_render_pt_tile_dev / mirt_render_pttest_tile_dev, mirt_shard_tile, mirt_render_pt_multi_shard(MIRT_SHARD_TILES).
Streams are keyed by the pixel of the FULL image and the global sample index and samples are added one by one, so
tiles rendered separately and put side by side must equal the full-frame render BIT FOR BIT."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stitched(ra, dev, cam, tiles, spp, pttest, **kw):
    import torch

    img = torch.zeros(cam.height, cam.width, 3, device="cuda")
    rays = samples = 0
    for t in tiles:
        acc = torch.zeros(t.height, t.width, 3, device="cuda")
        st = dev.render_tile_dev(cam, t, acc.data_ptr(), spp, pttest=pttest, **kw)
        torch.cuda.synchronize()
        img[t.y0:t.y0 + t.height, t.x0:t.x0 + t.width] = acc
        rays += int(st.n_rays)
        samples += int(st.n_samples)
    return img.cpu().numpy(), rays, samples


@pytest.mark.parametrize("cfg,pttest", [("Config_Glossy128.xml", False), ("Config_Glossy128.xml", True),
                                        ("Config_Fog128.xml", False), ("Config_Textured128.xml", False)])
def test_four_tiles_stitched_equal_full_frame(gpu, ra, cfg, pttest):
    scene = ra.HostScene.load_config(os.path.join(ROOT, "Data", cfg))
    gpu.upload(scene.describe())
    cam = scene.camera()
    spp = 12
    full, st = (gpu.render_pttest if pttest else gpu.render_pt)(cam, spp, seed=7)
    tiles = [ra.shard_tile(cam.width, cam.height, 4, k) for k in range(4)]
    assert sorted((t.x0, t.y0) for t in tiles) == [(0, 0), (0, 64), (64, 0), (64, 64)]
    got, rays, samples = _stitched(ra, gpu, cam, tiles, spp, pttest, seed=7)
    assert np.array_equal(got, full), float(np.abs(got - full).max())
    assert rays == int(st.n_rays) and samples == int(st.n_samples)
    # ragged windows (no multiple of 16, so the camera pass walks them row by row) and fewer samples in flight
    ragged = [ra.Tile(0, 0, 37, 128), ra.Tile(37, 0, 91, 50), ra.Tile(37, 50, 91, 78)]
    got, rays, _ = _stitched(ra, gpu, cam, ragged, spp, pttest, seed=7, spp_per_launch=5)
    assert np.array_equal(got, full), float(np.abs(got - full).max())
    assert rays == int(st.n_rays)
    scene.close()


def test_tile_arguments_are_checked(gpu, cornell, ra):
    import torch

    gpu.upload(cornell.describe())
    cam = cornell.camera()
    acc = torch.zeros(16, 16, 3, device="cuda")
    for bad in (ra.Tile(-1, 0, 16, 16), ra.Tile(120, 0, 16, 16), ra.Tile(0, 0, 0, 16), ra.Tile(0, 113, 16, 16)):
        with pytest.raises(ra.MirtError):
            gpu.render_tile_dev(cam, bad, acc.data_ptr(), 1)
    # the tiling covers every pixel exactly once, for awkward sizes too
    for (w, h, n) in ((128, 128, 8), (50, 37, 4), (1024, 1024, 7), (3, 2, 8), (640, 480, 6)):
        cover = np.zeros((h, w), np.int32)
        for k in range(n):
            t = ra.shard_tile(w, h, n, k)
            cover[t.y0:t.y0 + t.height, t.x0:t.x0 + t.width] += 1
        assert (cover == 1).all(), (w, h, n)


def test_multi_context_tile_sharding_is_bit_equal(gpu, cornell, ra):
    """mirt_render_pt_multi_shard(MIRT_SHARD_TILES): three contexts (all on device 0 here), one tile each, gathered"""
    gpu.upload(cornell.describe())
    cam = cornell.camera()
    spp = 9
    devs = [ra.Device(device=0) for _ in range(3)]
    for d in devs:
        d.upload(cornell.describe())
    for pttest in (False, True):
        one, st1 = (gpu.render_pttest if pttest else gpu.render_pt)(cam, spp, seed=11)
        img, st = ra.render_pt_multi(devs, cam, spp, seed=11, pttest=pttest, shard=ra.SHARD_TILES)
        assert np.array_equal(img, one)
        assert st.n_rays == st1.n_rays and st.n_samples == st1.n_samples
        same, _ = ra.render_pt_multi(devs, cam, spp, seed=11, pttest=pttest, shard=ra.SHARD_SAMPLES)
        assert np.abs(same - one).max() <= 1e-5 * max(1.0, float(one.max()))
    with pytest.raises(ra.MirtError):
        ra.render_pt_multi(devs, cam, spp, shard=5)
    for d in devs:
        d.close()
