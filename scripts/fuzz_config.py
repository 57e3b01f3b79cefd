"""One-off: equivalences the render entry points promise, on random scenes with random parameters (not part of the test
suite). Scenes: scripts/fuzz_media.py's generator (media, glass, emitters ...). For each: a random image size, a random
guillotine partition into tiles, random samples-per-launch, both integrators:
  (a) tiles rendered one by one and put side by side == the full frame, bit for bit, ray and sample counts equal;
  (b) the full frame with a random samples-per-launch (chained wavefronts) == the default, bit for bit;
  (c) two sample ranges [0, a) + [a, a + b) added == [0, a + b) up to the one float addition per pixel that differs.
  (d) MULTI=1: two or three contexts, tiles gathered == the frame bit for bit, sample shards reduced == the frame.
usage: [RICH=1] [MULTI=1] fuzz_config.py first_seed n_seeds"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
src = open(os.path.join(ROOT, "scripts", "fuzz_media.py")).read().split("dev = ra.Device(device=0)")[0]
ns = {"__file__": os.path.join(ROOT, "scripts", "fuzz_media.py")}
exec(compile(src, "fuzz_media_head", "exec"), ns)
import renderer_amd as ra


def partition(rng, x0, y0, w, h, depth):
    if depth == 0 or w * h < 8 or rng.random() < 0.2:
        return [ra.Tile(x0, y0, w, h)]
    if (rng.random() < 0.5 and w > 1) or h == 1:
        c = int(rng.integers(1, w))
        return partition(rng, x0, y0, c, h, depth - 1) + partition(rng, x0 + c, y0, w - c, h, depth - 1)
    c = int(rng.integers(1, h))
    return partition(rng, x0, y0, w, c, depth - 1) + partition(rng, x0, y0 + c, w, h - c, depth - 1)


dev = ra.Device(device=0)
more = [ra.Device(device=0) for _ in range(2)] if os.environ.get("MULTI") else []  # (d): three contexts on this GPU
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[1]) + int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    scene, kinds = ns["build"](rng)
    dev.upload(scene.describe())
    for m in more: m.upload(scene.describe())
    W, H = int(rng.integers(3, 90)), int(rng.integers(3, 90))
    w3 = rng.normal(size=3); w3 /= np.linalg.norm(w3)
    pos = tuple(rng.uniform(-1, 1, 3).tolist()) if rng.random() < 0.35 else tuple((w3 * 3.2).tolist())
    cam = ra.make_camera(W, H, 45, position=pos, focus=tuple(rng.uniform(-0.3, 0.3, 3).tolist()))
    spp = int(rng.integers(1, 14)); s0 = int(rng.integers(0, 50)); rseed = int(rng.integers(1, 1000))
    tiles = partition(rng, 0, 0, W, H, 3)
    line = "seed %d kinds %s %dx%d spp %d from %d, %d tiles:" % (seed, "".join(map(str, kinds)), W, H, spp, s0, len(tiles))
    for pttest in (False, True):
        render = dev.render_pttest if pttest else dev.render_pt
        full, st = render(cam, spp, spp_begin=s0, seed=rseed)
        img = torch.zeros(H, W, 3, device="cuda"); rays = samples = 0
        for t in tiles:
            acc = torch.zeros(t.height, t.width, 3, device="cuda")
            s = dev.render_tile_dev(cam, t, acc.data_ptr(), spp, spp_begin=s0, seed=rseed, pttest=pttest,
                                    spp_per_launch=int(rng.integers(0, spp + 2)))
            torch.cuda.synchronize()
            img[t.y0:t.y0 + t.height, t.x0:t.x0 + t.width] = acc
            rays += int(s.n_rays); samples += int(s.n_samples)
        a_ok = np.array_equal(img.cpu().numpy(), full, equal_nan=True) and rays == int(st.n_rays) and samples == int(st.n_samples)
        spl = int(rng.integers(1, spp + 1))
        ch, st2 = render(cam, spp, spp_begin=s0, seed=rseed, spp_per_launch=spl)
        b_ok = np.array_equal(ch, full, equal_nan=True) and int(st2.n_rays) == int(st.n_rays)
        a = int(rng.integers(0, spp + 1))
        p1 = render(cam, a, spp_begin=s0, seed=rseed)[0] if a else np.zeros_like(full)
        p2 = render(cam, spp - a, spp_begin=s0 + a, seed=rseed)[0] if spp - a else np.zeros_like(full)
        with np.errstate(invalid="ignore"):
            c_ok = np.allclose(p1 + p2, full, rtol=2e-5, atol=1e-6 * max(1.0, float(np.nanmax(np.abs(full)))), equal_nan=True)
        line += " %s a %s b(spl %d) %s c(%d+%d) %s |" % ("pttest" if pttest else "pt", "ok" if a_ok else "DIFFER", spl, "ok" if b_ok else "DIFFER",
                                                       a, spp - a, "ok" if c_ok else "DIFFER")
        bad += (not a_ok) + (not b_ok) + (not c_ok)
        if more:  # (d) several contexts: tiles gathered == the frame bit for bit; sample shards reduced == the frame to the additions
            k = int(rng.integers(2, 4))
            mt, s3 = ra.render_pt_multi([dev] + more[:k - 1], cam, spp, spp_begin=s0, seed=rseed, pttest=pttest, shard=ra.SHARD_TILES)
            d1 = np.array_equal(mt, full, equal_nan=True) and int(s3.n_rays) == int(st.n_rays)
            ms, s4 = ra.render_pt_multi([dev] + more[:k - 1], cam, spp, spp_begin=s0, seed=rseed, pttest=pttest)
            with np.errstate(invalid="ignore"):
                d2 = np.allclose(ms, full, rtol=2e-5, atol=1e-6 * max(1.0, float(np.nanmax(np.abs(full)))), equal_nan=True) and int(s4.n_rays) == int(st.n_rays)
            line += " d(%d ctx) %s %s |" % (k, "ok" if d1 else "DIFFER", "ok" if d2 else "DIFFER")
            bad += (not d1) + (not d2)
    print(line, flush=True)
    scene.close()
print("config fuzz: %d failed equivalences" % bad)
