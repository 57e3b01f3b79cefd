"""One-off: the closest-hit entry points against each other on random soups and ragged batch sizes (no oracle needed):
mirt_intersect (host buffers, staged), mirt_intersect_dev (device buffers, caller's stream), mirt_intersect_noalloc_dev
(static scratch, <= 4 Mi rays), the instrumented mirt_intersect_count_dev, and a batch split in two at a random place:
ids and t bit-equal in all three modes; mirt_visibility against mirt_visibility_dev likewise.
usage: fuzz_entry.py first_seed n_seeds"""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import renderer_amd as ra
import test_gpu_fuzz as F
dev = ra.Device(device=0)
bad = 0
ts = torch.cuda.Stream(); stream = C.c_void_p(ts.cuda_stream)
for seed in range(int(sys.argv[1]), int(sys.argv[1]) + int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    n_tri = int(10 ** rng.uniform(0.5, 4.0)); n_obj = int(rng.integers(1, 6)) if n_tri >= 40 else 1
    try:
        scene, v = F._soup(ra, rng, n_tri, n_obj)
    except ra.MirtError:
        continue
    dev.upload(scene.describe())
    n = int(rng.choice([1, 2, 63, 64, 65, 255, 257, 1000, 4097, int(rng.integers(1, 300000)), int(rng.integers(1, 300000))]))
    o, d, aux, last = F._rays(rng, v, n, n_obj, ra, tri_obj=scene.raw()["tri_obj"])
    to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    line = "seed %d tris %d rays %d:" % (seed, n_tri, n)
    for mode, a in ((0, None), (1, last), (2, aux)):
        th, ih = dev.intersect(o, d, a, mode)
        ta = torch.from_numpy(a).cuda() if a is not None else None
        ap = ta.data_ptr() if ta is not None else None
        res = {}
        for name in ("dev", "noalloc", "count", "split"):
            t = torch.full((n,), -7.0, dtype=torch.float32, device="cuda"); i = torch.full((n,), -7, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            if name == "dev":
                dev.intersect_dev(to.data_ptr(), td.data_ptr(), ap, n, mode, t.data_ptr(), i.data_ptr(), stream); ts.synchronize()
            elif name == "noalloc":
                dev.intersect_noalloc_dev(to.data_ptr(), td.data_ptr(), ap, n, mode, t.data_ptr(), i.data_ptr(), stream); ts.synchronize()
            elif name == "count":
                dev.intersect_count_dev(to.data_ptr(), td.data_ptr(), ap, n, mode, t.data_ptr(), i.data_ptr())
            else:  # two calls on column ranges of the same plane-major arrays are not expressible (stride = n): copy halves
                k = int(rng.integers(0, n + 1))
                for b, e in ((0, k), (k, n)):
                    if e > b:
                        o2, d2 = to[:, b:e].contiguous(), td[:, b:e].contiguous()
                        a2 = ta[b:e].contiguous() if ta is not None else None
                        t2 = torch.empty(e - b, dtype=torch.float32, device="cuda"); i2 = torch.empty(e - b, dtype=torch.int32, device="cuda")
                        torch.cuda.synchronize()
                        dev.intersect_dev(o2.data_ptr(), d2.data_ptr(), a2.data_ptr() if a2 is not None else None, e - b, mode, t2.data_ptr(), i2.data_ptr(), stream); ts.synchronize()
                        t[b:e] = t2; i[b:e] = i2
            torch.cuda.synchronize()
            ok = np.array_equal(i.cpu().numpy(), ih) and np.array_equal(t.cpu().numpy(), th, equal_nan=True)
            res[name] = ok; bad += not ok
        line += " m%d %s" % (mode, " ".join("%s:%s" % (k, "ok" if x else "DIFFER") for k, x in res.items()))
    lim = (rng.uniform(0, 3, n)).astype(np.float32)
    for mode, a in ((2, aux), (0, None)):
        vh = dev.visibility(o, d, a, lim, mode)
        ta = torch.from_numpy(a).cuda() if a is not None else None
        tl = torch.from_numpy(lim).cuda(); vv = torch.full((n,), 9, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
        dev.visibility_dev(to.data_ptr(), td.data_ptr(), ta.data_ptr() if ta is not None else None, tl.data_ptr(), n, mode, vv.data_ptr(), stream); ts.synchronize()
        ok = np.array_equal(vv.cpu().numpy(), vh); bad += not ok
        line += " v%d %s" % (mode, "ok" if ok else "DIFFER")
    print(line, flush=True)
    scene.close()
print("entry fuzz: %d differing results" % bad)
