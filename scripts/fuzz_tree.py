"""One-off, CPU only: the host surface's KDTree::build (renderer_amd/host/geometry.cpp, both strategies) against the
oracle's pointer restatement (oracle/oracle.cpp) on random triangle soups (tests/test_gpu_fuzz.py's generator: slivers,
axis-aligned, degenerate and duplicated triangles, any scale and offset) and on closed meshes: node boxes, child links
and leaf ranges bit for bit. usage: fuzz_tree.py first_seed n_seeds"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import renderer_amd as ra, orc
from renderer_amd import meshgen
import test_gpu_fuzz as F

bad = done = 0
for seed in range(int(sys.argv[1]), int(sys.argv[1]) + int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    n_tri = int(10 ** rng.uniform(0.5, 4.5))
    scale = float(10 ** rng.uniform(-3, 3))
    offset = tuple((rng.uniform(-1, 1, 3) * scale * 10 ** rng.uniform(-1, 1.5)).tolist()) if rng.random() < 0.6 else (0.0, 0.0, 0.0)
    try:
        if seed % 5 == 4:
            s = meshgen.mesh_scene(int(rng.integers(1, 6)), seed, smooth=bool(rng.integers(0, 2)))
        else:
            s, _ = F._soup(ra, rng, n_tri, int(rng.integers(1, 7)) if n_tri >= 40 else 1, scale, offset)
    except ra.MirtError as e:
        print("seed %d skipped: %s" % (seed, e)); continue
    o = orc.OracleScene(s.raw())
    line = "seed %d tris %d:" % (seed, s.n_tris)
    for strat in (1, 0):
        s.rebuild_tree(strat); o.rebuild(strat)
        a, b = s.tree(), o.flatten()
        same = s.n_nodes == o.n_nodes and s.depth == o.depth and all(np.array_equal(a[k], b[k]) for k in a)
        line += " %s nodes %d depth %d %s" % ("BEST" if strat else "LOOP", s.n_nodes, s.depth, "ok" if same else "DIFFER")
        bad += 0 if same else 1
    done += 1
    print(line, flush=True)
    o.close(); s.close()
print("tree fuzz: %d scenes, %d differing builds" % (done, bad))
