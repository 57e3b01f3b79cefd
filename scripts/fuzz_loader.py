"""One-off, CPU only: the host surface's loaders (config / scene / materials XML, .obj meshes, textures) on mutated copies
of the repository's Data/ files, each load in a child process: it must come back with a scene or with an error, not with a
signal or a hang. usage: fuzz_loader.py first_seed n_seeds"""
import os, shutil, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = "import sys; sys.path.insert(0, %r); import renderer_amd as ra\ntry:\n    s = ra.HostScene.load_config(sys.argv[1]); print('loaded', s.n_tris)\nexcept ra.MirtError as e:\n    print('refused', str(e)[:100])\n" % ROOT
configs = ["Config_Glossy128.xml", "Config_Fog128.xml", "Config_Textured128.xml", "Config_Bump128.xml", "Config_Balls.xml", "Config_Cube128.xml"]
stats = {"loaded": 0, "refused": 0, "signal": 0, "hang": 0, "other": 0}
for seed in range(int(sys.argv[1]), int(sys.argv[1]) + int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copytree(os.path.join(ROOT, "Data"), os.path.join(tmp, "Data"))
        files = []
        for d, _, fs in os.walk(os.path.join(tmp, "Data")):
            files += [os.path.join(d, f) for f in fs if f.endswith((".xml", ".obj", ".mtl", ".ppm", ".pfm", ".bmp", ".png", ".jpg", ".shp")) or "." not in f]
        victim = files[int(rng.integers(0, len(files)))] if rng.random() < 0.5 else os.path.join(tmp, "Data", ["Scene_CornellBox.xml", "Scene_Fog.xml", "Scene_Textured.xml", "Materials.xml"][int(rng.integers(0, 4))])
        data = bytearray(open(victim, "rb").read())
        kind = int(rng.integers(0, 5))
        if len(data) > 8:
            if kind == 0:
                del data[int(rng.integers(0, len(data))):]                        # truncated
            elif kind == 1:
                a = int(rng.integers(0, len(data))); del data[a:a + int(rng.integers(1, 200))]   # a piece missing
            elif kind == 2:
                for _ in range(int(rng.integers(1, 20))): data[int(rng.integers(0, len(data)))] = int(rng.integers(0, 256))  # bytes flipped
            elif kind == 3:
                a = int(rng.integers(0, len(data))); data[a:a] = data[a:a + int(rng.integers(1, 400))]    # a piece doubled
            else:
                txt = data.decode("latin1")                                        # a number replaced
                import re
                nums = list(re.finditer(r"-?\d+\.?\d*", txt))
                if nums:
                    m = nums[int(rng.integers(0, len(nums)))]
                    txt = txt[:m.start()] + ["nan", "-1", "0", "1e40", "99999999999", "", "abc", "1e-50"][int(rng.integers(0, 8))] + txt[m.end():]
                data = bytearray(txt.encode("latin1"))
        open(victim, "wb").write(data)
        cfg = os.path.join(tmp, "Data", configs[int(rng.integers(0, len(configs)))])
        try:
            r = subprocess.run([sys.executable, "-c", CHILD, cfg], capture_output=True, timeout=60)
            out = r.stdout.decode("latin1").strip().split("\n")[-1] if r.stdout else ""
            key = "signal" if r.returncode < 0 else ("loaded" if out.startswith("loaded") else ("refused" if out.startswith("refused") else "other"))
        except subprocess.TimeoutExpired:
            key, out, r = "hang", "", None
        stats[key] += 1
        if key in ("signal", "hang", "other"):
            print("seed %d %s: %s kind %d -> %s rc %s %s %s" % (seed, os.path.basename(cfg), os.path.relpath(victim, tmp), kind, key, r.returncode if r else None, out[:80], (r.stderr.decode("latin1")[-200:] if r else "")), flush=True)
print("loader fuzz:", stats)
