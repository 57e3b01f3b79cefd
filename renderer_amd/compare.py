"""compare — the reference's offline image-comparison tool (compare.cpp, its `main` is commented out
there) for the PFM series a progressive render leaves behind.

  python -m renderer_amd.compare REF.pfm DIR [MASK.txt]     # iteration / time / RMSE table

read_pfm clamps every value to [0, 1] like compare.cpp:260-283; rmse is CalculateRMSE
(compare.cpp:89-119): sqrt(sum over unmasked pixels of |rgb - rgb_ref|^2 / #pixels); rmse_table is
calcRMSEs (compare.cpp:167-251): every "*_iter_<N>_time_<T>*.pfm" in a folder against one reference,
sorted by iteration, printed as "iter time rmse"."""
import os
import sys

import numpy as np


def read_pfm(path, clamp=True):
    with open(path, "rb") as f:
        if f.readline().strip() != b"PF":
            raise ValueError("%s: not a colour PFM" % path)
        w, h = (int(x) for x in f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(w * h * 12), dtype="<f4" if scale < 0 else ">f4")
    if data.size != w * h * 3:
        raise ValueError("%s: truncated" % path)
    img = data.reshape(h, w, 3)[::-1].astype(np.float32)  # rows are stored bottom-to-top (MCRenderer.cpp:506-515)
    return np.clip(img, 0.0, 1.0) if clamp else img


def rmse(img, ref, mask=None):
    d2 = ((img.astype(np.float64) - ref.astype(np.float64)) ** 2).sum(-1)
    if mask is not None:
        d2 = d2[mask]
    return float(np.sqrt(d2.sum() / d2.size))


def read_mask(path, w, h):
    """compare.cpp:134-147: h rows of w integers, bottom row first; 1 = pixel counts"""
    v = np.loadtxt(path, dtype=np.int64).reshape(h, w)[::-1]
    return v == 1


def parse_iter_time(name):
    """compare.cpp:205-232: tokens split at '_' and '.'; the token after 'iter' / 'time' is the number"""
    it = tm = -1
    toks = name.replace(".", "_").split("_")
    for a, b in zip(toks, toks[1:]):
        if a == "iter":
            it = int(b) if b.lstrip("-").isdigit() else 0
        if a == "time":
            tm = int(b) if b.lstrip("-").isdigit() else 0
    return it, tm


def rmse_table(folder, ref_path, mask_path=None):
    ref = read_pfm(ref_path)
    mask = read_mask(mask_path, ref.shape[1], ref.shape[0]) if mask_path else None
    rows = []
    for name in os.listdir(folder):
        if not name.endswith(".pfm"):
            continue
        it, tm = parse_iter_time(name)
        if it == -1 or tm == -1:
            continue
        rows.append((it, tm, rmse(read_pfm(os.path.join(folder, name)), ref, mask)))
    rows.sort(key=lambda r: r[0])
    return rows


def main(argv):
    if len(argv) < 3:
        print(__doc__)
        return 2
    for it, tm, r in rmse_table(argv[2], argv[1], argv[3] if len(argv) > 3 else None):
        print("%d %d %.8f" % (it, tm, r))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
