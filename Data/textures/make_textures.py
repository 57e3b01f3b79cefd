#!/usr/bin/env python3
"""Authors the small .pfm textures of Data/Scene_Textured.xml (the reference's own textures are
photographs; these are synthetic patterns of the same file format, Texture.cpp:5-41: "PF" rgb /
"Pf" gray header, width height, a scale field, then raw little-endian fp32 texels in file order).

  checker.pfm  8 x 8 rgb   light / dark squares, albedo <= 0.8
  stripes.pfm  16 x 4 rgb  coloured vertical stripes
  coeff.pfm    4 x 4 gray  glossy exponents between 4 and 40 (read through its .x)
  bumps.pfm    16 x 16 rgb a smooth height field for <BumpTexture>
Run:  python3 Data/textures/make_textures.py   (writes next to this file; output is committed)
"""
import os
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))

def write(name, img):
    img = np.asarray(img, "<f4")
    h, w = img.shape[:2]
    with open(os.path.join(here, name), "wb") as f:
        f.write(("%s\n%d %d\n-1.000000\n" % ("PF" if img.ndim == 3 else "Pf", w, h)).encode())
        f.write(img.tobytes())

y, x = np.mgrid[0:8, 0:8]
chk = np.where(((x + y) & 1)[..., None] == 0, np.array([0.8, 0.75, 0.6]), np.array([0.15, 0.2, 0.35]))
write("checker.pfm", chk)
cols = np.array([[0.8, 0.2, 0.2], [0.8, 0.8, 0.2], [0.2, 0.8, 0.3], [0.2, 0.4, 0.8]])
write("stripes.pfm", np.broadcast_to(cols[(np.arange(16) // 2) % 4][None], (4, 16, 3)))
write("coeff.pfm", 4.0 + 36.0 * ((np.arange(16).reshape(4, 4) * 7) % 16) / 15.0)
# bumps.pfm: a height field (rgb, reduced to gray by the loader like every <BumpTexture>, ConfigManager.cpp:161)
by, bx = np.mgrid[0:16, 0:16]
hgt = 0.5 + 0.25 * np.sin(2 * np.pi * bx / 8.0) * np.cos(2 * np.pi * by / 16.0)
write("bumps.pfm", np.stack([hgt, 0.9 * hgt, 1.1 * hgt], -1))
print("wrote checker.pfm, stripes.pfm, coeff.pfm, bumps.pfm")
