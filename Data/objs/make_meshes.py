#!/usr/bin/env python3
"""Authors the three meshes the Cornell-box scenes use. The reference's own .obj files are not in
its tree (its .gitignore drops *.obj), so these are stand-ins that obey the reference loader's
rules (SimpleShape.cpp:65-189): triangles only, 3 numbers per v / vn line, faces as a//n.

  plane.obj  unit square on y = 0 spanning [-1,1]^2 in x,z, normal +y (2 triangles, one vn)
  ball.obj   UV sphere, 32 x 16 (960 triangles), per-vertex normals = positions
  cube.obj   [-1,1]^3, 12 triangles, outward winding, no vertex normals (flat)
  spot_light2.obj  two groups for <GroupObject> (Data/Scene_Fog.xml): "light", a square on y = 0 with normal
             +y, and "frame", a four-sided shade around it from y = 0 to y = 1 whose faces look inward
Run:  python3 Data/objs/make_meshes.py   (writes next to this file; output is committed)
"""
import math, os
here = os.path.dirname(os.path.abspath(__file__))

def write(name, verts, norms, faces):
    with open(os.path.join(here, name), "w") as f:
        f.write("# authored by Data/objs/make_meshes.py\n")
        for v in verts: f.write("v %.9g %.9g %.9g\n" % v)
        for n in norms: f.write("vn %.9g %.9g %.9g\n" % n)
        for fc in faces:
            if norms: f.write("f %d//%d %d//%d %d//%d\n" % (fc[0][0], fc[0][1], fc[1][0], fc[1][1], fc[2][0], fc[2][1]))
            else: f.write("f %d %d %d\n" % (fc[0][0], fc[1][0], fc[2][0]))

# plane: face normal (v1-v0)x(v2-v0) = +y
pv = [(-1.0, 0.0, -1.0), (-1.0, 0.0, 1.0), (1.0, 0.0, 1.0), (1.0, 0.0, -1.0)]
write("plane.obj", pv, [(0.0, 1.0, 0.0)], [((1, 1), (2, 1), (3, 1)), ((1, 1), (3, 1), (4, 1))])

# ball
NU, NV = 32, 16
verts, faces = [], []
verts.append((0.0, 1.0, 0.0))
for j in range(1, NV):
    th = math.pi * j / NV
    for i in range(NU):
        ph = 2 * math.pi * i / NU
        verts.append((math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph)))
verts.append((0.0, -1.0, 0.0))
def vid(j, i):  # 1-based
    if j == 0: return 1
    if j == NV: return len(verts)
    return 2 + (j - 1) * NU + (i % NU)
def tri(a, b, c):
    # orient outward: (b-a)x(c-a) . a > 0
    A, B, C = verts[a - 1], verts[b - 1], verts[c - 1]
    u = [B[k] - A[k] for k in range(3)]; w = [C[k] - A[k] for k in range(3)]
    n = (u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0])
    cen = [(A[k] + B[k] + C[k]) / 3 for k in range(3)]
    if sum(n[k] * cen[k] for k in range(3)) < 0: b, c = c, b
    faces.append(((a, a), (b, b), (c, c)))
for i in range(NU):
    tri(vid(0, i), vid(1, i), vid(1, i + 1))
    tri(vid(NV, i), vid(NV - 1, i + 1), vid(NV - 1, i))
for j in range(1, NV - 1):
    for i in range(NU):
        a, b, c, d = vid(j, i), vid(j, i + 1), vid(j + 1, i), vid(j + 1, i + 1)
        tri(a, c, d); tri(a, d, b)
write("ball.obj", verts, verts, faces)

# cube
cv = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)]
cv = [tuple(float(x) for x in v) for v in cv]
quads = [(1, 4, 3, 2), (5, 6, 7, 8), (1, 2, 6, 5), (4, 8, 7, 3), (1, 5, 8, 4), (2, 3, 7, 6)]
cf = []
for q in quads:
    for t in ((q[0], q[1], q[2]), (q[0], q[2], q[3])):
        A, B, C = cv[t[0] - 1], cv[t[1] - 1], cv[t[2] - 1]
        u = [B[k] - A[k] for k in range(3)]; w = [C[k] - A[k] for k in range(3)]
        n = (u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0])
        cen = [(A[k] + B[k] + C[k]) / 3 for k in range(3)]
        if sum(n[k] * cen[k] for k in range(3)) < 0: t = (t[0], t[2], t[1])
        cf.append(((t[0], 0), (t[1], 0), (t[2], 0)))
write("cube.obj", cv, [], cf)

# spot_light2: every group declares its own v / vn after its g line (the loader rebases a group's face
# indices to the counts at that line, SimpleShape.cpp:85-96, 142-155); indices stay file-global
with open(os.path.join(here, "spot_light2.obj"), "w") as f:
    f.write("# authored by Data/objs/make_meshes.py\n")
    f.write("g light\n")
    a = 0.6
    for v in ((-a, 0.0, -a), (-a, 0.0, a), (a, 0.0, a), (a, 0.0, -a)): f.write("v %.9g %.9g %.9g\n" % v)
    f.write("vn 0 1 0\n")
    f.write("f 1//1 2//1 3//1\nf 1//1 3//1 4//1\n")
    f.write("g frame\n")
    nv, nn = 4, 1
    sides = [((-1, 0, -1), (-1, 0, 1), (-1, 1, 1), (-1, 1, -1), (1, 0, 0)),
             ((1, 0, -1), (1, 0, 1), (1, 1, 1), (1, 1, -1), (-1, 0, 0)),
             ((-1, 0, -1), (1, 0, -1), (1, 1, -1), (-1, 1, -1), (0, 0, 1)),
             ((-1, 0, 1), (1, 0, 1), (1, 1, 1), (-1, 1, 1), (0, 0, -1))]
    for q in sides:
        for v in q[:4]: f.write("v %.9g %.9g %.9g\n" % tuple(float(x) for x in v))
        f.write("vn %.9g %.9g %.9g\n" % tuple(float(x) for x in q[4]))
    for k in range(4):
        b, n = nv + 4 * k, nn + 1 + k
        f.write("f %d//%d %d//%d %d//%d\n" % (b + 1, n, b + 2, n, b + 3, n))
        f.write("f %d//%d %d//%d %d//%d\n" % (b + 1, n, b + 3, n, b + 4, n))
# plane_uv / ball_uv: the same meshes with texture coordinates (faces as v/vt/vn). The plane maps
# (x, z) in [-1,1]^2 to (u, v) in [0,1]^2; the ball uses (longitude, latitude) with its own vt grid
# (the seam and the poles need more vt than v, which .obj's separate indices allow)
with open(os.path.join(here, "plane_uv.obj"), "w") as f:
    f.write("# authored by Data/objs/make_meshes.py\n")
    for v in pv: f.write("v %.9g %.9g %.9g\n" % v)
    for v in pv: f.write("vt %.9g %.9g 0\n" % ((v[0] + 1) / 2, (v[2] + 1) / 2))
    f.write("vn 0 1 0\n")
    f.write("f 1/1/1 2/2/1 3/3/1\nf 1/1/1 3/3/1 4/4/1\n")
with open(os.path.join(here, "ball_uv.obj"), "w") as f:
    f.write("# authored by Data/objs/make_meshes.py\n")
    for v in verts: f.write("v %.9g %.9g %.9g\n" % v)
    for j in range(NV + 1):
        for i in range(NU + 1): f.write("vt %.9g %.9g 0\n" % (i / NU, j / NV))
    for v in verts: f.write("vn %.9g %.9g %.9g\n" % v)
    def ring(idx):  # (j, i mod NU) of a 1-based vertex id
        if idx == 1: return 0, None
        if idx == len(verts): return NV, None
        return 1 + (idx - 2) // NU, (idx - 2) % NU
    for fc in faces:
        ids = [c[0] for c in fc]
        ji = [ring(a) for a in ids]
        cols = [i for (_, i) in ji if i is not None]
        wrap = max(cols) - min(cols) > NU // 2  # the face spans the seam: its column-0 corners use column NU
        lo = min(c + NU if wrap and c == 0 else c for c in cols)  # the pole's column: the face's first
        out = []
        for a, (j, i) in zip(ids, ji):
            col = lo if i is None else (i + NU if wrap and i == 0 else i)
            out.append("%d/%d/%d" % (a, 1 + j * (NU + 1) + col, a))
        f.write("f " + " ".join(out) + "\n")
print("wrote plane.obj (2 tris), ball.obj (%d tris), cube.obj (%d tris)" % (len(faces), len(cf)))
