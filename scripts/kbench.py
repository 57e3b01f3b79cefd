"""Kernel A/B timing: builds the bench.py ray batch once and times the two intersect launches.
usage: MIRT_LIB=path/to/libmirt_variant.so python3 scripts/kbench.py [reps]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import renderer_amd as ra
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
if os.environ.get("SCENE") == "mesh1m":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import meshgen
    scene = meshgen.mesh_scene(7); cam = ra.make_camera(1024, 1024, 700)
elif os.environ.get("SCENE") == "fog":
    scene = ra.HostScene.load_config(os.path.join(ROOT, "Data", "Config_Fog1024.xml")); cam = scene.camera()
else:
    scene = ra.HostScene.load_config(os.path.join(ROOT, "Data", "Config_Glossy1024.xml")); cam = scene.camera()
dev = ra.Device(device=0); dev.upload(scene.describe())
wh = cam.width * cam.height; n_pri = wh * 4; cap = 12 << 20
po = torch.empty((3, n_pri), dtype=torch.float32, device="cuda"); pd = torch.empty_like(po)
so = torch.empty((3, cap), dtype=torch.float32, device="cuda"); sd = torch.empty_like(so)
sa = torch.empty(cap, dtype=torch.int32, device="cuda"); torch.cuda.synchronize()
dev.generate_rays_dev(cam, 0, 4, 1, po.data_ptr(), pd.data_ptr())
n_sec = dev.capture_rays_dev(cam, 12, 1, 3, so.data_ptr(), sd.data_ptr(), sa.data_ptr(), cap)
if os.environ.get("SORT"):  # experiment: what a coherence reorder of the bounce batch would buy
    d_ = sd[:, :n_sec]; o_ = so[:, :n_sec]
    key = ((d_[0] < 0).to(torch.int64) | ((d_[1] < 0).to(torch.int64) << 1) | ((d_[2] < 0).to(torch.int64) << 2))
    if os.environ["SORT"] == "octant_cell":
        q = ((o_.clamp(-1, 1) + 1) * 3.999).to(torch.int64)  # 8^3 origin cells
        key = key * 512 + (q[0] * 64 + q[1] * 8 + q[2])
    perm = torch.sort(key, stable=True).indices
    so[:, :n_sec] = o_[:, perm]; sd[:, :n_sec] = d_[:, perm]; sa[:n_sec] = sa[:n_sec][perm]
    torch.cuda.synchronize()
t1 = torch.empty(cap, dtype=torch.float32, device="cuda"); i1 = torch.empty(cap, dtype=torch.int32, device="cuda")
ts = torch.cuda.Stream(); st = C.c_void_p(ts.cuda_stream); torch.cuda.synchronize()
def run(o, d, a, n, mode):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    dev.intersect_dev(o.data_ptr(), d.data_ptr(), a, n, mode, t1.data_ptr(), i1.data_ptr(), st)
    for k in range(reps):
        ev[k].record(ts)
        dev.intersect_dev(o.data_ptr(), d.data_ptr(), a, n, mode, t1.data_ptr(), i1.data_ptr(), st)
    ev[reps].record(ts); torch.cuda.synchronize()
    return np.median([ev[k].elapsed_time(ev[k + 1]) for k in range(reps)])
ms2 = run(so, sd, sa.data_ptr(), n_sec, 2); ms0 = run(po, pd, None, n_pri, 0)
chk = int(i1[:n_pri].to(torch.int64).sum().item())
print("%-40s sec %.3f ms (%.0f Mrays/s)  pri %.3f ms (%.0f Mrays/s)  total %.0f Mrays/s  chk %d" % (
    os.path.basename(os.environ.get("MIRT_LIB", "libmirt.so")), ms2, n_sec / ms2 / 1e3, ms0, n_pri / ms0 / 1e3,
    (n_sec + n_pri) / (ms0 + ms2) / 1e3, chk), flush=True)
if os.environ.get("AUDIT"):
    # (the audit entry point instruments the EXACT route unless MIRT_AUDIT_FAST is set: the product's route is the fast
    # one. `box` then holds the fast tree's tests, two per pair step, + k_verify's chain tests, a few per ray.)
    os.environ.setdefault("MIRT_AUDIT_FAST", "1")
    a = dev.intersect_audit_dev(so.data_ptr(), sd.data_ptr(), sa.data_ptr(), n_sec, 2, t1.data_ptr(), i1.data_ptr())
    print(a)
    print("per ray: box tests %.1f  triangle tests %.2f;  per 64 rays: node passes %.1f  leaf passes %.1f;  "
          "leaf pass lanes %.1f of 64" % (a["box"] / n_sec, a["tri"] / n_sec, 64.0 * a["wave_inner"] / n_sec,
                                          64.0 * a["wave_leaf"] / n_sec, a["tri"] / a["wave_leaf"]))
if os.environ.get("SORTEXP"):
    # experiment: how much does ray coherence buy? reorder the secondary batch by several keys
    n = n_sec
    O = so[:, :n]; Dd = sd[:, :n]
    lo = O.min(dim=1).values; hi = O.max(dim=1).values
    def morton(bits):
        q = ((O - lo[:, None]) / (hi - lo)[:, None].clamp_min(1e-9) * ((1 << bits) - 1e-3)).to(torch.int64)
        k = torch.zeros(n, dtype=torch.int64, device="cuda")
        for b in range(bits):
            for c in range(3):
                k |= ((q[c] >> b) & 1) << (3 * b + c)
        return k
    octant = ((Dd[0] < 0).to(torch.int64) | ((Dd[1] < 0).to(torch.int64) << 1) | ((Dd[2] < 0).to(torch.int64) << 2))
    keys = {"octant": octant, "octant+o3": (octant << 9) | morton(3), "o3+octant": (morton(3) << 3) | octant,
            "octant+o5": (octant << 15) | morton(5), "o4+octant": (morton(4) << 3) | octant}
    for name, key in keys.items():
        order = torch.argsort(key)
        o2 = torch.empty_like(so); d2 = torch.empty_like(sd); a2 = torch.empty_like(sa)
        o2[:, :n] = O[:, order]; d2[:, :n] = Dd[:, order]; a2[:n] = sa[:n][order]
        torch.cuda.synchronize()
        ms = run(o2, d2, a2.data_ptr(), n, 2)
        print("sorted by %-12s sec %.3f ms (%.0f Mrays/s)" % (name, ms, n / ms / 1e3), flush=True)
        if os.environ.get("AUDIT"):
            a = dev.intersect_audit_dev(o2.data_ptr(), d2.data_ptr(), a2.data_ptr(), n, 2, t1.data_ptr(), i1.data_ptr())
            print("   inner lane util %.3f  leaf lane util %.3f" % ((a["box"] / 2) / (64.0 * a["wave_inner"]), a["tri"] / (64.0 * a["wave_leaf"])))
        del o2, d2, a2
if os.environ.get("SWEEP"):
    # launch-size sweep: where is the latency floor of a small launch?
    for n in (64, 1024, 8192, 65536, 262144, 1 << 20, 1 << 22):
        ms = run(so, sd, sa.data_ptr(), n, 2)
        print("n %8d  %.4f ms  (%.1f Mrays/s)" % (n, ms, n / ms / 1e3), flush=True)
