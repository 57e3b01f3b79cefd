"""CPU tests of the host surface (XML/OBJ loading, tree build + flatten) and of the C-ABI library's
exports. No compute call is made here: there is no GPU in this environment."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import meshgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "Data")


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mirth?_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(ra):
    L = C.CDLL(ra.LIB_PATH)
    for header, known in (("mirt.h", ra.MIRT_SYMBOLS), ("mirt_host.h", ra.MIRTH_SYMBOLS)):
        names = _declared(header)
        assert sorted(known) == names, "bindings out of date for %s" % header
        for n in names:
            assert hasattr(L, n), "libmirt.so does not export %s" % n


def test_no_gpu_means_error_not_fallback(ra):
    """Without a device mirt_create must fail with MIRT_E_NO_DEVICE; nothing computes on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    rc = ra.lib().mirt_create(0, C.byref(ctx))
    assert rc == -2 and not ctx.value
    assert b"no CPU path" in ra.lib().mirt_last_error(None)
    with pytest.raises(ra.MirtError):
        ra.Device(device=0)


def test_no_scene_is_resident_without_a_device(cornell, mesh20k, ra):
    """IntersectionHIP records which scene it holds; without a context none is, and host-only edits of a
    scene (set_textures) re-stamp it (the device copy, if any, no longer describes it)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert ra.lib().mirth_is_resident(cornell.h) == 0 and ra.lib().mirth_is_resident(mesh20k.h) == 0
    assert ra.lib().mirth_is_resident(None) == 0
    with pytest.raises(ra.MirtError):  # a query uploads first, and that fails loudly without a GPU
        cornell.fill_intersect(np.zeros((1, 3), np.float32), np.ones((1, 3), np.float32), [-1], [-1])


def test_config_load_cornell(cornell, ra):
    assert (cornell.n_objects, cornell.n_tris, cornell.n_verts) == (8, 1932, 988)
    raw = cornell.raw()
    ob = raw["objects"]
    assert ob["type"].tolist() == [ra.OBJ_EMISSIVE, ra.OBJ_REFRACTIVE, ra.OBJ_REFRACTIVE, ra.OBJ_DIFFUSE,
                                   ra.OBJ_DIFFUSE, ra.OBJ_DIFFUSE, ra.OBJ_GLOSSY, ra.OBJ_DIFFUSE]
    assert ob["ior"][1] == -1.0 and ob["ior"][2] == 1.5  # Reflective = mirror, Refractive 1.5
    assert np.allclose(ob["color"][0], 50) and np.allclose(ob["color"][3], (0.75, 0.25, 0.25))
    assert ob["coeff"][6] == 5.0 and np.allclose(ob["decay"][2], 1)
    assert ob["tri_begin"].tolist() == [0, 2, 962, 1922, 1924, 1926, 1928, 1930]
    assert (ob["has_vn"] == 1).all()
    # readMatrix is row-major text (xmlHelper.cpp:4-16): the light is 0.4 x 0.4 at y = 0.99
    lv = raw["verts"][:4]
    assert np.allclose(lv[:, 1], 0.99) and np.allclose(np.abs(lv[:, [0, 2]]), 0.2)
    # ball 1: radius 0.4 around (-0.5,-0.6,-0.3)
    bv = raw["verts"][4:4 + 482]
    assert np.allclose(np.linalg.norm(bv - (-0.5, -0.6, -0.3), axis=1), 0.4, atol=1e-6)
    # normals go through the inverse-transpose: the mirrored light faces down, left wall faces +x
    assert np.allclose(raw["tri_vn"][0].reshape(3, 3), (0, -5, 0))
    assert np.allclose(raw["tri_vn"][1922].reshape(3, 3), (1, 0, 0))
    cam = cornell.camera()
    assert (cam.width, cam.height, cam.sight_dist) == (128, 128, 87)
    assert list(cam.position) == [0, 0, 2] and list(cam.up) == [0, 1, 0]
    assert cornell.render_cfg() == dict(max_depth=20, spp=64, seed=1)


def test_product_tree_equals_oracle_tree(cornell, cornell_oracle, mesh20k, mesh20k_oracle):
    """Two independent builds (array/stable-partition in the product, pointer/recursive in the
    oracle) of KDTree::splitBest + serializeForGPU must agree bit for bit."""
    for s, o in ((cornell, cornell_oracle), (mesh20k, mesh20k_oracle)):
        a, b = s.tree(), o.flatten()
        assert s.n_nodes == o.n_nodes and s.depth == o.depth
        for k in a:
            assert np.array_equal(a[k], b[k]), k


def test_loop_strategy_tree_equals_oracle_tree(ra, orc):
    """KDTree::build(LOOP) (splitLoop, KDTree.cpp:54-127): the product's array build and the oracle's pointer
    restatement agree bit for bit; the tree differs from BEST's, and building BEST again restores the loader's tree."""
    s = meshgen.mesh_scene(3)
    raw = s.raw()
    o = orc.OracleScene(raw)
    best = s.tree()
    s.rebuild_tree(ra.HostScene.LOOP)
    o.rebuild(0)
    a, b = s.tree(), o.flatten()
    assert s.n_nodes == o.n_nodes and s.depth == o.depth
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert s.n_nodes != len(best["node_right"]) or not np.array_equal(a["node_min"], best["node_min"])
    # the root splits on x (dim 0), its children on y: the left child's max x is the median vertex x
    xs = np.unique(raw["tris"].reshape(-1))
    med = np.sort(raw["verts"][xs, 0])[len(xs) >> 1]
    right = a["node_right"][0]
    assert a["node_min"][right][0] > med  # every triangle on the right lies wholly beyond the split
    s.rebuild_tree(ra.HostScene.BEST)
    o.rebuild(1)
    a, b = s.tree(), o.flatten()
    for k in a:
        assert np.array_equal(a[k], best[k]) and np.array_equal(a[k], b[k]), k
    with pytest.raises(ra.MirtError, match="strategy"):
        s.rebuild_tree(7)
    s.close()
    o.close()


def test_shp_mesh_loads_like_the_reference(ra, tmp_path):
    """SimpleShape::loadShape on a .shp file (SimpleShape.cpp:52-64): the two counts in the file are counts of
    floats / of indices (saveShape, :341-357), and the reference resizes its vec3 lists to them, so the lists come out
    three times too long with zero tails. The same mesh written as an .obj with those tails spelled out must load to
    the same scene, bit for bit."""
    rng = np.random.default_rng(5)
    V = rng.uniform(-2, 3, (7, 3)).astype(np.float32)
    F = np.array([[0, 1, 2], [2, 3, 4], [4, 5, 6], [6, 0, 3]], np.uint32)
    with open(tmp_path / "m.shp", "wb") as f:
        f.write(np.int32(V.size).tobytes() + V.tobytes() + np.int32(F.size).tobytes() + F.tobytes())
    with open(tmp_path / "m.obj", "w") as f:
        for v in V:
            f.write("v %s %s %s\n" % tuple(repr(float(x)) for x in v))
        for _ in range(2 * len(V)):
            f.write("v 0 0 0\n")
        for t in F:
            f.write("f %d %d %d\n" % tuple(int(i) + 1 for i in t))
        for _ in range(2 * len(F)):
            f.write("f 1 1 1\n")
    scene = """<Scene Name="S"><Camera Name="c"><width>8</width><height>4</height><sightDist>6</sightDist>
      <focus>(0, 0, 0)</focus><position>(0, 0, 3)</position><up>(0, 1, 0)</up></Camera>
      <Object><filePath>%s</filePath><Material><type>Diffuse</type><color>(0.1, 0.2, 0.3)</color></Material></Object>
      </Scene>"""
    raws = []
    for ext in ("shp", "obj"):
        (tmp_path / ("s_%s.xml" % ext)).write_text(scene % ("m." + ext))
        (tmp_path / ("c_%s.xml" % ext)).write_text("<Config><Scene><path>s_%s.xml</path><name>S</name><camera>c</camera>"
                                                   "</Scene><renderer>PT</renderer></Config>" % ext)
        s = ra.HostScene.load_config(str(tmp_path / ("c_%s.xml" % ext)))
        assert (s.n_verts, s.n_tris) == (3 * len(V), 3 * len(F))
        raws.append((s.raw(), s.tree()))
        s.close()
    for k in ("verts", "tris", "tri_obj", "tri_local"):
        assert np.array_equal(raws[0][0][k], raws[1][0][k]), k
    for k in raws[0][1]:
        assert np.array_equal(raws[0][1][k], raws[1][1][k]), k
    # the zero tail is part of the unitize box (SimpleShape.cpp:183): the origin maps inside [-1, 1]^3
    assert np.abs(raws[0][0]["verts"]).max() <= 1 + 1e-6
    (tmp_path / "short.shp").write_bytes(np.int32(30).tobytes() + V.tobytes()[:40])
    (tmp_path / "s_bad.xml").write_text(scene % "short.shp")
    (tmp_path / "c_bad.xml").write_text("<Config><Scene><path>s_bad.xml</path><name>S</name><camera>c</camera></Scene>"
                                        "<renderer>PT</renderer></Config>")
    with pytest.raises(ra.MirtError, match="truncated"):
        ra.HostScene.load_config(str(tmp_path / "c_bad.xml"))


def test_describe_is_consistent(cornell, ra):
    d = cornell.describe()
    assert d.struct_size == C.sizeof(ra.SceneDesc) and d.n_tris == 1932 and d.n_objects == 8
    assert d.n_nodes == cornell.n_nodes


def test_pttest_renderer_type_loads(ra):
    """<type>PTtest</type> (ConfigManager.cpp:285-288) selects the light-connection integrator"""
    s = ra.HostScene.load_config(os.path.join(DATA, "Config_Glossy128_PTtest.xml"))
    assert s.render_cfg() == dict(max_depth=20, spp=64, seed=1)
    assert s.n_tris == 1932
    s.close()


def test_config_errors(ra, tmp_path):
    with pytest.raises(ra.MirtError, match="cannot open"):
        ra.HostScene.load_config(str(tmp_path / "missing.xml"))
    cfg = tmp_path / "c.xml"
    cfg.write_text("<Config><Scene><path>%s/Scene_CornellBox.xml</path><name>Balls</name><camera>c1</camera>"
                   "</Scene><Renderer><type>VCM</type></Renderer></Config>" % DATA)
    # relative mesh / material paths resolve against the config's folder, so copy the data folder layout
    os.symlink(os.path.join(DATA, "objs"), tmp_path / "objs")
    os.symlink(os.path.join(DATA, "Materials.xml"), tmp_path / "Materials.xml")
    with pytest.raises(ra.MirtError, match="outside this build"):
        ra.HostScene.load_config(str(cfg))
    bad = tmp_path / "bad.xml"
    bad.write_text("<Config><Scene></Config>")
    with pytest.raises(ra.MirtError, match="xml"):
        ra.HostScene.load_config(str(bad))
    cfg2 = tmp_path / "c2.xml"
    cfg2.write_text("<Config><Scene><path>%s/Scene_CornellBox.xml</path><name>NoSuchScene</name><camera>c1</camera>"
                    "</Scene><Renderer><type>PT</type></Renderer></Config>" % DATA)
    # an unknown name falls back to the first scene without a Name -> none here -> error
    with pytest.raises(ra.MirtError, match="not found"):
        ra.HostScene.load_config(str(cfg2))


def test_inline_material_and_obj_rules(ra, tmp_path):
    """OBJ loader rules (SimpleShape.cpp:65-189): '#' skipped, 4-token lines only, a//n and plain
    faces, unitize to [-1,1]; inline <Material> fallback (ConfigManager.cpp:230-235)."""
    (tmp_path / "t.obj").write_text("# comment\nv 0 0 0\nv 4 0 0\nv 0 2 0\nv 0 0 1 9\nvn 0 0 1\nvt 0.5 0.5\n"
                                    "f 1 2 3\nf 1 3\ng grp\nf 1//1 3//1 4//1\n")
    (tmp_path / "s.xml").write_text("""<Scene Name="S"><Camera Name="c"><width>8</width><height>4</height>
      <sightDist>6</sightDist><focus>(0, 0, 0)</focus><position>(0, 0, 3)</position><up>(0, 1, 0)</up></Camera>
      <Object><filePath>t.obj</filePath><Material><type>Glossy</type><color>(0.1, 0.2, 0.3)</color>
      <coeff>7</coeff></Material></Object></Scene>""")
    (tmp_path / "c.xml").write_text("<Config><Scene><path>s.xml</path><name>S</name><camera>c</camera></Scene>"
                                    "<renderer>PT</renderer><maxDepth>5</maxDepth></Config>")
    s = ra.HostScene.load_config(str(tmp_path / "c.xml"))
    raw = s.raw()
    assert s.n_tris == 2 and s.n_verts == 4  # "f 1 3" has 3 tokens -> dropped; "v 0 0 1 9" keeps 3 numbers
    # the last face carried normals but the first did not -> normal index list is short -> flat shading
    assert raw["objects"]["has_vn"][0] == 0
    assert raw["objects"]["type"][0] == ra.OBJ_GLOSSY and raw["objects"]["coeff"][0] == 7
    # unitize: centre (2,1,0.5), longest extent 4 -> x in [-1,1]
    assert np.allclose(raw["verts"][:, 0].min(), -1) and np.allclose(raw["verts"][:, 0].max(), 1)
    assert np.allclose(raw["verts"][3], (-1, -0.5, 0.25))
    assert s.render_cfg()["max_depth"] == 5
    cam = s.camera()
    assert (cam.width, cam.height, cam.sight_dist) == (8, 4, 6)
    s.close()


def test_scene_from_arrays_validation(ra):
    m = meshgen.cornell_with_meshes(1)
    with pytest.raises(ra.MirtError, match="out of range"):
        bad = m["tris"].copy()
        bad[0, 0] = 10**6
        ra.HostScene.from_arrays(m["verts"], bad, m["tri_obj"], m["objects"], m["tri_vn"])
    with pytest.raises(ra.MirtError, match="ordered"):
        ra.HostScene.from_arrays(m["verts"], m["tris"], m["tri_obj"][::-1].copy(), m["objects"], m["tri_vn"])
    s = ra.HostScene.from_arrays(m["verts"], m["tris"], m["tri_obj"], m["objects"], m["tri_vn"])
    assert s.n_objects == 9 and s.n_tris == 12 + 3 * 80
    ob = s.raw()["objects"]
    assert ob["tri_count"].tolist() == [2] * 6 + [80] * 3
    s.close()


def test_pfm_roundtrip(ra, tmp_path):
    img = np.random.default_rng(0).random((5, 7, 3)).astype(np.float32)
    p = str(tmp_path / "a.pfm").encode()
    assert ra.lib().mirth_save_pfm(p, img.ctypes.data_as(C.c_void_p), 7, 5) == 0
    data = open(p, "rb").read()
    assert data.startswith(b"PF\n7 5\n-1.000000\n")  # MCRenderer.cpp:505
    body = np.frombuffer(data[len(b"PF\n7 5\n-1.000000\n"):], "<f4").reshape(5, 7, 3)
    assert np.array_equal(body[::-1], img)  # rows bottom-to-top, channels R,G,B
    back = np.zeros_like(img)
    wh = (C.c_int32 * 2)()
    assert ra.lib().mirth_load_pfm(p, back.ctypes.data_as(C.c_void_p), 35, C.byref(wh)) == 0
    assert list(wh) == [7, 5] and np.array_equal(back, img)


def test_pack_ids_and_shards(ra):
    a = ra.pack_ids([-1, 0, 7, 65533], [-1, 3, -1, 65533])
    assert a.view(np.uint32).tolist() == [0, 1 | (4 << 16), 8, 65534 | (65534 << 16)]
    for spp, ws in ((4096, 8), (10, 4), (3, 8), (64, 1)):
        parts = [ra.shard_samples(spp, ws, r) for r in range(ws)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == spp
        for (b0, c0), (b1, _) in zip(parts, parts[1:]):
            assert b0 + c0 == b1


def test_compare_tool(ra, tmp_path):
    """renderer_amd.compare = compare.cpp: clamped PFM read, RMSE over pixels, iter/time table"""
    from renderer_amd import compare

    rng = np.random.default_rng(4)
    ref = rng.uniform(0, 2, (6, 9, 3)).astype(np.float32)
    ra.save_pfm(str(tmp_path / "ref.pfm"), ref)
    got = compare.read_pfm(str(tmp_path / "ref.pfm"))
    assert np.array_equal(got, np.clip(ref, 0, 1))
    assert np.array_equal(compare.read_pfm(str(tmp_path / "ref.pfm"), clamp=False), ref)
    series = tmp_path / "series"
    series.mkdir()
    for it, tm, noise in ((10, 3, 0.05), (0, 0, 0.3), (5, 1, 0.1)):
        ra.save_pfm(str(series / ("img_iter_%d_time_%d.pfm" % (it, tm))), ref + noise)
    ra.save_pfm(str(series / "img.pfm"), ref)  # no iter/time in the name: ignored
    rows = compare.rmse_table(str(series), str(tmp_path / "ref.pfm"))
    assert [r[:2] for r in rows] == [(0, 0), (5, 1), (10, 3)]
    exp = [compare.rmse(np.clip(ref + n, 0, 1), np.clip(ref, 0, 1)) for n in (0.3, 0.1, 0.05)]
    assert np.allclose([r[2] for r in rows], exp)
    assert rows[0][2] > rows[1][2] > rows[2][2]
    assert compare.parse_iter_time("cornell_iter_15_time_7.pfm") == (15, 7)
    # mask file: bottom row first, 1 = counted
    mask = np.zeros((6, 9), np.int64)
    mask[0, :] = 1  # top image row
    np.savetxt(tmp_path / "mask.txt", mask[::-1], fmt="%d")
    a = np.clip(ref, 0, 1).copy()
    a[0] = 0
    assert np.isclose(compare.rmse(a, np.clip(ref, 0, 1), compare.read_mask(str(tmp_path / "mask.txt"), 9, 6)),
                      np.sqrt((np.clip(ref[0], 0, 1).astype(np.float64) ** 2).sum() / 9))


def test_group_object_split(ra, tmp_path):
    """<GroupObject> (ConfigManager.cpp:192-226, SimpleShape.cpp:85-96,186-194): one .obj split at its g
    lines; members come before the plain objects, take the group's transform (their own <transform> is
    overwritten, ConfigManager.cpp:32-35) and are scaled by the WHOLE file's unitize factor without being
    recentred (the unitize matrix carries its translation in the bottom row)."""
    s = ra.HostScene.load_config(os.path.join(DATA, "Config_SpotLight128.xml"))
    raw = s.raw()
    assert s.n_objects == 8 and s.n_tris == 2 + 8 + 960 + 5 * 2
    assert raw["objects"]["type"][0] == ra.OBJ_EMISSIVE and raw["objects"]["type"][1] == ra.OBJ_DIFFUSE
    tri_obj, tris, verts = raw["tri_obj"], raw["tris"].astype(np.int64), raw["verts"]
    light = verts[np.unique(tris[tri_obj == 0])]
    frame = verts[np.unique(tris[tri_obj == 1])]
    # group transform [0.2,0,0,0][0,-0.2,0,0.88][0,0,0.2,0]; the file spans [-1,1] x [0,1] x [-1,1]: factor 1
    assert np.allclose(light[:, 1], 0.88) and np.allclose(np.abs(light[:, [0, 2]]), 0.12)
    assert np.allclose(sorted(set(np.round(frame[:, 1], 6))), [0.68, 0.88]) and np.allclose(np.abs(frame[:, [0, 2]]), 0.2)
    # the light's shading normal points down (normal matrix of the y-flipping transform)
    vn = raw["tri_vn"][tri_obj == 0].reshape(-1, 3)
    assert (vn[:, 1] < 0).all() and np.allclose(vn[:, [0, 2]], 0)
    s.close()
    # a scaled, off-centre file: sub-shapes are multiplied by 2/maxLen but keep their offset
    obj = tmp_path / "g.obj"
    obj.write_text("g a\nv 2 0 0\nv 6 0 0\nv 2 4 0\nf 1 2 3\ng b\nv 2 0 1\nv 6 0 1\nv 2 4 1\nf 4 5 6\n")
    cfg = tmp_path / "c.xml"
    cfg.write_text("<Config><Scene><path>s.xml</path><name>S</name><camera>c</camera></Scene><renderer>PT</renderer></Config>")
    scn = tmp_path / "s.xml"
    scn.write_text("<Scene Name=\"S\"><Camera Name=\"c\"><width>4</width><height>4</height><sightDist>4</sightDist>"
                   "<focus>(0,0,0)</focus><position>(0,0,2)</position><up>(0,1,0)</up></Camera>"
                   "<GroupObject><filePath>%s</filePath>"
                   "<Object Name=\"b\"><Material><type>Diffuse</type><color>(1,1,1)</color></Material></Object>"
                   "<Object Name=\"a\"><Material><type>Emissive</type><color>(1,1,1)</color></Material></Object>"
                   "</GroupObject></Scene>" % obj)
    s = ra.HostScene.load_config(str(cfg))
    raw = s.raw()
    assert s.n_objects == 2 and raw["objects"]["type"][0] == ra.OBJ_DIFFUSE  # member order, not group order
    v = raw["verts"][raw["tris"].astype(np.int64)[0]]
    assert np.allclose(v, np.array([[2, 0, 1], [6, 0, 1], [2, 4, 1]]) * 0.5)  # maxLen 4: factor 2/4
    s.close()
    scn.write_text(scn.read_text().replace('Name="b"', 'Name="zz"'))
    with pytest.raises(ra.MirtError, match="no group of that name"):
        ra.HostScene.load_config(str(cfg))
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\ng a\nf 1 2 3\n")
    scn.write_text(scn.read_text().replace('Name="zz"', 'Name="a"'))
    with pytest.raises(ra.MirtError, match="declared before its g line"):
        ra.HostScene.load_config(str(cfg))


def test_phong_material_from_xml(ra, tmp_path):
    """<type>Phong</type> with <GlossyComponent> / <DiffuseComponent> (ConfigManager.cpp:48-58)"""
    (tmp_path / "t.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    (tmp_path / "s.xml").write_text("""<Scene Name="S"><Camera Name="c"><width>4</width><height>4</height>
      <sightDist>4</sightDist><focus>(0, 0, 0)</focus><position>(0, 0, 3)</position><up>(0, 1, 0)</up></Camera>
      <Object><filePath>t.obj</filePath><Material><type>Phong</type>
      <GlossyComponent><color>(0.3, 0.2, 0.1)</color><coeff>9</coeff></GlossyComponent>
      <DiffuseComponent><color>(0.4, 0.5, 0.6)</color></DiffuseComponent></Material></Object></Scene>""")
    (tmp_path / "c.xml").write_text("<Config><Scene><path>s.xml</path><name>S</name><camera>c</camera></Scene>"
                                    "<renderer>PT</renderer></Config>")
    s = ra.HostScene.load_config(str(tmp_path / "c.xml"))
    o = s.raw()["objects"][0]
    assert o["type"] == ra.OBJ_PHONG and o["coeff"] == 9
    assert np.allclose(o["color"], (0.4, 0.5, 0.6)) and np.allclose(o["color2"], (0.3, 0.2, 0.1))
    s.close()
    (tmp_path / "s.xml").write_text((tmp_path / "s.xml").read_text().replace("<DiffuseComponent>", "<X>").replace("</DiffuseComponent>", "</X>"))
    with pytest.raises(ra.MirtError, match="GlossyComponent"):
        ra.HostScene.load_config(str(tmp_path / "c.xml"))


def test_nan_corner_is_refused(ra):
    """The reference's tree build orders vertex coordinates; with a NaN among them there is no order and no defined
    answer. The host surface says so instead of building something."""
    m = meshgen.cornell_with_meshes(1)
    verts = m["verts"].copy()
    verts[int(m["tris"][5][1]), 2] = np.nan
    with pytest.raises(ra.MirtError, match="NaN"):
        ra.HostScene.from_arrays(verts, m["tris"], m["tri_obj"], m["objects"], m["tri_vn"])
    verts[int(m["tris"][5][1]), 2] = np.inf  # infinities order fine
    ra.HostScene.from_arrays(verts, m["tris"], m["tri_obj"], m["objects"], m["tri_vn"]).close()
