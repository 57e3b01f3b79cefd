"""mirt_render_pt_multi (include/mirt.h): sample shards over several contexts + one reduce, through the C-ABI. These
tests need ONE GPU: the contexts all sit on device 0, so the sharding arithmetic, the per-context threads and the
peer-copy + add reduce run on hardware here; on a multi-GPU node the same call spans the devices."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_two_contexts_equal_one(gpu, cornell, ra):
    cam = cornell.camera()
    gpu.upload(cornell.describe())
    spp = 7  # odd: shards of 4 and 3 samples
    ref, st1 = gpu.render_pt(cam, spp, seed=3)
    devs = [ra.Device(device=0) for _ in range(2)]
    for d in devs:
        d.upload(cornell.describe())
    img, st = ra.render_pt_multi(devs, cam, spp, seed=3)
    assert np.abs(img / spp - ref / spp).max() <= 1e-5 * max(1.0, float((ref / spp).max()))
    assert st.n_samples == st1.n_samples == cam.width * cam.height * spp and st.n_rays == st1.n_rays
    # a shard that starts in the middle of the sample range, and the same image by hand from two single renders
    a, _ = devs[0].render_pt(cam, 4, spp_begin=10, seed=3)
    b, _ = devs[1].render_pt(cam, 3, spp_begin=14, seed=3)
    both, _ = ra.render_pt_multi(devs, cam, 7, spp_begin=10, seed=3)
    assert np.array_equal(both, a + b)  # context order is the summation order
    # more contexts than samples: empty shards contribute nothing
    devs.append(ra.Device(device=0))
    devs[2].upload(cornell.describe())
    two, _ = ra.render_pt_multi(devs, cam, 2, seed=3)
    one, _ = gpu.render_pt(cam, 2, seed=3)
    assert np.abs(two - one).max() <= 1e-5 * max(1.0, float(one.max()))
    # the light-connection integrator through the same route
    pt_ref, _ = gpu.render_pttest(cam, 5, seed=2)
    pt_img, pst = ra.render_pt_multi(devs[:2], cam, 5, seed=2, pttest=True)
    assert np.abs(pt_img - pt_ref).max() <= 1e-5 * max(1.0, float(pt_ref.max())) and pst.n_shadow_rays > 0
    # errors: a context without a scene, the same context twice
    empty = ra.Device(device=0)
    with pytest.raises(ra.MirtError):
        ra.render_pt_multi([devs[0], empty], cam, 2)
    with pytest.raises(ra.MirtError):
        ra.render_pt_multi([devs[0], devs[0]], cam, 2)
    empty.close()
    for d in devs:
        d.close()


def test_host_surface_renders_over_several_contexts(cornell, ra):
    """IntersectionHIP::setDevices + PathTracerHIP::renderPixels (what `mirt_render --gpus N` does)"""
    assert ra.lib().mirt_device_count() >= 1
    one, st1 = cornell.render(spp=6, seed=4)
    ra.set_devices([0, 0, 0])
    try:
        three, st3 = cornell.render(spp=6, seed=4)
        assert np.abs(three - one).max() <= 1e-5 * max(1.0, float(one.max()))
        assert st3.n_samples == st1.n_samples and st3.n_rays == st1.n_rays
        # batch queries still go to the first context
        o = np.array([[0.0, 0.0, 2.0]], np.float32)
        d = np.array([[0.0, 0.0, -1.0]], np.float32)
        dist, tid, _, _ = cornell.fill_intersect(o, d, [-1], [-1])
        assert tid[0] >= 0 and dist[0] > 0
    finally:
        ra.set_devices([0])
    again, _ = cornell.render(spp=6, seed=4)
    assert np.array_equal(again, one)
