"""mirt_intersect_noalloc_dev (include/mirt.h): the closest-hit call whose scratch buffers come from the
library's static device array, for callers that cannot allocate. Through the C-ABI, against
mirt_intersect_dev on the same device buffers, bit for bit (ids and t), on two calls in a row, the second
with a smaller input: the array keeps the first call's queue words, list and stack levels."""
import numpy as np
import pytest

import meshgen

pytestmark = pytest.mark.gpu


def _pair(dev, ra, torch, o, d, aux, mode):
    """(t, id) of mirt_intersect_dev and of mirt_intersect_noalloc_dev on the same device buffers"""
    n = o.shape[1]
    to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    ta = torch.from_numpy(aux).cuda() if aux is not None else None
    out = []
    for fn in (dev.intersect_dev, dev.intersect_noalloc_dev):
        t = torch.full((n,), -7.0, dtype=torch.float32, device="cuda")
        i = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        fn(to.data_ptr(), td.data_ptr(), ta.data_ptr() if ta is not None else None, n, mode, t.data_ptr(), i.data_ptr())
        # the NULL stream argument means the context's own stream, which torch does not know: wait on the device
        torch.cuda.synchronize()
        out.append((t.cpu().numpy(), i.cpu().numpy()))
    return out


def test_noalloc_matches_intersect_dev(gpu, cornell, mesh20k, ra):
    import torch

    for scene, inside in ((cornell, 2), (mesh20k, 8)):
        gpu.upload(scene.describe())
        # first call: a large batch with adversarial rays mixed in (they fill the re-run list and use the deep
        # stack levels); second call: a smaller batch right after it
        o1, d1, a1 = meshgen.random_rays(1 << 20, 91, scene.n_objects, inside_obj=inside)
        d1[1, :50000] = 0.0  # exactly axis-parallel: outside the classifier's model, listed and re-run
        d1[:, :50000] /= np.maximum(np.linalg.norm(d1[:, :50000], axis=0, keepdims=True), 1e-30)
        o2, d2, a2 = meshgen.random_rays(100003, 92, scene.n_objects, inside_obj=inside)
        for mode in (ra.MODE_NO_SELF, ra.MODE_UNCONDITIONAL, ra.MODE_LAST_TID):
            for o, d, a in ((o1, d1, a1), (o2, d2, a2)):
                aux = None if mode == ra.MODE_UNCONDITIONAL else (a if mode == ra.MODE_NO_SELF else (a & 0x3FF))
                (t_ref, i_ref), (t_na, i_na) = _pair(gpu, ra, torch, o, d, aux, mode)
                assert (i_ref != -7).all() and (i_na != -7).all()
                assert np.array_equal(i_na, i_ref), "mode %d: %d ids differ" % (mode, (i_na != i_ref).sum())
                assert np.array_equal(t_na, t_ref)
                assert (i_ref >= 0).mean() > 0.5


def test_noalloc_limits(gpu, cornell, ra):
    import torch

    gpu.upload(cornell.describe())
    n = ra.NOALLOC_MAX_RAYS + 1
    buf = torch.zeros((3, n), dtype=torch.float32, device="cuda")
    t = torch.empty(n, dtype=torch.float32, device="cuda")
    i = torch.empty(n, dtype=torch.int32, device="cuda")
    with pytest.raises(ra.MirtError, match="MIRT_NOALLOC_MAX_RAYS"):
        gpu.intersect_noalloc_dev(buf.data_ptr(), buf.data_ptr(), None, n, ra.MODE_UNCONDITIONAL, t.data_ptr(), i.data_ptr())
    # the largest batch it takes, and an empty one
    o, d, a = meshgen.random_rays(ra.NOALLOC_MAX_RAYS, 93, cornell.n_objects, inside_obj=2)
    (t_ref, i_ref), (t_na, i_na) = _pair(gpu, ra, torch, o, d, a, ra.MODE_NO_SELF)
    assert np.array_equal(i_na, i_ref) and np.array_equal(t_na, t_ref)
    gpu.intersect_noalloc_dev(buf.data_ptr(), buf.data_ptr(), None, 0, ra.MODE_UNCONDITIONAL, t.data_ptr(), i.data_ptr())


def test_noalloc_allocates_nothing_and_caps_the_grid(cornell, ra):
    """The entry point's contract itself: on a FRESH context the first closest-hit call ever is the
    no-alloc one with the largest batch it takes, and the device's free memory must not move (mirt_intersect_dev would
    hipMalloc its list buffer at that point). Then the same with MIRT_LDS_LEVELS=1 at upload, which leaves almost every
    stack level in the static spill slice so that the persistent grid has to be cut down to fit its 32 MiB."""
    import os
    import torch

    o, d, a = meshgen.random_rays(ra.NOALLOC_MAX_RAYS, 94, cornell.n_objects, inside_obj=2)
    d[1, :30000] = 0.0  # rays for the re-run list
    d[:, :30000] /= np.maximum(np.linalg.norm(d[:, :30000], axis=0, keepdims=True), 1e-30)
    n = o.shape[1]
    to, td, ta = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), torch.from_numpy(a).cuda()
    t = torch.empty(n, dtype=torch.float32, device="cuda")
    i = torch.empty(n, dtype=torch.int32, device="cuda")
    ref_dev = ra.Device(device=0)
    ref_dev.upload(cornell.describe())
    t_ref = torch.empty_like(t)
    i_ref = torch.empty_like(i)
    ref_dev.intersect_dev(to.data_ptr(), td.data_ptr(), ta.data_ptr(), n, ra.MODE_NO_SELF, t_ref.data_ptr(), i_ref.data_ptr())
    torch.cuda.synchronize()
    for levels in (None, "1"):
        if levels:
            os.environ["MIRT_LDS_LEVELS"] = levels
        try:
            dev = ra.Device(device=0)
            dev.upload(cornell.describe())
        finally:
            os.environ.pop("MIRT_LDS_LEVELS", None)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        for mode, aux in ((ra.MODE_NO_SELF, ta.data_ptr()), (ra.MODE_UNCONDITIONAL, None)):
            dev.intersect_noalloc_dev(to.data_ptr(), td.data_ptr(), aux, n, mode, t.data_ptr(), i.data_ptr())
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0, "mirt_intersect_noalloc_dev changed the device's free memory"
        dev.intersect_noalloc_dev(to.data_ptr(), td.data_ptr(), ta.data_ptr(), n, ra.MODE_NO_SELF, t.data_ptr(), i.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(i, i_ref) and torch.equal(t, t_ref), levels
        dev.close()
    ref_dev.close()
