import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The world-size-2 tests run their ranks in fresh processes. Those are forked by multiprocessing's fork server,
    # which is started HERE, before anything in this process has touched the GPU: a process that holds the GPU must
    # not exec another program (and the fork server's children are plain forks of a clean interpreter, no exec).
    import multiprocessing as mp
    from multiprocessing import forkserver

    try:
        mp.set_start_method("forkserver")
    except RuntimeError:
        pass
    forkserver.ensure_running()


DATA = os.path.join(ROOT, "Data")


@pytest.fixture(scope="session")
def ra():
    import renderer_amd

    renderer_amd.lib()
    return renderer_amd


@pytest.fixture(scope="session")
def orc():
    import orc as _orc

    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def cornell(ra):
    """BallsWithGlossyWalls at 128x128: diffuse, glossy, mirror, glass, emitter."""
    s = ra.HostScene.load_config(os.path.join(DATA, "Config_Glossy128.xml"))
    yield s
    s.close()


@pytest.fixture(scope="session")
def cornell_oracle(cornell, orc):
    o = orc.OracleScene(cornell.raw())
    yield o
    o.close()


@pytest.fixture(scope="session")
def mesh20k(ra):
    import meshgen

    s = meshgen.mesh_scene(4)  # 3 x 5120 + 12 triangles
    yield s
    s.close()


@pytest.fixture(scope="session")
def mesh20k_oracle(mesh20k, orc):
    o = orc.OracleScene(mesh20k.raw())
    yield o
    o.close()


@pytest.fixture(scope="session")
def gpu(ra):
    """One MirtContext on device 0; fails loudly (no skip, no CPU fallback) without a GPU."""
    dev = ra.Device(device=0)
    yield dev
    dev.close()


def rmse(a, b):
    """compare.cpp:89-119 of the reference: clamp to [0,1], sqrt(mean over pixels of |d rgb|^2)"""
    d = np.clip(a, 0, 1) - np.clip(b, 0, 1)
    return float(np.sqrt((d * d).sum(-1).mean()))
