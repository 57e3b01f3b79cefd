"""World size 2 with the PRODUCT: two fresh processes, ranks 0 / 1 of a gloo group, both on GPU 0.
Each rank renders its shard with Device.render_*_dev (sample shards, then image tiles; PathTracer and PathTracerTest) and
the collective of renderer_amd.multi_gpu combines them on rank 0. On the 8-GPU node the same code runs one rank per GPU
over RCCL (bench.py); here the collective is gloo because both ranks share the one GPU of the box."""
import json
import os

import numpy as np
import pytest

import dist_worker
from conftest import rmse

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_render_shards_and_combine(tmp_path, gpu, cornell, cornell_oracle, ra):
    import torch

    spp = 7  # odd: sample shards of 4 and 3
    cfg = os.path.join(ROOT, "Data", "Config_Glossy128.xml")
    codes = dist_worker.run_ranks(dist_worker.gpu_render_rank, 2, (cfg, spp, str(tmp_path)))
    assert codes == [0, 0], codes
    cam = cornell.camera()
    gpu.upload(cornell.describe())
    rays = json.load(open(tmp_path / "rays.json"))
    for pttest in (False, True):
        name = "pttest" if pttest else "pt"
        one, st = (gpu.render_pttest if pttest else gpu.render_pt)(cam, spp, seed=3)
        by_samples = np.load(tmp_path / (name + "_samples.npy"))
        by_tiles = np.load(tmp_path / (name + "_tiles.npy"))
        # sample shards: equal up to the fp32 order of the cross-rank sum; tiles: no arithmetic at all
        assert np.abs(by_samples - one).max() <= 1e-5 * max(1.0, float(one.max())), name
        assert np.array_equal(by_tiles, one), name
        for shard in ("samples", "tiles"):
            assert rays["%s_%s" % (name, shard)] == [int(st.n_rays), int(st.n_samples)], (name, shard)
    # ... and the two-rank image against the oracle, by the bounds of tests/test_gpu_pt.py::_parity
    c, _ = cornell_oracle.render(cam, spp, seed=3, threads=16)
    c2, _ = cornell_oracle.render(cam, spp, seed=103, threads=16)
    g = np.load(tmp_path / "pt_samples.npy")
    noise = rmse(c / spp, c2 / spp)
    assert rmse(g / spp, c / spp) <= 0.02 * noise
    assert rmse(g / spp, c2 / spp) <= 1.15 * noise
    torch.cuda.synchronize()


def test_bench_two_ranks_one_json_line(tmp_path):
    """bench.py --gpus 2 (the driver's multi-GPU command shape), ranks over gloo on the one GPU: one JSON line from
    rank 0 with n_gpus 2, a whole-job value and the sample-sharded path-tracing section"""
    out = tmp_path / "bench2.json"
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--pt-spp", "32", "--no-cpu", "--workload", "cornell",
            "--backend", "gloo"]
    codes = dist_worker.run_ranks(dist_worker.bench_rank, 2, (argv, str(out)))
    assert codes == [0, 0], codes
    lines = [ln for ln in open(out).read().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    # (two ranks share the GPU here, so the committed one-rank profile is stale for this run: frac may be null, never > 1)
    assert rec["roofline"]["bound"] == "valu_issue" and (rec["roofline"]["frac"] is None or rec["roofline"]["frac"] <= 1.0)
    assert "cpu_baseline" not in rec
    assert rec["path_tracing"]["value"] > 0 and "x2" in rec["path_tracing"]["workload"]
