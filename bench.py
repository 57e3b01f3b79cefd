#!/usr/bin/env python3
"""bench.py — closest-hit throughput of the hot path on N MI355X GPUs (one process per GPU).

  python bench.py --gpus 1 --steps 10 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): the Cornell box "BallsWithGlossyWalls" (1 932 triangles) at
1024x1024; one STEP = one pass of the closest-hit kernel over a 16 Mi-ray batch already resident in
HBM: 4 Mi camera rays (1024^2 x 4 jittered samples, unconditional) + 12 Mi secondary rays captured
from bounce depths 1-3 of the wavefront path tracer (NoSelfIntersection predicate). With N > 1 every
rank runs its own batch (different sample indices): the path shards by ray with no data-path
collective, so scaling is "weak" and value = N x rays / max-over-ranks time.
A second, untimed-for-`value` section renders the same scene with the full wavefront path tracer,
sample-sharded over the ranks with one RCCL reduce of the radiance sums, and reports Msamples/s.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import renderer_amd as ra  # noqa: E402
from renderer_amd.multi_gpu import reduce_radiance_sums  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
PRIMARY_SPP = 4
N_SECONDARY = 12 << 20


def algorithmic_bytes(n_rays, n_box, n_tri):
    """36 B per ray in+out, 32 B per node-box test, 48 B per triangle test."""
    return 36 * n_rays + 32 * n_box + 48 * n_tri


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pt-spp", type=int, default=4096,
                    help="spp of the path-tracing section, whole job (BASELINE.json configs[2]: 1024x1024 x 4096 spp)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--workload", choices=("cornell", "mesh1m", "fog"), default="cornell",
                    help="cornell = BASELINE.json configs[1] (the headline); mesh1m = SURVEY.md C4, three "
                         "displaced icospheres (983 052 triangles) inside the Cornell walls; fog = SURVEY.md "
                         "C5, BallsInFog (homogeneous medium around the whole scene)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload == "cornell":
        scene = ra.HostScene.load_config(os.path.join(ROOT, "Data", "Config_Glossy1024.xml"))
        cam = scene.camera()
        scene_name = "BallsWithGlossyWalls"
    elif args.workload == "fog":
        scene = ra.HostScene.load_config(os.path.join(ROOT, "Data", "Config_Fog1024.xml"))
        cam = scene.camera()
        scene_name = "BallsInFog"
    else:
        from renderer_amd import meshgen  # procedural mesh generator
        scene = meshgen.mesh_scene(7)
        cam = ra.make_camera(1024, 1024, 700)
        scene_name = "DisplacedIcospheres(level 7 x3: diffuse / glossy 5 / glass 1.5)"
    dev = ra.Device(device=local)
    dev.upload(scene.describe())
    wh = cam.width * cam.height
    # a stream of our own: torch.cuda.Event only sees the stream it is recorded on, and a NULL
    # stream argument means "the context's stream" to mirt_intersect_dev
    tstream = torch.cuda.Stream()
    stream = C.c_void_p(tstream.cuda_stream)

    # ---- the ray batch, built on the device ------------------------------------------------------
    n_pri = wh * PRIMARY_SPP
    po = torch.empty((3, n_pri), dtype=torch.float32, device="cuda")
    pd = torch.empty_like(po)
    so = torch.empty((3, N_SECONDARY), dtype=torch.float32, device="cuda")
    sd = torch.empty_like(so)
    sa = torch.empty(N_SECONDARY, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    spp0 = rank * 64  # distinct sample indices per rank
    dev.generate_rays_dev(cam, spp0, PRIMARY_SPP, 1, po.data_ptr(), pd.data_ptr())
    n_sec = dev.capture_rays_dev(cam, 12, 1, 3, so.data_ptr(), sd.data_ptr(), sa.data_ptr(), N_SECONDARY,
                                 spp_begin=spp0, seed=1)
    if n_sec < N_SECONDARY:  # keep planes contiguous for the count actually captured
        so = so[:, :n_sec].contiguous()
        sd = sd[:, :n_sec].contiguous()
        sa = sa[:n_sec].contiguous()
    n_rays = n_pri + n_sec
    pt_ = torch.empty(n_pri, dtype=torch.float32, device="cuda")
    pi_ = torch.empty(n_pri, dtype=torch.int32, device="cuda")
    st_ = torch.empty(n_sec, dtype=torch.float32, device="cuda")
    si_ = torch.empty(n_sec, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()

    # ---- algorithmic work of this batch (instrumented launch, never timed) -------------------------
    pa = dev.intersect_audit_dev(po.data_ptr(), pd.data_ptr(), None, n_pri, ra.MODE_UNCONDITIONAL,
                                 pt_.data_ptr(), pi_.data_ptr())
    sa_ = dev.intersect_audit_dev(so.data_ptr(), sd.data_ptr(), sa.data_ptr(), n_sec, ra.MODE_NO_SELF,
                                  st_.data_ptr(), si_.data_ptr())
    hit_rate = float(((pi_ >= 0).sum() + (si_ >= 0).sum()).item()) / n_rays

    def step(events=None):
        if events is not None:
            events[0].record(tstream)
        dev.intersect_dev(po.data_ptr(), pd.data_ptr(), None, n_pri, ra.MODE_UNCONDITIONAL, pt_.data_ptr(),
                          pi_.data_ptr(), stream)
        if events is not None:
            events[1].record(tstream)
        dev.intersect_dev(so.data_ptr(), sd.data_ptr(), sa.data_ptr(), n_sec, ra.MODE_NO_SELF, st_.data_ptr(),
                          si_.data_ptr(), stream)
        if events is not None:
            events[2].record(tstream)

    for _ in range(args.warmup):
        step()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(evs[k])
    barrier()
    elapsed = time.perf_counter() - t0
    # ---- algorithmic work of the path that was timed: the product path instrumented (fast tree + verified
    # chains + re-run in mode 2; whichever path modes 0 / 1 settled on), never timed. pa / sa_ above are the
    # reference tree walked exactly, which the classifier statistics refer to.
    pb, ptr_ = dev.intersect_count_dev(po.data_ptr(), pd.data_ptr(), None, n_pri, ra.MODE_UNCONDITIONAL,
                                       pt_.data_ptr(), pi_.data_ptr())
    sb, str_ = dev.intersect_count_dev(so.data_ptr(), sd.data_ptr(), sa.data_ptr(), n_sec, ra.MODE_NO_SELF,
                                       st_.data_ptr(), si_.data_ptr())
    pri_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs])) if args.steps else 0.0
    sec_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in evs])) if args.steps else 0.0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    value = world * n_rays * args.steps / elapsed / 1e6 if args.steps else 0.0

    # ---- path tracing section: sample-sharded, one RCCL reduce of the radiance sums ----------------
    pt_spp = max(args.pt_spp, world)
    s_begin, s_count = ra.shard_samples(pt_spp, world, rank)
    acc = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
    dev.render_pt_dev(cam, acc.data_ptr(), min(s_count, 4), spp_begin=s_begin, seed=1)  # warm-up
    acc.zero_()
    barrier()
    t0 = time.perf_counter()
    pst = dev.render_pt_dev(cam, acc.data_ptr(), s_count, spp_begin=s_begin, seed=1)
    reduce_radiance_sums(acc, dst=0)
    barrier()
    pt_elapsed = time.perf_counter() - t0
    pt_stats = torch.tensor([pt_elapsed, float(pst.n_rays), pst.gpu_ms, pst.intersect_ms], dtype=torch.float64,
                            device="cuda")
    pt_max = pt_stats.clone()
    if world > 1:
        dist.all_reduce(pt_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(pt_stats, op=dist.ReduceOp.SUM)
    pt_elapsed = float(pt_max[0].item())
    pt_rays = float(pt_stats[1].item())
    image_mean = (acc / pt_spp).mean(dim=(0, 1)).tolist() if rank == 0 else None

    if rank == 0:
        out = {
            "metric": "closest_hit_Mrays_per_s",
            "value": round(value, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s_closest_hit_16Mi: %s (%d tris, %d nodes), 1024x1024; "
                            "%d camera rays (mode 0) + %d secondary rays of bounce depth 1-3 (mode 2, "
                            "NoSelfIntersection) per GPU per step, resident in HBM" % (args.workload, scene_name, scene.n_tris,
                                                                                     scene.n_nodes, n_pri, n_sec),
                "rays_per_gpu_per_step": n_rays,
                "hit_rate": round(hit_rate, 4),
                "box_tests_per_ray": round((pb + sb) / n_rays, 3),
                "tri_tests_per_ray": round((ptr_ + str_) / n_rays, 3),
                "reference_tree_box_tests_per_ray": round((pa["box"] + sa_["box"]) / n_rays, 3),
                "classifier": {"unproven_box_fraction": round((pa["ambiguous"] + sa_["ambiguous"]) / (pa["box"] + sa_["box"]), 5),
                               "proven_decisions_disagreeing_with_exact_test": pa["mismatch"] + sa_["mismatch"],
                               "exact_box_tests_per_ray": round((pa["walks"] + sa_["walks"]) / n_rays, 5)},
                "parallelism": "ray-sharded x%d, no collective" % world,
            },
        }
        # dominant kernel: k_intersect<NO_SELF> over the secondary batch
        sec_bytes = algorithmic_bytes(n_sec, sb, str_)
        ach = sec_bytes / (sec_ms * 1e-3) / 1e9 if sec_ms > 0 else 0.0
        # HBM-side bytes of one launch of this kernel on this batch: PMC counters cannot be read from
        # inside this process, so the figure comes from the committed rocprofv3 --pmc passes of this
        # same command (profiles/, scripts/profile_gpu.sh); null when that profile is of another batch
        traffic_bytes, traffic_src = None, None
        tj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_current_traffic.json")
        if os.path.exists(tj) and n_sec == 12582912 and args.workload == "cornell":
            ks = json.load(open(tj))["kernels"]
            # one closest-hit call in the predicate mode = fast-tree launch + k_verify + exact re-run
            parts = [v for n, v in sorted(ks.items())
                     if n.startswith("k_intersect<2, false, false, false, true") or n.startswith("k_verify<false, false")
                     or n.startswith("k_rerun<false")]
            if parts:
                traffic_bytes = int(sum(k["read_bytes"] + k["write_bytes"] for k in parts))
                traffic_src = ("profiles/r1_current_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, mean per "
                               "dispatch, summed over the %d kernels of one call)" % len(parts))
        out["roofline"] = {
            "bound": "hbm",
            "kernel": "one mirt_intersect_dev call in MODE_NO_SELF on the secondary batch (%d rays): k_intersect<NO_SELF, FAST> "
                      "over the fast tree (the dominant launch, ~90 %% of kernel_ms) + k_verify + k_rerun, the exact re-run "
                      "of the rays it rejected, one wave per ray" % n_sec,
            "achieved": round(ach, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": traffic_bytes,
            "traffic_source": traffic_src,
            "kernel_ms": round(sec_ms, 4),
            "algorithmic_bytes_per_launch": sec_bytes,
            "bytes_per_ray": round(sec_bytes / n_sec, 1),
            "compulsory_bytes_per_ray": 36,
            "primary_kernel_ms": round(pri_ms, 4),
            "primary_achieved_GBs": round(algorithmic_bytes(n_pri, pb, ptr_) / (pri_ms * 1e-3) / 1e9, 1) if pri_ms > 0 else 0.0,
        }
        out["path_tracing"] = {
            "metric": "path_traced_Msamples_per_s",
            "value": round(wh * pt_spp / pt_elapsed / 1e6, 2),
            "unit": "Msamples/s",
            "workload": scene_name + " 1024x1024, %d spp whole job, maxDepth 20, sample-sharded x%d%s"
                        % (pt_spp, world, ", RCCL reduce of 12 MiB fp32 sums" if world > 1 else ""),
            "Mrays_per_s": round(pt_rays / pt_elapsed / 1e6, 2),
            "rays_per_sample": round(pt_rays / (wh * pt_spp), 3),
            "seconds": round(pt_elapsed, 4),
            "image_mean_rgb": [round(x, 5) for x in image_mean],
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"], out["path_tracing"]["cpu_baseline"] = cpu_baseline(scene, cam, po, pd, so, sd, sa)
        print(json.dumps(out), flush=True)
    dev.close()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(scene, cam, po, pd, so, sd, sa):
    """The CPU oracle (a port of the reference algorithm, oracle/oracle.cpp) on the box's host cores,
    on a bounded sample of the same workload. Test infrastructure used as the baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc

    threads = min(16, os.cpu_count() or 1)  # the box's CPU share for one GPU is 16
    oracle = orc.OracleScene(scene.raw())
    n1, n2 = min(4 << 20, po.shape[1]), min(8 << 20, so.shape[1])  # ~20 s of CPU work on 16 threads
    o1, d1 = po[:, :n1].cpu().numpy().copy(), pd[:, :n1].cpu().numpy().copy()
    o2, d2 = so[:, :n2].cpu().numpy().copy(), sd[:, :n2].cpu().numpy().copy()
    a2 = sa[:n2].cpu().numpy().copy()
    t0 = time.perf_counter()
    oracle.intersect(o1, d1, None, 0, threads=threads)
    oracle.intersect(o2, d2, a2, 2, threads=threads)
    dt = time.perf_counter() - t0
    hit = {"value": round((o1.shape[1] + o2.shape[1]) / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads,
           "kind": "port",
           "sample": "first %d camera + first %d secondary rays of the same batch, reference traversal order "
                     "(both children, no pruning), OpenMP over rays, %.1f s wall" % (o1.shape[1], o2.shape[1], dt)}
    sight = max(1, int(cam.sight_dist) * 256 // int(cam.width))  # same field of view at 256x256
    small = ra.make_camera(256, 256, sight, position=tuple(cam.position), focus=tuple(cam.focus), up=tuple(cam.up))
    t0 = time.perf_counter()
    _, nr = oracle.render(small, 64, seed=1, threads=threads)
    dt = time.perf_counter() - t0
    pt = {"value": round(256 * 256 * 64 / dt / 1e6, 3), "unit": "Msamples/s", "cores": threads, "kind": "port",
          "sample": "same scene and view at 256x256 (sightDist %d) x 64 spp, OpenMP over pixels, %.1f s wall" % (sight, dt)}
    oracle.close()
    return hit, pt


if __name__ == "__main__":
    main()
