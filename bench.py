#!/usr/bin/env python3
"""bench.py — closest-hit throughput of the hot path on N MI355X GPUs (one process per GPU).

  python bench.py --gpus 1 --steps 10 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json configs[1]): the Cornell box "BallsWithGlossyWalls" (1 932 triangles) at
1024x1024; one STEP = one pass of the closest-hit service over a 16 Mi-ray batch already resident in HBM:
4 Mi camera rays (1024^2 x 4 jittered samples, unconditional) + 12 Mi secondary rays captured from bounce
depths 1-3 of the wavefront path tracer (NoSelfIntersection predicate). With N > 1 every rank runs its own batch
(different sample indices): the path shards by ray with no data-path collective, so scaling is "weak" and
value = N x rays / max-over-ranks time.

The ONE JSON line rank 0 prints carries the whole metric ("Cornell box, 1M-tri mesh" + the fog scene):
  top level       the Cornell batch: value, roofline, cpu_baseline (BASELINE.json configs[1])
  path_tracing    the same scene through the full wavefront path tracer, 1024^2 x 4096 spp, sample-sharded over the
                  ranks with one RCCL reduce of the radiance sums (configs[2]); cpu_baseline at configs[0]'s
                  512^2 x 64 spp
  mesh1m          the same 16 Mi-ray batch shape on three displaced icospheres, 983 052 triangles (configs[3])
  fog             the same on BallsInFog (homogeneous medium), plus its path-traced Msamples/s (configs[4])
--workload cornell|mesh1m|fog runs one section alone (what the rocprofv3 passes under profiles/ are taken from).
Each section's oracle leg (the CPU baseline, N = 1 only) also checks the GPU's answers on its sample.
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
PROFILE_ROUND = "r2"


def algorithmic_bytes(n_rays, n_box, n_tri):
    """36 B per ray in+out, 32 B per node-box test, 48 B per triangle test."""
    return 36 * n_rays + 32 * n_box + 48 * n_tri


def build_scene(workload):
    if workload == "cornell":
        scene = ra.HostScene.load_config(os.path.join(ROOT, "Data", "Config_Glossy1024.xml"))
        return scene, scene.camera(), "BallsWithGlossyWalls"
    if workload == "fog":
        scene = ra.HostScene.load_config(os.path.join(ROOT, "Data", "Config_Fog1024.xml"))
        return scene, scene.camera(), "BallsInFog"
    from renderer_amd import meshgen  # procedural mesh generator

    scene = meshgen.mesh_scene(7)
    return scene, ra.make_camera(1024, 1024, 700), "DisplacedIcospheres(level 7 x3: diffuse / glossy 5 / glass 1.5)"


def committed_profile(workload):
    """HBM-side bytes and issue-slot figures of one closest-hit call in the predicate mode on this workload's batch,
    from the committed rocprofv3 passes of `bench.py --workload X` (PMC counters cannot be read from inside this
    process): profiles/<round>_<workload>_traffic.json, written by scripts/summarize_profile.py."""
    path = os.path.join(ROOT, "profiles", "%s_%s_traffic.json" % (PROFILE_ROUND, workload))
    if not os.path.exists(path):
        return None, None
    return json.load(open(path)), os.path.relpath(path, ROOT)


def closest_hit_section(workload, args, rank, world, local, barrier, tstream, want_cpu):
    """Builds the 16 Mi-ray batch of `workload` on the device, times args.steps passes of the closest-hit service
    over it and returns (section dict, objects the caller still needs)."""
    t_build = time.perf_counter()
    scene, cam, scene_name = build_scene(workload)
    build_s = time.perf_counter() - t_build
    dev = ra.Device(device=local)
    dev.upload(scene.describe())
    wh = cam.width * cam.height
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

    # ---- the reference tree walked exactly, instrumented (never timed): classifier statistics -----------
    pa = dev.intersect_audit_dev(po.data_ptr(), pd.data_ptr(), None, n_pri, ra.MODE_UNCONDITIONAL,
                                 pt_.data_ptr(), pi_.data_ptr())
    sa_ = dev.intersect_audit_dev(so.data_ptr(), sd.data_ptr(), sa.data_ptr(), n_sec, ra.MODE_NO_SELF,
                                  st_.data_ptr(), si_.data_ptr())

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

    # (at least one untimed pass: the first large call on a scene grows the library's list buffer once)
    for _ in range(max(args.warmup, 1)):
        step()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(evs[k])
    barrier()
    elapsed = time.perf_counter() - t0
    hit_rate = float(((pi_ >= 0).sum() + (si_ >= 0).sum()).item()) / n_rays  # of the timed passes' own answers
    # ---- algorithmic work of the route that was timed: the product path instrumented (fast tree + verified
    # chains + re-run, in every mode), never timed.
    ct = torch.empty(max(n_pri, n_sec), dtype=torch.float32, device="cuda")
    ci = torch.empty(max(n_pri, n_sec), dtype=torch.int32, device="cuda")
    pb, ptr_ = dev.intersect_count_dev(po.data_ptr(), pd.data_ptr(), None, n_pri, ra.MODE_UNCONDITIONAL,
                                       ct.data_ptr(), ci.data_ptr())
    sb, str_ = dev.intersect_count_dev(so.data_ptr(), sd.data_ptr(), sa.data_ptr(), n_sec, ra.MODE_NO_SELF,
                                       ct.data_ptr(), ci.data_ptr())
    del ct, ci
    pri_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs])) if args.steps else 0.0
    sec_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in evs])) if args.steps else 0.0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    value = world * n_rays * args.steps / elapsed / 1e6 if args.steps else 0.0

    sec = {
        "metric": "closest_hit_Mrays_per_s",
        "value": round(value, 2),
        "unit": "Mrays/s",
        "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
        "config": {
            "workload": "%s_closest_hit_16Mi: %s (%d tris, %d nodes), 1024x1024; "
                        "%d camera rays (mode 0) + %d secondary rays of bounce depth 1-3 (mode 2, "
                        "NoSelfIntersection) per GPU per step, resident in HBM" % (workload, scene_name, scene.n_tris,
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
            "host_scene_build_s": round(build_s, 2),
        },
    }
    # ---- roofline of the dominant launch sequence: one closest-hit call in the predicate mode ---------------
    sec_bytes = algorithmic_bytes(n_sec, sb, str_)
    ach = sec_bytes / (sec_ms * 1e-3) / 1e9 if sec_ms > 0 else 0.0
    prof, prof_path = committed_profile(workload)
    traffic_bytes, traffic_src, issue = None, None, None
    if prof and n_sec == N_SECONDARY:
        ks = prof["kernels"]
        # one closest-hit call in the predicate mode = fast-tree launch + k_verify + exact re-run
        names = [n for n in sorted(ks) if n.startswith("k_intersect<2, false, false, false, true")
                 or n.startswith("k_verify<false, false") or n.startswith("k_rerun<false, 2")]
        if names:
            traffic_bytes = int(sum(ks[n]["read_bytes"] + ks[n]["write_bytes"] for n in names))
            traffic_src = ("%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, mean per dispatch, summed over the %d "
                           "kernels of one call)" % (prof_path, len(names)))
            main = ks[names[0]]
            if main.get("valu_issue_frac") is not None:
                issue = {"kernel": names[0], "valu_issue_frac": main["valu_issue_frac"],
                         "lanes_active_of_64": main["lanes_active_of_64"], "l2_hit_rate": main.get("l2_hit_rate"),
                         "definition": "valu_issue_frac = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE per "
                                       "XCD); lanes_active_of_64 = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU (as "
                                       "scripts/summarize_profile.py prints them)",
                         "source": prof_path}
    sec["roofline"] = {
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
        # the bound that actually binds: the scene is cache-resident, the kernel saturates VALU issue under divergence
        "issue": issue,
        "binding": ("valu_issue: the tree and triangles are served from L1/L2 (traffic << algorithmic bytes) and the dominant "
                    "kernel fills the VALU issue slots at a fraction of the lanes (see `issue`); the HBM figure above is "
                    "the contract's algorithmic-bytes rate, not a physical limit on this workload"),
    }
    if want_cpu:
        sec["cpu_baseline"], chk = cpu_closest_hit(scene, po, pd, so, sd, sa, pt_, pi_, st_, si_)
        sec["config"].update(chk)
    return sec, (scene, cam, scene_name, dev)


def path_tracing_section(dev, cam, scene_name, pt_spp, rank, world, barrier):
    wh = cam.width * cam.height
    pt_spp = max(pt_spp, world)
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
    if rank != 0:
        return None
    image_mean = (acc / pt_spp).mean(dim=(0, 1)).tolist()
    return {
        "metric": "path_traced_Msamples_per_s",
        "value": round(wh * pt_spp / pt_elapsed / 1e6, 2),
        "unit": "Msamples/s",
        "workload": scene_name + " %dx%d, %d spp whole job, maxDepth 20, sample-sharded x%d%s"
                    % (cam.width, cam.height, pt_spp, world,
                       ", RCCL reduce of %d MiB fp32 sums" % (wh * 12 >> 20) if world > 1 else ""),
        "Mrays_per_s": round(pt_rays / pt_elapsed / 1e6, 2),
        "rays_per_sample": round(pt_rays / (wh * pt_spp), 3),
        "seconds": round(pt_elapsed, 4),
        "intersect_share_of_gpu_time": round(float(pt_stats[3].item()) / max(float(pt_stats[2].item()), 1e-9), 3),
        "image_mean_rgb": [round(x, 5) for x in image_mean],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pt-spp", type=int, default=4096,
                    help="spp of the path-tracing section, whole job (BASELINE.json configs[2]: 1024x1024 x 4096 spp)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--workload", choices=("all", "cornell", "mesh1m", "fog"), default="all",
                    help="all = the Cornell headline + path tracing, then the mesh1m and fog sections; one name = that "
                         "section alone as the top level (what profiles/ are taken from): cornell = BASELINE.json "
                         "configs[1]; mesh1m = SURVEY.md C4, three displaced icospheres (983 052 triangles) inside the "
                         "Cornell walls; fog = SURVEY.md C5, BallsInFog (homogeneous medium around the whole scene)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the "
                                                      "multi-rank flow on a box with fewer GPUs than ranks)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if args.backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)  # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # a stream of our own: torch.cuda.Event only sees the stream it is recorded on, and a NULL
    # stream argument means "the context's stream" to mirt_intersect_dev
    tstream = torch.cuda.Stream()
    want_cpu = world == 1 and not args.no_cpu
    first = "cornell" if args.workload == "all" else args.workload
    head, (scene, cam, scene_name, dev) = closest_hit_section(first, args, rank, world, local, barrier, tstream, want_cpu)
    out = {
        "metric": head["metric"],
        "value": head["value"],
        "unit": head["unit"],
        "n_gpus": world,
        "steps": args.steps,
        "warmup": max(args.warmup, 1),
        "ms_per_step": head["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": head["config"],
        "roofline": head["roofline"],
    }
    if "cpu_baseline" in head:
        out["cpu_baseline"] = head["cpu_baseline"]
    pt = path_tracing_section(dev, cam, scene_name, args.pt_spp if first == "cornell" else max(args.pt_spp // 8, world),
                              rank, world, barrier)
    if rank == 0:
        out["path_tracing"] = pt
        if want_cpu and first == "cornell":
            out["path_tracing"]["cpu_baseline"] = cpu_path_tracing(scene, cam)
    dev.close()
    scene.close()
    if args.workload == "all":
        for w in ("mesh1m", "fog"):
            torch.cuda.empty_cache()
            sec, (scene, cam, scene_name, dev) = closest_hit_section(w, args, rank, world, local, barrier, tstream, want_cpu)
            # fog: configs[4], the medium's path tracing (distance + phase sampling inside the wavefront loop);
            # mesh1m: the metric names path-traced samples/s on the 1 M-triangle mesh too
            if w in ("fog", "mesh1m"):
                fpt = path_tracing_section(dev, cam, scene_name, max(args.pt_spp // 8, world), rank, world, barrier)
                if rank == 0:
                    sec["path_tracing"] = fpt
            dev.close()
            scene.close()
            if rank == 0:
                out[w] = sec
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _oracle(scene):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc

    return orc.OracleScene(scene.raw())


def cpu_closest_hit(scene, po, pd, so, sd, sa, pt_, pi_, st_, si_):
    """The CPU oracle (a port of the reference algorithm, oracle/oracle.cpp) on the box's host cores, on a bounded
    sample of the same batch; its answers also check the GPU's on that sample (outside every timed region).
    Test infrastructure used as the baseline and the checker only."""
    threads = min(16, os.cpu_count() or 1)  # the box's CPU share for one GPU is 16
    oracle = _oracle(scene)
    big = scene.n_tris > 100000  # slower per ray on the 1 M-triangle tree: a smaller sample
    n1, n2 = min((1 << 20) if big else (4 << 20), po.shape[1]), min((3 << 20) if big else (12 << 20), so.shape[1])
    o1, d1 = po[:, :n1].cpu().numpy().copy(), pd[:, :n1].cpu().numpy().copy()
    o2, d2 = so[:, :n2].cpu().numpy().copy(), sd[:, :n2].cpu().numpy().copy()
    a2 = sa[:n2].cpu().numpy().copy()
    t0 = time.perf_counter()
    t1c, i1c = oracle.intersect(o1, d1, None, 0, threads=threads)
    t2c, i2c = oracle.intersect(o2, d2, a2, 2, threads=threads)
    dt = time.perf_counter() - t0
    oracle.close()
    hit = {"value": round((n1 + n2) / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": "first %d camera + first %d secondary rays of the same batch, reference traversal order "
                     "(both children, no pruning), OpenMP over rays, %.1f s wall" % (n1, n2, dt)}
    # ---- the GPU's answers on the same rays (the timed passes' outputs) against the oracle's ---------------
    mism, rel = 0, 0.0
    for tg, ig, tc, ic in ((pt_[:n1].cpu().numpy(), pi_[:n1].cpu().numpy(), t1c, i1c),
                           (st_[:n2].cpu().numpy(), si_[:n2].cpu().numpy(), t2c, i2c)):
        mism += int((ig != ic).sum())
        both = (ig == ic) & (ic >= 0)
        if both.any():
            rel = max(rel, float((np.abs(tg[both].astype(np.float64) - tc[both]) / np.maximum(tc[both], 1e-30)).max()))
    chk = {"oracle_check": {"rays_compared": n1 + n2, "id_mismatches": mism, "t_max_rel_err": rel,
                            "what": "hit ids / t of the timed passes vs the CPU oracle on the cpu_baseline sample "
                                    "(tests/test_gpu_fullbatch.py compares whole batches and classifies mismatches)"}}
    return hit, chk


def cpu_path_tracing(scene, cam):
    """BASELINE.json configs[0]: the scene at 512x512 x 64 spp (same field of view) through the oracle's restatement
    of the reference's CPU path tracer, OpenMP over pixels."""
    threads = min(16, os.cpu_count() or 1)
    oracle = _oracle(scene)
    res = 512
    sight = max(1, int(cam.sight_dist) * res // int(cam.width))
    small = ra.make_camera(res, res, sight, position=tuple(cam.position), focus=tuple(cam.focus), up=tuple(cam.up))
    t0 = time.perf_counter()
    oracle.render(small, 64, seed=1, threads=threads)
    dt = time.perf_counter() - t0
    oracle.close()
    return {"value": round(res * res * 64 / dt / 1e6, 3), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": "same scene and view at %dx%d (sightDist %d) x 64 spp = BASELINE.json configs[0], OpenMP over "
                      "pixels, %.1f s wall" % (res, res, sight, dt)}


if __name__ == "__main__":
    main()
