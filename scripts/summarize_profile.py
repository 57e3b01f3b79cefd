#!/usr/bin/env python3
"""Summarises a scripts/profile_gpu.sh output directory into one text file (for profiles/).
usage: summarize_profile.py gpurun_out/prof_r1 [traffic.json] > profiles/r1_summary.txt
With a second argument the HBM-side bytes per dispatch of every kernel are also written as JSON; bench.py reports
them as roofline.traffic. FETCH_SIZE on gfx950 is TCC_EA0_RDREQ x 64 B: half of the bytes of a wide streaming read
(one request per 128 B: the guide's x2) but exactly the bytes of a gather of 64-B records (one request per record),
which is what the traversal kernels mostly issue (profiles/r3_fetch_factor.txt, scripts/experiments/fetch_factor.hip).
A kernel mixes both shapes, so read bytes are given as a range: read_bytes = raw FETCH_SIZE (every request a 64-B
record), read_bytes_x2 = twice that (every request a 128-B streaming request; also what the fabric moves if it moves
whole 128-B lines per request)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

d = sys.argv[1]
def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("mirt::", "")
    return name[:70]
cmd = None
for line in open(os.path.join(d, "kt.log"), errors="replace") if os.path.exists(os.path.join(d, "kt.log")) else []:
    m = re.search(r"\[rocprofv3\] '(.*)' ::", line)
    if m:
        cmd = re.sub(r"/\S*/repo/", "", m.group(1))
print("# rocprofv3 summary of: %s (scripts/%s), directory %s" % (
    cmd or "python3 bench.py --steps N --warmup 1 --no-cpu --pt-spp 8 [--workload X]",
    "profile_pt.sh" if "ptprof" in d else "profile_gpu.sh", d))
avg_us = {}
for f in sorted(glob.glob(os.path.join(d, "kt", "*", "*_kernel_stats.csv")))[-1:]:
    print("\n## kernel trace --stats (per kernel: calls, total ms, average us, % of GPU time)")
    for r in csv.DictReader(open(f)):
        avg_us[short(r["Name"])] = float(r["AverageNs"]) / 1e3
        if float(r["Percentage"]) < 0.05: continue
        print("%-72s calls %5s total %10.3f ms avg %12.3f us  %6.2f%%" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
agg = defaultdict(lambda: defaultdict(float)); calls = defaultdict(lambda: defaultdict(int)); meta = {}
for f in glob.glob(os.path.join(d, "*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if "k_" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]] += 1
        meta[k] = (r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])
print("\n## PMC counters, mean per dispatch (separate --pmc passes)")
traffic = {}
for k in sorted(agg):
    m = meta[k]
    print("\n%s  [wg %s, LDS %s B, VGPR %s, SGPR %s, scratch %s]" % ((k,) + m))
    c = {n: agg[k][n] / calls[k][n] for n in agg[k]}
    for n in sorted(c): print("    %-26s %18.1f   (dispatches %d)" % (n, c[n], calls[k][n]))
    g = c.get
    if g("SQ_WAVE_CYCLES"):
        print("    -> wave-cycle split: active %.1f%%  wait_any %.1f%%  wait_inst %.1f%%" % (100 * g("SQ_ACTIVE_INST_ANY", 0) / g("SQ_WAVE_CYCLES"), 100 * g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES"), 100 * g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES")))
    if g("SQ_INSTS_VALU") and g("SQ_THREAD_CYCLES_VALU"):
        print("    -> VALU lane utilisation (thread-cycles / (insts*64 lanes... see guide)): SQ_THREAD_CYCLES_VALU/SQ_INSTS_VALU = %.2f" % (g("SQ_THREAD_CYCLES_VALU") / g("SQ_INSTS_VALU")))
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and (g("TCC_HIT_sum") + g("TCC_MISS_sum")) > 0:
        print("    -> L2 hit rate %.2f%%" % (100 * g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
    if g("FETCH_SIZE") is not None:
        print("    -> HBM-side read  : FETCH_SIZE %.1f KiB raw = %.3f MB if every request is a 64-B record gather, %.3f MB if every request is a 128-B streaming request (profiles/r3_fetch_factor.txt)" % (g("FETCH_SIZE"), g("FETCH_SIZE") * 1024 / 1e6, 2 * g("FETCH_SIZE") * 1024 / 1e6))
    if g("WRITE_SIZE") is not None:
        print("    -> HBM-side write : WRITE_SIZE %.1f KiB = %.3f MB" % (g("WRITE_SIZE"), g("WRITE_SIZE") * 1024 / 1e6))
    issue_frac = lanes = l2 = None
    if g("SQ_INSTS_VALU") and g("GRBM_GUI_ACTIVE"):
        # a wave64 VALU instruction occupies its SIMD (16 lanes) for 4 cycles; 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs
        issue_frac = g("SQ_INSTS_VALU") * 4.0 / (1024.0 * g("GRBM_GUI_ACTIVE") / 8.0)
        print("    -> VALU issue slots used: SQ_INSTS_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) = %.3f" % issue_frac)
    if g("SQ_INSTS_VALU") and g("SQ_THREAD_CYCLES_VALU"):
        lanes = g("SQ_THREAD_CYCLES_VALU") / g("SQ_INSTS_VALU")
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and (g("TCC_HIT_sum") + g("TCC_MISS_sum")) > 0:
        l2 = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        traffic[k] = {"read_bytes": g("FETCH_SIZE") * 1024, "read_bytes_x2": 2 * g("FETCH_SIZE") * 1024,
                      "write_bytes": g("WRITE_SIZE") * 1024,
                      "dispatches": calls[k]["FETCH_SIZE"], "avg_us": avg_us.get(k),
                      "valu_issue_frac": None if issue_frac is None else round(issue_frac, 4),
                      "lanes_active_of_64": None if lanes is None else round(lanes, 2),
                      "l2_hit_rate": None if l2 is None else round(l2, 4)}
if len(sys.argv) > 2:
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* / GRBM_GUI_ACTIVE / TCC_HIT,MISS (separate passes) and "
                         "--kernel-trace --stats (avg_us) of scripts/profile_gpu.sh; means per dispatch; read_bytes = raw "
                         "FETCH_SIZE (64-B gathers are counted exactly), read_bytes_x2 = doubled (streaming reads are counted "
                         "at half): profiles/r3_fetch_factor.txt", "kernels": traffic},
              open(sys.argv[2], "w"), indent=1)
