"""Prints the kernel timeline of the LAST wavefront in a rocprofv3 --kernel-trace csv (start offset, duration, gap)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last k_generate starts the last wavefront
gi = max(i for i, r in enumerate(rows) if "k_generate" in r["Kernel_Name"])
t0 = int(rows[gi]["Start_Timestamp"]); prev_end = t0
import re
for r in rows[gi:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("mirt::", "")[:48]
    print("%9.1f us  dur %8.1f  gap %6.1f  grid %8s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "?")), name))
    prev_end = e
print("total %.1f us" % ((prev_end - t0) / 1e3))
