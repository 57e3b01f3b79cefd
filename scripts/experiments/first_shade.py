import csv, glob, sys, re
for d in sys.argv[1:]:
    f = sorted(glob.glob(d + "/*/*_kernel_trace.csv"))[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    out = []
    for i, r in enumerate(rows):
        if "k_generate" in r["Kernel_Name"]:
            for q in rows[i:i + 12]:
                if "k_shade" in q["Kernel_Name"]:
                    out.append((int(q["End_Timestamp"]) - int(q["Start_Timestamp"])) / 1e3); break
    print(d, "first k_shade after each camera pass (us):", [round(x, 1) for x in out])
