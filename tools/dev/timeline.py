#!/usr/bin/env python3
"""Dev: compact timeline of the K1 kernels from a rocprofv3 --kernel-trace csv (start / end in ms relative to the first kernel of the window).
usage: timeline.py <dir> [last_n_kernels=60]"""
import csv, glob, os, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) > 3e5][-n:]   # kernels longer than 0.3 ms
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:22]
    print(f"{name:22s} q{r.get('Queue_Id','?'):>3s} {s:9.3f} -> {e:9.3f}  ({e - s:7.3f} ms) grid {r.get('Grid_Size','?')}")
