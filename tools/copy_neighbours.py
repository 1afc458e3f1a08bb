#!/usr/bin/env python3
"""kernel_trace.csv (rocprofv3 --kernel-trace) -> which kernels run right before / after every
__amd_rocclr_copyBuffer dispatch (device copies issued by torch / the runtime), as histograms."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])[:70], r.get("Queue_Id", "?")) for r in rows)
prev, nxt = collections.Counter(), collections.Counter()
n = 0
for i, (t, name, q) in enumerate(ev):
    if "copyBuffer" in name:
        n += 1
        j = i - 1
        while j >= 0 and "copyBuffer" in ev[j][1]:
            j -= 1
        k = i + 1
        while k < len(ev) and "copyBuffer" in ev[k][1]:
            k += 1
        prev[ev[j][1] if j >= 0 else "-"] += 1
        nxt[ev[k][1] if k < len(ev) else "-"] += 1
print(f"{n} copyBuffer dispatches of {len(ev)}")
print("== preceded by")
for k, c in prev.most_common(15):
    print(f"{c:5d}  {k}")
print("== followed by")
for k, c in nxt.most_common(15):
    print(f"{c:5d}  {k}")
