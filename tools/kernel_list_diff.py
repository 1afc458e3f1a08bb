#!/usr/bin/env python3
"""kernel-name-level difference of two rocprofv3 --kernel-trace --stats runs: usage kernel_list_diff.py <dirA> <dirB>
(each searched for *kernel_stats.csv).  Prints kernels present in only one run and kernels whose call counts differ."""
import csv, glob, os, re, sys


def load(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\s+", " ", r["Name"])[:110]
            out[name] = out.get(name, 0) + int(r["Calls"])
    return out


a, b = load(sys.argv[1]), load(sys.argv[2])
per = sys.argv[3] if len(sys.argv) > 3 else None   # a kernel launched once per step (e.g. lars_apply_kernel): normalise to launches per step
na = next((v for k, v in a.items() if per and per in k), 1)
nb = next((v for k, v in b.items() if per and per in k), 1)
print(f"# A = {sys.argv[1]} ({len(a)} kernels, {sum(a.values())} launches, {na} steps)  B = {sys.argv[2]} ({len(b)} kernels, {sum(b.values())} launches, {nb} steps)")
print("# launches per step:  A       B   kernel (only those that differ)")
for k in sorted(set(a) | set(b)):
    x, y = a.get(k, 0) / na, b.get(k, 0) / nb
    if abs(x - y) > 1e-9:
        print(f"{x:9.1f} {y:9.1f}  {k}")
