#!/usr/bin/env python3
"""Per-kernel register / scratch summary of a `-Rpass-analysis=kernel-resource-usage` dump.
usage: kernel_usage.py file.usage.txt [substring]"""
import re, sys
txt = open(sys.argv[1]).read()
key = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
rows = {}
for ln in txt.split("\n"):
    m = re.search(r"remark:\s+Function Name: (\S+)", ln)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\])?: (\S+)", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for k, v in rows.items():
    if key in k:
        print(f"{k[:70]:70s} vgpr {v.get('VGPRs','?'):>4s} agpr {v.get('AGPRs','?'):>3s} sgpr {v.get('TotalSGPRs','?'):>3s} "
              f"scratch {v.get('ScratchSize','?'):>4s} spill {v.get('VGPRs Spill','?'):>3s} occ {v.get('Occupancy','?')}")
