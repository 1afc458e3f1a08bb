#!/usr/bin/env python3
"""kernel_trace.csv (rocprofv3 --kernel-trace) -> GPU busy / idle accounting over the timed steps:
union of kernel intervals, total idle time, histogram of idle gaps, time with 2+ kernels resident."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# keep the last 60 % of the trace (timed steps; skips warm-up / allocation phases)
t_lo = ev[0][0] + int(0.4 * (ev[-1][1] - ev[0][0]))
ev = [e for e in ev if e[0] >= t_lo]
span = ev[-1][1] - ev[0][0]
busy = 0
gaps = []
cur_s, cur_e = ev[0][0], ev[0][1]
for s, e, _ in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in ev)
print(f"span {span/1e6:.2f} ms, busy(union) {busy/1e6:.2f} ms ({100*busy/span:.1f} %), sum of kernel durations {tot/1e6:.2f} ms "
      f"(overlap {100*(tot-busy)/span:.1f} % of span), idle {sum(gaps)/1e6:.2f} ms in {len(gaps)} gaps")
bins = [(0, 2), (2, 5), (5, 10), (10, 50), (50, 10**9)]
for lo, hi in bins:
    g = [x for x in gaps if lo * 1000 <= x < hi * 1000]
    print(f"  gaps {lo:>3}-{hi if hi < 10**9 else 'inf':>3} us: {len(g):5d}  total {sum(g)/1e6:.3f} ms")
