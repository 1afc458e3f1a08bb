#!/usr/bin/env python3
"""kernel_trace.csv of a few MAE steps (rocprofv3 --kernel-trace) -> per step: when the last data-gradient
kernel of the backward pass ends, when the last weight-gradient (TN / slab reduce) kernel ends, when the
optimizer kernel starts — i.e. how long the weight-gradient stream trails behind the main stream — and how busy
the chip is over the step (sum of kernel durations / step time)."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])) for r in rows)
opt = [i for i, e in enumerate(ev) if "adamw_arena_kernel" in e[2]]
for a, b in zip(opt[:-1], opt[1:]):
    seg = ev[a + 1:b]          # kernels between two optimizer launches = one step
    t0, t1 = ev[a][1], ev[b][0]
    tn = [e for e in seg if "gemm_bf16_tn" in e[2] or "slab_reduce" in e[2]]
    main = [e for e in seg if not ("gemm_bf16_tn" in e[2] or "slab_reduce" in e[2] or "copyBuffer" in e[2])]
    last_main = max(e[1] for e in main)
    last_tn = max(e[1] for e in tn)
    first_tn = min(e[0] for e in tn)
    busy = sum(e[1] - e[0] for e in seg)
    print(f"step {(t1 - t0) / 1e6:6.2f} ms (optimizer end -> next optimizer start): last main-stream kernel ends at "
          f"{(last_main - t0) / 1e6:6.2f}, last weight-gradient kernel at {(last_tn - t0) / 1e6:6.2f} "
          f"(first at {(first_tn - t0) / 1e6:5.2f}); sum of kernel time {busy / 1e6:6.2f} ms; "
          f"weight-gradient kernels {sum(e[1] - e[0] for e in tn) / 1e6:5.2f} ms")
