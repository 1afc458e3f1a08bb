#!/usr/bin/env python3
"""pmc_summary.txt (tools/gpu_pmc.sh) -> profiles/pmc_traffic.json: HBM bytes per launch of the
kernel kinds bench.py's `roofline` object reports.  FETCH_SIZE / WRITE_SIZE are in KiB per dispatch;
FETCH_SIZE is doubled (MI355X_MICROARCH.md §HBM: on gfx950 the counter sees half of the read
traffic), both are averaged over the dispatches of every kernel of a kind, weighted by count."""
import json
import re
import sys

KINDS = {"gemm_bf16_nt": ("gemm_bf16_nt256_kernel", "gemm_bf16_nt_kernel"),
         "gemm_bf16_tn": ("gemm_bf16_tn256k_kernel", "gemm_bf16_tn256_kernel", "gemm_bf16_tn_kernel"),
         "attn_fwd_bf16": ("attn_fwd_bf16_kernel", "attn_long_fwd_kernel"),
         "attn_bwd_bf16": ("attn_bwd_bf16_kernel", "attn_long_bwd_dq_kernel", "attn_long_bwd_dkv_kernel")}


def tree_hash():
    """the commit the counters were taken on (+dirty when the work tree differs), or None outside a git checkout"""
    import subprocess
    try:
        h = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
        dirty = subprocess.run(["git", "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True).stdout.strip()
        return h + ("+dirty" if dirty else "")
    except Exception:
        return None


def main(src, dst, note):
    rd = {k: 0.0 for k in KINDS}
    wr = {k: 0.0 for k in KINDS}
    cnt = {k: 0 for k in KINDS}
    for line in open(src):
        m = re.search(r"dispatches=(\d+)", line)
        if not m:
            continue
        n = int(m.group(1))
        name = line[:m.start()]
        kind = next((k for k, pats in KINDS.items() if any(p in name for p in pats)), None)
        if kind is None:
            continue
        f = re.search(r"FETCH_SIZE=([0-9.e+]+)", line)
        w = re.search(r"WRITE_SIZE=([0-9.e+]+)", line)
        if not f or not w:
            continue
        rd[kind] += 2.0 * float(f.group(1)) * 1024 * n
        wr[kind] += float(w.group(1)) * 1024 * n
        cnt[kind] += n
    out = {"source": note, "tree": tree_hash(),
           "hbm_bytes_per_launch": {k: int((rd[k] + wr[k]) / cnt[k]) for k in KINDS if cnt[k]},
           "read_bytes_per_launch": {k: int(rd[k] / cnt[k]) for k in KINDS if cnt[k]},
           "write_bytes_per_launch": {k: int(wr[k] / cnt[k]) for k in KINDS if cnt[k]},
           "dispatches_seen": {k: cnt[k] for k in KINDS if cnt[k]}}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
