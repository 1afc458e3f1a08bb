#!/usr/bin/env python3
"""HBM traffic of the split-K weight-gradient (TN) launches of the MAE ViT-B step, predicted from the launch
geometry alone: the tile walk and xcd_remap of csrc/gemm_tn256.hip (consecutive logical workgroups share an XCD
and its L2: a chunk that holds r m-tiles and c n-tiles of one K-split reads r + c operand slabs), plus the fp32
slabs (written by the GEMM, read and written once more by slab_reduce_kernel).  Prints, per paired launch, the
prediction for the shipped split count (SSL4GIE_TN_FILL = 75 % of 240 CUs) and for every other split count.
Checked against the PMC measurement: 405 MB per launch predicted, 397 MB measured (profiles/pmc_traffic.json)."""
import math


def model(prods, K, cus=240, fill=0.75, splits=None):
    tiles = []
    for p, (M, N) in enumerate(prods):
        tm, tn = math.ceil(M / 256), math.ceil(N / 256)
        for t in range(tm * tn):  # the shorter tile dimension is the inner one (gemm_tn256.hip)
            tiles.append((p, t % tm, t // tm) if tm < tn else (p, t // tn, t % tn))
    nt, nkt = len(tiles), K // 64
    if splits is None:
        target = int(cus * fill) if len(prods) > 1 else cus
        splits = max(1, min((target + nt // 2) // nt, nkt // 8, 64))
    G = nt * splits
    q, r = divmod(G, 8)
    lo, slabs_read = 0, 0
    for x in range(8):  # xcd_remap: 8 contiguous chunks of logical ids
        hi = lo + q + (1 if x < r else 0)
        seen = set()
        for lid in range(lo, hi):
            s, (p, mt, nt_) = lid // nt, tiles[lid % nt]
            seen.add((s, p, "A", mt)); seen.add((s, p, "B", nt_))
        slabs_read += len(seen)
        lo = hi
    reads = slabs_read * 256 * (K / splits) * 2
    ideal = sum(M + N for M, N in prods) * K * 2
    out = sum(M * N for M, N in prods) * 4
    slab = out * splits * 2 + out if splits > 1 else out
    return reads, ideal, splits, slab, G


def main():
    enc, dec = 12800, 50432
    cases = {"enc qkv+proj": ([(2304, 768), (768, 768)], enc, 12), "enc fc1+fc2": ([(3072, 768), (768, 3072)], enc, 12),
             "dec qkv+proj": ([(1536, 512), (512, 512)], dec, 8), "dec fc1+fc2": ([(2048, 512), (512, 2048)], dec, 8)}
    tot = n_launch = 0
    for name, (prods, K, n) in cases.items():
        rb, ideal, s, slab, G = model(prods, K)
        print(f"{name:13s} shipped: {s:2d} splits, {G:3d} workgroups: operands {rb / 1e6:6.1f} MB (x{rb / ideal:.2f} of "
              f"{ideal / 1e6:.1f}) + slabs / output {slab / 1e6:6.1f} MB = {(rb + slab) / 1e6:6.1f} MB")
        tot += n * (rb + slab); n_launch += n
        row = []
        for ss in range(1, 17):
            if K // 64 // ss < 8:
                break
            rb2, _, _, sl2, G2 = model(prods, K, splits=ss)
            row.append(f"{ss}:{(rb2 + sl2) / 1e6:.0f}({G2})")
        print("    MB per launch by split count (workgroups): " + " ".join(row))
    print(f"per launch, over the {n_launch} paired launches of a step: {tot / n_launch / 1e6:.0f} MB")


if __name__ == "__main__":
    main()
