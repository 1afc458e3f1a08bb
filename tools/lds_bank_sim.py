#!/usr/bin/env python3
"""LDS bank-conflict calculator for gfx950 access patterns (MI355X_MICROARCH.md §LDS).

Given each lane's byte address for one wave-instruction, returns the LDS cycles it takes
(ideal = number of lane groups).  Used to validate / search the XOR swizzles in csrc/.
"""
import itertools

B128_GROUPS = [
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
    list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
    list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64)),
]
HALF_GROUPS = [list(range(0, 32)), list(range(32, 64))]


def cycles(addrs, width, groups, nbanks=64):
    total = 0
    for grp in groups:
        per_bank = {}
        for l in grp:
            a = addrs[l]
            for w in range(width // 4):
                bank = ((a // 4) + w) % nbanks
                per_bank.setdefault(bank, set()).add((a // 4) + w)
        total += max(len(v) for v in per_bank.values())
    return total


def b128_cycles(addrs):
    return cycles(addrs, 16, B128_GROUPS)


def tr_b16_cycles(addrs):  # ds_read_b64_tr_b16: 8 B per lane, 2 x 32 groups, 64 banks
    return cycles(addrs, 8, HALF_GROUPS)


def b64_cycles(addrs):
    return cycles(addrs, 8, HALF_GROUPS)


def row_read_addrs(row_bytes, swz, chunk0, row0=0):
    """MFMA 16x16x32 row-fragment read: lane l -> row l&15, logical chunk chunk0 + (l>>4)."""
    out = []
    for l in range(64):
        r = row0 + (l & 15)
        c = chunk0 + (l >> 4)
        out.append(r * row_bytes + ((c ^ swz(r)) << 4))
    return out


def tr_read_addrs(row_bytes, swz, krow0, x0, second=False):
    """ds_read_b64_tr_b16 fragment read (see tr_frag in gemm.hip): lane l, g=l>>4 ->
    row krow0 + 8g (+4) + q, logical chunk x0/8 + (p>>1), +8*(p&1)."""
    out = []
    for l in range(64):
        g, q, p = l >> 4, (l & 15) >> 2, l & 3
        r = krow0 + 8 * g + (4 if second else 0) + q
        c = (x0 >> 3) + (p >> 1)
        out.append(r * row_bytes + ((c ^ swz(r)) << 4) + ((p & 1) << 3))
    return out


def attn_tr_read_addrs(row_bytes, swz, key0, d0, second=False):
    """attention V / K^T tr read: lane (g,q,p) -> row key0 + (16 if second) + 4g + q,
    logical chunk d0/8 + (p>>1)."""
    out = []
    for l in range(64):
        g, q, p = l >> 4, (l & 15) >> 2, l & 3
        r = key0 + (16 if second else 0) + 4 * g + q
        c = (d0 >> 3) + (p >> 1)
        out.append(r * row_bytes + ((c ^ swz(r)) << 4) + ((p & 1) << 3))
    return out


if __name__ == "__main__":
    nt = lambda r: (r >> 1) & 7
    print("NT 128B rows, b128:", [b128_cycles(row_read_addrs(128, nt, c0, r0))
                                  for c0 in (0, 4) for r0 in (0, 16, 64)])
    tn = lambda k: ((k & 3) | ((k >> 1) & 4)) << 1
    print("TN 256B rows, tr:", [tr_b16_cycles(tr_read_addrs(256, tn, k0, x0, s))
                                for k0 in (0, 32) for x0 in (0, 16, 112) for s in (False, True)])


# ---- direct 3x3 convolution (csrc/conv_direct.hip): halo reads of the tap loop, weight reads, the
# staged output tile.  Geometry = (TB, TH, TW) with TB * TH * TW = 256; formulas mirror the kernel.
def direct_x_swizzle(tw, hy, hx):
    if tw == 8:
        return ((hx >> 2) & 1) | ((hy & 1) << 1)
    return ((hx >> 1) & 3) if tw == 16 else ((hx >> 2) & 3)


def direct_x_read_cycles(tb, th, tw):
    """worst ds_read_b128 cycles (ideal 4) of the forward kernel's halo reads over all waves, taps,
    operand halves and k-steps"""
    hh, hw = th + 2, tw + 2
    worst = 0
    for wave in range(4):
        for pb in range(2):
            for tap in range(9):
                dy, dx = divmod(tap, 3)
                for ks in range(2):
                    addrs = []
                    for lane in range(64):
                        p = (2 * wave + pb) * 32 + (lane & 31)
                        im, r, c = p // (th * tw), (p // tw) % th, p % tw
                        hy, hx = r + dy, c + dx
                        lin = (im * hh + hy) * hw + hx
                        addrs.append(lin * 64 + (((ks * 2 + (lane >> 5)) ^ direct_x_swizzle(tw, hy, hx)) << 4))
                    worst = max(worst, b128_cycles(addrs))
    return worst


def direct_w_read_cycles(ncb=2):
    worst = 0
    for n in range(ncb):
        for tap in range(9):
            for ks in range(2):
                addrs = []
                for lane in range(64):
                    row = n * 32 + (lane & 31)
                    cc = tap * 4 + ks * 2 + (lane >> 5)
                    addrs.append(row * 576 + ((cc ^ ((row >> 2) & 3)) << 4))
                worst = max(worst, b128_cycles(addrs))
    return worst


def direct_out_cycles(ncb):
    """(worst ds_write_b64 cycles, ideal 2; worst ds_read_b128 cycles, ideal 4) of the staged output"""
    rowb = ncb * 64
    slots = rowb // 8
    sm = slots - 1
    sw = lambda p: (p ^ (p >> 2)) & sm
    ww = wr = 0
    for wave in range(4):
        for pb in range(2):
            for n in range(ncb):
                for q in range(4):
                    addrs = []
                    for lane in range(64):
                        p = (2 * wave + pb) * 32 + (lane & 31)
                        slot = n * 8 + 2 * q + (lane >> 5)
                        addrs.append(p * rowb + ((slot ^ sw(p)) << 3))
                    ww = max(ww, b64_cycles(addrs))
    for i in range(slots // 2):
        for w in range(4):
            addrs = []
            for lane in range(64):
                idx = w * 64 + lane + i * 256
                p, j = idx // (slots // 2), idx % (slots // 2)
                addrs.append(p * rowb + ((j ^ (sw(p) >> 1)) << 4))
            wr = max(wr, b128_cycles(addrs))
    return ww, wr
