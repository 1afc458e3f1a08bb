#!/usr/bin/env python3
"""Condensed outline of one kernel's ISA (waits, barriers, MFMA, LDS, VMEM, branches).
usage: asm_outline.py file.s kernel_substring [max_lines]"""
import re, sys
src = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(\S*' + re.escape(key) + r'\S*):', src, re.M)
start = m.start()
end = src.index('.Lfunc_end', start)
out = []
for ln in src[start:end].split('\n'):
    s = ln.strip()
    if re.match(r'\.LBB', s):
        out.append(s.split()[0]); continue
    m = re.match(r'(s_waitcnt|s_barrier|global_load_lds\w*|v_mfma\w*|ds_read\w*|ds_write\w*|s_cbranch\w*|global_store\w*|global_load\w*|buffer_\w+|scratch_\w+|v_exp\w*)\b', s)
    if m:
        k = m.group(1)
        if k == 's_waitcnt' or k.startswith('s_cbranch'): k = s.split(';')[0].strip()
        out.append(k)
res = []; prev = None; cnt = 0
for o in out:
    if o == prev: cnt += 1
    else:
        if prev is not None: res.append(f"{prev} x{cnt}" if cnt > 1 else prev)
        prev = o; cnt = 1
res.append(f"{prev} x{cnt}")
n = int(sys.argv[3]) if len(sys.argv) > 3 else 200
print('\n'.join(res[:n]))
