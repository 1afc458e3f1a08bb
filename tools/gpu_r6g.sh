#!/bin/bash
# round 6: hd-32 attention backward with software-pipelined pair loops — parity, A/B against the round-5 loops (variant library pipe0)
set -u
out=gpurun_out/r6g; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_syncbn_direct.py -x -q -m gpu 2>&1 | grep -v "amdgpu.ids" | grep "^E  " | cut -c1-1200 | head -20 > $out/syncbn_fail.log
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_blocks_well_conditioned.py -x -q -m gpu -k "attn or attention or block" 2>&1 | grep -v "amdgpu.ids" | tail -4 | tee $out/tests.log
for lib in "" xpipe0 "" xpipe0; do echo "== lib=${lib:-release}"; SSL4GIE_DEBUG_LIB=$lib ATTN_CASES=mae.dec:256:197:16:32,det32:64:256:24:32,small32:256:100:16:32 timeout -k 10 200 python tools/attn_bench.py 2>&1 | grep -v "amdgpu\|DEBUG" | tee -a $out/attn_ab.log; done
python - <<'PY' 2>&1 | grep -v amdgpu | tee $out/bits.log
import os, subprocess, sys, torch
# bit-equality of the two libraries on the production shape
code = """
import torch, sys
sys.path.insert(0, %r)
from ssl4gie_amd import ops
torch.manual_seed(1)
B, N, H, hd = 64, 197, 16, 32
qkv = torch.randn(B, N, 3 * H * hd).bfloat16().cuda()
out, lse = ops.attn_fwd(qkv, B, N, H, hd)
dout = torch.randn(B, N, H * hd).bfloat16().cuda()
d = ops.attn_bwd(qkv, out, dout, lse, B, N, H, hd)
torch.save(d.cpu(), sys.argv[1])
""" % os.getcwd()
for lib, f in (("", "/tmp/a.pt"), ("xpipe0", "/tmp/b.pt")):
    subprocess.run([sys.executable, "-c", code, f], check=True, env=dict(os.environ, SSL4GIE_DEBUG_LIB=lib))
a, b = torch.load("/tmp/a.pt"), torch.load("/tmp/b.pt")
print("pipelined == round-5 loops, bit for bit:", bool(torch.equal(a, b)), "finite:", bool(torch.isfinite(a.float()).all()))
PY
bash tools/gpu_sweep.sh r6g "SSL4GIE_DEBUG_LIB=" "SSL4GIE_DEBUG_LIB=xpipe0"
