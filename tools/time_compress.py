#!/usr/bin/env python3
"""Time only the compress kernel (no correctness check): for A/B experiments with diagnostic library builds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench, zig_lz4_amd as zl
dist = sys.argv[1] if len(sys.argv) > 1 else "text"
nblocks = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
dev = torch.device("cuda:0"); block = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
slot = (zl.compressBound(block) + 15) // 16 * 16
inp = bench.make_device_blocks(dist, nblocks, block, dev, seed=1)
ar = torch.arange(nblocks, dtype=torch.int64, device=dev)
in_len = torch.full((nblocks,), block, dtype=torch.int32, device=dev)
cap = torch.full((nblocks,), slot, dtype=torch.int32, device=dev)
comp = torch.empty(nblocks * slot, dtype=torch.uint8, device=dev)
res = torch.empty(nblocks, dtype=torch.int64, device=dev)
ts = []
for it in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); zl.batch_compress_fast(inp, ar * block, in_len, comp, ar * slot, cap, res, block, 1); e1.record()
    torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("%s %s blocks=%d  kernel ms %s  -> %.1f GiB/s  sum(csize)=%d" % (os.environ.get("ZLZ4_AMD_LIB", "default"), dist, nblocks,
      ["%.1f" % t for t in ts], nblocks * block / min(ts) * 1e3 / 2**30, int(res.clamp(min=0).sum())))
