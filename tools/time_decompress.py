#!/usr/bin/env python3
"""Time only the decompress kernel on a pre-compressed batch (A/B experiments via env knobs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench, zig_lz4_amd as zl
dist = sys.argv[1] if len(sys.argv) > 1 else "text"
nblocks = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0"); block = 65536
slot = (zl.compressBound(block) + 15) // 16 * 16
inp = bench.make_device_blocks(dist, nblocks, block, dev, seed=1)
ar = torch.arange(nblocks, dtype=torch.int64, device=dev)
in_len = torch.full((nblocks,), block, dtype=torch.int32, device=dev)
cap = torch.full((nblocks,), slot, dtype=torch.int32, device=dev)
comp = torch.empty(nblocks * slot, dtype=torch.uint8, device=dev)
res = torch.empty(nblocks, dtype=torch.int64, device=dev)
zl.batch_compress_fast(inp, ar * block, in_len, comp, ar * slot, cap, res, block, 1)
torch.cuda.synchronize()
clen = res.to(torch.int32)
out = torch.empty_like(inp); ds = torch.empty(nblocks, dtype=torch.int64, device=dev)
ts = []
for it in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); zl.batch_decompress_safe(comp, ar * slot, clen, out, ar * block, in_len, ds); e1.record()
    torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ok = bool((ds == block).all()) and torch.equal(out, inp)
print("lanes=%s %s blocks=%d  ms %s -> %.1f GiB/s  roundtrip_ok=%s" % (os.environ.get("ZLZ4_DECOMP_LANES", "auto"), dist, nblocks,
      ["%.1f" % t for t in ts], nblocks * block / min(ts) * 1e3 / 2**30, ok))
