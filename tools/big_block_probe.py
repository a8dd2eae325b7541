#!/usr/bin/env python3
"""Times compressHC on a few LARGE blocks (default 4 MiB: the HBM-link variants of the search kernels) of zeros, of random
content repeated every 256 / 5000 bytes, and of text, and checks the round trip (levels <= 9).
usage: python tools/big_block_probe.py [level [nblocks [block bytes]]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench, zig_lz4_amd as zl
level = int(sys.argv[1]) if len(sys.argv) > 1 else 9
nblocks = int(sys.argv[2]) if len(sys.argv) > 2 else 16
block = int(sys.argv[3]) if len(sys.argv) > 3 else 4 << 20
dev = torch.device("cuda:0")
slot = (zl.compressBound(block) + 15) // 16 * 16
ar = torch.arange(nblocks, dtype=torch.int64, device=dev)
in_len = torch.full((nblocks,), block, dtype=torch.int32, device=dev)
cap = torch.full((nblocks,), slot, dtype=torch.int32, device=dev)
ws = torch.empty(zl.batch_compress_hc_workspace(nblocks, block), dtype=torch.uint8, device=dev)
g = torch.Generator(device=dev); g.manual_seed(7)
def periodic(P):
    pat = torch.randint(0, 256, (nblocks, P), dtype=torch.uint8, device=dev, generator=g)
    return pat.repeat(1, block // P + 1)[:, :block].contiguous()
cases = [("zeros", torch.zeros((nblocks, block), dtype=torch.uint8, device=dev)), ("period 256", periodic(256)), ("period 5000", periodic(5000)),
         ("text", bench.make_device_blocks("text", nblocks, block, dev, seed=1).view(nblocks, block))]
for name, inp in cases:
    comp = torch.empty(nblocks * slot, dtype=torch.uint8, device=dev)
    res = torch.empty(nblocks, dtype=torch.int64, device=dev)
    best = 1e9
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); zl.batch_compress_hc(inp, ar * block, in_len, comp, ar * slot, cap, res, block, level, ws); e1.record()
        torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    ok = "-"
    if level <= 9:
        out = torch.empty_like(inp); ds = torch.empty(nblocks, dtype=torch.int64, device=dev)
        zl.batch_decompress_safe(comp, ar * slot, res.to(torch.int32), out, ar * block, in_len, ds); torch.cuda.synchronize()
        ok = bool((ds == block).all()) and torch.equal(out, inp)
    print("level %d, %d x %d KiB of %-12s %9.1f ms  %8.2f GiB/s  ratio %.1f  round trip %s" % (
        level, nblocks, block >> 10, name + ":", best, nblocks * block / 2**30 / best * 1e3, nblocks * block / max(1, int(res.clamp(min=0).sum())), ok), flush=True)
