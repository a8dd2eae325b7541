#!/usr/bin/env python3
"""Looks for performance cliffs: times fast compress, decompress and compressHC level 9 on batches of 64 KiB blocks made of a
random pattern repeated with period P (P = 1 .. 40000), i.e. the overlap / long-match / long-chain paths that the text
benchmarks never load.  Round trips are checked.  usage: python tools/cliff_probe.py [nblocks [period,period,... [hc level]]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, zig_lz4_amd as zl
nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0"); block = 65536
slot = (zl.compressBound(block) + 15) // 16 * 16
ar = torch.arange(nblocks, dtype=torch.int64, device=dev)
in_len = torch.full((nblocks,), block, dtype=torch.int32, device=dev)
cap = torch.full((nblocks,), slot, dtype=torch.int32, device=dev)
ws = torch.empty(zl.batch_compress_hc_workspace(nblocks, block), dtype=torch.uint8, device=dev)
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
g = torch.Generator(device=dev); g.manual_seed(5)
level = int(sys.argv[3]) if len(sys.argv) > 3 else 9
periods = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 2, 3, 7, 16, 40, 63, 64, 100, 255, 256, 257, 1000, 1024, 4096, 5000, 40000]
for P in periods:
    pat = torch.randint(0, 256, (nblocks, P), dtype=torch.uint8, device=dev, generator=g)
    inp = pat.repeat(1, block // P + 1)[:, :block].contiguous()
    comp = torch.empty(nblocks * slot, dtype=torch.uint8, device=dev)
    res = torch.empty(nblocks, dtype=torch.int64, device=dev)
    out = torch.empty_like(inp); ds = torch.empty(nblocks, dtype=torch.int64, device=dev)
    tc = timed(lambda: zl.batch_compress_fast(inp, ar * block, in_len, comp, ar * slot, cap, res, block, 1))
    clen = res.to(torch.int32)
    td = timed(lambda: zl.batch_decompress_safe(comp, ar * slot, clen, out, ar * block, in_len, ds))
    ok1 = bool((ds == block).all()) and torch.equal(out, inp)
    th = timed(lambda: zl.batch_compress_hc(inp, ar * block, in_len, comp, ar * slot, cap, res, block, level, ws), reps=2)
    clen = res.to(torch.int32)
    zl.batch_decompress_safe(comp, ar * slot, clen, out, ar * block, in_len, ds); torch.cuda.synchronize()
    ok2 = bool((ds == block).all()) and torch.equal(out, inp)
    gib = nblocks * block / 2**30
    print("period %5d: fast %8.1f GiB/s  decode %8.1f GiB/s  hc%d %8.2f GiB/s  round trips %s %s" % (P, gib / tc * 1e3, gib / td * 1e3, level, gib / th * 1e3, ok1, ok2 or level >= 10), flush=True)
