#!/usr/bin/env python3
"""Diagnostic: run the -DZLZ4_STAMPS build of the wave decoder on a batch and print where its wave-cycles go.
Usage: ZLZ4_DECOMP_LANE_MIN=100000000 ZLZ4_AMD_LIB=zig-lz4_amd/libzlz4_amd_stamps.so python tools/stamp_decode.py [dist] [nblocks]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
import zig_lz4_amd as zl

dist = sys.argv[1] if len(sys.argv) > 1 else "text"
nblocks = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
dev = torch.device("cuda:0")
block = 65536
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
out = torch.empty_like(inp)
ds = torch.empty(nblocks, dtype=torch.int64, device=dev)
L = zl.lib()
L.zlz4_debug_read_dstamps.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 16)()
for it in range(2):
    L.zlz4_debug_read_dstamps(buf, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    zl.batch_decompress_safe(comp, ar * slot, clen, out, ar * block, in_len, ds)
    e1.record()
    torch.cuda.synchronize()
    L.zlz4_debug_read_dstamps(buf, 0)
names = ["window wait + parse", "token walk", "scan + checks", "literal store", "match loads issue",
         "after last batch", "single-sequence paths / loop", "deferred match stores (load wait)"]
tot = sum(buf[i] for i in range(8))
nb = max(1, buf[8])
print("dist %s  blocks %d  kernel %.2f ms  round trip ok %s" % (dist, nblocks, e0.elapsed_time(e1), bool(torch.equal(out, inp))))
print("batches %d (%.1f / block)  sequences in batches %d (%.2f / batch)  resolve rounds %.2f / batch  single-path sequences %d" % (
    buf[8], buf[8] / nblocks, buf[9], buf[9] / nb, buf[10] / nb, buf[11]))
for i, n in enumerate(names):
    print("  %-32s %6.2f %%   %8.0f cycles/batch" % (n, 100.0 * buf[i] / tot, buf[i] / nb))
print("  total wave-cycles/batch %.0f  (s_memtime ticks at 100 MHz: x ~24 for shader cycles)" % (tot / nb))
