#!/usr/bin/env python3
"""Diagnostic: run the -DZLZ4_STAMPS build of the compress kernel on a batch and print where its
wave-cycles go (per phase).  Usage: ZLZ4_AMD_LIB=zig-lz4_amd/libzlz4_amd_stamps.so python tools/stamp_profile.py"""
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
block = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
slot = (zl.compressBound(block) + 15) // 16 * 16
inp = bench.make_device_blocks(dist, nblocks, block, dev, seed=1)
ar = torch.arange(nblocks, dtype=torch.int64, device=dev)
in_len = torch.full((nblocks,), block, dtype=torch.int32, device=dev)
cap = torch.full((nblocks,), slot, dtype=torch.int32, device=dev)
comp = torch.empty(nblocks * slot, dtype=torch.uint8, device=dev)
res = torch.empty(nblocks, dtype=torch.int64, device=dev)
L = zl.lib()
L.zlz4_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 24)()
for it in range(2):
    L.zlz4_debug_read_stamps(buf, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    zl.batch_compress_fast(inp, ar * block, in_len, comp, ar * slot, cap, res, block, 1)
    e1.record()
    torch.cuda.synchronize()
    L.zlz4_debug_read_stamps(buf, 0)
names = {0: "table init", 1: "forward load + hash", 3: "table read + gather issue", 2: "put / read-back / groups",
         6: "vector prep (vo, mlo, J/S)", 8: "fast-run scalar loop", 9: "exact steps / loop misc", 10: "flush (emission)",
         4: "after loop", 5: "restore / commit", 7: "tail literals + generic path"}
tot = sum(buf[i] for i in range(12))
nw = max(1, buf[16])
print("dist %s  blocks %d  kernel %.2f ms  windows %d (%.1f / block)" % (dist, nblocks, e0.elapsed_time(e1), buf[16], buf[16] / nblocks))
print("per window: fast-run sequences %.2f  exact matches %.2f  exact failed probes %.2f  flushes %.2f ; generic-path sequences %d" % (
    buf[17] / nw, buf[18] / nw, buf[19] / nw, buf[21] / nw, buf[20]))
for i, n in names.items():
    print("  %-32s %6.2f %%   %8.0f cycles/window" % (n, 100.0 * buf[i] / tot, buf[i] / nw))
print("  total wave-cycles/window %.0f" % (tot / nw))
