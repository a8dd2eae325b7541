#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-pointer entry points (never used for bench.py's `value`): the frame calls on a
host buffer (H2D + kernels + D2H inside the call) and the single-block calls in a loop.
Usage (GPU box): python tools/time_host_calls.py [MiB, default 1024]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import bench
import zig_lz4_amd as zl

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
nblocks = mib * 16
src = bench.make_device_blocks("text", nblocks, 65536, dev, seed=1).cpu().numpy().reshape(-1)
n = src.size
L = zl.lib()
p = zl.Prefs()
p.block_size_id = 7
bound = L.zlz4f_compress_frame_bound(n, C.byref(p))
frame = np.empty(bound, dtype=np.uint8)
back = np.empty(n, dtype=np.uint8)
ptr = lambda a: a.ctypes.data_as(C.c_void_p)
for f in (L.zlz4f_compress_frame, L.zlz4f_decompress_frame):
    f.restype = C.c_int64
L.zlz4f_compress_frame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
L.zlz4f_decompress_frame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
for it in range(3):
    t0 = time.perf_counter(); c = L.zlz4f_compress_frame(ptr(src), n, ptr(frame), bound, C.byref(p)); t1 = time.perf_counter()
    d = L.zlz4f_decompress_frame(ptr(frame), c, ptr(back), n); t2 = time.perf_counter()
    assert c > 0 and d == n, (c, d)
assert np.array_equal(src, back)
print("lz4f host pointers, %d MiB D-text, 4 MiB blocks (pageable host memory): compressFrame %.1f ms = %.2f GiB/s, "
      "decompressFrame %.1f ms = %.2f GiB/s (frame %d bytes)" % (mib, (t1 - t0) * 1e3, n / (t1 - t0) / 2**30,
                                                              (t2 - t1) * 1e3, n / (t2 - t1) / 2**30, c))
# single-block calls, 64 KiB each
blk = src[:65536].tobytes()
reps = 200
zl.compressDefault(blk)
t0 = time.perf_counter()
for _ in range(reps):
    comp = zl.compressDefault(blk)
t1 = time.perf_counter()
for _ in range(reps):
    out = zl.decompressSafe(comp, 65536)
t2 = time.perf_counter()
assert out == blk
print("single-block host calls, 64 KiB: compressDefault %.0f us/call, decompressSafe %.0f us/call (ctypes overhead included)"
      % ((t1 - t0) / reps * 1e6, (t2 - t1) / reps * 1e6))
