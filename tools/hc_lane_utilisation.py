"""Diagnostic (-DZLZ4_STAMPS build): how many of the 64 lanes of the HC search kernel are still walking their chain
per loop trip (k_hc_search: levels 10..12), and how its long counts go.
Usage: ZLZ4_AMD_LIB=zig-lz4_amd/libzlz4_amd_stamps.so python tools/hc_lane_utilisation.py [level] [dist | periodN]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench, zig_lz4_amd as zl
dev = torch.device("cuda:0"); nblocks = 512; block = 65536
level = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dist = sys.argv[2] if len(sys.argv) > 2 else "text"
if dist.startswith("period"):                      # periodN: random content repeated every N bytes
    P = int(dist[6:]); g = torch.Generator(device=dev); g.manual_seed(5)
    inp = torch.randint(0, 256, (nblocks, P), dtype=torch.uint8, device=dev, generator=g).repeat(1, block // P + 1)[:, :block].contiguous()
else:
    inp = bench.make_device_blocks(dist, nblocks, block, dev, seed=1)
slot = (zl.compressBound(block) + 15) // 16 * 16
ar = torch.arange(nblocks, dtype=torch.int64, device=dev)
in_len = torch.full((nblocks,), block, dtype=torch.int32, device=dev)
cap = torch.full((nblocks,), slot, dtype=torch.int32, device=dev)
comp = torch.empty(nblocks * slot, dtype=torch.uint8, device=dev)
res = torch.empty(nblocks, dtype=torch.int64, device=dev)
ws = torch.empty(zl.batch_compress_hc_workspace(nblocks, block), dtype=torch.uint8, device=dev)
import time
t0 = time.perf_counter()
zl.batch_compress_hc(inp, ar * block, in_len, comp, ar * slot, cap, res, block, level, ws)
torch.cuda.synchronize()
print("%.1f ms for %d blocks (first call, stamps build)" % ((time.perf_counter() - t0) * 1e3, nblocks))
L = zl.lib(); buf = (C.c_ulonglong * 16)()
L.zlz4_debug_read_hstamps(buf)
print("level %d D-%s" % (level, dist))
print("long counts per block: asked %.0f  counted by the wavefront %.0f (%.1f steps of 1 KiB each)" % (buf[4] / nblocks, buf[2] / nblocks, buf[3] / max(1, buf[2])))
print("per wavefront: search loop %.0f cycles, of which long counts %.0f" % (buf[5] / (nblocks * 1024), buf[6] / (nblocks * 1024)))
print("lane steps %d  wave-iteration lane slots %d  utilisation %.3f  steps/position %.1f" % (buf[0], buf[1], buf[0] / max(1, buf[1]), buf[0] / (nblocks * block)))
