"""Diagnostic (-DZLZ4_STAMPS build): where k_hc_seg_search's loop trips go.
Usage: ZLZ4_AMD_LIB=zig-lz4_amd/libzlz4_amd_stamps.so python tools/hc_seg_stamps.py [level] [dist | periodN]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench, zig_lz4_amd as zl
level = int(sys.argv[1]) if len(sys.argv) > 1 else 9
dist = sys.argv[2] if len(sys.argv) > 2 else "text"
dev = torch.device("cuda:0"); nblocks = 4096; block = 65536
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
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    zl.batch_compress_hc(inp, ar * block, in_len, comp, ar * slot, cap, res, block, level, ws)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
L = zl.lib(); buf = (C.c_ulonglong * 16)()
L.zlz4_debug_read_hstamps(buf)
chain, trips, fetch, walks, ta, tf, tc = [buf[i] / 2 for i in range(7)]      # two passes
waves = nblocks * 16
print("level %d D-%s: %.1f ms for %d blocks (stamps build)" % (level, dist, dt * 1e3, nblocks))
print("per block: walks %.0f  wave-trips %.0f (%.0f per wave)  chain lane-trips %.0f (util %.3f)  fetch lane-trips %.0f (util %.3f)"
      % (walks / nblocks, trips / nblocks, trips / waves, chain / nblocks, chain / (trips * 64), fetch / nblocks, fetch / (trips * 64)))
print("cycles per trip: assign %.0f  fetch %.0f  chain %.0f   (per wave total %.0f)" % (ta / trips, tf / trips, tc / trips, (ta + tf + tc) / waves))
print("per block: long counts %.1f (%.1f steps of 1 KiB each)  answered by a counted run %.1f  pattern analyses %.1f ; long-count section %.0f cycles per wave ; slowest wave of the batch %.0f cycles (both passes)"
      % (buf[7] / 2 / nblocks, buf[10] / max(1, buf[7]), buf[8] / 2 / nblocks, buf[9] / 2 / nblocks, buf[11] / 2 / waves, buf[12]))
print("long counts not answered, per block: other distance in the slot %.1f  below the counted run %.1f  above it %.1f" % (buf[13] / 2 / nblocks, buf[14] / 2 / nblocks, buf[15] / 2 / nblocks))
