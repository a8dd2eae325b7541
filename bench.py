#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X LZ4 block codec on BASELINE.json's configs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|cfg5] [--dist text|...]

Default workload = BASELINE.json configs[1]: 65 536 independent 64 KiB blocks (4 GiB),
compressDefault, then decompressSafe of the result (the metric is "block
compress+decompress").  One step = one pass of both kernels over the batch; inputs are
resident in HBM before the timed region.  `value` = uncompressed GiB processed per second
by the whole job (all ranks), over compress+decompress time (round trip); the per-kernel
rates are reported next to it.  For N > 1 every rank runs the same per-GPU batch on its
own data (independent blocks: no collective on the data path, weak scaling).

The JSON line also carries
  roofline      for the dominant kernel (compress): algorithmic bytes (N read + C written)
                per launch / average launch duration from HIP events on the launch stream,
                against the 8 TB/s HBM3E peak;
  cpu_baseline  oracle/ (C restatement of the reference, "port") timed on this box's host
                cores on a bounded sample of the same blocks (rank 0, N == 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
GIB = float(1 << 30)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_device_blocks(dist, nblocks, block_size, dev, seed):
    """[nblocks, block_size] uint8 on the device, every block distinct.

    text: a CPU-generated pool (tests/datagen.py) is expanded on the GPU by giving every
    repetition its own random byte-substitution table -- LZ4's behaviour is invariant under
    a bijection of byte values, so all repetitions have the statistics of the pool while no
    two blocks hold the same bytes (no cache aliasing between them)."""
    import datagen as dg
    g = torch.Generator(device="cpu")
    g.manual_seed(1000 + seed)
    if dist == "zero":
        return torch.zeros((nblocks, block_size), dtype=torch.uint8, device=dev)
    if dist == "ramp":
        row = (torch.arange(block_size, device=dev) % 256).to(torch.uint8)
        return row.unsqueeze(0).repeat(nblocks, 1).contiguous()
    gd = torch.Generator(device=dev)
    gd.manual_seed(2000 + seed)
    if dist == "random":
        return torch.randint(0, 256, (nblocks, block_size), dtype=torch.uint8, device=dev, generator=gd)
    if dist == "mixed":
        t = torch.randint(0, 256, (nblocks, block_size), dtype=torch.uint8, device=dev, generator=gd)
        t[:, : block_size // 2] = ord("X")
        return t
    assert dist in ("text", "reptext")
    pool_blocks = min(nblocks, max(1, (32 << 20) // block_size))
    pool = torch.from_numpy(dg.make_blocks(dist, pool_blocks, block_size, seed=seed)).to(dev)
    out = torch.empty((nblocks, block_size), dtype=torch.uint8, device=dev)
    done, rep = 0, 0
    idx = pool.long()
    while done < nblocks:
        n = min(pool_blocks, nblocks - done)
        lut = torch.randperm(256, generator=g).to(torch.uint8).to(dev) if rep else torch.arange(256, dtype=torch.uint8, device=dev)
        shift = (rep * 7919) % pool_blocks
        src = torch.roll(idx, shifts=shift, dims=0)[:n] if shift else idx[:n]
        out[done:done + n] = lut[src]
        done += n
        rep += 1
    del idx
    return out


def cpu_baseline(sample_blocks, block_size, slot, hc_level=None):
    """Time the oracle (C restatement of the reference) on host cores: 1 thread and all cores."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding as ob
    L = ob.lib()
    data = np.ascontiguousarray(sample_blocks)
    nb = data.shape[0]
    comp = np.zeros(nb * slot, dtype=np.uint8)
    sizes = np.zeros(nb, dtype=np.int64)
    dec = np.zeros(nb * block_size, dtype=np.uint8)
    dsz = np.zeros(nb, dtype=np.int64)

    def run(lo, hi):
        n = hi - lo
        if hc_level is None:
            L.zo_batch_compress_default(data[lo:].ctypes.data, block_size, n, comp[lo * slot:].ctypes.data, slot,
                                        sizes[lo:].ctypes.data)
        else:
            L.zo_batch_compress_hc(data[lo:].ctypes.data, block_size, n, comp[lo * slot:].ctypes.data, slot,
                                   sizes[lo:].ctypes.data, hc_level)

    def rund(lo, hi):
        L.zo_batch_decompress_safe(comp[lo * slot:].ctypes.data, slot, sizes[lo:].ctypes.data, hi - lo,
                                   dec[lo * block_size:].ctypes.data, block_size, dsz[lo:].ctypes.data)

    t0 = time.perf_counter(); run(0, nb); tc1 = time.perf_counter() - t0
    t0 = time.perf_counter(); rund(0, nb); td1 = time.perf_counter() - t0
    assert (dec.reshape(nb, block_size) == data).all(), "oracle round trip failed"
    ncores = min(os.cpu_count() or 1, 64)
    try:
        ncores = min(ncores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    parts = [(i * nb // ncores, (i + 1) * nb // ncores) for i in range(ncores)]
    with ThreadPoolExecutor(ncores) as ex:     # ctypes releases the GIL inside the C calls
        t0 = time.perf_counter(); list(ex.map(lambda p: run(*p), parts)); tcn = time.perf_counter() - t0
        t0 = time.perf_counter(); list(ex.map(lambda p: rund(*p), parts)); tdn = time.perf_counter() - t0
    nbytes = nb * block_size
    return {
        "value": nbytes / (tc1 + td1) / GIB, "unit": "GiB/s", "cores": 1, "kind": "port",
        "sample": "%d of the benchmark's %d-byte blocks (%.0f MiB): oracle/ compress then decompress, 1 thread"
                  % (nb, block_size, nbytes / 2**20),
        "compress_gibs": nbytes / tc1 / GIB, "decompress_gibs": nbytes / td1 / GIB,
        "all_cores": {"cores": ncores, "value": nbytes / (tcn + tdn) / GIB,
                      "compress_gibs": nbytes / tcn / GIB, "decompress_gibs": nbytes / tdn / GIB},
        "compressed_sizes_head": [int(x) for x in sizes[:4]],
    }


def launch_ranks(n):
    """One fresh child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run sets them),
    same command line.  Rank 0's stdout (the JSON line) is forwarded; a failing rank ends the others and the exit
    code is non-zero."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    buf = []
    t = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    t.start()
    rc, deadline = 0, time.time() + 3600
    while any(p.poll() is None for p in procs):
        bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad or time.time() > deadline:
            # a rank that died leaves the others in a barrier: end them (exact PIDs, nothing by pattern)
            for q in procs:
                if q.poll() is None:
                    q.kill()
            for r, c in bad:
                log("rank %d exited with %d" % (r, c))
            rc = 1
            break
        time.sleep(0.2)
    for r, p in enumerate(procs):
        c = p.wait()
        if c != 0 and rc == 0:
            log("rank %d exited with %d" % (r, c))
            rc = 1
    t.join(timeout=10)
    out0 = buf[0] if buf else b""
    for ln in out0.decode(errors="replace").splitlines():      # stdout carries the JSON line only; library chatter -> stderr
        print(ln, file=sys.stdout if ln.startswith("{") else sys.stderr, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--dist", default="text", choices=["text", "reptext", "ramp", "mixed", "random", "zero"])
    ap.add_argument("--blocks", type=int, default=0, help="override the number of blocks per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-sample-mib", type=int, default=0)
    ap.add_argument("--level", type=int, default=9, help="compressHC level for cfg4 (2..12)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks here.  Nothing above has touched the GPU
        # (importing torch does not), and this process never does: it only waits and passes rank 0's JSON line on.
        raise SystemExit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log("warning: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    dist_on = world > 1
    # ZLZ4_BENCH_REHEARSE=1: rehearsal of the multi-rank control flow on a box with ONE GPU -- every rank uses cuda:0 and
    # the barrier / max-reduction run over gloo (RCCL refuses two ranks on one device).  The value of such a run is
    # meaningless; it only proves that the N > 1 path runs end to end.
    rehearse = os.environ.get("ZLZ4_BENCH_REHEARSE") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if dist_on:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            td.init_process_group("gloo", rank=rank, world_size=world)
        else:
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import zig_lz4_amd as zl     # raises if libzlz4_amd.so is missing: no fallback
    if not zl.device_available():
        raise SystemExit("libzlz4_amd.so: no usable gfx950 device")

    defaults = {"cfg2": 65536, "cfg3": 1 << 20, "cfg4": 16384, "cfg5": 1024}
    block = (4 << 20) if args.workload == "cfg5" else 65536
    nblocks = args.blocks or defaults[args.workload]
    hc_level = args.level if args.workload == "cfg4" else None
    slot = (zl.compressBound(block) + 15) // 16 * 16          # 65 809 -> 65 824
    names = {
        "cfg2": "configs[1]: %d x 64 KiB blocks, compressDefault then decompressSafe (round trip), D-%s",
        "cfg3": "configs[2]: decompressSafe only, %d pre-compressed 64 KiB blocks, D-%s",
        "cfg4": "configs[3]: compressHC level " + str(args.level) + " then decompressSafe, %d x 64 KiB blocks, D-%s",
        "cfg5": "configs[4]: lz4f frame, %d independent 4 MiB blocks per GPU, compressFrame then decompressFrame, D-%s",
    }
    workload = names[args.workload] % (nblocks, args.dist)
    total_n = nblocks * block
    decomp_only = args.workload == "cfg3"
    frame_mode = args.workload == "cfg5"

    if frame_mode:
        # ---- config 5: one frame per GPU shard, device-resident source and destination ----
        t0 = time.time()
        inp = make_device_blocks(args.dist, nblocks, block, dev, seed=rank + 1).reshape(-1)
        torch.cuda.synchronize()
        log("[rank %d] generated %.2f GiB of D-%s in %.1f s" % (rank, total_n / GIB, args.dist, time.time() - t0))
        prefs = zl.Prefs()
        prefs.block_size_id = 7          # max4MB
        prefs.block_mode = 1             # independent
        bound = zl.lz4f.compressFrameBound(total_n, prefs)
        frame = torch.empty(bound, dtype=torch.uint8, device=dev)
        out = torch.empty(total_n, dtype=torch.uint8, device=dev)
        L = zl.lib()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        fsize = [0]

        # One logical frame of world x nblocks blocks: rank r holds blocks [r*nblocks, (r+1)*nblocks) (shard.shard_range)
        # and produces / decodes its own segment of the frame; rank 0's segment starts with the frame header, the last
        # rank's ends with the end mark.  With one rank the segment is the whole frame.
        seg_flags = (zl.lz4f.SEG_FIRST if rank == 0 else 0) | (zl.lz4f.SEG_LAST if rank == world - 1 else 0)

        def do_compress():
            r = L.zlz4f_compress_frame_segment_device(st, C.c_void_p(inp.data_ptr()), total_n, C.c_void_p(frame.data_ptr()),
                                                      bound, C.byref(prefs), seg_flags)
            assert r > 0, "compressFrame failed: %s" % zl.error_name(r)
            fsize[0] = r

        def do_decompress():
            r = L.zlz4f_decompress_frame_segment_device(st, C.c_void_p(frame.data_ptr()), fsize[0], C.c_void_p(out.data_ptr()),
                                                        total_n, C.byref(prefs), seg_flags)
            assert r == total_n, "decompressFrame failed: %s" % zl.error_name(r)

        do_compress()
        do_decompress()
        torch.cuda.synchronize()
        assert torch.equal(out, inp), "frame round trip mismatch"
        total_c = fsize[0]
        if dist_on:
            # the only cross-rank step of the real flow: where each segment goes in the assembled frame (untimed here)
            from zig_lz4_amd import shard
            sizes = [torch.zeros(1, dtype=torch.int64, device="cpu" if rehearse else dev) for _ in range(world)]
            td.all_gather(sizes, torch.tensor([fsize[0]], dtype=torch.int64, device="cpu" if rehearse else dev))
            offs, end = shard.segment_offsets([int(x) for x in sizes], 0)
            log("[rank %d] segment %d bytes at frame offset %d of %d" % (rank, fsize[0], offs[rank], end))
        sample_src = inp.reshape(nblocks, block)
        csize_head = None
    elif decomp_only:
        # ---- config 3: compress chunk by chunk (untimed), keep only the compressed slots, time decompression ----
        chunk = min(nblocks, 65536)
        comp = torch.empty(nblocks * slot, dtype=torch.uint8, device=dev)
        csize = torch.empty(nblocks, dtype=torch.int64, device=dev)
        ar = torch.arange(nblocks, dtype=torch.int64, device=dev)
        in_off = ar * block
        in_len = torch.full((nblocks,), block, dtype=torch.int32, device=dev)
        slot_off = ar * slot
        slot_cap = torch.full((nblocks,), slot, dtype=torch.int32, device=dev)
        t0 = time.time()
        for c0 in range(0, nblocks, chunk):
            n = min(chunk, nblocks - c0)
            part = make_device_blocks(args.dist, n, block, dev, seed=(rank + 1) * 1000 + c0 // chunk)
            zl.batch_compress_fast(part, in_off[:n], in_len[:n], comp[c0 * slot:], slot_off[:n], slot_cap[:n],
                                   csize[c0:c0 + n], block, 1)
            torch.cuda.synchronize()
            del part
        log("[rank %d] pre-compressed %.1f GiB in %.1f s" % (rank, total_n / GIB, time.time() - t0))
        assert int(csize.min()) > 0
        clen32 = csize.to(torch.int32)
        out = torch.empty((nblocks, block), dtype=torch.uint8, device=dev)
        dsize = torch.empty(nblocks, dtype=torch.int64, device=dev)

        def do_compress():
            pass

        def do_decompress():
            zl.batch_decompress_safe(comp, slot_off, clen32, out, in_off, in_len, dsize)

        do_decompress()
        torch.cuda.synchronize()
        assert bool((dsize == block).all()), "decompress size mismatch"
        for c0 in range(0, nblocks, chunk):      # regenerate each chunk and compare
            n = min(chunk, nblocks - c0)
            part = make_device_blocks(args.dist, n, block, dev, seed=(rank + 1) * 1000 + c0 // chunk)
            assert torch.equal(out[c0:c0 + n], part), "round trip mismatch in chunk %d" % (c0 // chunk)
            sample_src = part if c0 == 0 else sample_src
        total_c = int(csize.sum())
        csize_head = [int(x) for x in csize[:4].cpu()]
    else:
        # ---- configs 2 and 4: compress then decompress the same resident batch ----
        t0 = time.time()
        inp = make_device_blocks(args.dist, nblocks, block, dev, seed=rank + 1)
        torch.cuda.synchronize()
        log("[rank %d] generated %.2f GiB of D-%s in %.1f s" % (rank, total_n / GIB, args.dist, time.time() - t0))
        ar = torch.arange(nblocks, dtype=torch.int64, device=dev)
        in_off = ar * block
        in_len = torch.full((nblocks,), block, dtype=torch.int32, device=dev)
        slot_off = ar * slot
        slot_cap = torch.full((nblocks,), slot, dtype=torch.int32, device=dev)
        comp = torch.empty(nblocks * slot, dtype=torch.uint8, device=dev)
        csize = torch.empty(nblocks, dtype=torch.int64, device=dev)
        out = torch.empty((nblocks, block), dtype=torch.uint8, device=dev)
        dsize = torch.empty(nblocks, dtype=torch.int64, device=dev)
        ws = None
        if hc_level is not None:
            ws = torch.empty(zl.batch_compress_hc_workspace(nblocks, block), dtype=torch.uint8, device=dev)
        clen = [None]

        def do_compress():
            if hc_level is None:
                zl.batch_compress_fast(inp, in_off, in_len, comp, slot_off, slot_cap, csize, block, 1)
            else:
                zl.batch_compress_hc(inp, in_off, in_len, comp, slot_off, slot_cap, csize, block, hc_level, ws)

        def do_decompress():
            zl.batch_decompress_safe(comp, slot_off, clen[0], out, in_off, in_len, dsize)

        # untimed first pass: produces the compressed batch and checks the round trip
        do_compress()
        torch.cuda.synchronize()
        if hc_level in (10, 11, 12):
            # the reference's lz4opt early-encode walk can leave ip below the anchor, where the reference itself underflows;
            # oracle and HIP path both report OutputTooSmall for such blocks
            log("[rank %d] level %d: %d of %d blocks report an error (reference behaviour)" % (rank, hc_level, int((csize <= 0).sum()), nblocks))
        else:
            assert int(csize.min()) > 0, "compress reported an error: %d" % int(csize.min())
        clen[0] = csize.clamp(min=0).to(torch.int32)
        do_decompress()
        torch.cuda.synchronize()
        if hc_level in (10, 11):
            # the reference's lz4opt early-encode branch (src/lz4hc.zig:1207-1256) emits streams that do not always
            # decode (every 64 KiB D-text block at level 10, a few percent at level 11); bit-exactness is checked
            # against the oracle's bytes by the cpu_baseline leg / tests, not by a round trip
            nbad = int(((dsize != block) | (out != inp).any(dim=1)).sum())
            log("[rank %d] level %d: %d of %d blocks do not round-trip (reference behaviour)" % (rank, hc_level, nbad, nblocks))
        else:
            assert bool((dsize == block).all()), "decompress size mismatch"
            assert torch.equal(out, inp), "round trip mismatch"
        total_c = int(csize.clamp(min=0).sum())
        sample_src = inp
        csize_head = [int(x) for x in csize[:4].cpu()]
    log("[rank %d] round trip ok, ratio %.3f" % (rank, total_n / total_c))

    for _ in range(args.warmup):
        do_compress()
        do_decompress()
    torch.cuda.synchronize()
    if dist_on:
        td.barrier()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        do_compress()
        ev[k][1].record()
        do_decompress()
        ev[k][2].record()
    torch.cuda.synchronize()
    if dist_on:
        td.barrier()
    elapsed = time.perf_counter() - t_start
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())
    tc_ms = float(np.mean([ev[k][0].elapsed_time(ev[k][1]) for k in range(args.steps)]))
    td_ms = float(np.mean([ev[k][1].elapsed_time(ev[k][2]) for k in range(args.steps)]))

    if rank == 0:
        value = world * total_n * args.steps / elapsed / GIB
        comp_gibs = None if decomp_only else total_n / (tc_ms * 1e-3) / GIB
        dec_gibs = total_n / (td_ms * 1e-3) / GIB
        hc_kernels = ("zlz4::k_hc_mid_serial" if args.level <= 2 else
                      "HC pipeline: k_hc_build_links + k_hc_seg_search<4> + k_hc_parse_emit (rounds of 4096 blocks, emit beside the next round)" if args.level <= 9 else
                      "HC pipeline: k_hc_build_links + k_hc_search + k_hc_opt_parse_wave (rounds of 4096 blocks, parse beside the next round's search)")
        # the decoder has two builds: lane-per-dword match copies for batches that fill the chip, 16 bytes per sequence lane below
        dec_kernel = "zlz4::k_decompress_safe<%s>" % ("true, true, true" if nblocks >= 6144 else "true, false, false")
        kc = {"cfg2": "zlz4::k_compress_fast<uint16_t, 0>", "cfg4": hc_kernels,
              "cfg5": "zlz4::k_compress_fast<uint32_t, 2>", "cfg3": None}[args.workload]
        if decomp_only:
            dom, dom_ms = dec_kernel, td_ms
        else:
            dom, dom_ms = kc, tc_ms
        # HBM traffic from rocprofv3 PMC passes of this same command (FETCH_SIZE / WRITE_SIZE collected in separate
        # runs, tools/pmc_run.sh); bytes per launch, null when no profile of this workload/distribution is committed
        traffic = {}
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            tkey = "%s/%s/%d" % (args.workload, args.dist, nblocks)
            if args.workload == "cfg4":
                tkey += "/L%d" % args.level          # the HC levels run different kernels
            traffic = tj.get(tkey, {})
        except Exception:
            pass
        algo = total_n + total_c
        achieved = algo / (dom_ms * 1e-3) / 1e9
        res = {
            "metric": "GiB/s uncompressed, block compress+decompress",
            "value": value, "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "blocks_per_gpu": nblocks, "block_bytes": block,
                       "uncompressed_bytes_per_gpu": total_n, "compressed_bytes_per_gpu": total_c,
                       "ratio": total_n / total_c, "distribution": "D-" + args.dist,
                       "sharding": "independent blocks, contiguous range per GPU, no collective"},
            "compress_gibs_per_gpu": comp_gibs, "decompress_gibs_per_gpu": dec_gibs,
            "compress_ms": None if decomp_only else tc_ms, "decompress_ms": td_ms,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic.get("decompress" if decomp_only else "compress"),
                         "traffic_source": (None if not traffic else
                                            "from profile, not measured in this run: " + str(traffic.get("source"))),
                         "algorithmic_bytes_per_launch": algo, "avg_launch_ms": dom_ms},
            "roofline_decompress": {"bound": "hbm", "kernel": dec_kernel,
                                    "achieved": algo / (td_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": algo / (td_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "traffic": traffic.get("decompress"), "avg_launch_ms": td_ms},
        }
        # achieved fraction of HBM bandwidth from the FETCH/WRITE counters themselves (BASELINE.json's "HBM-% achieved")
        for key, ms in (("roofline", dom_ms), ("roofline_decompress", td_ms)):
            tr = res[key]["traffic"]
            res[key]["traffic_frac"] = None if tr is None else tr / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if frame_mode:
            res["roofline"]["note"] = ("frame calls = descriptor + batch compress + plan + scatter kernels plus "
                                       "hipMalloc/hipFree of the slot arena inside the call; avg_launch_ms is the whole call")
        if world == 1 and not args.no_cpu:
            mib = args.cpu_sample_mib or (1024 if hc_level is None else 96)
            nsamp = max(1, min(nblocks, mib * (1 << 20) // block))
            sample = sample_src[:nsamp].cpu().numpy()
            cb = cpu_baseline(sample, block, slot, hc_level)
            # the same sample through the HIP path must give the same compressed sizes
            if csize_head is not None:
                assert cb["compressed_sizes_head"] == csize_head[:len(cb["compressed_sizes_head"])], "HIP vs oracle size mismatch"
            res["cpu_baseline"] = cb
        print(json.dumps(res), flush=True)
    if dist_on:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
