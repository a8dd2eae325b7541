"""One lz4f frame over several ranks on the HIP path (BASELINE configs[4] flow; run on the GPU box: pytest -m gpu).

Two (and three) fresh child processes -- gloo for the exchange, all of them on cuda:0 because the box has one GPU --
each compress the block range shard_range() gives them with zlz4f_compress_frame_segment_device, exchange ONLY the
segment sizes to learn where their bytes go, and the assembled frame must equal byte for byte the frame a single
process produces (and the oracle's).  Each rank then decodes its own segment back to its input range."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, tmp, level, block_checksum):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    import zig_lz4_amd as zl
    from zig_lz4_amd import shard
    import datagen as dg
    dev = torch.device("cuda:0")
    bs = 65536
    data = bytes(dg.text_bytes(9 * bs + 1234, 42)) + bytes(dg.random_bytes(bs, 43)) + b"\0" * 70000 + bytes(dg.text_bytes(3 * bs + 17, 44))
    nblocks = (len(data) + bs - 1) // bs
    lo, hi = shard.shard_range(nblocks, rank, world)
    mine = data[lo * bs: min(len(data), hi * bs)]
    p = zl.Prefs(); p.block_size_id = 4; p.block_mode = 1; p.block_checksum = block_checksum; p.compression_level = level
    flags = (zl.lz4f.SEG_FIRST if rank == 0 else 0) | (zl.lz4f.SEG_LAST if rank == world - 1 else 0)
    d_src = torch.from_numpy(np.frombuffer(mine, dtype=np.uint8).copy()).to(dev)
    d_seg = torch.empty(zl.lz4f.compressFrameBound(len(mine), p), dtype=torch.uint8, device=dev)
    n = zl.lz4f.compressFrameSegmentDevice(d_src, d_seg, p, flags)
    seg = d_seg[:n].cpu().numpy().tobytes()
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([n], dtype=torch.int64))                 # the only exchange the data path needs
    offs, end = shard.segment_offsets([int(s) for s in sizes], 0)               # rank 0's segment includes the header
    # reference frame: one process, whole input (rank 0 computes it on the device, everyone checks its own slice)
    d_all = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev)
    d_whole = torch.empty(zl.lz4f.compressFrameBound(len(data), p), dtype=torch.uint8, device=dev)
    wn = zl.lz4f.compressFrameDevice(d_all, d_whole, p)
    whole = d_whole[:wn].cpu().numpy().tobytes()
    assert end == len(whole), (end, len(whole))
    assert whole[offs[rank]: offs[rank] + n] == seg, "rank %d: segment differs from the single-process frame" % rank
    if rank == 0:
        from oracle import binding as o
        op = o.Prefs(); op.block_size_id = 4; op.block_mode = 1; op.block_checksum = block_checksum; op.compression_level = level
        assert o.compress_frame(data, op) == whole
    # decode the rank's own segment
    d_out = torch.empty(len(mine), dtype=torch.uint8, device=dev)
    got = zl.lz4f.decompressFrameSegmentDevice(d_seg, n, d_out, p, flags)
    assert got == len(mine) and d_out.cpu().numpy().tobytes() == mine
    open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,level,bc", [(2, 0, 0), (3, 0, 1), (2, 9, 1)])
def test_sharded_frame_on_device_equals_single_process_frame(tmp_path, world, level, bc):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), level, bc), nprocs=world, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(world))


def test_segment_argument_checks(zl, gpu):
    p = zl.Prefs(); p.block_size_id = 4; p.content_checksum = 1
    src = torch.zeros(65536 + 5, dtype=torch.uint8, device=gpu)
    dst = torch.empty(zl.lz4f.compressFrameBound(src.numel(), p), dtype=torch.uint8, device=gpu)
    with pytest.raises(zl.Lz4Error) as e:                      # content checksum cannot be split over ranks
        zl.lz4f.compressFrameSegmentDevice(src, dst, p, zl.lz4f.SEG_FIRST)
    assert e.value.name == "Unsupported"
    p.content_checksum = 0
    with pytest.raises(zl.Lz4Error) as e:                      # only the last segment may end with a short block
        zl.lz4f.compressFrameSegmentDevice(src, dst, p, zl.lz4f.SEG_FIRST)
    assert e.value.name == "ParameterInvalid"
    whole = zl.lz4f.compressFrameSegmentDevice(src, dst, p, zl.lz4f.SEG_FIRST | zl.lz4f.SEG_LAST)
    assert whole == zl.lz4f.compressFrameDevice(src, torch.empty_like(dst), p)
