"""N > 1 path on CPU: two gloo ranks shard the blocks of one frame, exchange only their segment sizes and
assemble a frame that is byte-identical to the single-process one.  (On CPU the per-block compressor is the
oracle; on the GPU box the same shard/offset logic feeds the HIP batch kernels, see bench.py.)"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import zig_lz4_amd  # noqa: F401  (package import; no compute)
    from zig_lz4_amd import shard
    from oracle import binding as o
    import datagen as dg
    bs = 65536
    data = bytes(dg.text_bytes(5 * bs + 1234, 42)) + bytes(dg.random_bytes(bs, 43)) + b"\0" * 70000
    nblocks = (len(data) + bs - 1) // bs
    lo, hi = shard.shard_range(nblocks, rank, world)
    raws = [data[i * bs:(i + 1) * bs] for i in range(lo, hi)]
    seg = shard.block_segment([(o.compress_default(r), r) for r in raws], [len(r) for r in raws], True, o.xxh32)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([len(seg)], dtype=torch.int64))          # the only cross-rank exchange
    p = o.Prefs(); p.block_checksum = 1
    whole = o.compress_frame(data, p)
    offs, end = shard.segment_offsets([int(s) for s in sizes], 7)
    assert whole[offs[rank]:offs[rank] + len(seg)] == seg, "rank %d segment differs" % rank
    assert whole[end:end + 4] == b"\0\0\0\0" and end + 4 == len(whole)
    covered = sorted(shard.shard_range(nblocks, r, world) for r in range(world))
    assert covered[0][0] == 0 and covered[-1][1] == nblocks and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_frame_equals_single_process_frame(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(world))


def test_shard_ranges_cover_everything():
    sys.path.insert(0, ROOT)
    from zig_lz4_amd import shard
    for nb in (0, 1, 7, 8, 9, 8192, 65536, 1000003):
        for w in (1, 2, 4, 8):
            r = [shard.shard_range(nb, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == nb and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_bench_launcher_starts_its_own_ranks_and_reports_failures():
    """`python bench.py --gpus 2` without a launcher starts two child ranks itself (RANK / WORLD_SIZE / MASTER_* set as
    torch.distributed.run would) and never touches the GPU in the parent.  On this box the children cannot find a GPU:
    the parent must come back with a non-zero exit code, name the failed rank and print no JSON line."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--blocks", "64"],
                       capture_output=True, text=True, env=env, timeout=600)
    import torch
    if torch.cuda.is_available():          # (on a GPU box this is a real two-rank run on one device: not what is tested here)
        return
    assert r.returncode != 0
    assert "exited with" in r.stderr and "rank" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
