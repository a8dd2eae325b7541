"""BASELINE.json's big shapes on the HIP path (run on the GPU box: pytest -m gpu).

  * configs[4]: lz4f frames with 4 MiB independent blocks -- several blocks, a stored block, checksums on and off, fast
    and level 9 -- byte-compared with the oracle (src/lz4f.zig:354-446, stored fallback :407-417), and the full per-GPU
    slice (1024 x 4 MiB) as a round trip through the device-resident frame calls;
  * configs[2]: a decompressSafe batch whose output offsets cross 2^32 (and 2^33).
"""
import numpy as np
import pytest
import torch

import datagen as dg

pytestmark = pytest.mark.gpu

MIB = 1 << 20


def _prefs(P, **kw):
    p = P()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _frame_inputs():
    text = bytes(dg.text_bytes(9 * MIB + 5, 21))                         # 3 blocks, the last one 1 MiB + 5
    mixed = bytes(dg.text_bytes(4 * MIB, 22)) + bytes(dg.random_bytes(4 * MIB, 23)) + bytes(4 * MIB) + b"tail"
    return [("text 9 MiB + 5", text), ("text | random (stored) | zero | tail", mixed)]


@pytest.mark.parametrize("level", [0, 9])
@pytest.mark.parametrize("cks", [(0, 0), (1, 1), (1, 0)])
def test_frame_4mib_blocks_bit_exact(zl, oracle, gpu, level, cks):
    for name, b in _frame_inputs():
        if level == 9 and name.startswith("text 9"):
            b = b[: 8 * MIB + 77]                                       # keep the CPU oracle's level-9 time bounded
        kw = dict(block_size_id=7, block_mode=1, block_checksum=cks[0], content_checksum=cks[1], compression_level=level)
        want = oracle.compress_frame(b, _prefs(oracle.Prefs, **kw))
        got = zl.lz4f.compressFrame(b, _prefs(zl.Prefs, **kw))
        assert got == want, "%s level %d cks %s: %d vs %d bytes" % (name, level, cks, len(got), len(want))
        if "stored" in name:
            # second block header has the uncompressed flag (src/lz4f.zig:411-414)
            hs = zl.lz4f.headerSize(got)
            first = int.from_bytes(got[hs:hs + 4], "little")
            second_at = hs + 4 + (first & 0x7FFFFFFF) + (4 if cks[0] else 0)
            assert int.from_bytes(got[second_at:second_at + 4], "little") == 0x80000000 | (4 * MIB)
        assert zl.lz4f.decompressFrame(got, len(b)) == b
        assert oracle.decompress_frame(got, len(b)) == b


def test_frame_config4_full_slice_round_trip(zl, gpu):
    """1024 x 4 MiB independent blocks (one GPU's share of configs[4]) through zlz4f_*_frame_device."""
    import bench
    nblocks, block = 1024, 4 * MIB
    inp = bench.make_device_blocks("text", nblocks, block, gpu, seed=5).reshape(-1)
    inp[7 * block: 8 * block] = torch.randint(0, 256, (block,), dtype=torch.uint8, device=gpu)   # one stored block
    p = _prefs(zl.Prefs, block_size_id=7, block_mode=1)
    bound = zl.lz4f.compressFrameBound(inp.numel(), p)
    frame = torch.empty(bound, dtype=torch.uint8, device=gpu)
    n = zl.lz4f.compressFrameDevice(inp, frame, p)
    assert 0 < n < inp.numel()
    out = torch.empty_like(inp)
    assert zl.lz4f.decompressFrameDevice(frame, n, out) == inp.numel()
    assert torch.equal(out, inp)
    # block chain: 1024 data blocks, block 7 stored, end mark at the end (host walk over the block headers)
    head = frame[:32].cpu().numpy().tobytes()
    pos = zl.lz4f.headerSize(head)
    sizes = []
    for _ in range(nblocks):
        h = int.from_bytes(frame[pos:pos + 4].cpu().numpy().tobytes(), "little")
        sizes.append(h)
        pos += 4 + (h & 0x7FFFFFFF)
    assert sizes[7] == 0x80000000 | block and all(s < block for i, s in enumerate(sizes) if i != 7)
    assert int.from_bytes(frame[pos:pos + 4].cpu().numpy().tobytes(), "little") == 0 and pos + 4 == n
    del frame, out, inp
    zl.lib().zlz4_release_device_cache()
    torch.cuda.empty_cache()


def test_decompress_batch_output_offsets_cross_2_pow_32(zl, gpu):
    """196 608 x 64 KiB blocks = 12 GiB of output: offsets beyond 2^32 and 2^33 (configs[2] is this shape x 5.3)."""
    import bench
    nblocks, block = 196608, 65536
    slot = (zl.compressBound(block) + 15) // 16 * 16
    ar = torch.arange(nblocks, dtype=torch.int64, device=gpu)
    in_off, slot_off = ar * block, ar * slot
    in_len = torch.full((nblocks,), block, dtype=torch.int32, device=gpu)
    slot_cap = torch.full((nblocks,), slot, dtype=torch.int32, device=gpu)
    comp = torch.empty(nblocks * slot, dtype=torch.uint8, device=gpu)
    csize = torch.empty(nblocks, dtype=torch.int64, device=gpu)
    chunk = 65536
    for c0 in range(0, nblocks, chunk):
        part = bench.make_device_blocks("text", chunk, block, gpu, seed=100 + c0 // chunk)
        zl.batch_compress_fast(part, in_off[:chunk], in_len[:chunk], comp[c0 * slot:], slot_off[:chunk], slot_cap[:chunk],
                               csize[c0:c0 + chunk], block, 1)
        torch.cuda.synchronize()
        del part
    assert int(csize.min()) > 0 and int(csize.max()) < block
    out = torch.empty((nblocks, block), dtype=torch.uint8, device=gpu)
    dsize = torch.full((nblocks,), -999, dtype=torch.int64, device=gpu)
    assert int(in_off[-1]) + block > (1 << 33)
    zl.batch_decompress_safe(comp, slot_off, csize.to(torch.int32), out, in_off, in_len, dsize)
    torch.cuda.synchronize()
    assert bool((dsize == block).all())
    for c0 in range(0, nblocks, chunk):
        part = bench.make_device_blocks("text", chunk, block, gpu, seed=100 + c0 // chunk)
        assert torch.equal(out[c0:c0 + chunk], part), "chunk %d" % (c0 // chunk)
        del part
    del out, comp
    torch.cuda.empty_cache()


def test_hc_long_runs_are_not_quadratic(zl, oracle, gpu):
    """Level 9 on blocks that are one long run: every speculative walk of the parse-aware search would count the same
    huge match; the frontier of the walk from 0 retires them and the count itself is wave-cooperative.  Bit-exact and
    bounded in time (the search-every-position kernel of round 1 needed minutes for the 4 MiB case)."""
    import time
    import gpu_harness as gh
    items = [bytes(4 << 20), b"\xAB" * (1 << 20) + bytes(dg.text_bytes(3000, 5)), bytes(65536), b"ab" * 40000,
             bytes(dg.text_bytes(50000, 6)) + bytes(200000)]
    t0 = time.time()
    got = gh.compress_hc(zl, items, gpu, 9)
    dt = time.time() - t0
    for b, (n, c) in zip(items, got):
        assert c == oracle.compress_hc(b, 9), len(b)
    assert dt < 20.0, "level 9 on long runs took %.1f s" % dt
