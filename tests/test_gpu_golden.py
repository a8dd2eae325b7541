"""HIP path vs the committed golden fixtures (tests/golden/*.json) and full-size property checks at the
BASELINE.json configuration sizes (too large for the oracle: round trip + checksum-of-sizes + idempotence)."""
import hashlib
import json
import os

import pytest
import torch

import cases
import gpu_harness as gh

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_hip_matches_committed_golden_vectors(zl, gpu):
    gold = {v["name"]: v for v in json.load(open(os.path.join(HERE, "golden", "oracle_vectors.json")))["vectors"]}
    ins = cases.reference_test_inputs() + list(cases.kat_inputs().items()) + cases.seeded_cases()
    items = [b for _, b in ins]
    fast = gh.compress_fast(zl, items, gpu)
    hc = {lvl: gh.compress_hc(zl, items, gpu, lvl) for lvl in (3, 6, 9, 2, 10, 11, 12)}
    for k, (name, b) in enumerate(ins):
        v = gold[name]
        assert (fast[k][0], hashlib.sha256(fast[k][1]).hexdigest()) == (v["fast"]["len"], v["fast"]["sha256"]), name
        for lvl in (3, 6, 9, 2, 10, 11, 12):
            if "hc%d" % lvl in v:
                assert hashlib.sha256(hc[lvl][k][1]).hexdigest() == v["hc%d" % lvl]["sha256"], (name, lvl)
        if len(b) <= 20000:
            assert hashlib.sha256(zl.lz4f.compressFrame(b)).hexdigest() == v["frame_default"]["sha256"], name


def test_hip_matches_appendix_b_kats(zl, gpu):
    kat = json.load(open(os.path.join(HERE, "golden", "kat_appendix_b.json")))
    from test_oracle import _kat_input
    fns = {"compressDefault": lambda b: zl.compressDefault(b), "compressHC9": lambda b: zl.compressHC(b, 9),
           "compressHC8": lambda b: zl.compressHC(b, 8), "compressFrame": lambda b: zl.lz4f.compressFrame(b)}
    for v in kat["vectors"]:
        data = _kat_input(kat["inputs"][v["input"]])
        out = fns[v["fn"]](data)
        if "hex" in v:
            assert out.hex() == v["hex"], (v["input"], v["fn"])
        else:
            assert hashlib.sha256(out).hexdigest() == v["sha256"]


def test_hip_matches_round2_kats(zl, gpu):
    """kat_round2.json: level 2, levels 10-12, acceleration 7, blocks > 64 KiB, a checksummed frame."""
    from test_oracle import _check_round2, _round2_fn
    kat = json.load(open(os.path.join(HERE, "golden", "kat_round2.json")))
    _check_round2(kat, lambda fn: _round2_fn(zl.compressFast, zl.compressHC, zl.lz4f.compressFrame, zl.Prefs, fn))


def _roundtrip(zl, dev, dist, nblocks, block, hc_level=None):
    import bench
    inp = bench.make_device_blocks(dist, nblocks, block, dev, seed=77)
    slot = (zl.compressBound(block) + 15) // 16 * 16
    ar = torch.arange(nblocks, dtype=torch.int64, device=dev)
    in_len = torch.full((nblocks,), block, dtype=torch.int32, device=dev)
    cap = torch.full((nblocks,), slot, dtype=torch.int32, device=dev)
    comp = torch.empty(nblocks * slot, dtype=torch.uint8, device=dev)
    cs = torch.empty(nblocks, dtype=torch.int64, device=dev)
    if hc_level is None:
        zl.batch_compress_fast(inp, ar * block, in_len, comp, ar * slot, cap, cs, block, 1)
    else:
        ws = torch.empty(zl.batch_compress_hc_workspace(nblocks, block), dtype=torch.uint8, device=dev)
        zl.batch_compress_hc(inp, ar * block, in_len, comp, ar * slot, cap, cs, block, hc_level, ws)
    torch.cuda.synchronize()
    assert int(cs.min()) > 0 and int(cs.max()) <= zl.compressBound(block)
    out = torch.empty_like(inp)
    ds = torch.empty(nblocks, dtype=torch.int64, device=dev)
    zl.batch_decompress_safe(comp, ar * slot, cs.to(torch.int32), out, ar * block, in_len, ds)
    torch.cuda.synchronize()
    assert bool((ds == block).all()) and torch.equal(out, inp)
    # idempotence: a second compression of the same input gives the same sizes (deterministic kernels)
    cs2 = torch.empty_like(cs)
    comp2 = torch.empty_like(comp)
    if hc_level is None:
        zl.batch_compress_fast(inp, ar * block, in_len, comp2, ar * slot, cap, cs2, block, 1)
        torch.cuda.synchronize()
        assert torch.equal(cs, cs2)
    return int(cs.sum())


def test_config2_full_size_round_trip(zl, gpu):
    """BASELINE.json configs[1]: 65 536 x 64 KiB blocks (4 GiB), compressDefault -> decompressSafe."""
    total_c = _roundtrip(zl, gpu, "text", 65536, 65536)
    assert 1.5 < 65536 * 65536 / total_c < 3.0


@pytest.mark.parametrize("dist", ["mixed", "random", "zero", "ramp"])
def test_config2_other_distributions_round_trip(zl, gpu, dist):
    _roundtrip(zl, gpu, dist, 8192, 65536)


def test_config4_hc9_round_trip(zl, gpu):
    """BASELINE.json configs[3] shape (level 9, 64 KiB blocks) at 2048 blocks."""
    _roundtrip(zl, gpu, "text", 2048, 65536, hc_level=9)


def test_large_blocks_4mib_round_trip(zl, gpu):
    """config 5 block size: 4 MiB blocks use the 32-bit table variants of both compressors."""
    _roundtrip(zl, gpu, "text", 24, 4 << 20)
    _roundtrip(zl, gpu, "text", 6, 4 << 20, hc_level=9)


def test_frame_device_api_many_blocks(zl, gpu):
    """lz4f on a device-resident 1.2 GiB buffer with 64 KiB blocks (18 500 blocks + a short last one): the frame
    decoder then runs the one-lane-per-block kernel with exact per-block capacities; checksums on."""
    import ctypes as C
    import bench
    nblocks, block = 18500, 65536
    inp = bench.make_device_blocks("text", nblocks, block, gpu, seed=5).reshape(-1)
    inp = torch.cat([inp, inp[:12345]])
    n = inp.numel()
    prefs = zl.Prefs()
    prefs.block_size_id = 4
    prefs.block_checksum = 1
    prefs.content_checksum = 1
    bound = zl.lz4f.compressFrameBound(n, prefs)
    frame = torch.empty(bound, dtype=torch.uint8, device=gpu)
    out = torch.full((n + 64,), 0x5A, dtype=torch.uint8, device=gpu)
    L = zl.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fs = L.zlz4f_compress_frame_device(st, C.c_void_p(inp.data_ptr()), n, C.c_void_p(frame.data_ptr()), bound, C.byref(prefs))
    assert fs > 0, zl.error_name(fs)
    r = L.zlz4f_decompress_frame_device(st, C.c_void_p(frame.data_ptr()), fs, C.c_void_p(out.data_ptr()), n)
    assert r == n, zl.error_name(r)
    assert torch.equal(out[:n], inp) and bool((out[n:] == 0x5A).all())
    # corrupt one payload byte in the middle of the frame: block checksum must catch it (src/lz4f.zig:594-598)
    frame[fs // 2] ^= 0x40
    r = L.zlz4f_decompress_frame_device(st, C.c_void_p(frame.data_ptr()), fs, C.c_void_p(out.data_ptr()), n)
    assert zl.error_name(r) == "BlockChecksumInvalid"
