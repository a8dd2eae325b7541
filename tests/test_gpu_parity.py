"""HIP path vs the oracle, bit-exact, through the C ABI (run on the GPU box: pytest -m gpu)."""
import numpy as np
import pytest

import cases
import datagen as dg
import gpu_harness as gh

pytestmark = pytest.mark.gpu

ALL_INPUTS = cases.reference_test_inputs() + list(cases.kat_inputs().items()) + cases.seeded_cases(heavy=True)


def _cmp(names, got, want):
    bad = []
    for name, (n, data), w in zip(names, got, want):
        if isinstance(w, int):
            if n != w:
                bad.append("%s: status %d, oracle %d" % (name, n, w))
        elif n != len(w) or data != w:
            first = next((i for i, (a, b) in enumerate(zip(data, w)) if a != b), min(len(data), len(w)))
            bad.append("%s: size %d vs oracle %d, first diff at %d" % (name, n, len(w), first))
    assert not bad, "%d/%d mismatches: %s" % (len(bad), len(names), "; ".join(bad[:8]))


def test_compress_default_bit_exact(zl, oracle, gpu):
    names = [n for n, _ in ALL_INPUTS]
    items = [b for _, b in ALL_INPUTS]
    got = gh.compress_fast(zl, items, gpu)
    want = [oracle.compress_default(b) for b in items]
    _cmp(names, got, want)


@pytest.mark.parametrize("accel", [2, 7, 63, 64, 65, 1000, 65537, 0])
def test_compress_fast_acceleration_bit_exact(zl, oracle, gpu, accel):
    sel = [(n, b) for n, b in ALL_INPUTS if len(b) <= 70000]
    got = gh.compress_fast(zl, [b for _, b in sel], gpu, accel=accel)
    want = [oracle.compress_fast(b, accel) for _, b in sel]
    _cmp([n for n, _ in sel], got, want)


def test_decompress_safe_bit_exact(zl, oracle, gpu):
    names, comp, caps, want = [], [], [], []
    for n, b in ALL_INPUTS:
        for lvl in (0, 9):
            c = oracle.compress_default(b) if lvl == 0 else oracle.compress_hc(b, 9)
            names.append("%s/l%d" % (n, lvl))
            comp.append(c)
            caps.append(len(b))
            want.append(oracle.decompress_safe(c, len(b)))
            assert want[-1] == b
    got = gh.decompress(zl, comp, caps, gpu)
    _cmp(names, got, want)


def test_decompress_capacity_larger_and_smaller(zl, oracle, gpu):
    names, comp, caps, want = [], [], [], []
    for n, b in ALL_INPUTS[:40] + cases.seeded_cases()[-30:]:
        c = oracle.compress_default(b)
        for cap in (len(b) + 100, max(0, len(b) - 1), len(b) // 2, 0, 1):
            names.append("%s/cap%d" % (n, cap))
            comp.append(c)
            caps.append(cap)
            want.append(oracle.decompress_safe(c, cap))
    got = gh.decompress(zl, comp, caps, gpu)
    # on error only the status is specified; on success bytes must match
    _cmp(names, got, want)


def test_decompress_malformed_status_parity(zl, oracle, gpu):
    """Truncations, bit flips and hand-made bad streams: same status class as the reference algorithm."""
    base = [oracle.compress_default(b) for _, b in ALL_INPUTS if 20 <= len(b) <= 5000][:30]
    rng = np.random.default_rng(1234)
    names, comp, caps = [], [], []
    for i, c in enumerate(base):
        for cut in (1, 2, 3, len(c) // 2, len(c) - 1):
            names.append("trunc%d/%d" % (i, cut)); comp.append(c[:cut]); caps.append(8192)
        for k in range(6):
            m = bytearray(c)
            pos = int(rng.integers(0, len(m)))
            m[pos] ^= 1 << int(rng.integers(0, 8))
            names.append("flip%d/%d" % (i, k)); comp.append(bytes(m)); caps.append(8192)
    hand = [b"\x00", b"\x10", b"\x10A", b"\x1fA\x00\x00", b"\x00\x00\x00", b"\x10A\x01", b"\x10A\x02\x00",
            b"\xf0", b"\xf0\xff", b"\xf0\xff\xff\x00", b"\x0f\x01\x00", b"\x1fA\x01\x00\xff", b"\x40ABCD\x04\x00",
            b"\x4fABCD\x04\x00\xff\xff\x00", b"\x40ABCD\x05\x00", b"\x00\x01\x00", b"\x11A\x01\x00\x10"]
    for i, h in enumerate(hand):
        for cap in (0, 3, 4, 64, 100000):
            names.append("hand%d/cap%d" % (i, cap)); comp.append(h); caps.append(cap)
    want = []
    for c, cap in zip(comp, caps):
        w = oracle.decompress_safe(c, cap)
        want.append(w)
    got = gh.decompress(zl, comp, caps, gpu)
    _cmp(names, got, want)


@pytest.mark.parametrize("level", [9, 3, 4, 5, 6, 7, 8, 2, 10, 11, 12])
def test_compress_hc_bit_exact(zl, oracle, gpu, level):
    """levels 3-9: hash chain kernels; 2: lz4mid; 10-12: lz4opt (incl. the reference's level-10 early-encode quirk)."""
    sel = ALL_INPUTS if level in (9, 2) else [(n, b) for n, b in ALL_INPUTS if len(b) <= 70000]
    got = gh.compress_hc(zl, [b for _, b in sel], gpu, level)
    want = [oracle.compress_hc(b, level) for _, b in sel]
    _cmp([n for n, _ in sel], got, want)


@pytest.mark.parametrize("level", [3, 9, 12])
def test_compress_hc_link_build_step_boundaries(zl, oracle, gpu, level):
    """k_hc_build_links hands 256-position steps round robin to four wavefronts and flushes its staging area every 4096
    positions: block sizes whose number of insertable positions (n - 11) sits on and around those boundaries, on both
    sides of the 16-bit / 32-bit link switch (65547 / 65548 bytes)."""
    sizes = [266, 267, 268, 523, 1034, 1035, 1036, 1291, 4106, 4107, 4108, 8203, 12299, 12300, 65547, 65548, 69643, 69644]
    if level != 12:
        sizes += [131083, 131084]
    names, items = [], []
    for dist in ("text", "mixed", "reptext"):
        for n in sizes:
            names.append("%s/%d" % (dist, n)); items.append(bytes(dg.GENERATORS[dist](n, 31 + n)))
    got = gh.compress_hc(zl, items, gpu, level)
    want = [oracle.compress_hc(b, level) for b in items]
    _cmp(names, got, want)


def test_compress_output_too_small_parity(zl, oracle, gpu):
    """Destination smaller than the bound: identical OutputTooSmall / success decision (Appendix A Q10, H7)."""
    names, items, caps = [], [], []
    for n, b in ALL_INPUTS:
        if not (13 <= len(b) <= 20000):
            continue
        full = len(oracle.compress_default(b))
        for cap in (full, full - 1, full + 1, full // 2, 1, 0, 2, 3):
            names.append("%s/cap%d" % (n, cap)); items.append(b); caps.append(max(0, cap))
    got = gh.compress_fast(zl, items, gpu, caps=caps)
    want = [oracle.compress_default(b, cap=c) for b, c in zip(items, caps)]
    _cmp(names, got, want)


def test_batch_of_64k_blocks_all_distributions(zl, oracle, gpu):
    """configs[1]-shaped batch (64 KiB blocks) at a size the oracle finishes in seconds."""
    for dist in dg.GENERATORS:
        blocks = dg.make_blocks(dist, 96, 65536, seed=21)
        items = [bytes(b) for b in blocks]
        got = gh.compress_fast(zl, items, gpu)
        want = [oracle.compress_default(b) for b in items]
        _cmp(["%s#%d" % (dist, i) for i in range(len(items))], got, want)
        dec = gh.decompress(zl, want, [65536] * len(items), gpu)
        _cmp(["%s#%d" % (dist, i) for i in range(len(items))], dec, items)


def test_single_buffer_entry_points(zl, oracle, gpu):
    """The root.zig-shaped host-pointer calls (stage -> kernel -> copy back)."""
    for name, b in cases.reference_test_inputs()[:12]:
        c = zl.compressDefault(b)
        assert c == oracle.compress_default(b), name
        assert zl.decompressSafe(c, len(b)) == b, name
        h = zl.compressHC(b, 9)
        assert h == oracle.compress_hc(b, 9), name
        assert zl.decompressSafe(h, len(b) + 7) == b, name
    for name, b in cases.reference_test_inputs()[:8]:          # src/test_compat.zig:112-123 uses levels 2..12
        for lvl in (2, 10, 12, 1, 0, 13, -3):                   # <2 -> 9, >12 -> 12 (src/lz4hc.zig:1445)
            assert zl.compressHC(b, lvl) == oracle.compress_hc(b, lvl), (name, lvl)
    with pytest.raises(zl.Lz4Error) as e:
        zl.decompressSafe(b"\x1fA\x00\x00", 100)
    assert e.value.name == "CorruptedData"
    with pytest.raises(zl.Lz4Error) as e:
        zl.compressDefault(b"A" * 100, dst_cap=3)
    assert e.value.name == "OutputTooSmall"


def test_rank4_entry_points(zl, oracle, gpu):
    """compressFastExtState, compressDestSize, decompressSafePartial, sizeofState."""
    assert zl.sizeofState() == 16384
    b = bytes(dg.text_bytes(50000, 1))
    assert zl.compressFastExtState(16384, b, 1) == oracle.compress_fast_ext_state(16384, b, 1)
    assert zl.compressFastExtState(20000, b, 9) == oracle.compress_fast(b, 9)
    with pytest.raises(zl.Lz4Error) as e:
        zl.compressFastExtState(16383, b, 1)
    assert e.value.name == "InvalidState" and oracle.compress_fast_ext_state(16383, b, 1) == -5
    for data in (b, bytes(dg.random_bytes(30000, 2)), b"\0" * 40000, b[:100], b""):
        for cap in (0, 1, 5, 13, 100, 1000, 10000, len(data), len(data) + 200, 70000):
            want_r, want_consumed = oracle.compress_dest_size(data, cap)
            out, consumed = zl.compressDestSize(data, cap)
            assert (len(out), consumed) == (want_r, want_consumed), (len(data), cap)
            if consumed:
                assert out == oracle.compress_default(data[:consumed], cap=cap), (len(data), cap)
                assert zl.decompressSafe(out, consumed) == data[:consumed]
    c = oracle.compress_default(b)
    for cap, target in ((50000, 50000), (50000, 100), (100, 200), (50000, 0), (60000, 55000), (0, 0), (50000, 49999)):
        want = oracle.decompress_safe_partial(c, cap, target)
        try:
            got = zl.decompressSafePartial(c, cap, target)
        except zl.Lz4Error as e:
            got = e.code
        assert got == want, (cap, target)
    for stream in (b"\x00", b"\x10A", b"\x00\x01\x00", b"\x0f\x01\x00", b"\xf0", b"\x00\x00\x00"):
        want = oracle.decompress_safe_partial(stream, 10, 0)
        try:
            got = zl.decompressSafePartial(stream, 10, 0)
        except zl.Lz4Error as e:
            got = e.code
        assert got == want, stream


def test_decoder_batch_path_malformed_and_capacity_stress(zl, oracle, gpu):
    """The wave decoder's batch path only runs on streams of >= 68 bytes with output room to spare, i.e. not on the
    small cases above.  Text blocks of 4..64 KiB (fast, accelerated and HC streams): several hundred corruptions
    (bit flips, byte replacements, 0x00 / 0xFF runs that fake offsets of 0 and 255-chains), truncations and a
    capacity sweep around the exact size.  Status must equal the oracle's; on success the bytes too."""
    rng = np.random.default_rng(20240607)
    blocks = [bytes(dg.text_bytes(n, 77 + i)) for i, n in enumerate((4096, 20000, 65536, 65536))]
    blocks.append(bytes(dg.mixed_bytes(65536, 5)))
    names, comp, caps = [], [], []
    for bi, b in enumerate(blocks):
        streams = [oracle.compress_default(b), oracle.compress_fast(b, 7), oracle.compress_hc(b, 9)]
        for si, c in enumerate(streams):
            n = len(b)
            for cap in (n, n + 1, n + 31, n + 32, n - 1, n - 4, n - 17, n - 31, n - 32, n - 33, n - 100, n // 2,
                        int(rng.integers(1, n)), int(rng.integers(1, n))):
                names.append("b%d/s%d/cap%d" % (bi, si, cap)); comp.append(c); caps.append(cap)
            for k in range(24):
                m = bytearray(c)
                pos = int(rng.integers(0, len(m)))
                kind = k % 6
                if kind == 0: m[pos] ^= 1 << int(rng.integers(0, 8))
                elif kind == 1: m[pos] = int(rng.integers(0, 256))
                elif kind == 2: m[pos:pos + 2] = b"\x00\x00"              # an offset of 0 somewhere
                elif kind == 3: m[pos:pos + 4] = b"\xff\xff\xff\xff"      # 255-chains / huge lengths
                elif kind == 4: m[pos] = 0xF0 | (m[pos] & 15)             # literal-length extension
                else: m[pos] = (m[pos] & 0xF0) | 15                       # match-length extension
                names.append("b%d/s%d/corrupt%d@%d" % (bi, si, kind, pos)); comp.append(bytes(m)); caps.append(n)
            for k in range(6):
                cut = int(rng.integers(1, len(c)))
                names.append("b%d/s%d/trunc%d" % (bi, si, cut)); comp.append(c[:cut]); caps.append(n)
    want = [oracle.decompress_safe(c, cap) for c, cap in zip(comp, caps)]
    assert sum(isinstance(w, int) for w in want) > 100 and sum(not isinstance(w, int) for w in want) > 50
    got = gh.decompress(zl, comp, caps, gpu)
    _cmp(names, got, want)


def test_batch_max_in_len_understated(zl, oracle, gpu):
    """A block longer than the call's max_in_len is refused (InvalidState) and its neighbours are unaffected."""
    import torch
    items = [bytes(dg.text_bytes(3000, 1)), bytes(dg.text_bytes(70000, 2)), bytes(dg.text_bytes(5000, 3)), bytes(dg.text_bytes(66000, 4))]
    for kind, level, lie in (("fast", 0, 8192), ("hc", 9, 8192), ("hc", 12, 8192), ("hc", 2, 8192), ("hc", 9, 65536), ("hc", 2, 65536),
                             ("fast", 0, 65536)):
        buf, offs, lens = gh._pack(items)
        caps = np.array([zl.compressBound(len(b)) for b in items], dtype=np.int64)
        out_offs = np.concatenate([[0], np.cumsum((caps + 15) // 16 * 16 + 64)[:-1]]).astype(np.int64)
        d_in = torch.from_numpy(buf).to(gpu)
        d_out = torch.zeros(int(out_offs[-1] + caps[-1] + 80), dtype=torch.uint8, device=gpu)
        res = torch.full((len(items),), -999, dtype=torch.int64, device=gpu)
        a = (torch.from_numpy(offs).to(gpu), torch.from_numpy(lens.astype(np.uint32).view(np.int32)).to(gpu), d_out,
             torch.from_numpy(out_offs).to(gpu), torch.from_numpy(caps.astype(np.uint32).view(np.int32)).to(gpu), res)
        if kind == "fast":
            zl.batch_compress_fast(d_in, *a, lie, 1)
        else:
            ws = torch.empty(max(16, zl.batch_compress_hc_workspace(len(items), lie)), dtype=torch.uint8, device=gpu)
            zl.batch_compress_hc(d_in, *a, lie, level, ws)
        torch.cuda.synchronize()
        r = res.cpu().numpy()
        o = d_out.cpu().numpy()
        for i, b in enumerate(items):
            if len(b) > lie:
                assert r[i] == -5, (kind, level, lie, i, r[i])
            else:
                want = oracle.compress_default(b) if kind == "fast" else oracle.compress_hc(b, level)
                assert r[i] == len(want) and bytes(o[out_offs[i]: out_offs[i] + r[i]]) == want, (kind, level, lie, i)


def test_compress_hc_ext_state(zl, oracle, gpu):
    """lz4hc.compressHCExtState / sizeofStateHC (src/lz4hc.zig:1457-1494) with a fresh context: the level rules differ
    from compressHC's (level < 1 -> 9, level 1 -> the table's lz4mid row, > 12 -> 12; dst.len == 0 -> OutputTooSmall)."""
    n_state = zl.sizeofStateHC()
    assert n_state >= 32768 * 4 + 65536 * 2
    b = bytes(dg.text_bytes(30000, 77))
    for level, same_as in ((9, 9), (0, 9), (-3, 9), (1, 2), (2, 2), (3, 3), (12, 12), (40, 12)):
        assert zl.compressHCExtState(n_state, b, level) == oracle.compress_hc(b, same_as), level
    assert zl.compressHCExtState(n_state, b"", 9) == b""
    for bad_state, cap, name in ((n_state - 1, None, "InvalidState"), (n_state, 0, "OutputTooSmall")):
        with pytest.raises(zl.Lz4Error) as e:
            zl.compressHCExtState(bad_state, b, 9, dst_cap=cap)
        assert e.value.name == name


def test_batch_calls_capture_into_a_hip_graph(zl, oracle, gpu):
    """the batch calls allocate nothing and only enqueue work, so a caller can capture them into
    a hipGraph (launch-bound inner loops).  compressFast + compressHC(9) (which forks to its side stream and joins)
    + decompressSafe are captured once and replayed on new input; results must equal the oracle's."""
    import torch
    nblocks, block = 64, 65536
    slot = (zl.compressBound(block) + 15) // 16 * 16
    ar = torch.arange(nblocks, dtype=torch.int64, device=gpu)
    in_off, slot_off = ar * block, ar * slot
    in_len = torch.full((nblocks,), block, dtype=torch.int32, device=gpu)
    cap = torch.full((nblocks,), slot, dtype=torch.int32, device=gpu)
    inp = torch.zeros((nblocks, block), dtype=torch.uint8, device=gpu)
    comp_f = torch.zeros(nblocks * slot, dtype=torch.uint8, device=gpu)
    comp_h = torch.zeros(nblocks * slot, dtype=torch.uint8, device=gpu)
    out = torch.zeros((nblocks, block), dtype=torch.uint8, device=gpu)
    r_f = torch.zeros(nblocks, dtype=torch.int64, device=gpu)
    r_h = torch.zeros(nblocks, dtype=torch.int64, device=gpu)
    r_d = torch.zeros(nblocks, dtype=torch.int64, device=gpu)
    clen = torch.zeros(nblocks, dtype=torch.int32, device=gpu)
    ws = torch.empty(zl.batch_compress_hc_workspace(nblocks, block), dtype=torch.uint8, device=gpu)

    def work():
        zl.batch_compress_fast(inp, in_off, in_len, comp_f, slot_off, cap, r_f, block, 1)
        zl.batch_compress_hc(inp, in_off, in_len, comp_h, slot_off, cap, r_h, block, 9, ws)
        clen.copy_(r_h.to(torch.int32))
        zl.batch_decompress_safe(comp_h, slot_off, clen, out, in_off, in_len, r_d)

    first = torch.from_numpy(dg.make_blocks("text", nblocks, block, seed=31)).to(gpu)
    inp.copy_(first)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        work()                                   # warm-up outside the capture (lazy module load, side stream creation)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        work()
    second = torch.from_numpy(dg.make_blocks("mixed", nblocks, block, seed=32)).to(gpu)
    second[: nblocks // 2] = torch.from_numpy(dg.make_blocks("text", nblocks // 2, block, seed=33)).to(gpu)
    inp.copy_(second)
    for t in (comp_f, comp_h, out, r_f, r_h, r_d):
        t.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, inp) and bool((r_d == block).all())
    host = second.cpu().numpy()
    cf, ch = comp_f.cpu().numpy(), comp_h.cpu().numpy()
    for i in (0, 7, nblocks // 2, nblocks - 1):
        b = bytes(host[i])
        assert bytes(cf[i * slot: i * slot + int(r_f[i])]) == oracle.compress_default(b), i
        assert bytes(ch[i * slot: i * slot + int(r_h[i])]) == oracle.compress_hc(b, 9), i


def test_decompress_long_overlapping_matches_of_every_period_class(zl, oracle, gpu):
    """Matches that overlap their own output (offset < match length, src/lz4.zig:235-241) with periods below 16, of
    16..63, 64..1023 (copied by doubling: offset, 2 x offset, 4 x offset ... bytes at a time) and >= 1024 bytes, at lengths
    that are and are not multiples of 16 or of the period: random period content repeated, compressed by the oracle
    (fast and HC give different offset / length splits), decoded on the device."""
    rng = np.random.default_rng(99)
    items = []
    for period in (1, 2, 3, 7, 15, 16, 17, 31, 63, 64, 65, 100, 255, 256, 257, 500, 1000, 1023, 1024, 1025, 3000):
        for total in (period * 3 + 5, 4096 + period, 65536, 30011):
            pat = rng.integers(0, 256, period, dtype=np.uint8).tobytes()
            head = rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8).tobytes()
            items.append((head + pat * (total // period + 2))[:max(total, 13)])
    comp = [oracle.compress_default(b) for b in items] + [oracle.compress_hc(b, 9) for b in items]
    caps = [len(b) for b in items] * 2
    got = gh.decompress(zl, comp, caps, gpu)
    _cmp(["periodic%d" % i for i in range(len(comp))], got, items + items)


@pytest.mark.parametrize("level", [9, 4, 12])
def test_compress_hc_periodic_inputs(zl, oracle, gpu, level):
    """Blocks with a period (random content repeated every 1 .. 40000 bytes, some with noise in the middle or two periods
    in a row): every start point's first match is the same long run, which the search kernel counts once and shares
    between its walks (counted runs, k_hc_seg_search in zlz4_compress_hc.hip; at level 12 k_hc_search, which searches
    every position, shares it between the lanes of a wavefront and skips candidates that cannot be longer).  Bytes vs the oracle
    (insertAndFindBestMatch / lz4Count, src/lz4hc.zig:540-640, :234-264)."""
    rng = np.random.default_rng(4242 + level)
    items = []
    for period in (1, 2, 3, 5, 16, 40, 63, 64, 100, 255, 256, 257, 1000, 1024, 4096, 5000, 40000):
        for total in (65536, 30011, period * 2 + 70):
            pat = rng.integers(0, 256, period, dtype=np.uint8).tobytes()
            b = bytearray((pat * (total // period + 2))[:max(total, 13)])
            items.append(bytes(b))
            if total > 20000:
                for _ in range(3):                                # a few damaged bytes: runs that end early
                    b[int(rng.integers(0, len(b)))] ^= 0x55
                items.append(bytes(b))
                pat2 = rng.integers(0, 256, max(1, period // 2 + 1), dtype=np.uint8).tobytes()
                half = len(b) // 2
                items.append(bytes(b[:half]) + (pat2 * (half // len(pat2) + 2))[:len(b) - half])   # two periods in a row
    got = gh.compress_hc(zl, items, gpu, level)
    want = [oracle.compress_hc(b, level) for b in items]
    _cmp(["periodic%d" % i for i in range(len(items))], got, want)


@pytest.mark.parametrize("level", [9, 12])
def test_compress_hc_periodic_large_blocks(zl, oracle, gpu, level):
    """The same kind of input in blocks > 64 KiB, i.e. through the search kernels' HBM-link variants (32-bit links, the
    counted runs packed 16 + 24 + 24 bits): periods 1, 256, 5000, with and without damaged bytes, 150 .. 300 KB."""
    rng = np.random.default_rng(777 + level)
    items = []
    for period, total in ((1, 300000), (256, 200000), (5000, 150000), (256, 262144 + 17)):
        pat = rng.integers(0, 256, period, dtype=np.uint8).tobytes()
        b = bytearray((pat * (total // period + 2))[:total])
        items.append(bytes(b))
        for _ in range(4):
            b[int(rng.integers(0, len(b)))] ^= 0x3C
        items.append(bytes(b))
    got = gh.compress_hc(zl, items, gpu, level)
    want = [oracle.compress_hc(b, level) for b in items]
    _cmp(["periodic_large%d" % i for i in range(len(items))], got, want)
