"""lz4f frame wrapper on the HIP path vs the oracle (bit-exact) and vs the `lz4` CLI (interop, as the
reference's src/test_compat.zig does).  Run on the GPU box: pytest -m gpu."""
import os
import shutil
import subprocess
import tempfile

import pytest

import cases
import datagen as dg

pytestmark = pytest.mark.gpu

LZ4_CLI = shutil.which("lz4") or ("/opt/conda/bin/lz4" if os.path.exists("/opt/conda/bin/lz4") else None)


def _prefs(P, **kw):
    p = P()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _pref_matrix():
    out = [dict()]
    for bsid in (0, 4, 5, 6, 7):
        out.append(dict(block_size_id=bsid, block_mode=1))
    out.append(dict(block_checksum=1))
    out.append(dict(content_checksum=1))
    out.append(dict(block_checksum=1, content_checksum=1, block_size_id=4, content_size=12345, dict_id=7))
    out.append(dict(compression_level=9))
    out.append(dict(compression_level=3, block_checksum=1))
    out.append(dict(compression_level=1, content_checksum=1))      # level 1 -> compressHC clamps to 9
    out.append(dict(compression_level=-5))                          # negative = fast
    out.append(dict(compression_level=2))                           # lz4mid
    out.append(dict(compression_level=12, block_checksum=1))        # lz4opt
    out.append(dict(compression_level=10))                          # lz4opt, targetLength 64
    return out


def _status(fn, *a):
    try:
        r = fn(*a)
        return len(r), r
    except Exception as e:      # zl.Lz4Error
        return e.code, None


def _inputs():
    ins = [(n, b) for n, b in cases.reference_test_inputs()]
    ins.append(("text 300000", bytes(dg.text_bytes(300000, 3))))
    ins.append(("random 70000 (stored blocks)", bytes(dg.random_bytes(70000, 4))))
    ins.append(("mixed 200001", bytes(dg.mixed_bytes(200001, 5))))
    ins.append(("exactly 65536", bytes(dg.text_bytes(65536, 6))))
    ins.append(("65537", bytes(dg.text_bytes(65537, 7))))
    return ins


def test_compress_frame_bit_exact(zl, oracle, gpu):
    bad = []
    for kw in _pref_matrix():
        for name, b in _inputs():
            if len(b) > 400000 and kw.get("compression_level", 0) > 0:
                continue
            want = oracle.compress_frame(b, _prefs(oracle.Prefs, **kw))
            got = zl.lz4f.compressFrame(b, _prefs(zl.Prefs, **kw))
            if got != want:
                bad.append("%s %s: %d vs %d bytes" % (kw, name, len(got), len(want)))
    assert not bad, bad[:6]


def test_compress_frame_bound_and_dst_too_small(zl, oracle, gpu):
    for kw in _pref_matrix():
        for n in (0, 1, 65536, 65537, 1 << 20, (1 << 22) + 5):
            assert zl.lz4f.compressFrameBound(n, _prefs(zl.Prefs, **kw)) == oracle.compress_frame_bound(n, _prefs(oracle.Prefs, **kw))
    b = b"A" * 1000
    with pytest.raises(zl.Lz4Error) as e:
        zl.lz4f.compressFrame(b, None, dst_cap=zl.lz4f.compressFrameBound(len(b)) - 1)
    assert e.value.name == "DstMaxSizeTooSmall"                      # src/lz4f.zig:363-366


def test_decompress_frame_bit_exact(zl, oracle, gpu):
    for kw in _pref_matrix():
        for name, b in _inputs():
            if len(b) > 400000 and kw.get("compression_level", 0) > 0:
                continue
            f = oracle.compress_frame(b, _prefs(oracle.Prefs, **kw))
            want = oracle.decompress_frame(f, len(b))
            if kw.get("compression_level") != 10:        # the reference's level-10 stream is not always decodable
                assert want == b
            for cap in (len(b), len(b) + 1000):
                got = _status(zl.lz4f.decompressFrame, f, cap)
                if isinstance(want, int):
                    assert got[0] == want, (kw, name, cap)
                elif cap == len(b):
                    assert got[1] == want, (kw, name, cap)


def _status(fn, *a):
    try:
        r = fn(*a)
        return len(r), r
    except Exception as e:      # zl.Lz4Error
        return e.code, None


def test_decompress_frame_error_parity(zl, oracle, gpu):
    """Corrupted / truncated / short-destination frames: same error as the reference algorithm (oracle)."""
    import numpy as np
    rng = np.random.default_rng(99)
    frames = []
    for kw in (dict(), dict(block_checksum=1, content_checksum=1), dict(content_checksum=1, block_size_id=4)):
        for name, b in _inputs()[:12] + _inputs()[-4:]:
            frames.append((b, oracle.compress_frame(b, _prefs(oracle.Prefs, **kw))))
    checked = 0
    for b, f in frames:
        variants = [f[:k] for k in (0, 3, 6, 7, 8, 10, len(f) // 2, len(f) - 1, len(f) - 4, len(f) - 5) if 0 <= k <= len(f)]
        for _ in range(6):
            m = bytearray(f)
            if m:
                m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
            variants.append(bytes(m))
        variants.append(f + b"trailing garbage")                     # ignored after the end mark (:563-575)
        for v in variants:
            for cap in (len(b), max(0, len(b) - 1), len(b) // 2, 0):
                w = oracle.decompress_frame(v, cap)
                g_code, g = _status(zl.lz4f.decompressFrame, v, cap)
                if isinstance(w, int):
                    assert g_code == w, ("status", len(v), cap, g_code, w)
                else:
                    assert g == w, ("bytes", len(v), cap)
                checked += 1
    assert checked > 1000


@pytest.mark.skipif(LZ4_CLI is None, reason="lz4 CLI not installed")
def test_cli_interop_both_directions(zl, gpu):
    """src/test_compat.zig groups 1-3: our frames decode with `lz4 -d`; `lz4` frames decode with ours."""
    six = [b"Hello World!", b"ABCDEFGH" * 125, cases.LOREM, bytes(dg.random_bytes(256, 12345)), b"",
           bytes(i % 256 for i in range(100000))]
    with tempfile.TemporaryDirectory() as td:
        def cli_decode(frame):
            a, b_ = os.path.join(td, "a.lz4"), os.path.join(td, "a.out")
            open(a, "wb").write(frame)
            subprocess.check_call([LZ4_CLI, "-d", "-f", "-q", a, b_])
            return open(b_, "rb").read()

        def cli_encode(data):
            a, b_ = os.path.join(td, "b.in"), os.path.join(td, "b.lz4")
            open(a, "wb").write(data)
            subprocess.check_call([LZ4_CLI, "-f", "-q", a, b_])
            return open(b_, "rb").read()

        for d in six:                                                # group 1 (:84-92)
            assert cli_decode(zl.lz4f.compressFrame(d)) == d
        for d in six:                                                # group 2 (:98-106)
            assert zl.lz4f.decompressFrame(cli_encode(d), len(d) + 16) == d
        for lvl in range(2, 13):                                     # group 3 (:112-123): levels 2..12
            p = zl.Prefs()
            p.compression_level = lvl
            assert cli_decode(zl.lz4f.compressFrame(six[1], p)) == six[1]
        p = zl.Prefs(); p.block_checksum = 1; p.content_checksum = 1; p.block_size_id = 5; p.block_mode = 1
        big = bytes(dg.text_bytes(700000, 8))
        assert cli_decode(zl.lz4f.compressFrame(big, p)) == big
