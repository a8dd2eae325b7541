"""CPU tests of the oracle (oracle/lz4_oracle.c): SURVEY Appendix B KATs, every assertion the reference's own
tests make on this path (round trips, inequalities, interop, corruption detection), liblz4 / lz4-CLI cross checks,
and the frozen golden fixtures."""
import ctypes as C
import hashlib
import json
import os
import shutil
import subprocess
import tempfile

import pytest

import cases
import datagen as dg

HERE = os.path.dirname(os.path.abspath(__file__))
LZ4_CLI = shutil.which("lz4") or ("/opt/conda/bin/lz4" if os.path.exists("/opt/conda/bin/lz4") else None)


def _kat_input(spec):
    if "ascii" in spec:
        return spec["ascii"].encode()
    if "repeat" in spec:
        return spec["repeat"][0].encode() * spec["repeat"][1]
    if "concat" in spec:
        return b"".join(_kat_input(s) for s in spec["concat"])
    if spec.get("special") == "S1":
        return cases.kat_inputs()["S1"]
    if "gen" in spec:
        kind, n, seed = spec["gen"]
        return bytes(dg.GENERATORS[kind](n, seed))
    raise ValueError(spec)


def _round2_fn(mod_fast, mod_hc, mod_frame, prefs_cls, fn):
    """kat_round2.json function names -> a callable on bytes."""
    if fn == "compressDefault":
        return lambda b: mod_fast(b, 1)
    if fn.startswith("compressFast"):
        return lambda b, a=int(fn[len("compressFast"):]): mod_fast(b, a)
    if fn.startswith("compressHC"):
        return lambda b, l=int(fn[len("compressHC"):]): mod_hc(b, l)
    if fn == "compressFrameChecksums":
        def frame(b):
            p = prefs_cls(); p.block_size_id = 4; p.block_mode = 1; p.block_checksum = 1; p.content_checksum = 1
            return mod_frame(b, p)
        return frame
    raise ValueError(fn)


def _check_round2(kat, call):
    for v in kat["vectors"]:
        data = _kat_input(kat["inputs"][v["input"]])
        assert hashlib.sha256(data).hexdigest() == v["in_sha256"], v["input"]
        out = call(v["fn"])(data)
        assert len(out) == v["len"], (v["input"], v["fn"], len(out))
        if "hex" in v:
            assert out.hex() == v["hex"], (v["input"], v["fn"])
        else:
            assert hashlib.sha256(out).hexdigest() == v["sha256"], (v["input"], v["fn"])


def test_round2_known_answers(oracle):
    """Level 2, levels 10-12, acceleration 7, blocks > 64 KiB, a checksummed frame: three hand traces and the vectors on
    which the independent Python restatement and this oracle agree (tests/golden/make_kat_round2.py)."""
    kat = json.load(open(os.path.join(HERE, "golden", "kat_round2.json")))
    assert sum(v["provenance"] == "hand" for v in kat["vectors"]) >= 3
    _check_round2(kat, lambda fn: _round2_fn(oracle.compress_fast, oracle.compress_hc, oracle.compress_frame, oracle.Prefs, fn))


def test_appendix_b_known_answers(oracle):
    kat = json.load(open(os.path.join(HERE, "golden", "kat_appendix_b.json")))
    fns = {
        "compressDefault": oracle.compress_default,
        "compressHC9": lambda b: oracle.compress_hc(b, 9),
        "compressHC8": lambda b: oracle.compress_hc(b, 8),
        "compressFrame": lambda b: oracle.compress_frame(b),
    }
    for v in kat["vectors"]:
        data = _kat_input(kat["inputs"][v["input"]])
        out = fns[v["fn"]](data)
        if "hex" in v:
            assert out.hex() == v["hex"], (v["input"], v["fn"])
        else:
            assert hashlib.sha256(data).hexdigest() == v["in_sha256"]
            assert len(out) == v["len"] and hashlib.sha256(out).hexdigest() == v["sha256"], (v["input"], v["fn"])
    s1 = oracle.compress_default(cases.kat_inputs()["S1"])
    assert s1[:2] == b"\xff\x57" and s1[104:108] == bytes([0x5B, 0, 0, 0xD0])     # Appendix B, S1 row


def test_reference_round_trips(oracle):
    """src/test.zig (six cases), src/test_compat.zig inputs, src/test_lz4hc.zig inputs: decompress(compress(x)) == x."""
    for name, b in cases.reference_test_inputs():
        c = oracle.compress_default(b)
        assert oracle.decompress_safe(c, len(b)) == b, name
        for lvl in (3, 9):
            h = oracle.compress_hc(b, lvl)
            assert oracle.decompress_safe(h, len(b)) == b, (name, lvl)


def test_reference_inequalities(oracle):
    rep = b"ABCD" * 1000
    assert len(oracle.compress_hc(rep, 9)) < len(rep)                        # src/test_lz4hc.zig:81
    rnd = bytes(dg.random_bytes(1000, 12345))
    assert len(oracle.compress_hc(rnd, 9)) >= len(rnd)                       # src/test_lz4hc.zig:142
    for lvl in range(3, 9):                                                  # deeper search never hurts on this input
        assert len(oracle.compress_hc(cases.LOREM * 20, lvl + 1)) <= len(oracle.compress_hc(cases.LOREM * 20, lvl)) + 8
    t = bytes(dg.text_bytes(30000, 2)) + cases.LOREM * 20
    s9, s10, s11, s12 = (len(oracle.compress_hc(t, l)) for l in (9, 10, 11, 12))
    assert s10 <= s9 and s11 <= s10 and s12 <= s11                           # src/test_lz4hc.zig:424-426
    for lvl in (2, 11, 12):                                                  # every level the reference routes (:72-86)
        for name, b in cases.reference_test_inputs():
            if len(b) <= 100000:
                assert oracle.decompress_safe(oracle.compress_hc(b, lvl), len(b)) == b, (name, lvl)
    # level 10 round trips the reference itself asserts: "ABCD" x 500 at every level 2..12 (src/test_lz4hc.zig:151-186) and
    # the 500-byte multi-pattern buffer at 10 (:375-433, with size10 <= size9 etc.) -- the early-encode defect of lz4opt
    #  must not show on them
    abcd500 = b"ABCD" * 500
    for lvl in range(2, 13):
        assert oracle.decompress_safe(oracle.compress_hc(abcd500, lvl), len(abcd500)) == abcd500, lvl
    pats = [b"ABCD", b"XYZ", b"PQR", b"123", b"abc"]
    multi = (b"".join(p_ * 3 for p_ in pats) * 40)[:500]
    sizes = {l: len(oracle.compress_hc(multi, l)) for l in (9, 10, 11, 12)}
    assert sizes[10] <= sizes[9] and sizes[11] <= sizes[10] and sizes[12] <= sizes[11]
    for lvl in (10, 11, 12):
        assert oracle.decompress_safe(oracle.compress_hc(multi, lvl), len(multi)) == multi, lvl
    assert oracle.compress_default(b"") == b"" and oracle.compress_hc(b"", 9) == b""   # src/test.zig:182, lz4hc.zig:1443
    assert oracle.decompress_safe(b"", 10) == b"" and oracle.decompress_safe(b"\x10A", 0) == b""   # lz4.zig:97-98


def test_frame_reference_assertions(oracle):
    data = bytes((i // 16) % 256 for i in range(1 << 20))                    # src/test_lz4f.zig:98-105
    f = oracle.compress_frame(data)
    assert f[:4] == bytes([0x04, 0x22, 0x4D, 0x18])                          # magic, src/test_lz4f.zig:50-51
    assert oracle.decompress_frame(f, len(data)) == data
    p = oracle.Prefs(); p.content_checksum = 1
    g = bytearray(oracle.compress_frame(b"A" * 1000, p))
    g[-1] ^= 0xFF                                                            # src/test_lz4f.zig:168-179
    assert oracle.decompress_frame(bytes(g), 1000) == -118                   # ContentChecksumInvalid
    for bsid in (4, 5, 6, 7):                                                # src/test_lz4f.zig:216-255
        q = oracle.Prefs(); q.block_size_id = bsid
        assert oracle.decompress_frame(oracle.compress_frame(b"A" * 1000, q), 1000) == b"A" * 1000
    assert oracle.header_size(f) == 7
    assert oracle.xxh32(b"\x40\x40") >> 8 & 0xFF == 0xC0                     # SURVEY 8(c) header checksum pin


def test_xxh32_matches_python_xxhash(oracle):
    xxhash = pytest.importorskip("xxhash")
    for n in (0, 1, 3, 4, 15, 16, 17, 31, 32, 33, 1000, 65536):
        b = bytes(dg.random_bytes(n, n + 1))
        assert oracle.xxh32(b) == xxhash.xxh32(b, seed=0).intdigest()


def _liblz4():
    for name in ("liblz4.so.1", "/lib/x86_64-linux-gnu/liblz4.so.1", "/opt/conda/lib/liblz4.so.1"):
        try:
            L = C.CDLL(name)
            L.LZ4_decompress_safe.restype = C.c_int
            L.LZ4_decompress_safe.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
            return L
        except OSError:
            continue
    return None


def test_liblz4_decodes_every_oracle_stream(oracle):
    """Format validity: C liblz4 decodes every oracle block back to its input (it is NOT a compression oracle)."""
    L = _liblz4()
    if L is None:
        pytest.skip("liblz4 not installed")
    for name, b in cases.reference_test_inputs() + cases.seeded_cases():
        if not b:
            continue
        for c in (oracle.compress_default(b), oracle.compress_hc(b, 9), oracle.compress_fast(b, 17)):
            out = C.create_string_buffer(len(b) + 8)
            r = L.LZ4_decompress_safe(c, out, len(c), len(b))
            assert r == len(b) and out.raw[:r] == b, name


@pytest.mark.skipif(LZ4_CLI is None, reason="lz4 CLI not installed")
def test_cli_interop_oracle(oracle):
    """src/test_compat.zig groups 1-3 reproduced with the oracle."""
    six = [b"Hello World!", b"ABCDEFGH" * 125, cases.LOREM, bytes(dg.random_bytes(256, 12345)), b"",
           bytes(i % 256 for i in range(100000))]
    with tempfile.TemporaryDirectory() as td:
        a, o = os.path.join(td, "a"), os.path.join(td, "o")
        for d in six:
            open(a, "wb").write(oracle.compress_frame(d))
            subprocess.check_call([LZ4_CLI, "-d", "-f", "-q", a, o])
            assert open(o, "rb").read() == d
            open(a, "wb").write(d)
            subprocess.check_call([LZ4_CLI, "-f", "-q", a, o])
            assert oracle.decompress_frame(open(o, "rb").read(), len(d) + 8) == d
        for lvl in range(2, 13):                 # src/test_compat.zig:112-123 group 3: levels 2..12
            p = oracle.Prefs(); p.compression_level = lvl
            open(a, "wb").write(oracle.compress_frame(six[1], p))
            subprocess.check_call([LZ4_CLI, "-d", "-f", "-q", a, o])
            assert open(o, "rb").read() == six[1]


def test_golden_fixtures_still_match(oracle):
    """tests/golden/oracle_vectors.json was frozen from this oracle: any drift of oracle or generators shows here."""
    gold = {v["name"]: v for v in json.load(open(os.path.join(HERE, "golden", "oracle_vectors.json")))["vectors"]}
    n = 0
    for name, b in cases.reference_test_inputs() + list(cases.kat_inputs().items()) + cases.seeded_cases():
        v = gold[name]
        assert v["len"] == len(b) and v["in_sha256"] == hashlib.sha256(b).hexdigest(), name
        c = oracle.compress_default(b)
        assert (len(c), hashlib.sha256(c).hexdigest()) == (v["fast"]["len"], v["fast"]["sha256"]), name
        if len(b) <= 5000:
            h = oracle.compress_hc(b, 9)
            assert hashlib.sha256(h).hexdigest() == v["hc9"]["sha256"], name
        n += 1
    assert n >= 250


def test_acceleration_schedule_literal_vs_closed_form():
    """The kernel's closed-form probe schedule (zlz4_compress_fast.hip header) equals the literal loop of
    src/lz4.zig:321-338 for every acceleration class."""
    def literal(F0, a, L):
        step, nb, fwd, out = a, a, F0, []
        while True:
            ip = fwd; fwd += step; step = nb >> 6; nb += 1
            if fwd > L:
                return out
            if not out or out[-1] != ip:
                out.append(ip)

    def S(x):
        q, r = x >> 6, x & 63
        return 32 * q * (q - 1) + q * r

    def closed(F0, a, L):
        c, out, u = max(64, a), [], 0
        while True:
            if u == 0: pos, st = F0, a
            elif u == 1: pos, st = F0 + a, a >> 6
            else:
                x = c + u - 1; pos, st = F0 + a + S(x) - S(c), x >> 6
            if pos + st > L:
                return out
            out.append(pos); u += 1

    for a in (1, 2, 3, 7, 31, 62, 63, 64, 65, 66, 100, 128, 129, 1000, 65537):
        for F0 in (1, 5, 50):
            for L in (F0 + 1, F0 + 2, F0 + 3, F0 + a, F0 + a + 1, F0 + 70, F0 + 200, F0 + 5000, F0 + 100000):
                assert literal(F0, a, L) == closed(F0, a, L), (a, F0, L)
