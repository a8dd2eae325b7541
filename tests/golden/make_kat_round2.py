#!/usr/bin/env python3
"""Writes tests/golden/kat_round2.json: known-answer vectors for the parts of the path Appendix B does not reach
(level 2 `compressMID`, levels 10-12 `compressOptimal`, acceleration > 1, a block > 64 KiB, a checksummed frame).

Provenance -- stated in the fixture as well:
  * "hand": traced by hand from the Zig source; the trace is in the vector's `trace` field.
  * "two-restatements": produced by tools/pyref/zig_lz4_pyref.py (independent Python restatement written straight from
    the Zig) and required to equal the C oracle's output (oracle/lz4_oracle.c) -- this script refuses to write a vector
    on which the two disagree -- and, where the stream is decodable, to decode to its input with C liblz4.
The reference itself cannot run here (no zig toolchain), so none of this is reference OUTPUT: parity stays "unpinned".
Run from the repo root in the build container:  python tests/golden/make_kat_round2.py
"""
import ctypes
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools", "pyref"))
import datagen as dg  # noqa: E402
import zig_lz4_pyref as pr  # noqa: E402
from oracle import binding as o  # noqa: E402

_lz4 = ctypes.CDLL("liblz4.so.1")


def liblz4_decodes(comp, raw):
    out = ctypes.create_string_buffer(max(1, len(raw)))
    r = _lz4.LZ4_decompress_safe(comp, out, len(comp), len(raw))
    return r == len(raw) and out.raw[:len(raw)] == raw


INPUTS = {
    "A7": {"ascii": "XYZabcdWQRSTabcdUVWXYZ0123456789"},
    "M2": {"ascii": "abcdefghabcdefghabcdefgh0123"},
    "O1": {"concat": [{"ascii": "the quick brown fox jumps over the lazy dog. "}, {"ascii": "the quick brown cat jumps over the lazy dog! "},
                      {"ascii": "a quick brown fox naps under the lazy dog. "}, {"ascii": "0123456789"}]},
    "T70000": {"gen": ["text", 70000, 1]},
    "T140001": {"gen": ["text", 140001, 2]},
    "R3000": {"gen": ["reptext", 3000, 9]},
}


def build(spec):
    if "ascii" in spec:
        return spec["ascii"].encode()
    if "concat" in spec:
        return b"".join(build(s) for s in spec["concat"])
    if "gen" in spec:
        kind, n, seed = spec["gen"]
        return bytes(dg.GENERATORS[kind](n, seed))
    raise ValueError(spec)


HAND = {
    ("A7", "compressFast7"): (
        "f011" + b"XYZabcdWQRSTabcdUVWXYZ0123456789".hex(),
        "acceleration 7 (src/lz4.zig:321-338): the search probes position 1, then jumps by `step` = 7 to position 8, stalls "
        "there while step = searchMatchNb >> 6 is 0 (searchMatchNb 8..63: the re-probes see match == ip and fail "
        "`match < ip`, :346), then walks 9, 10, ... one by one.  Positions 2..7 are never probed, so 'abcd' at 3 is never "
        "put(); the probe at 12 ('abcd') reads an empty slot.  forwardIp passes mflimitPlusOne = 20 before anything "
        "matches -> finishCompression: one literal run of 32 = token F0, length byte 32 - 15 = 0x11, 32 bytes."),
    ("A7", "compressFast1"): (
        "c0" + b"XYZabcdWQRST".hex() + "0900" + "f001" + b"UVWXYZ0123456789".hex(),
        "acceleration 1: positions 1..11 are probed and put(); the probe at 12 finds 'abcd' put at 3 (:342-348), "
        "12 literals, offset 9, forward extension stops at once ('U' != 'W', :405-413) -> token C0; put(16) and the "
        "search restarts at 17 (:438-442), no further match before forwardIp > 20 -> last run of 16: F0 01."),
    ("M2", "compressHC2"): (
        "9a" + b"abcdefgha".hex() + "0800" + "50" + b"h0123".hex(),
        "compressMID (src/lz4hc.zig:733-934): ip 0..7 find empty tables and put index 0..7; at ip 8 both tables return 0 "
        "for 'abcdefg'/'abcd' -- index 0 is indistinguishable from empty (`pos8 > 0`, :744) -- so no match; at ip 9 the "
        "8-byte table returns 1, lz4Count(ip, match, matchlimit = 23) = 14 (:750), encodeSequence with 9 literals, "
        "offset 8, token 0x9A (14 - 4 = 10); ip = 23 > mflimit = 16 ends the loop; last literals 'h0123' -> 0x50."),
}

VECTORS = [
    ("A7", "compressFast7"), ("A7", "compressFast1"), ("M2", "compressHC2"), ("M2", "compressHC9"),
    ("O1", "compressHC2"), ("O1", "compressHC9"), ("O1", "compressHC10"), ("O1", "compressHC11"), ("O1", "compressHC12"),
    ("O1", "compressFast7"),
    ("R3000", "compressHC2"), ("R3000", "compressHC10"), ("R3000", "compressHC12"), ("R3000", "compressFast7"),
    ("T70000", "compressDefault"), ("T70000", "compressFast7"), ("T70000", "compressHC2"), ("T70000", "compressHC9"),
    ("T140001", "compressDefault"),
    ("O1", "compressFrameChecksums"), ("T70000", "compressFrameChecksums"),
]


def run(fn, data):
    if fn == "compressDefault":
        return pr.compress_fast(data, 1), o.compress_default(data)
    if fn.startswith("compressFast"):
        a = int(fn[len("compressFast"):])
        return pr.compress_fast(data, a), o.compress_fast(data, a)
    if fn.startswith("compressHC"):
        lvl = int(fn[len("compressHC"):])
        return pr.compress_hc(data, lvl), o.compress_hc(data, lvl)
    if fn == "compressFrameChecksums":
        # frame from the Python restatement's blocks (64 KiB blocks, block + content checksums, src/lz4f.zig:354-446)
        import xxhash
        p = o.Prefs(); p.block_size_id = 4; p.block_mode = 1; p.block_checksum = 1; p.content_checksum = 1
        hdr = bytes([0x04, 0x22, 0x4D, 0x18, 0x40 | 0x20 | 0x10 | 0x04, 0x40])
        hdr += bytes([(xxhash.xxh32(hdr[4:], seed=0).intdigest() >> 8) & 0xFF])
        out = bytearray(hdr)
        for i in range(0, len(data), 65536):
            raw = data[i:i + 65536]
            c = pr.compress_fast(raw, 1)
            stored = len(c) >= len(raw)
            body = raw if stored else c
            out += (len(body) | (0x80000000 if stored else 0)).to_bytes(4, "little") + body
            out += xxhash.xxh32(body, seed=0).intdigest().to_bytes(4, "little")
        out += b"\0\0\0\0" + xxhash.xxh32(data, seed=0).intdigest().to_bytes(4, "little")
        return bytes(out), o.compress_frame(data, p)
    raise ValueError(fn)


def main():
    vectors = []
    for name, fn in VECTORS:
        data = build(INPUTS[name])
        got_py, got_c = run(fn, data)
        assert got_py == got_c, "restatements disagree on %s %s" % (name, fn)
        v = {"input": name, "fn": fn, "len": len(got_c), "in_sha256": hashlib.sha256(data).hexdigest()}
        if (name, fn) in HAND:
            hexs, trace = HAND[(name, fn)]
            assert got_c.hex() == hexs, "hand trace disagrees with the restatements on %s %s: %s" % (name, fn, got_c.hex())
            v["provenance"] = "hand"; v["trace"] = trace
        else:
            v["provenance"] = "two-restatements"
        if len(got_c) <= 200:
            v["hex"] = got_c.hex()
        else:
            v["sha256"] = hashlib.sha256(got_c).hexdigest()
        if fn != "compressFrameChecksums":
            v["decodes_with_liblz4"] = bool(liblz4_decodes(got_c, data))
            if not fn.startswith("compressHC1"):
                assert v["decodes_with_liblz4"], (name, fn)
        else:
            assert o.decompress_frame(got_c, len(data)) == data
        vectors.append(v)
        print(name, fn, len(got_c), v["provenance"], v.get("decodes_with_liblz4"))
    doc = {"source": "tests/golden/make_kat_round2.py: 'hand' = traced by hand from the Zig source (trace in the vector); "
                     "'two-restatements' = tools/pyref/zig_lz4_pyref.py (independent Python restatement of src/lz4.zig:292-519, "
                     "src/lz4hc.zig:126-1489) and oracle/lz4_oracle.c agree byte for byte; decodable streams also decode with C "
                     "liblz4 1.9.3.  NOT reference output (no zig toolchain): parity unpinned.",
           "inputs": INPUTS, "vectors": vectors}
    with open(os.path.join(ROOT, "tests", "golden", "kat_round2.json"), "w") as f:
        json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
