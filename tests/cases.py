"""Input cases used by both the oracle tests and the GPU parity tests.

The named cases reproduce the inputs of the reference's own tests (cited per case);
the seeded ones add sizes/distributions around every boundary of the algorithm.
"""
import numpy as np

import datagen as dg

LOREM = (b"Lorem ipsum dolor sit amet, consectetur adipiscing elit.\n"
         b"Sed do eiusmod tempor incididunt ut labore et dolore magna aliqua.\n"
         b"Ut enim ad minim veniam, quis nostrud exercitation ullamco laboris.")   # src/test_compat.zig:12-16


def reference_test_inputs():
    """(name, bytes) for every input the reference's tests feed the hot path."""
    c = []
    c.append(("test.zig:19 AAAA", b"AAAA"))
    c.append(("test.zig:58 sentence", b"Hello, World! This is a test of LZ4 compression in Zig. Hello!"))
    c.append(("test.zig:127 160xA", b"A" * 160))
    c.append(("test.zig:182 empty", b""))
    c.append(("test.zig:208 ABC", b"ABC"))
    c.append(("test.zig:239 ramp10000", bytes(i % 256 for i in range(10000))))
    c.append(("test_compat.zig small", b"Hello World!"))
    c.append(("test_compat.zig repeated", b"ABCDEFGH" * 125))
    c.append(("test_compat.zig text", LOREM))
    c.append(("test_compat.zig random256", bytes(dg.random_bytes(256, 12345))))
    c.append(("test_compat.zig large", bytes(i % 256 for i in range(100000))))
    c.append(("test_lz4hc.zig ABCDx1000", b"ABCD" * 1000))
    c.append(("test_lz4hc.zig ABCDx500", b"ABCD" * 500))
    c.append(("test_lz4hc.zig random1000", bytes(dg.random_bytes(1000, 12345))))
    for n in (1, 10, 100, 1000, 10000):                                # src/test_lz4hc.zig:191-227
        half = n // 2
        c.append(("test_lz4hc.zig halfX %d" % n, b"X" * half + bytes(dg.random_bytes(n - half, 54321))))
    c.append(("test_lz4hc.zig TestDatax250", b"TestData" * 250))
    c.append(("test_lz4hc.zig pat1", b"A" * 1000))
    c.append(("test_lz4hc.zig pat2", b"AB" * 1000))
    c.append(("test_lz4hc.zig pat4", b"ABCD" * 1000))
    c.append(("test_lz4f.zig 1MiB (i/16)%256", bytes((i // 16) % 256 for i in range(1 << 20))))
    c.append(("test_lz4f.zig Ax1000", b"A" * 1000))
    return c


def kat_inputs():
    """Inputs of SURVEY.md Appendix B."""
    u = bytes((37 * i + 11) % 251 for i in range(256))
    return {
        "P1": b"A" * 600 + b"B" + b"A" * 400 + b"0123456789ab",
        "S1": u[0:101] + u[10:30] + bytes(range(0xF0, 0xFD)),
    }


def boundary_sizes():
    return [0, 1, 4, 11, 12, 13, 14, 15, 16, 17, 20, 27, 28, 29, 31, 32, 33, 63, 64, 65, 66, 67, 68, 79, 80, 127, 128,
            129, 255, 256, 257, 269, 270, 271, 1023, 1024, 1025, 4095, 4096, 4097, 65535, 65536]


def seeded_cases(max_size=65536, heavy=False):
    """(name, bytes): every distribution x boundary sizes, plus structured adversarial inputs."""
    out = []
    for dist in dg.GENERATORS:
        for n in boundary_sizes():
            if n <= max_size:
                out.append(("%s/%d" % (dist, n), bytes(dg.GENERATORS[dist](n, 7 + n))))
    # periodic data with every small period (overlap copies / duplicate-hash groups)
    for period in (1, 2, 3, 4, 5, 7, 8, 13, 16, 31, 63, 64, 65, 100, 255, 256, 1000):
        unit = bytes(dg.random_bytes(period, 900 + period))
        out.append(("period%d" % period, (unit * (5000 // period + 2))[:5000]))
    # long literal runs followed by one far match (exercises the skip schedule at large strides)
    r = bytes(dg.random_bytes(40000, 77))
    out.append(("random+copy", r + r[100:300] + bytes(dg.random_bytes(50, 78))))
    out.append(("random+copy-far", r + bytes(dg.random_bytes(25000, 79)) + r[5:4000]))
    # text with injected long repeats
    t = bytes(dg.text_bytes(30000, 5))
    out.append(("text+self", t + t[1000:9000] + t[:123]))
    # runs of different bytes (RLE) with literal islands
    rle = b"".join(bytes([65 + (i % 7)]) * (3 + 37 * i % 900) + bytes(dg.random_bytes(i % 11, i)) for i in range(60))
    out.append(("rle-islands", rle))
    # match that runs to the very end of the block / last-literals boundary
    for tail in range(0, 20):
        out.append(("zeros+tail%d" % tail, b"\0" * 300 + bytes(dg.random_bytes(tail, 3))))
    if heavy:
        for dist in ("text", "mixed", "random"):
            out.append(("%s/262144" % dist, bytes(dg.GENERATORS[dist](262144, 11))))
        out.append(("text/1MiB+1", bytes(dg.text_bytes((1 << 20) + 1, 12))))
        out.append(("zero/300000", b"\0" * 300000))
        out.append(("period70000", (bytes(dg.random_bytes(70000, 4)) * 3)[:200001]))
    return out
