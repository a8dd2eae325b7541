"""Cross-check of the C oracle (oracle/lz4_oracle.c) against an INDEPENDENT Python restatement written straight
from the Zig source (tools/pyref/zig_lz4_pyref.py).  CPU only, build container and GPU box alike (neither needs the
reference at run time).  Two restatements agreeing is not the reference agreeing -- the status stays "parity
unpinned"  -- but a transcription slip in one of them would have to be repeated in the other."""
import os
import sys

import numpy as np
import pytest

import cases
import datagen as dg

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "pyref"))
import zig_lz4_pyref as pr  # noqa: E402


_SEEDED = None


def _corpus(max_len):
    global _SEEDED
    if _SEEDED is None:          # (the text generators build their tables per call: generate the cases once)
        _SEEDED = cases.seeded_cases(max_size=4097)
    out = [(n, b) for n, b in cases.reference_test_inputs() if len(b) <= max_len]
    out += [(n, b) for n, b in cases.kat_inputs().items() if len(b) <= max_len]
    out += [(n, b) for n, b in _SEEDED if len(b) <= max_len]
    rng = np.random.default_rng(2024)
    for i in range(40):      # copies of earlier slices at random distances, runs, noise (the fuzz tool's recipe, small)
        n = int(rng.integers(20, min(max_len, 6000)))
        buf = bytearray(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
        for _ in range(int(rng.integers(1, 12))):
            a = int(rng.integers(0, n)); ln = int(rng.integers(4, 200)); d = int(rng.integers(1, max(2, a + 1)))
            for k in range(a, min(n, a + ln)):
                if k - d >= 0:
                    buf[k] = buf[k - d]
        if i % 5 == 0:
            a = int(rng.integers(0, n)); buf[a:a + int(rng.integers(4, 300))] = bytes([int(rng.integers(0, 256))]) * min(n - a, int(rng.integers(4, 300)))
        out.append(("fuzz%d" % i, bytes(buf[:n])))
    return out


@pytest.mark.parametrize("accel", [1, 7, 65])
def test_fast_matches_python_restatement(oracle, accel):
    bad = [n for n, b in _corpus(45000) if oracle.compress_fast(b, accel) != pr.compress_fast(b, accel)]
    assert not bad, bad[:8]


def test_fast_larger_than_64k(oracle):
    for seed, n in ((1, 70000),):
        b = bytes(dg.text_bytes(n, seed))
        assert oracle.compress_default(b) == pr.compress_fast(b, 1)


@pytest.mark.parametrize("level", [2, 3, 6, 9])
def test_hc_matches_python_restatement(oracle, level):
    bad = [n for n, b in _corpus(10000) if oracle.compress_hc(b, level) != pr.compress_hc(b, level)]
    assert not bad, bad[:8]


@pytest.mark.parametrize("level", [10, 11, 12])
def test_optimal_matches_python_restatement(oracle, level):
    bad = []
    for n, b in _corpus(5000):
        want = oracle.compress_hc(b, level)
        try:
            got = pr.compress_hc(b, level)
        except (OverflowError, ValueError, IndexError):
            got = -1          # the reference's `ip - anchor` / `iend - anchor` underflow: the oracle reports OutputTooSmall
        if got != want:
            bad.append(n)
    assert not bad, bad[:8]
