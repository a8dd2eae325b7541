#!/usr/bin/env python3
"""Randomised parity sweep, GPU vs oracle (beyond the fixed cases of tests/): inputs are concatenations of random
segments -- text, repeated text, runs, ramps, noise, copies of earlier slices at random distances and lengths (which
force long matches, long literal runs and every length-extension path) -- compressed with compressDefault /
compressFast(accel) / compressHC(level) and decoded again; every byte and status is compared with oracle/.
Two steps, so that the slow oracle runs where no GPU time is billed:
  python tools/fuzz_parity.py --prepare ROUNDS SEED tests/_fuzz_cache/f.npz     (CPU)
  python tools/fuzz_parity.py --check tests/_fuzz_cache/f.npz                   (GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import datagen as dg
from oracle import binding as oracle


_POOLS = {}


def _pool(kind):
    if kind not in _POOLS:
        gen = {"text": dg.text_bytes, "rep": dg.reptext_bytes, "rand": dg.random_bytes}[kind]
        _POOLS[kind] = bytes(gen(1 << 20, 4242))
    return _POOLS[kind]


def make_input(rng, n):
    out = bytearray()
    while len(out) < n:
        kind = int(rng.integers(0, 8))
        ln = int(rng.choice([1, 2, 3, 4, 5, 8, 13, 15, 16, 17, 19, 31, 47, 48, 64, 100, 255, 270, 300, 1000, 5000]))
        if kind in (0, 1, 4):
            pool = _pool({0: "text", 1: "rand", 4: "rep"}[kind])
            at = int(rng.integers(0, len(pool) - ln))
            seg = pool[at:at + ln]
        elif kind == 2:
            seg = bytes([int(rng.integers(0, 256))]) * ln
        elif kind == 3:
            step = int(rng.integers(1, 7))
            seg = bytes((i * step) & 255 for i in range(ln))
        elif len(out) >= 4:                                    # copy of an earlier slice (kinds 5..7), LZ77 style
            dist = int(rng.integers(1, min(len(out), 70000) + 1))
            start = len(out) - dist
            seg = bytes(out[start:start + ln]) if dist >= ln else (bytes(out[start:]) * (ln // dist + 1))[:ln]
        else:
            seg = b"abcd"
        out += seg
    return bytes(out[:n])


def prepare(rounds, seed, path):
    """CPU side (no GPU needed): inputs and the oracle's outputs, pickled-free (one .npz of byte arrays)."""
    rng = np.random.default_rng(seed)
    store = {}
    for r in range(rounds):
        sizes = [int(rng.choice([13, 64, 200, 1000, 4096, 20000, 65536, 65536, 65547, 100000])) for _ in range(40)]
        items = [make_input(rng, n) for n in sizes]
        accel = int(rng.choice([1, 1, 1, 2, 7, 64, 65, 1000]))
        level = int(rng.choice([3, 6, 9, 9, 2, 4, 10, 12]))
        store["r%d_meta" % r] = np.array([len(items), accel, level], dtype=np.int64)
        ub = 0
        for i, b in enumerate(items):
            store["r%d_in%d" % (r, i)] = np.frombuffer(b, dtype=np.uint8)
            store["r%d_f%d" % (r, i)] = np.frombuffer(oracle.compress_fast(b, accel), dtype=np.uint8)
            oracle.hc_reference_ub()
            h = oracle.compress_hc(b, level)
            if isinstance(h, int):      # the lz4opt levels can run into the reference's defect #1: error code, no bytes
                store["r%d_herr%d" % (r, i)] = np.array([h], dtype=np.int64)
                h = b""
            store["r%d_h%d" % (r, i)] = np.frombuffer(h, dtype=np.uint8)
            ub += 1 if oracle.hc_reference_ub() else 0
        print("prepared round %d (accel %d, level %d; %d inputs on which the reference's HC path has no defined output, "
              "see oracle/lz4_oracle.c)" % (r, accel, level, ub), flush=True)
    store["rounds"] = np.array([rounds], dtype=np.int64)
    np.savez_compressed(path, **store)


def check(path):
    import torch
    import gpu_harness as gh
    import zig_lz4_amd as zl
    dev = torch.device("cuda:0")
    z = np.load(path)
    rng = np.random.default_rng(7)
    total = 0
    for r in range(int(z["rounds"][0])):
        n_items, accel, level = (int(v) for v in z["r%d_meta" % r])
        items = [z["r%d_in%d" % (r, i)].tobytes() for i in range(n_items)]
        want_f = [z["r%d_f%d" % (r, i)].tobytes() for i in range(n_items)]
        want_h = [z["r%d_h%d" % (r, i)].tobytes() for i in range(n_items)]
        got_f = gh.compress_fast(zl, items, dev, accel=accel)
        for i, ((n, data), w) in enumerate(zip(got_f, want_f)):
            assert n == len(w) and data == w, "round %d fast accel %d item %d (size %d): GPU %d vs oracle %d bytes" % (r, accel, i, len(items[i]), n, len(w))
        got_h = gh.compress_hc(zl, items, dev, level)
        for i, ((n, data), w) in enumerate(zip(got_h, want_h)):
            key = "r%d_herr%d" % (r, i)
            if key in z.files:
                assert n == int(z[key][0]), "round %d hc level %d item %d: GPU status %d vs oracle %d" % (r, level, i, n, int(z[key][0]))
                continue
            assert n == len(w) and data == w, "round %d hc level %d item %d: GPU %d vs oracle %d bytes" % (r, level, i, n, len(w))
        hs = want_h if level < 10 else []      # the reference's lz4opt levels may emit undecodable streams
        got_d = gh.decompress(zl, want_f + hs, [len(b) for b in items] * (2 if hs else 1), dev)
        for i, ((n, data), b) in enumerate(zip(got_d, items + (items if hs else []))):
            assert n == len(b) and data == b, "round %d decode item %d: %d vs %d" % (r, i, n, len(b))
        # short capacities: status parity with the oracle (cheap on the CPU)
        caps2 = [max(0, len(b) - int(rng.integers(1, 40))) for b in items]
        want_d2 = [oracle.decompress_safe(c, cap) for c, cap in zip(want_f, caps2)]
        got_d2 = gh.decompress(zl, want_f, caps2, dev)
        for i, ((n, data), w) in enumerate(zip(got_d2, want_d2)):
            if isinstance(w, int):
                assert n == w, "round %d short-cap decode item %d: status %d vs %d" % (r, i, n, w)
            else:
                assert n == len(w) and data == w
        total += n_items
        print("round %d ok: %d inputs, accel %d, hc level %d" % (r, n_items, accel, level), flush=True)
    print("fuzz parity ok: %d inputs" % total)


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--prepare":
        prepare(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    elif len(sys.argv) >= 3 and sys.argv[1] == "--check":
        check(sys.argv[2])
    else:
        print(__doc__)


if __name__ == "__main__":
    main()
