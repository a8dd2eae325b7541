"""Synthetic block generators shared by tests/ and bench.py (the reference defines no corpus).

Counter-based splitmix64 (vectorised in numpy) so every byte is a pure function of
(seed, index): the same data is produced here, on the GPU box and in any later round.
Distributions follow SURVEY.md section 8(d): D-text (headline, LZ4 ratio ~2:1), D-ramp
(i % 256, reference src/test.zig:244-246), D-mixed (half constant / half random,
reference src/test_lz4hc.zig:203-206), D-random, D-zero.
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(x):
    """Vectorised splitmix64 finaliser over a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = (x + GOLDEN).astype(np.uint64)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def u64_stream(seed, n, start=0):
    """n pseudo-random uint64, element i = splitmix64(seed*GOLDEN + start + i)."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * GOLDEN
        idx = np.arange(start, start + n, dtype=np.uint64)
        return splitmix64(base + idx)


def random_bytes(n, seed):
    w = u64_stream(seed, (n + 7) // 8)
    return w.view(np.uint8)[:n].copy()


# ---- D-text: Zipf-distributed words from a seeded vocabulary ---------------------------------
_VOCAB = 2048
_LETTERS = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)


def _vocabulary():
    r = u64_stream(0xD1C7, _VOCAB * 16).reshape(_VOCAB, 16)
    lens = (2 + (r[:, 0] % np.uint64(9))).astype(np.int64)              # 2..10 letters
    # letter choice skewed towards the front of _LETTERS (product of two uniforms)
    li = ((r[:, 1:13] % np.uint64(26)) * ((r[:, 1:13] >> np.uint64(8)) % np.uint64(26)) // np.uint64(26)).astype(np.int64)
    words = _LETTERS[li]                                                  # [V, 12]
    mat = np.full((_VOCAB, 12), ord(" "), dtype=np.uint8)
    for k in range(10):
        m = lens > k
        mat[m, k] = words[m, k]
    # word k occupies mat[k, :lens[k]] followed by one separator
    ranks = np.arange(_VOCAB, dtype=np.float64)
    p = 1.0 / (ranks + 1.0) ** 1.5      # tuned so compressDefault gives ~2:1 on 64 KiB blocks
    cdf = np.cumsum(p / p.sum())
    return mat, lens, cdf


_VOC = None


def text_bytes(n, seed):
    global _VOC
    if _VOC is None:
        _VOC = _vocabulary()
    mat, lens, cdf = _VOC
    out = np.empty(n, dtype=np.uint8)
    filled = 0
    chunk_words = 1 << 20
    ctr = 0
    while filled < n:
        r = u64_stream(seed ^ 0x7E47, chunk_words, start=ctr)
        ctr += chunk_words
        u = (r >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
        wi = np.searchsorted(cdf, u).clip(0, _VOCAB - 1)
        wl = lens[wi] + 1                                               # + separator
        # every ~12th separator becomes ". " / ",\n" style punctuation for a little variety
        ends = np.cumsum(wl)
        total = int(ends[-1])
        starts = ends - wl
        rep = np.repeat(np.arange(chunk_words, dtype=np.int64), wl)
        within = np.arange(total, dtype=np.int64) - np.repeat(starts, wl)
        buf = mat[wi[rep], np.minimum(within, 11)]
        sep = within == (wl[rep] - 1)
        buf[sep] = ord(" ")
        punct = sep & ((r[rep] & np.uint64(0xF)) == np.uint64(0))
        buf[punct] = ord("\n")
        take = min(total, n - filled)
        out[filled:filled + take] = buf[:take]
        filled += take
    return out


def ramp_bytes(n, seed=0):
    return (np.arange(n, dtype=np.int64) % 256).astype(np.uint8)


def mixed_bytes(n, seed):
    out = random_bytes(n, seed)
    out[: n // 2] = ord("X")
    return out


def zero_bytes(n, seed=0):
    return np.zeros(n, dtype=np.uint8)


GENERATORS = {
    "text": text_bytes,
    "ramp": ramp_bytes,
    "mixed": mixed_bytes,
    "random": random_bytes,
    "zero": zero_bytes,
}


def make_blocks(dist, nblocks, block_size, seed=1):
    """[nblocks, block_size] uint8; every block is different (except ramp/zero, which are constant by definition)."""
    if dist == "mixed":     # half constant / half random inside EVERY block
        data = random_bytes(nblocks * block_size, seed).reshape(nblocks, block_size)
        data[:, : block_size // 2] = ord("X")
        return data
    data = GENERATORS[dist](nblocks * block_size, seed)
    return data.reshape(nblocks, block_size)
