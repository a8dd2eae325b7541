"""Synthetic block generators shared by tests/ and bench.py (the reference defines no corpus).

Counter-based splitmix64 (vectorised in numpy) so every byte is a pure function of
(seed, index): the same data is produced here, on the GPU box and in any later round.
Distributions follow SURVEY.md section 8(d): D-text (headline, LZ4 ratio ~2:1), D-reptext (short-range
repetitive text), D-ramp
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


# ---- word material shared by the two text distributions ---------------------------------------
_LETTERS = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)


def _word_matrix(nwords, tag):
    """[nwords, 12] uint8 words (space padded) and their lengths (1..11, English-like mean ~5)."""
    r = u64_stream(tag, nwords * 16).reshape(nwords, 16)
    # frequent (low-rank) words are short, rare ones long, as in natural language: 1..11 letters
    rank = np.arange(nwords, dtype=np.float64)
    lens = (1 + (np.log2(rank + 2.0) * 0.55).astype(np.int64) + (r[:, 0] % np.uint64(4)).astype(np.int64)).clip(1, 11)
    li = ((r[:, 1:13] % np.uint64(26)) * ((r[:, 1:13] >> np.uint64(8)) % np.uint64(26)) // np.uint64(26)).astype(np.int64)
    mat = np.full((nwords, 12), ord(" "), dtype=np.uint8)
    words = _LETTERS[li]
    for k in range(11):
        m = lens > k
        mat[m, k] = words[m, k]
    return mat, lens


def _zipf_cdf(n, shift, expo):
    p = 1.0 / (np.arange(n, dtype=np.float64) + shift) ** expo
    return np.cumsum(p / p.sum())


def _uniform01(r):
    return (r >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _words_to_bytes(mat, lens, wi, r):
    """Concatenate words wi (indices into mat) with separators; every ~16th separator is a newline."""
    wl = lens[wi] + 1
    ends = np.cumsum(wl)
    total = int(ends[-1])
    starts = ends - wl
    rep = np.repeat(np.arange(wi.shape[0], dtype=np.int64), wl)
    within = np.arange(total, dtype=np.int64) - np.repeat(starts, wl)
    buf = mat[wi[rep], np.minimum(within, 11)]
    sep = within == (wl[rep] - 1)
    buf[sep] = ord(" ")
    buf[sep & ((r[rep] & np.uint64(0xF)) == np.uint64(0))] = ord("\n")
    return buf


# ---- D-text (headline): natural-language-like statistics ----------------------------------------
# 16384-word vocabulary with a Zipf(1.0) law (most frequent word ~7 %), and a table of 4096 recurring
# phrases (2..7 words) that make up part of the stream -- the long-range redundancy (markup, idioms,
# boilerplate) that gives real text its ~2:1 LZ4 ratio on 64 KiB blocks with ~17-byte sequences.
_TV, _TP = 16384, 4096
_TXT = None


def _text_tables():
    mat, lens = _word_matrix(_TV, 0x7E87)
    wcdf = _zipf_cdf(_TV, 2.0, 1.0)
    pr = u64_stream(0x9A5E, _TP * 8).reshape(_TP, 8)
    plen = (2 + (pr[:, 0] % np.uint64(6))).astype(np.int64)                     # 2..7 words
    pw = np.searchsorted(wcdf, _uniform01(pr[:, 1:8])).clip(0, _TV - 1)         # [P, 7] word ids
    pcdf = _zipf_cdf(_TP, 4.0, 1.25)
    return mat, lens, wcdf, plen, pw, pcdf


def text_bytes(n, seed, phrase_prob=0.72):
    global _TXT
    if _TXT is None:
        _TXT = _text_tables()
    mat, lens, wcdf, plen, pw, pcdf = _TXT
    out = np.empty(n, dtype=np.uint8)
    # (counter-based stream: the bytes do not depend on the chunk size; small requests use small chunks)
    filled, ctr, chunk = 0, 0, min(1 << 18, max(256, n // 8 + 64))
    while filled < n:
        r = u64_stream(seed ^ 0x7E47, chunk * 2, start=ctr).reshape(chunk, 2)
        ctr += chunk * 2
        u0, u1 = _uniform01(r[:, 0]), _uniform01(r[:, 1])
        is_phrase = u0 < phrase_prob
        single = np.searchsorted(wcdf, u1).clip(0, _TV - 1)
        pid = np.searchsorted(pcdf, u1).clip(0, _TP - 1)
        cnt = np.where(is_phrase, plen[pid], 1)                                 # words contributed by each unit
        ends = np.cumsum(cnt)
        total = int(ends[-1])
        rep = np.repeat(np.arange(chunk, dtype=np.int64), cnt)
        within = np.arange(total, dtype=np.int64) - np.repeat(ends - cnt, cnt)
        wi = np.where(is_phrase[rep], pw[pid[rep], np.minimum(within, 6)], single[rep])
        buf = _words_to_bytes(mat, lens, wi, splitmix64(r[rep, 0] + within.astype(np.uint64)))
        take = min(buf.shape[0], n - filled)
        out[filled:filled + take] = buf[:take]
        filled += take
    return out


# ---- D-reptext: short-range repetitive text (round-1's first generator) --------------------------
# 2048 words, Zipf(1.5): the top word has probability ~0.38, so 4-grams recur every few bytes, LZ4
# sequences average ~9 bytes and hash chains are saturated.  Kept as a stress distribution.
_RV = 2048
_REP = None


def reptext_bytes(n, seed):
    global _REP
    if _REP is None:
        r = u64_stream(0xD1C7, _RV * 16).reshape(_RV, 16)
        lens = (2 + (r[:, 0] % np.uint64(9))).astype(np.int64)
        li = ((r[:, 1:13] % np.uint64(26)) * ((r[:, 1:13] >> np.uint64(8)) % np.uint64(26)) // np.uint64(26)).astype(np.int64)
        words = _LETTERS[li]
        mat = np.full((_RV, 12), ord(" "), dtype=np.uint8)
        for k in range(10):
            m = lens > k
            mat[m, k] = words[m, k]
        _REP = (mat, lens, _zipf_cdf(_RV, 1.0, 1.5))
    mat, lens, cdf = _REP
    out = np.empty(n, dtype=np.uint8)
    filled, ctr, chunk = 0, 0, min(1 << 20, max(256, n // 4 + 64))
    while filled < n:
        r = u64_stream(seed ^ 0x7E47, chunk, start=ctr)
        ctr += chunk
        wi = np.searchsorted(cdf, _uniform01(r)).clip(0, _RV - 1)
        buf = _words_to_bytes(mat, lens, wi, r)
        take = min(buf.shape[0], n - filled)
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
    "reptext": reptext_bytes,
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
