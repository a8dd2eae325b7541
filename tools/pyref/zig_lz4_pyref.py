"""Independent Python restatement of jedisct1/zig-lz4's block compressors -- BUILD-CONTAINER-ONLY cross-check.

Written straight from the Zig source (file:line cited per function, relative to /root/reference/), without looking at
oracle/lz4_oracle.c: a second, slow, obviously-literal statement of the same algorithms, used by
tests/test_pyref_crosscheck.py to cross-check the C oracle on a corpus of small inputs.  It is test infrastructure like
the oracle (never imported by the product, the bench or the GPU tests) and pins nothing by itself: the reference
cannot be run here, so two restatements that agree are still "parity unpinned" -- they only make a transcription slip
in either of them unlikely to survive.

Conventions: `dst` is unbounded (compressBound-sized in practice), so the OutputTooSmall paths are not restated;
`opt[]` of the optimal parser starts zeroed and keeps stale entries between rounds (the reference leaves it undefined,
src/lz4hc.zig:1079).
"""
M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF
MINMATCH, LASTLITERALS, MFLIMIT = 4, 5, 12            # src/lz4.zig:12-15
ML_BITS, ML_MASK, RUN_MASK = 4, 15, 15                # src/lz4.zig:17-21
DIST_MAX = 65535                                      # src/lz4.zig:24-25
HASH_MUL = 2654435761                                 # src/lz4.zig:44
HASH_MUL_64 = 58295818150454627                       # src/lz4hc.zig:50
OPT_NUM = 4096                                        # src/lz4hc.zig:42


def rd32(b, i):
    return int.from_bytes(b[i:i + 4], "little")


def rd64(b, i):
    return int.from_bytes(b[i:i + 8], "little")


def _put_len(out, n):                                 # the 255-run after a saturated nibble (value n = len - 15)
    while n >= 255:
        out.append(255)
        n -= 255
    out.append(n)


def _last_literals(out, src, anchor):                 # finishCompression src/lz4.zig:484-519 / lz4hc.zig:1035-1061
    lit = len(src) - anchor
    if lit == 0:
        return
    if lit >= RUN_MASK:
        out.append(RUN_MASK << ML_BITS)
        _put_len(out, lit - RUN_MASK)
    else:
        out.append(lit << ML_BITS)
    out += src[anchor:]


# ------------------------------------------------------------------------------------------------ lz4.zig
def compress_fast(src, acceleration=1):
    """src/lz4.zig:292-447."""
    src = bytes(src)
    n = len(src)
    out = bytearray()
    if n == 0:
        return bytes(out)                             # :299
    if n < MFLIMIT + 1:                               # :302-304 compressAsLiterals :449-482
        if n >= RUN_MASK:
            out.append(RUN_MASK << ML_BITS)
            _put_len(out, n - RUN_MASK)
        else:
            out.append(n << ML_BITS)
        out += src
        return bytes(out)
    table = [0] * 4096                                # :307
    ip, anchor = 0, 0
    mflimit_plus_one = n - MFLIMIT                    # :313
    match_limit = n - LASTLITERALS                    # :314
    ip += 1                                           # :317
    while ip < mflimit_plus_one:                      # :320
        accel = min(max(acceleration, 1), 65537)      # :321
        step = accel
        search_nb = accel
        forward = ip
        while True:                                   # :329
            ip = forward
            forward += step
            step = search_nb >> 6
            search_nb += 1
            if forward > mflimit_plus_one:            # :335
                _last_literals(out, src, anchor)
                return bytes(out)
            seq = rd32(src, ip)
            h = ((seq * HASH_MUL) & M32) >> 20        # :75-77
            match = table[h]
            valid = match > 0 and match < ip and match + DIST_MAX >= ip and rd32(src, match) == seq   # :345-348
            table[h] = ip                             # :350
            if valid:
                break
        lit = ip - anchor                             # :360
        token_pos = len(out)
        out.append(0)
        if lit >= RUN_MASK:                           # :368-385
            out[token_pos] = RUN_MASK << ML_BITS
            _put_len(out, lit - RUN_MASK)
        else:
            out[token_pos] = lit << ML_BITS
        out += src[anchor:ip]                         # :390
        out += (ip - match).to_bytes(2, "little")     # :395-397
        ip += MINMATCH
        match += MINMATCH
        ml = 0
        while ip < match_limit and src[ip] == src[match]:   # :405-413
            ip += 1
            match += 1
            ml += 1
        if ml >= ML_MASK:                             # :416-432
            out[token_pos] |= ML_MASK
            _put_len(out, ml - ML_MASK)
        else:
            out[token_pos] |= ml
        anchor = ip                                   # :435
        if ip < mflimit_plus_one:                     # :438-442
            table[((rd32(src, ip) * HASH_MUL) & M32) >> 20] = ip
            ip += 1
    _last_literals(out, src, anchor)                  # :446
    return bytes(out)


# ------------------------------------------------------------------------------------------------ lz4hc.zig
def _hash_hc(seq):                                    # :129-131
    return ((seq * HASH_MUL) & M32) >> (32 - 15)


def _hash_mid4(seq):                                  # :139-141
    return ((seq * HASH_MUL) & M32) >> (32 - 14)


def _hash_mid8(src, i):                               # :149-157 (lower 56 bits of an 8-byte read)
    masked = (rd64(src, i) << 8) & M64
    return ((masked * HASH_MUL_64) & M64) >> (64 - 14)


def _count(src, a, b, limit):                         # lz4Count :234-264: common bytes of a.. and b.., a < limit
    c = 0
    while a < limit and src[a] == src[b]:
        a += 1
        b += 1
        c += 1
    return c


def _count_pattern(src, ip, iend, pattern32):         # :170-199
    start = ip
    p64 = pattern32 | (pattern32 << 32)
    while ip + 7 < iend:
        diff = rd64(src, ip) ^ p64
        if diff == 0:
            ip += 8
        else:
            return ip + ((diff & -diff).bit_length() - 1) // 8 - start
    pb = pattern32
    while ip < iend:
        if src[ip] != (pb & 0xFF):
            break
        ip += 1
        pb >>= 8
        if pb == 0:
            pb = pattern32
    return ip - start


def _reverse_count_pattern(src, ip, ilow, pattern):   # :202-222
    start = ip
    while ip >= ilow + 4:
        if rd32(src, ip - 4) != pattern:
            break
        ip -= 4
    pbytes = pattern.to_bytes(4, "little")
    idx = 3
    while ip > ilow:
        if src[ip - 1] != pbytes[idx]:
            break
        ip -= 1
        idx = 3 if idx == 0 else idx - 1
    return start - ip


class _Ctx:                                           # Context :391-419 (one-shot use: prefixStart = src, all limits 0)
    def __init__(self):
        self.hash = [0] * 32768
        self.chain = [0] * 65536
        self.next_to_update = 0


def _insert_hc(ctx, src, target):                     # :491-510
    idx = ctx.next_to_update
    while idx < target:
        h = _hash_hc(rd32(src, idx))
        prev = ctx.hash[h]
        delta = DIST_MAX + 1 if prev > idx else idx - prev
        ctx.chain[idx & 0xFFFF] = DIST_MAX if delta > DIST_MAX else delta
        ctx.hash[h] = idx
        idx += 1
    ctx.next_to_update = target


def _wider_match(ctx, src, ip, ihigh, longest, max_attempts, pattern_analysis):
    """insertAndGetWiderMatch :538-681 with iLowLimit == ip (no backward extension) -> (len, off)."""
    ip_index = ip
    lowest = 0 if (0 + DIST_MAX + 1 > ip_index) else ip_index - DIST_MAX     # :553-554
    nb = max_attempts
    pattern = rd32(src, ip)
    r_len, r_off = longest, 0
    mi = ctx.hash[_hash_hc(pattern)]                  # :563
    if mi == 0:
        return r_len, r_off                           # :566-568
    while mi > 0 and nb > 0:                          # :571
        if mi > ip_index or ip_index - mi > DIST_MAX:
            break                                     # :573
        nb -= 1
        if mi >= lowest:                              # :579
            if rd32(src, mi) == pattern:              # :586
                mlt = MINMATCH + _count(src, ip + MINMATCH, mi + MINMATCH, ihigh)
                if mlt > r_len:                       # :607 (back == 0)
                    r_len, r_off = mlt, ip_index - mi
                    if mlt > max_attempts:
                        break                         # :613
        delta = ctx.chain[mi & 0xFFFF]                # :619
        if delta == 0 or delta > mi:
            break
        mi -= delta
    if pattern_analysis and r_len > 0:                # :626
        delta = ctx.chain[mi & 0xFFFF]
        if delta == 1:
            if (pattern & 0xFFFF) == (pattern >> 16) and (pattern & 0xFF) == (pattern >> 24):   # :225-228
                src_pat = _count_pattern(src, ip + 4, ihigh, pattern) + 4
                cand = mi - 1                         # :636 (mi >= 1 here for inputs <= 64 KiB: chain[0] == 0)
                if cand >= lowest and cand >= 0:
                    if rd32(src, cand) == pattern:
                        fwd = _count_pattern(src, cand + 4, ihigh, pattern) + 4
                        back = _reverse_count_pattern(src, cand, 0, pattern)
                        limited_back = cand - max(cand - back, lowest)
                        seg = limited_back + fwd
                        max_ml = min(seg, src_pat)
                        if seg >= src_pat and fwd <= src_pat:
                            new_mi = cand + fwd - src_pat
                        else:
                            new_mi = cand - limited_back
                        if max_ml > r_len and ip_index - new_mi <= DIST_MAX:
                            r_len, r_off = max_ml, ip_index - new_mi
    return r_len, r_off


def _encode_sequence(out, src, ip, anchor, mlen, off):   # :308-386 -> new ip (= new anchor)
    lit = ip - anchor
    token_pos = len(out)
    out.append(0)
    if lit >= RUN_MASK:
        out[token_pos] = RUN_MASK << ML_BITS
        _put_len(out, lit - RUN_MASK)
    else:
        out[token_pos] = lit << ML_BITS
    out += src[anchor:ip]
    out += off.to_bytes(2, "little")
    code = mlen - MINMATCH
    if code >= ML_MASK:
        out[token_pos] += ML_MASK
        rem = code - ML_MASK
        while rem >= 510:
            out += b"\xff\xff"
            rem -= 510
        if rem >= 255:
            out.append(255)
            rem -= 255
        out.append(rem)
    else:
        out[token_pos] += code
    return ip + mlen


def _encode_literals(src):                            # :1394-1425
    out = bytearray()
    n = len(src)
    if n >= RUN_MASK:
        out.append(RUN_MASK << ML_BITS)
        _put_len(out, n - RUN_MASK)
    else:
        out.append(n << ML_BITS)
    out += src
    return bytes(out)


def _hash_chain(src, max_attempts):                   # compressHashChain :976-1064
    n = len(src)
    if n < MFLIMIT + 1:
        return _encode_literals(src)
    ctx = _Ctx()
    out = bytearray()
    ip = anchor = 0
    mflimit, matchlimit = n - MFLIMIT, n - LASTLITERALS
    pattern_analysis = max_attempts > 128             # :983
    while ip <= mflimit:                              # :1009
        _insert_hc(ctx, src, ip)                      # insertAndFindBestMatch :514-535
        mlen, off = _wider_match(ctx, src, ip, matchlimit, MINMATCH - 1, max_attempts, pattern_analysis)
        if mlen < MINMATCH or off == 0:
            ip += 1
            continue
        ip = anchor = _encode_sequence(out, src, ip, anchor, mlen, off)
    _last_literals(out, src, anchor)
    return bytes(out)


def _mid(src):                                        # compressMID :687-971
    n = len(src)
    if n < MFLIMIT + 1:
        return _encode_literals(src)
    out = bytearray()
    ip = anchor = 0
    mflimit, matchlimit, ilimit = n - MFLIMIT, n - LASTLITERALS, n - 8
    h4t, h8t = [0] * 16384, [0] * 16384

    def fill_begin(ip, final_idx):                    # :763-771 / :871-879 (ip may differ from final_idx by one!)
        if ip + 1 <= ilimit:
            h8t[_hash_mid8(src, ip + 1)] = final_idx + 1
        if ip + 2 <= ilimit:
            h8t[_hash_mid8(src, ip + 2)] = final_idx + 2
        if ip + 1 <= ilimit:
            h4t[_hash_mid4(rd32(src, ip + 1))] = final_idx + 1

    def fill_end(e):                                  # :786-813 / :897-924 (e = ip after the match)
        if e - 2 < ilimit:
            if e > 5 and e - 5 <= ilimit:
                h8t[_hash_mid8(src, e - 5)] = e - 5
            if e - 3 <= ilimit:
                h8t[_hash_mid8(src, e - 3)] = e - 3
            if e - 2 <= ilimit:
                h8t[_hash_mid8(src, e - 2)] = e - 2
                h4t[_hash_mid4(rd32(src, e - 2))] = e - 2
            if e - 1 <= ilimit:
                h4t[_hash_mid4(rd32(src, e - 1))] = e - 1

    while ip <= mflimit:                              # :733
        ip_index = ip
        h8 = _hash_mid8(src, ip)                      # long match :739-817
        pos8 = h8t[h8]
        h8t[h8] = ip_index
        if pos8 > 0 and ip_index - pos8 <= DIST_MAX and pos8 < ip:
            mlt = _count(src, ip, pos8, matchlimit)
            if mlt >= MINMATCH:
                fill_begin(ip, ip_index)
                ip = anchor = _encode_sequence(out, src, ip, anchor, mlt, ip_index - pos8)
                fill_end(ip)
                continue
        h4 = _hash_mid4(rd32(src, ip))                # short match :823-930
        pos4 = h4t[h4]
        h4t[h4] = ip_index
        if pos4 > 0 and ip_index - pos4 <= DIST_MAX and pos4 < ip:
            mlen = _count(src, ip, pos4, matchlimit)
            if mlen >= MINMATCH:
                dist = ip_index - pos4
                if ip < mflimit:                      # :836-855
                    h8n = _hash_mid8(src, ip + 1)
                    pos8n = h8t[h8n]
                    m2d = ip_index + 1 - pos8n
                    if m2d <= DIST_MAX and pos8n > 0 and pos8n < ip + 1:
                        ml2 = _count(src, ip + 1, pos8n, matchlimit)
                        if ml2 > mlen:
                            h8t[h8n] = ip_index + 1
                            ip += 1
                            mlen, dist = ml2, m2d
                fill_begin(ip, ip_index)              # finalIpIndex4 = ipIndex, not ip (:869)
                ip = anchor = _encode_sequence(out, src, ip, anchor, mlen, dist)
                fill_end(ip)
                continue
        ip += 1 + ((ip - anchor) >> 9)                # :933-934
    _last_literals(out, src, anchor)
    return bytes(out)


def _lit_price(litlen):                               # :463-469
    p = litlen
    if litlen >= RUN_MASK:
        p += 1 + (litlen - RUN_MASK) // 255
    return p


def _seq_price(litlen, mlen):                         # :473-483
    p = 1 + 2 + _lit_price(litlen)
    if mlen >= ML_MASK + MINMATCH:
        p += 1 + (mlen - (ML_MASK + MINMATCH)) // 255
    return p


def _optimal(src, nb_searches, sufficient_len):       # compressOptimal :1068-1391
    n = len(src)
    if n < MFLIMIT + 1:
        return _encode_literals(src)
    TRAIL = 3
    opt = [[0, 0, 0, 0] for _ in range(OPT_NUM + TRAIL)]      # [price, off, mlen, litlen]
    PRICE, OFF, MLEN, LITLEN = 0, 1, 2, 3
    ctx = _Ctx()
    out = bytearray()
    ip = anchor = 0
    mflimit, matchlimit = n - MFLIMIT, n - LASTLITERALS
    if sufficient_len >= OPT_NUM:
        sufficient_len = OPT_NUM - 1

    def trailing(lmp):
        for add in range(1, TRAIL + 1):
            e = opt[lmp + add]
            e[MLEN], e[OFF], e[LITLEN] = 1, 0, add
            e[PRICE] = opt[lmp][PRICE] + _lit_price(add)

    while ip <= mflimit:                              # :1111 outer
        llen = ip - anchor
        _insert_hc(ctx, src, ip)
        f_len, f_off = _wider_match(ctx, src, ip, matchlimit, MINMATCH - 1, nb_searches, True)
        if f_len == 0:                                # :1127 (never: len starts at 3)
            ip += 1
            continue
        if f_len > sufficient_len:                    # :1133
            ip = anchor = _encode_sequence(out, src, ip, anchor, f_len, f_off)
            continue
        for r in range(MINMATCH):                     # :1150-1157
            opt[r][MLEN], opt[r][OFF], opt[r][LITLEN], opt[r][PRICE] = 1, 0, llen + r, _lit_price(llen + r)
        for ml in range(MINMATCH, f_len + 1):         # :1160-1169
            opt[ml][MLEN], opt[ml][OFF], opt[ml][LITLEN], opt[ml][PRICE] = ml, f_off, llen, _seq_price(llen, ml)
        last = f_len
        trailing(last)                                # :1174-1180
        cur = 1
        restart = False
        while cur < last:                             # :1183
            cur_ptr = ip + cur
            if cur_ptr > mflimit:
                break
            if opt[cur + 1][PRICE] <= opt[cur][PRICE]:
                cur += 1
                continue
            _insert_hc(ctx, src, cur_ptr)
            n_len, n_off = _wider_match(ctx, src, cur_ptr, matchlimit, MINMATCH - 1, nb_searches, True)
            if n_len == 0:
                cur += 1
                continue
            if n_len > sufficient_len or n_len + cur >= OPT_NUM:     # :1207-1256 (forward walk, as written)
                rp = 0
                while rp < cur:
                    ml, off = opt[rp][MLEN], opt[rp][OFF]
                    if ml == 1:
                        ip += 1
                        rp += 1
                        continue
                    rp += ml
                    ip = anchor = _encode_sequence(out, src, ip, anchor, ml, off)
                ip = anchor = _encode_sequence(out, src, ip, anchor, n_len, n_off)
                restart = True
                break
            base = opt[cur][LITLEN]                   # :1259-1270
            for litlen in range(1, MINMATCH):
                price = opt[cur][PRICE] - _lit_price(base) + _lit_price(base + litlen)
                pos = cur + litlen
                if price < opt[pos][PRICE]:
                    opt[pos][MLEN], opt[pos][OFF], opt[pos][LITLEN], opt[pos][PRICE] = 1, 0, base + litlen, price
            for ml in range(MINMATCH, n_len + 1):     # :1273-1302
                pos = cur + ml
                if opt[cur][MLEN] == 1:
                    ll = opt[cur][LITLEN]
                    price = opt[cur - ll][PRICE] if cur > ll else 0
                    price += _seq_price(ll, ml)
                else:
                    ll = 0
                    price = opt[cur][PRICE] + _seq_price(0, ml)
                if pos > last + TRAIL or price <= opt[pos][PRICE]:
                    if ml == n_len and last < pos:
                        last = pos
                    opt[pos][MLEN], opt[pos][OFF], opt[pos][LITLEN], opt[pos][PRICE] = ml, n_off, ll, price
            trailing(last)                            # :1305-1311
            cur += 1
        if restart:
            continue
        best_mlen, best_off = opt[last][MLEN], opt[last][OFF]       # :1315-1332
        cand = last - best_mlen
        sel_ml, sel_off = best_mlen, best_off
        while True:
            nxt_ml, nxt_off = opt[cand][MLEN], opt[cand][OFF]
            opt[cand][MLEN], opt[cand][OFF] = sel_ml, sel_off
            sel_ml, sel_off = nxt_ml, nxt_off
            if nxt_ml > cand:
                break
            cand -= nxt_ml
        r = 0                                         # :1335-1358
        while r < last:
            ml, off = opt[r][MLEN], opt[r][OFF]
            if ml == 1:
                ip += 1
                r += 1
                continue
            r += ml
            ip = anchor = _encode_sequence(out, src, ip, anchor, ml, off)
    if anchor > n:
        raise OverflowError("reference underflow: anchor past the end of the input (iend - anchor)")
    _last_literals(out, src, anchor)
    return bytes(out)


def compress_hc(src, level):
    """compressHC src/lz4hc.zig:1440-1453 -> compressHCExtState :1457-1489, level table :72-86."""
    src = bytes(src)
    if len(src) == 0:
        return b""
    level = 9 if level < 2 else (12 if level > 12 else level)
    if level == 2:
        return _mid(src)
    if level <= 9:
        return _hash_chain(src, 1 << (level - 1))
    nb, target = {10: (96, 64), 11: (512, 128), 12: (16384, OPT_NUM)}[level]
    return _optimal(src, nb, target)
