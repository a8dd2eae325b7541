#!/usr/bin/env python3
"""Lane-level Python emulation of k_compress_fast's window path + generic path (debug aid).
Mirrors the window logic of zig-lz4_amd/csrc/zlz4_compress_fast.hip as it was when the window path was brought up
(12-byte in-register compare, hand-over at lane 49, one flush per sequence): the later refinements of the kernel
(48-byte compare levels, extension bytes in the run flush, precomputed fast-run steps, hand-over at lane 64) change
how many sequences a window resolves and how they are emitted, not the bytes, so the emulator still produces the
oracle's output and remains the place to try a change of the lane logic on the CPU first."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

M32 = 0xFFFFFFFF
def rd32(b, p): return int.from_bytes(b[p:p+4], "little")
def hash4(x): return ((x * 2654435761) & M32) >> 20
def ext_len_bytes(v): return 1 + (v - 15) // 255 if v >= 15 else 0
def ext_bytes(v):
    r = v - 15; return bytes([255] * (r // 255) + [r % 255])
def ctz(x): return (x & -x).bit_length() - 1
def msb(x): return x.bit_length() - 1
def S(x):
    q, r = x >> 6, x & 63; return 32 * q * (q - 1) + q * r

def extend(src, m_pos, m_cand, mlen, match_limit):
    ip, mt = m_pos + 4 + mlen, m_cand + 4 + mlen
    while ip < match_limit and src[ip] == src[mt]:
        ip += 1; mt += 1; mlen += 1
    return mlen

def compress(src, accel=1, dst_len=None, RESTART=48, trace=False):
    n = len(src)
    if n == 0: return b""
    if n < 13:
        return bytes([min(n,15) << 4]) + (ext_bytes(n) if n >= 15 else b"") + src
    dst_len = dst_len if dst_len is not None else n + n // 255 + 16
    table = [0] * 4096
    L, match_limit = n - 12, n - 5
    anchor, op, F0, has_ins = 0, 0, 1, False
    out = bytearray(dst_len + 64)
    cbase = max(64, accel); s_cbase = S(cbase)
    def emit_general(lit_src_pos, lit, offset, mlen):
        nonlocal op
        tok = (min(lit, 15) << 4) | min(mlen, 15)
        b = bytes([tok]) + (ext_bytes(lit) if lit >= 15 else b"") + src[lit_src_pos:lit_src_pos+lit] + offset.to_bytes(2, "little") + (ext_bytes(mlen) if mlen >= 15 else b"")
        out[op:op+len(b)] = b; op += len(b)
    while F0 < L:
        ub = -1
        if accel == 1 and F0 == anchor + 1 and (has_ins or anchor == 0) and anchor + 192 < L:
            A = anchor
            pos = [A + i for i in range(64)]
            wr = [has_ins or i > 0 for i in range(64)]
            fwd = [[rd32(src, pos[i] + 4*k) for k in range(4)] for i in range(64)]
            h = [hash4(fwd[i][0]) for i in range(64)]
            old = [table[h[i]] if wr[i] else 0 for i in range(64)]
            for i in range(64):
                if wr[i]: table[h[i]] = pos[i]      # race winner: any; emulate "last lane wins"
            grp = [0] * 64
            for i in range(64):
                g = 0
                for k in range(64):
                    if wr[i] and wr[k] and h[k] == h[i]: g |= 1 << k
                grp[i] = g if g else (1 << i)
                if not wr[i]: grp[i] = 1 << i
            old_ok = [wr[i] and old[i] > 0 and old[i] + 65535 >= pos[i] for i in range(64)]
            cold = [[rd32(src, old[i] + 4*k) for k in range(4)] if old_ok[i] else [0]*4 for i in range(64)]
            vo = [old_ok[i] and cold[i][0] == fwd[i][0] for i in range(64)]
            mlo = []
            for i in range(64):
                x1, x2, x3 = fwd[i][1]^cold[i][1], fwd[i][2]^cold[i][2], fwd[i][3]^cold[i][3]
                mlo.append(ctz(x1)>>3 if x1 else (4+(ctz(x2)>>3) if x2 else (8+(ctz(x3)>>3) if x3 else 12)))
            single = [grp[i] == (1 << i) for i in range(64)]
            wrmask = sum(1 << i for i in range(64) if wr[i])
            cfast = sum(1 << i for i in range(64) if vo[i] and mlo[i] < 12)
            slow = sum(1 << i for i in range(64) if wr[i] and (((not single[i]) and not (vo[i] and mlo[i] < 12)) or (vo[i] and mlo[i] >= 12)))
            nsing = sum(1 << i for i in range(64) if wr[i] and not single[i])
            def first_ge(mask, i):
                m = mask >> i
                return i + ctz(m) if m else 64
            J = [first_ge(cfast, i) for i in range(64)]
            Sx = [first_ge(slow, i) for i in range(64)]
            v_end = [i + 4 + mlo[i] for i in range(64)]
            E = [v_end[J[i] & 63] for i in range(64)]
            f, a, nseq, covered = 1, 0, 0, 0
            continue_generic = False
            tight = dst_len - op < 512
            while True:
                # ---- fast run: collect match lanes with a minimal scalar loop ----
                a0, op0, mm_run = a, op, 0
                while f < 64 and (f < 49 or (nseq == 0 and mm_run == 0)):
                    j, sl = J[f], Sx[f]
                    if tight or sl <= j or j - a >= 15 or (nsing >> j) & 1: break
                    mm_run |= 1 << j
                    e = v_end[j]; a = e; f = e + 1
                if mm_run:
                    # ---- vector flush of the run ----
                    jlast = msb(mm_run)
                    litmask = 0; cov_run = 0
                    info = []
                    for i in range(64):
                        mb = mm_run & ((1 << i) - 1)
                        has_prev = mb != 0
                        pend = v_end[msb(mb)] if has_prev else a0
                        cov = has_prev and i < pend
                        is_m = (mm_run >> i) & 1
                        is_lit = i >= a0 and i < jlast and not cov and not is_m
                        if is_lit: litmask |= 1 << i
                        if cov: cov_run |= 1 << i
                        info.append((pend, is_m, is_lit))
                    for i in range(64):
                        pend, is_m, is_lit = info[i]
                        k = bin(mm_run & ((1 << i) - 1)).count("1")
                        Lb = bin(litmask & ((1 << i) - 1)).count("1")
                        if is_lit: out[op0 + 3 * k + Lb + 1] = fwd[i][0] & 0xFF
                        if is_m:
                            lit_k = i - pend
                            out[op0 + 3 * k + Lb - lit_k] = (lit_k << 4) | mlo[i]
                            o2 = op0 + 3 * k + Lb + 1
                            out[o2:o2 + 2] = ((pos[i] - old[i]) & 0xFFFF).to_bytes(2, "little")
                            if trace: print("  fast seq A=%d j=%d cand=%d lit=%d mlen=%d" % (A, i, old[i], lit_k, mlo[i]))
                    op = op0 + 3 * bin(mm_run).count("1") + bin(litmask).count("1")
                    covered |= cov_run
                    nseq += bin(mm_run).count("1")
                if f >= 64 or (f >= 49 and nseq > 0):
                    if nseq == 0: continue_generic = True
                    break
                # ---- exact step for the search at probe lane f ----
                j, sl = J[f], Sx[f]
                x = sl if sl < j else j
                if x >= 64:
                    if nseq == 0: continue_generic = True
                    break
                pm = 0
                if (nsing >> x) & 1:
                    pm = grp[x] & wrmask & ~covered & ((1 << x) - 1)
                if pm:
                    pr = msb(pm); ok = fwd[pr][0] == fwd[x][0]; m_cand = A + pr; c = fwd[pr]
                else:
                    ok = vo[x]; m_cand = old[x]; c = cold[x]
                if not ok:
                    f = x + 1; continue
                j = x; m_pos = A + j
                xa = ((fwd[j][2] ^ c[2]) << 32) | (fwd[j][1] ^ c[1]); xb = fwd[j][3] ^ c[3]
                if xa: mlen = ctz(xa) >> 3
                elif xb: mlen = 8 + (ctz(xb) >> 3)
                else: mlen = extend(src, m_pos, m_cand, 12, match_limit)
                lit = j - a
                if trace: print("  slow seq A=%d j=%d cand=%d lit=%d mlen=%d" % (A, j, m_cand, lit, mlen))
                emit_general(A + a, lit, m_pos - m_cand, mlen)
                e = j + 4 + mlen
                hi = min(e, 64)
                covered |= ((1 << hi) - 1) & ~((2 << j) - 1)
                nseq += 1
                a = e
                if e >= 64: break
                f = e + 1
            anchor = A + a
            inside = [bool((covered >> i) & 1) for i in range(64)]
            f_end = 64 if continue_generic else (64 if a >= 64 else a + 1)
            ins = sum(1 << i for i in range(64) if wr[i] and i < f_end and not inside[i])
            for i in range(64):
                if wr[i] and not (ins >> i) & 1: table[h[i]] = old[i]
            for i in range(64):
                if (ins >> i) & 1 and (grp[i] & ins & ~((1 << i) - 1) & ~(1 << i)) == 0: table[h[i]] = pos[i]
            if not continue_generic:
                if anchor < L: has_ins, F0 = True, anchor + 1
                else: has_ins, F0 = False, L
                continue
            ub = 63
        # generic (serial emulation of one search from probe index ub)
        found = False
        u = ub
        while True:
            if u < 0:
                if has_ins:
                    p = F0 - 1; table[hash4(rd32(src, p))] = p
                u = 0; continue
            if u == 0: pos_, st = F0, accel
            elif u == 1: pos_, st = F0 + accel, accel >> 6
            else:
                x = cbase + u - 1; pos_, st = F0 + accel + S(x) - s_cbase, x >> 6
            if pos_ + st > L: break
            hh = hash4(rd32(src, pos_)); cand = table[hh]
            valid = cand > 0 and cand < pos_ and cand + 65535 >= pos_ and rd32(src, cand) == rd32(src, pos_)
            table[hh] = pos_
            if valid: found = True; break
            u += 1
        if not found: break
        mlen = extend(src, pos_, cand, 0, match_limit)
        if trace: print("  generic seq pos=%d cand=%d lit=%d mlen=%d" % (pos_, cand, pos_ - anchor, mlen))
        emit_general(anchor, pos_ - anchor, pos_ - cand, mlen)
        end = pos_ + 4 + mlen; anchor = end
        if end < L: has_ins, F0 = True, end + 1
        else: has_ins, F0 = False, L
    lit = n - anchor
    if lit:
        b = bytes([min(lit,15) << 4]) + (ext_bytes(lit) if lit >= 15 else b"") + src[anchor:]
        out[op:op+len(b)] = b; op += len(b)
    return bytes(out[:op])

if __name__ == "__main__":
    import cases
    from oracle import binding as o
    bad = 0
    allc = cases.reference_test_inputs() + list(cases.kat_inputs().items()) + cases.seeded_cases()
    for name, b in allc:
        if len(b) > 20000: continue
        w = o.compress_default(b); g = compress(b)
        if g != w:
            bad += 1
            first = next((i for i, (x, y) in enumerate(zip(g, w)) if x != y), min(len(g), len(w)))
            print("MISMATCH", name, len(g), len(w), "first diff", first)
    print("bad", bad, "of", len(allc))
