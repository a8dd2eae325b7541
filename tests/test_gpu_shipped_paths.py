"""Paths of the SHIPPED library (libzlz4_amd.so, no tuning knobs) that only large batches reach, against the oracle:

* the decoder build the headline uses -- `k_decompress_safe<true, true>`, chosen by the launcher for >= 6144 blocks
  (zig-lz4_amd/csrc/zlz4_decompress.hip) -- with malformed, truncated and short-capacity blocks inside such a batch
  (status order of src/lz4.zig:111-247);
* levels 10..12 beyond one chunk of the HC pipeline (> 8192 blocks: the second trip of launch_hc_chunked's loop,
  zig-lz4_amd/csrc/zlz4_compress_hc.hip);
* the GPU branch of the C++ host mirror (tests/host_mirror_check.cpp: batch round trip through csrc/host/zlz4.hpp).
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

import datagen as dg
import gpu_harness as gh

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cmp(names, got, want):
    bad = []
    for name, (n, data), w in zip(names, got, want):
        if isinstance(w, int):
            if n != w:
                bad.append("%s: status %d, oracle %d" % (name, n, w))
        elif n != len(w) or data != w:
            first = next((i for i, (a, b) in enumerate(zip(data, w)) if a != b), min(len(data), len(w)))
            bad.append("%s: size %d vs oracle %d, first diff at %d" % (name, n, len(w), first))
    assert not bad, "%d/%d mismatches: %s" % (len(bad), len(names), "; ".join(bad[:8]))


def test_large_batch_decoder_malformed_and_capacity_statuses(zl, oracle, gpu):
    """7000 blocks in ONE decode call of the shipped library (>= 6144 -> the lane-copy decoder build).  Blocks of
    600..6000 bytes (text, repetitive text, mixed), fast / accelerated / HC streams; every 9th block is damaged (bit
    flip, byte replacement, offset 0, 255-chain, forced length extensions), truncated, or given a capacity around its
    exact size.  Status equals the oracle's for every block, bytes too where it decodes."""
    rng = np.random.default_rng(20261005)
    nblocks = 7000
    names, comp, caps = [], [], []
    for i in range(nblocks):
        n = int(rng.integers(600, 6000))
        kind = i % 3
        b = bytes((dg.text_bytes, dg.reptext_bytes, dg.mixed_bytes)[kind](n, 1000 + i))
        c = (oracle.compress_default(b), oracle.compress_fast(b, 5), oracle.compress_hc(b, 6))[(i // 3) % 3]
        cap, what = n, "ok"
        if i % 9 == 4:
            m = bytearray(c)
            pos = int(rng.integers(0, len(m)))
            k = (i // 9) % 6
            if k == 0: m[pos] ^= 1 << int(rng.integers(0, 8))
            elif k == 1: m[pos] = int(rng.integers(0, 256))
            elif k == 2: m[pos:pos + 2] = b"\x00\x00"
            elif k == 3: m[pos:pos + 4] = b"\xff\xff\xff\xff"
            elif k == 4: m[pos] = 0xF0 | (m[pos] & 15)
            else: m[pos] = (m[pos] & 0xF0) | 15
            c, what = bytes(m), "corrupt%d@%d" % (k, pos)
        elif i % 9 == 7:
            cut = int(rng.integers(1, len(c)))
            c, what = c[:cut], "trunc%d" % cut
        elif i % 9 == 2:
            cap = n + int(rng.choice([-1, -4, -17, -31, -32, -33, -100, 1, 31, 32, -n // 2]))
            what = "cap%d" % cap
        names.append("blk%d/n%d/%s" % (i, n, what)); comp.append(c); caps.append(cap)
    want = [oracle.decompress_safe(c, cap) for c, cap in zip(comp, caps)]
    nerr = sum(isinstance(w, int) for w in want)
    assert nerr > 600 and len(want) - nerr > 5000, nerr
    got = gh.decompress(zl, comp, caps, gpu)
    _cmp(names, got, want)


@pytest.mark.parametrize("level", [12, 10])
def test_optimal_levels_beyond_one_chunk(zl, oracle, gpu, level):
    """8500 small blocks (> 8192: a second chunk in launch_hc_chunked) at levels 12 and 10, bytes vs the oracle
    (compressOptimal, src/lz4hc.zig:1068-1391; levels 10..12 are compared, never round-tripped -- DESIGN.md section 2)."""
    rng = np.random.default_rng(7 + level)
    items = []
    for i in range(8500):
        n = int(rng.integers(200, 900))
        items.append(bytes((dg.text_bytes, dg.reptext_bytes)[i & 1](n, 50000 + i)))
    items[8300] = bytes(dg.text_bytes(20000, 99))          # one larger block in the second chunk
    items[100] = b""                                      # and the edge sizes in the first
    items[101] = b"abcabcabcabc"
    got = gh.compress_hc(zl, items, gpu, level)
    want = [oracle.compress_hc(b, level) for b in items]
    _cmp(["blk%d" % i for i in range(len(items))], got, want)


def test_cpp_host_mirror_gpu_branch(zl, gpu, tmp_path):
    """tests/host_mirror_check.cpp built with g++ against the shipped library, run ON the GPU box: single-block and batch
    round trips through zig-lz4_amd/csrc/host/zlz4.hpp (the compiled mirror of src/root.zig:1-57)."""
    if shutil.which("g++") is None:
        pytest.skip("no g++ on this box: the C++ mirror's GPU branch was NOT exercised")
    exe = str(tmp_path / "hmc")
    libdir = os.path.dirname(zl.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "host_mirror_check.cpp"),
                           "-L", libdir, "-lzlz4_amd", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0 and "host mirror ok" in out.stdout, out.stdout + out.stderr
    assert "batch" in out.stdout, "the GPU branch did not run: " + out.stdout


def test_batch_verify_flags_the_streams_that_do_not_round_trip(zl, oracle, gpu):
    """zlz4_batch_verify (the opt-in safety net for levels 10..12, include/zlz4_amd.h): level 9 streams all verify; at
    level 10 the blocks whose reference stream does not decode to the input (the oracle's own decoder decides) are
    flagged with ZLZ4_ERR_VERIFY and the others are not; negative compress results pass through."""
    import torch
    items = [bytes(dg.text_bytes(n, 300 + i)) for i, n in enumerate((65536, 40000, 65536, 3000, 65536, 12, 0, 65536))]
    for level in (9, 10):
        buf, offs, lens = gh._pack(items)
        caps = np.array([zl.compressBound(len(b)) for b in items], dtype=np.int64)
        caps[3] = 20                                            # one block that cannot fit: OutputTooSmall passes through
        out_offs = np.concatenate([[0], np.cumsum((caps + 15) // 16 * 16 + 64)[:-1]]).astype(np.int64)
        d_in = torch.from_numpy(buf).to(gpu)
        d_out = torch.zeros(int(out_offs[-1] + caps[-1] + 80), dtype=torch.uint8, device=gpu)
        res = torch.full((len(items),), -999, dtype=torch.int64, device=gpu)
        t_off, t_len = torch.from_numpy(offs).to(gpu), torch.from_numpy(lens.astype(np.uint32).view(np.int32)).to(gpu)
        t_ooff, t_cap = torch.from_numpy(out_offs).to(gpu), torch.from_numpy(caps.astype(np.uint32).view(np.int32)).to(gpu)
        ws = torch.empty(max(16, zl.batch_compress_hc_workspace(len(items), 65536)), dtype=torch.uint8, device=gpu)
        zl.batch_compress_hc(d_in, t_off, t_len, d_out, t_ooff, t_cap, res, 65536, level, ws)
        ver = torch.full((len(items),), -777, dtype=torch.int64, device=gpu)
        nbad = zl.batch_verify(d_in, t_off, t_len, d_out, t_ooff, res, ver)
        r, v, o = res.cpu().numpy(), ver.cpu().numpy(), d_out.cpu().numpy()
        want_bad = 0
        for i, b in enumerate(items):
            if r[i] < 0:
                assert v[i] == r[i], (level, i, r[i], v[i])
                continue
            stream = bytes(o[out_offs[i]: out_offs[i] + r[i]])
            dec = oracle.decompress_safe(stream, len(b))
            good = (not isinstance(dec, int)) and dec == b
            want_bad += 0 if good else 1
            assert v[i] == (r[i] if good else -9), (level, i, r[i], v[i], good)
        assert nbad == want_bad, (level, nbad, want_bad)
        assert r[3] == -1
        if level == 9:
            assert nbad == 0
        else:
            assert nbad > 0, "level 10 is expected to lose ordinary 64 KiB text blocks (DESIGN.md section 2)"
