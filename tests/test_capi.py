"""The C-ABI library loads and exports every symbol include/zlz4_amd.h declares (no compute calls: no GPU here)."""
import os
import re

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "zlz4_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zlz4f?_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(zl):
    L = zl.lib()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(L, name), "libzlz4_amd.so does not export %s" % name
    assert set(declared) == set(zl.SYMBOLS), "Python binding and header disagree"


def test_library_is_hip_only(zl):
    """The product must not link the oracle and must carry gfx950 code objects."""
    import subprocess
    out = subprocess.run(["readelf", "-d", zl.LIB_PATH], capture_output=True, text=True).stdout
    assert "libamdhip64" in out and "oracle" not in out
    blob = open(zl.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"zo_compress" not in blob
    # the shipped library reads no environment variable (tuning knobs live in libzlz4_amd_tuning.so only)
    syms = subprocess.run(["nm", "-D", "--undefined-only", zl.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in syms


def test_pure_arithmetic_entry_points(zl, oracle):
    for n in (0, 1, 12, 13, 255, 256, 65536, 4194304, 0x7E000000, 0x7E000001):
        assert zl.compressBound(n) == oracle.compress_bound(n)                      # src/lz4.zig:80-83
    assert zl.compressBound(65536) == 65809 and zl.compressBound(4194304) == 4210768
    for bsid in (0, 4, 5, 6, 7):
        for n in (0, 1, 65536, 65537, 10_000_000):
            for bc in (0, 1):
                for cc in (0, 1):
                    p, q = zl.Prefs(), oracle.Prefs()
                    for x in (p, q):
                        x.block_size_id, x.block_checksum, x.content_checksum = bsid, bc, cc
                    assert zl.lz4f.compressFrameBound(n, p) == oracle.compress_frame_bound(n, q)
    frame = oracle.compress_frame(b"A" * 100)
    assert zl.lz4f.headerSize(frame) == 7 == oracle.header_size(frame)
    p = oracle.Prefs(); p.content_size = 100; p.dict_id = 5
    assert zl.lz4f.headerSize(oracle.compress_frame(b"A" * 100, p)) == 19
    assert zl.lz4f.headerSize(bytes([0x50, 0x2A, 0x4D, 0x18, 0, 0, 0, 0])) == 8      # skippable, lz4f.zig:459-462
    for bad, name in ((b"abc", "FrameHeaderIncomplete"), (b"\0" * 8, "FrameTypeUnknown")):
        try:
            zl.lz4f.headerSize(bad)
            assert False
        except zl.Lz4Error as e:
            assert e.name == name


def test_hc_workspace_size_is_bounded_and_monotone(zl):
    """zlz4_batch_compress_hc_workspace is host arithmetic: it grows with the batch up to the round size (8192 blocks of
    64 KiB: 448 KiB each), stays near 6 GiB for large blocks, and never shrinks when the batch grows."""
    per_block = 65536 * 6 + 65584                     # links + results, the level-10-12 price table
    assert zl.batch_compress_hc_workspace(1, 65536) == per_block
    assert zl.batch_compress_hc_workspace(8192, 65536) == 8192 * per_block
    assert zl.batch_compress_hc_workspace(16384, 65536) == 8192 * per_block       # longer batches run in rounds
    assert zl.batch_compress_hc_workspace(1 << 20, 65536) == 8192 * per_block
    prev = 0
    for nb in (1, 2, 100, 1000, 4096, 8192, 8193, 100000):
        w = zl.batch_compress_hc_workspace(nb, 65536)
        assert w >= prev
        prev = w
    big = zl.batch_compress_hc_workspace(1024, 4 << 20)
    assert (4 << 20) * 12 <= big <= (6 << 30) + (64 << 20)


def test_error_names_follow_the_reference(zl):
    names = {-1: "OutputTooSmall", -2: "InputTooLarge", -3: "CorruptedData", -4: "DecompressionFailed",
             -5: "InvalidState", -6: "AllocationFailed", -101: "Generic", -111: "DstMaxSizeTooSmall",
             -114: "FrameSizeWrong", -116: "DecompressionFailed", -117: "HeaderChecksumInvalid",
             -118: "ContentChecksumInvalid", -107: "BlockChecksumInvalid"}
    for code, name in names.items():
        assert zl.error_name(code) == name
    assert zl.lib().zlz4_version_string().startswith(b"zlz4-amd")


def test_no_device_means_loud_failure(zl):
    """On a box without a gfx950 GPU the compute entry points must fail (DeviceError), never fall back."""
    if zl.device_available():
        return
    for fn, args in ((zl.compressDefault, (b"A" * 100,)), (zl.compressHC, (b"A" * 100, 9)),
                     (zl.decompressSafe, (b"\x10A", 10)), (zl.lz4f.compressFrame, (b"A" * 100,)),
                     (zl.lz4f.decompressFrame, (bytes.fromhex("04224d184040c000000000"), 10))):
        try:
            fn(*args)
            assert False, "computed without a device"
        except zl.Lz4Error as e:
            assert e.name == "DeviceError"


def test_cpp_host_mirror_compiles_links_and_runs(zl, tmp_path):
    """zig-lz4_amd/csrc/host/zlz4.hpp (the compiled-language mirror of src/root.zig) against the shared library."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        import pytest
        pytest.skip("no g++")
    exe = str(tmp_path / "hmc")
    libdir = os.path.dirname(zl.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "host_mirror_check.cpp"),
                           "-L", libdir, "-lzlz4_amd", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "host mirror ok" in out.stdout, out.stdout + out.stderr


def test_decoder_hand_issued_loads_are_not_touched_before_their_wait():
    """tools/check_decoder_asm.py: the decoder's asm window prefetch must not be read, copied or spilled by compiler code
    before the asm s_waitcnt that covers it (re-run after any toolchain change)."""
    import shutil
    import subprocess
    import sys
    import pytest
    if not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_decoder_asm.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_zig_facade_declares_the_reference_names():
    """zig/root.zig cannot be compiled here (no zig toolchain), so at least its text is checked: every public name of
    src/root.zig:1-57 and the lz4f names a drop-in caller uses (src/lz4f.zig:31-57 `Error` with all 23 members and
    `isError`, :64-79 `BlockSizeID.toBlockSize`, :100-111 `FrameInfo.frameType`) are declared, and every extern
    function it binds is exported by the library."""
    import re
    txt = open(os.path.join(ROOT, "zig-lz4_amd", "zig", "root.zig")).read()
    for name in ("compressBound", "compressDefault", "compressFast", "decompressSafe", "compressHC", "compressHCExtState",
                 "MINMATCH", "LZ4_MAX_INPUT_SIZE", "LZ4_DISTANCE_MAX", "LZ4HC_CLEVEL_MIN", "LZ4HC_CLEVEL_DEFAULT",
                 "LZ4HC_CLEVEL_MAX"):
        assert re.search(r"pub (const|fn) %s\b" % name, txt), name
    m = re.search(r"pub const Error = error\{(.*?)\};", txt[txt.index("pub const lz4f = struct"):], re.S)
    assert m, "lz4f.Error"
    members = set(re.findall(r"\w+", m.group(1)))
    ref = """Generic MaxBlockSizeInvalid BlockModeInvalid ParameterInvalid CompressionLevelInvalid HeaderVersionWrong
             BlockChecksumInvalid ReservedFlagSet AllocationFailed SrcSizeTooLarge DstMaxSizeTooSmall FrameHeaderIncomplete
             FrameTypeUnknown FrameSizeWrong SrcPtrWrong DecompressionFailed HeaderChecksumInvalid ContentChecksumInvalid
             FrameDecodingAlreadyStarted CompressionStateUninitialized ParameterNull MaxCode OutOfMemory""".split()
    assert len(ref) == 23 and not [r for r in ref if r not in members], [r for r in ref if r not in members]
    for frag in ("pub fn isError(code: usize) bool", "pub fn toBlockSize(self: BlockSizeID) Error!usize",
                 "frameType: FrameType = .frame", "pub const FrameType = enum(u1)"):
        assert frag in txt, frag
    import zig_lz4_amd
    L = zig_lz4_amd.lib()
    for fn in set(re.findall(r'extern (?:"c" )?fn (zlz4f?_\w+)\(', txt)):
        assert hasattr(L, fn), "root.zig binds %s, which the library does not export" % fn
