//! Drop-in replacement for jedisct1/zig-lz4's `src/root.zig` (reference src/root.zig:1-57):
//! same `pub` names, same slices-in / error-union-out signatures, but every hot-path call is
//! forwarded to the MI355X library `libzlz4_amd.so` through its C ABI (include/zlz4_amd.h).
//! Host code stays pure Zig; link with `-lzlz4_amd` (see INTEGRATION.md).
//!
//! NOTE: this file could not be compiled in the build container (no zig toolchain); it is the
//! binding a maintainer adds, kept deliberately mechanical.

const std = @import("std");

// ---- C ABI (include/zlz4_amd.h) ----
extern "c" fn zlz4_compress_bound(input_size: usize) usize;
extern "c" fn zlz4_compress_default(src: [*]const u8, src_len: usize, dst: [*]u8, dst_cap: usize) i64;
extern "c" fn zlz4_compress_fast(src: [*]const u8, src_len: usize, dst: [*]u8, dst_cap: usize, acceleration: u32) i64;
extern "c" fn zlz4_compress_hc(src: [*]const u8, src_len: usize, dst: [*]u8, dst_cap: usize, level: i32) i64;
extern "c" fn zlz4_decompress_safe(src: [*]const u8, src_len: usize, dst: [*]u8, dst_cap: usize) i64;
extern "c" fn zlz4_decompress_safe_partial(src: [*]const u8, src_len: usize, dst: [*]u8, dst_cap: usize, target: usize) i64;
extern "c" fn zlz4_sizeof_state() usize;
extern "c" fn zlz4_compress_fast_ext_state(state: [*]u8, state_len: usize, src: [*]const u8, src_len: usize, dst: [*]u8, dst_cap: usize, acceleration: u32) i64;
extern "c" fn zlz4_compress_dest_size(src: [*]const u8, dst: [*]u8, dst_cap: usize, src_size: *usize) i64;
extern "c" fn zlz4_sizeof_state_hc() usize;
extern "c" fn zlz4_compress_hc_ext_state(state: [*]u8, state_len: usize, src: [*]const u8, src_len: usize, dst: [*]u8, dst_cap: usize, level: i32) i64;

// the data-parallel hot path: device pointers, one wavefront (HC: one workgroup) per block, results per block
extern "c" fn zlz4_batch_compress_fast(stream: ?*anyopaque, d_in: [*]const u8, d_in_off: [*]const u64, d_in_len: [*]const u32, d_out: [*]u8, d_out_off: [*]const u64, d_out_cap: [*]const u32, d_result: [*]i64, nblocks: u32, max_in_len: u32, acceleration: u32) i32;
extern "c" fn zlz4_batch_decompress_safe(stream: ?*anyopaque, d_in: [*]const u8, d_in_off: [*]const u64, d_in_len: [*]const u32, d_out: [*]u8, d_out_off: [*]const u64, d_out_cap: [*]const u32, d_result: [*]i64, nblocks: u32) i32;
extern "c" fn zlz4_batch_compress_hc_workspace(nblocks: u32, max_in_len: u32) usize;
extern "c" fn zlz4_batch_compress_hc(stream: ?*anyopaque, d_in: [*]const u8, d_in_off: [*]const u64, d_in_len: [*]const u32, d_out: [*]u8, d_out_off: [*]const u64, d_out_cap: [*]const u32, d_result: [*]i64, nblocks: u32, max_in_len: u32, level: i32, d_workspace: ?*anyopaque, workspace_bytes: usize) i32;
extern "c" fn zlz4_batch_verify(stream: ?*anyopaque, d_in: [*]const u8, d_in_off: [*]const u64, d_in_len: [*]const u32, d_comp: [*]const u8, d_comp_off: [*]const u64, d_comp_result: [*]const i64, d_verify: [*]i64, nblocks: u32) i64;

pub const CPrefs = extern struct {
    block_size_id: u32 = 0,
    block_mode: u32 = 0,
    content_checksum: u32 = 0,
    block_checksum: u32 = 0,
    content_size: u64 = 0,
    dict_id: u32 = 0,
    compression_level: i32 = 0,
};
extern "c" fn zlz4f_compress_frame_bound(src_size: usize, prefs: ?*const CPrefs) usize;
extern "c" fn zlz4f_compress_frame(src: [*]const u8, src_len: usize, dst: [*]u8, dst_cap: usize, prefs: ?*const CPrefs) i64;
extern "c" fn zlz4f_decompress_frame(src: [*]const u8, src_len: usize, dst: [*]u8, dst_cap: usize) i64;
extern "c" fn zlz4f_header_size(src: [*]const u8, src_len: usize) i64;
extern "c" fn zlz4f_compress_frame_device(stream: ?*anyopaque, d_src: [*]const u8, src_len: usize, d_dst: [*]u8, dst_cap: usize, prefs: ?*const CPrefs) i64;
extern "c" fn zlz4f_decompress_frame_device(stream: ?*anyopaque, d_src: [*]const u8, src_len: usize, d_dst: [*]u8, dst_cap: usize) i64;
extern "c" fn zlz4f_compress_frame_segment_device(stream: ?*anyopaque, d_src: [*]const u8, src_len: usize, d_dst: [*]u8, dst_cap: usize, prefs: ?*const CPrefs, segment_flags: u32) i64;
extern "c" fn zlz4f_decompress_frame_segment_device(stream: ?*anyopaque, d_src: [*]const u8, src_len: usize, d_dst: [*]u8, dst_cap: usize, prefs: ?*const CPrefs, segment_flags: u32) i64;

// ---- constants (reference src/lz4.zig:12-25, src/lz4hc.zig:28-31) ----
pub const MINMATCH = 4;
pub const LZ4_MAX_INPUT_SIZE = 0x7E000000;
pub const LZ4_DISTANCE_MAX = 65535;
pub const LZ4HC_CLEVEL_MIN = 2;
pub const LZ4HC_CLEVEL_DEFAULT = 9;
pub const LZ4HC_CLEVEL_MAX = 12;

/// reference src/lz4.zig:48-55 plus the two device-side additions
pub const Error = error{
    OutputTooSmall,
    InputTooLarge,
    CorruptedData,
    DecompressionFailed,
    InvalidState,
    AllocationFailed,
    DeviceError,
    Unsupported,
};

fn mapBlock(code: i64) Error!usize {
    if (code >= 0) return @intCast(code);
    return switch (code) {
        -1 => error.OutputTooSmall,
        -2 => error.InputTooLarge,
        -3 => error.CorruptedData,
        -4 => error.DecompressionFailed,
        -5 => error.InvalidState,
        -6 => error.AllocationFailed,
        -8 => error.Unsupported,
        else => error.DeviceError,
    };
}

pub fn compressBound(inputSize: usize) usize {
    return zlz4_compress_bound(inputSize);
}
pub fn compressDefault(src: []const u8, dst: []u8) Error!usize {
    return mapBlock(zlz4_compress_default(src.ptr, src.len, dst.ptr, dst.len));
}
pub fn compressFast(src: []const u8, dst: []u8, acceleration: u32) Error!usize {
    return mapBlock(zlz4_compress_fast(src.ptr, src.len, dst.ptr, dst.len, acceleration));
}
pub fn decompressSafe(src: []const u8, dst: []u8) Error!usize {
    return mapBlock(zlz4_decompress_safe(src.ptr, src.len, dst.ptr, dst.len));
}
/// reference src/lz4.zig:619-621
pub fn decompressSafePartial(src: []const u8, dst: []u8, targetOutputSize: usize) Error!usize {
    return mapBlock(zlz4_decompress_safe_partial(src.ptr, src.len, dst.ptr, dst.len, targetOutputSize));
}
/// reference src/lz4.zig:524-526
pub fn sizeofState() usize {
    return zlz4_sizeof_state();
}
/// reference src/lz4.zig:531-546
pub fn compressFastExtState(state: []u8, src: []const u8, dst: []u8, acceleration: u32) Error!usize {
    return mapBlock(zlz4_compress_fast_ext_state(state.ptr, state.len, src.ptr, src.len, dst.ptr, dst.len, acceleration));
}
/// reference src/lz4.zig:551-616 (srcSizePtr: in = available, out = consumed)
pub fn compressDestSize(src: []const u8, dst: []u8, srcSizePtr: *usize) Error!usize {
    return mapBlock(zlz4_compress_dest_size(src.ptr, dst.ptr, dst.len, srcSizePtr));
}
pub fn compressHC(src: []const u8, dst: []u8, compressionLevel: i32) Error!usize {
    return mapBlock(zlz4_compress_hc(src.ptr, src.len, dst.ptr, dst.len, compressionLevel));
}
/// reference src/lz4hc.zig:1492-1494
pub fn sizeofStateHC() usize {
    return zlz4_sizeof_state_hc();
}
/// reference src/lz4hc.zig:1457-1489.  The reference takes `*Context`; its tables live on the device here, so the
/// context is passed as the bytes it occupies (a fresh `Context.init()`, as compressHC itself uses, :1450).
pub fn compressHCExtState(ctx: []u8, src: []const u8, dst: []u8, compressionLevel: i32) Error!usize {
    return mapBlock(zlz4_compress_hc_ext_state(ctx.ptr, ctx.len, src.ptr, src.len, dst.ptr, dst.len, compressionLevel));
}

/// `@import("lz4").lz4.compressDefault(...)` and `.lz4hc.compressHC(...)` keep working (reference src/root.zig:3-5)
const root = @This();
pub const lz4 = struct {
    pub const Error = root.Error;
    pub const MINMATCH = 4;
    pub const LZ4_MAX_INPUT_SIZE = 0x7E000000;
    pub const LZ4_DISTANCE_MAX = 65535;
    pub const compressBound = root.compressBound;
    pub const compressDefault = root.compressDefault;
    pub const compressFast = root.compressFast;
    pub const compressDestSize = root.compressDestSize;
    pub const decompressSafe = root.decompressSafe;
    pub const decompressSafePartial = root.decompressSafePartial;
    pub const sizeofState = root.sizeofState;
    pub const compressFastExtState = root.compressFastExtState;
};
pub const lz4hc = struct {
    pub const LZ4HC_CLEVEL_MIN = 2;
    pub const LZ4HC_CLEVEL_DEFAULT = 9;
    pub const LZ4HC_CLEVEL_MAX = 12;
    pub const compressHC = root.compressHC;
    pub const compressHCExtState = root.compressHCExtState;
    pub const sizeofStateHC = root.sizeofStateHC;
};

/// The hot path itself (no counterpart in the reference, whose calls take one block): many independent blocks per
/// call, all pointers DEVICE pointers (`[*]` = raw device address), asynchronous on `stream` (a hipStream_t, null =
/// default stream).  `descs` is the slice-of-slices view a Zig caller has, flattened into the four descriptor arrays
/// the C ABI takes; they live in device memory like the payload.
pub const device = struct {
    pub const Blocks = struct {
        in: [*]const u8, // input arena
        in_off: [*]const u64, // per block: offset into `in`
        in_len: [*]const u32, // per block: bytes
        out: [*]u8, // output arena
        out_off: [*]const u64, // per block: offset into `out`
        out_cap: [*]const u32, // per block: capacity
        result: [*]i64, // per block: bytes written or -(lz4.Error index)
        nblocks: u32,
    };
    fn mapLaunch(rc: i32) Error!void {
        if (rc == 0) return;
        _ = try mapBlock(rc);
    }
    /// batch form of compressFast (src/lz4.zig:292-447); every in_len[i] <= max_in_len
    pub fn compressFastBatch(stream: ?*anyopaque, b: Blocks, max_in_len: u32, acceleration: u32) Error!void {
        return mapLaunch(zlz4_batch_compress_fast(stream, b.in, b.in_off, b.in_len, b.out, b.out_off, b.out_cap, b.result, b.nblocks, max_in_len, acceleration));
    }
    /// batch form of decompressSafe (src/lz4.zig:257-259)
    pub fn decompressSafeBatch(stream: ?*anyopaque, b: Blocks) Error!void {
        return mapLaunch(zlz4_batch_decompress_safe(stream, b.in, b.in_off, b.in_len, b.out, b.out_off, b.out_cap, b.result, b.nblocks));
    }
    pub fn compressHCWorkspace(nblocks: u32, max_in_len: u32) usize {
        return zlz4_batch_compress_hc_workspace(nblocks, max_in_len);
    }
    /// batch form of compressHC (src/lz4hc.zig:1440-1453); `workspace` = device memory of compressHCWorkspace() bytes
    pub fn compressHCBatch(stream: ?*anyopaque, b: Blocks, max_in_len: u32, level: i32, workspace: ?*anyopaque, workspace_bytes: usize) Error!void {
        return mapLaunch(zlz4_batch_compress_hc(stream, b.in, b.in_off, b.in_len, b.out, b.out_off, b.out_cap, b.result, b.nblocks, max_in_len, level, workspace, workspace_bytes));
    }
    /// opt-in check for levels 10..12 (include/zlz4_amd.h): decodes the batch a compress call produced (`b` as passed to
    /// it) and compares with the input; verify[i] = b.result[i] or -9; returns the number of blocks that do not round-trip
    pub fn verifyBatch(stream: ?*anyopaque, b: Blocks, verify: [*]i64) Error!usize {
        return mapBlock(zlz4_batch_verify(stream, b.in, b.in_off, b.in_len, b.out, b.out_off, b.result, verify, b.nblocks));
    }
};

/// Mirror of the `lz4f` namespace (reference src/lz4f.zig); enum/struct shapes as in :64-122.
pub const lz4f = struct {
    pub const MAGICNUMBER: u32 = 0x184D2204;
    /// reference src/lz4f.zig:57-59
    pub fn isError(code: usize) bool {
        return code > @as(usize, @bitCast(@as(isize, -65536)));
    }
    pub const BlockSizeID = enum(u3) {
        default = 0,
        max64KB = 4,
        max256KB = 5,
        max1MB = 6,
        max4MB = 7,

        /// reference src/lz4f.zig:71-78
        pub fn toBlockSize(self: BlockSizeID) Error!usize {
            return switch (self) {
                .default, .max64KB => 64 * 1024,
                .max256KB => 256 * 1024,
                .max1MB => 1024 * 1024,
                .max4MB => 4 * 1024 * 1024,
            };
        }
    };
    pub const BlockMode = enum(u1) { linked = 0, independent = 1 };
    pub const ContentChecksum = enum(u1) { disabled = 0, enabled = 1 };
    pub const BlockChecksum = enum(u1) { disabled = 0, enabled = 1 };
    pub const FrameType = enum(u1) { frame = 0, skippableFrame = 1 };
    pub const FrameInfo = struct {
        blockSizeID: BlockSizeID = .default,
        blockMode: BlockMode = .linked,
        contentChecksumFlag: ContentChecksum = .disabled,
        frameType: FrameType = .frame,   // reference :100-111; compressFrame always writes .frame (:304-351)
        contentSize: u64 = 0,
        dictID: u32 = 0,
        blockChecksumFlag: BlockChecksum = .disabled,
    };
    pub const Preferences = struct {
        frameInfo: FrameInfo = .{},
        compressionLevel: i32 = 0,
        autoFlush: bool = false,
        favorDecSpeed: bool = false,
    };
    /// reference src/lz4f.zig:31-55: every member, under the reference's name (`lz4f.Error`), so that callers that
    /// switch on it or name `lz4f.Error!usize` compile unchanged; + the two device additions
    pub const Error = error{
        Generic, MaxBlockSizeInvalid, BlockModeInvalid, ParameterInvalid, CompressionLevelInvalid, HeaderVersionWrong,
        BlockChecksumInvalid, ReservedFlagSet, AllocationFailed, SrcSizeTooLarge, DstMaxSizeTooSmall,
        FrameHeaderIncomplete, FrameTypeUnknown, FrameSizeWrong, SrcPtrWrong, DecompressionFailed,
        HeaderChecksumInvalid, ContentChecksumInvalid, FrameDecodingAlreadyStarted, CompressionStateUninitialized,
        ParameterNull, MaxCode, OutOfMemory,
        DeviceError, Unsupported,
    };
    pub const FrameError = Error;   // the name earlier revisions of this facade used
    fn mapFrame(code: i64) Error!usize {
        if (code >= 0) return @intCast(code);
        return switch (code) {
            -101 => error.Generic, -102 => error.MaxBlockSizeInvalid, -106 => error.HeaderVersionWrong,
            -107 => error.BlockChecksumInvalid, -108 => error.ReservedFlagSet, -109 => error.AllocationFailed,
            -110 => error.SrcSizeTooLarge, -111 => error.DstMaxSizeTooSmall, -112 => error.FrameHeaderIncomplete,
            -113 => error.FrameTypeUnknown, -114 => error.FrameSizeWrong, -116 => error.DecompressionFailed,
            -117 => error.HeaderChecksumInvalid, -118 => error.ContentChecksumInvalid, -8 => error.Unsupported,
            else => error.DeviceError,
        };
    }
    fn toC(p: Preferences) CPrefs {
        return .{
            .block_size_id = @intFromEnum(p.frameInfo.blockSizeID),
            .block_mode = @intFromEnum(p.frameInfo.blockMode),
            .content_checksum = @intFromEnum(p.frameInfo.contentChecksumFlag),
            .block_checksum = @intFromEnum(p.frameInfo.blockChecksumFlag),
            .content_size = p.frameInfo.contentSize,
            .dict_id = p.frameInfo.dictID,
            .compression_level = p.compressionLevel,
        };
    }
    pub fn compressFrameBound(srcSize: usize, prefs: ?Preferences) usize {
        if (prefs) |p| { const c = toC(p); return zlz4f_compress_frame_bound(srcSize, &c); }
        return zlz4f_compress_frame_bound(srcSize, null);
    }
    /// the allocator argument of the reference (unused there, src/lz4f.zig:443) is kept for source compatibility
    pub fn compressFrame(allocator: std.mem.Allocator, src: []const u8, dst: []u8, prefs: ?Preferences) Error!usize {
        _ = allocator;
        if (prefs) |p| { const c = toC(p); return mapFrame(zlz4f_compress_frame(src.ptr, src.len, dst.ptr, dst.len, &c)); }
        return mapFrame(zlz4f_compress_frame(src.ptr, src.len, dst.ptr, dst.len, null));
    }
    pub fn decompressFrame(allocator: std.mem.Allocator, src: []const u8, dst: []u8) Error!usize {
        _ = allocator;
        return mapFrame(zlz4f_decompress_frame(src.ptr, src.len, dst.ptr, dst.len));
    }
    pub fn headerSize(src: []const u8) Error!usize {
        return mapFrame(zlz4f_header_size(src.ptr, src.len));
    }

    /// Device-resident frames (BASELINE configs[4]): `d_src` / `d_dst` are device pointers.
    pub const SEG_FIRST: u32 = 1;
    pub const SEG_LAST: u32 = 2;
    pub fn compressFrameDevice(stream: ?*anyopaque, d_src: [*]const u8, src_len: usize, d_dst: [*]u8, dst_cap: usize, prefs: ?Preferences) Error!usize {
        if (prefs) |p| { const c = toC(p); return mapFrame(zlz4f_compress_frame_device(stream, d_src, src_len, d_dst, dst_cap, &c)); }
        return mapFrame(zlz4f_compress_frame_device(stream, d_src, src_len, d_dst, dst_cap, null));
    }
    pub fn decompressFrameDevice(stream: ?*anyopaque, d_src: [*]const u8, src_len: usize, d_dst: [*]u8, dst_cap: usize) Error!usize {
        return mapFrame(zlz4f_decompress_frame_device(stream, d_src, src_len, d_dst, dst_cap));
    }
    /// one rank's block segment of a frame spread over several GPUs (include/zlz4_amd.h)
    pub fn compressFrameSegmentDevice(stream: ?*anyopaque, d_src: [*]const u8, src_len: usize, d_dst: [*]u8, dst_cap: usize, prefs: Preferences, segment_flags: u32) Error!usize {
        const c = toC(prefs);
        return mapFrame(zlz4f_compress_frame_segment_device(stream, d_src, src_len, d_dst, dst_cap, &c, segment_flags));
    }
    pub fn decompressFrameSegmentDevice(stream: ?*anyopaque, d_src: [*]const u8, src_len: usize, d_dst: [*]u8, dst_cap: usize, prefs: Preferences, segment_flags: u32) Error!usize {
        const c = toC(prefs);
        return mapFrame(zlz4f_decompress_frame_segment_device(stream, d_src, src_len, d_dst, dst_cap, &c, segment_flags));
    }
};
