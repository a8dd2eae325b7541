// zlz4.hpp -- C++ host-side mirror of the reference's public facade (src/root.zig:1-57) over the C ABI.
//
// The reference is compiled Zig and no zig toolchain exists in the build image, so the compiled-language
// host layer above the C ABI is this header (the Zig binding itself is zig/root.zig, delivered as source).
// Same names and argument meaning as the reference; Zig error unions become zlz4::Result { value, error }.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "../../../include/zlz4_amd.h"

namespace zlz4 {

struct Result {
    std::size_t value = 0;   // bytes written when ok()
    std::int64_t error = 0;  // 0 or a ZLZ4_ERR_* / ZLZ4F_ERR_* code
    bool ok() const { return error == 0; }
    std::string error_name() const { return zlz4_error_name(error); }
};
inline Result wrap(std::int64_t r) { return r >= 0 ? Result{(std::size_t)r, 0} : Result{0, r}; }

constexpr int MINMATCH = ZLZ4_MINMATCH;
constexpr std::uint32_t LZ4_MAX_INPUT_SIZE = ZLZ4_MAX_INPUT_SIZE;
constexpr std::uint32_t LZ4_DISTANCE_MAX = ZLZ4_DISTANCE_MAX;
constexpr int LZ4HC_CLEVEL_MIN = ZLZ4HC_CLEVEL_MIN, LZ4HC_CLEVEL_DEFAULT = ZLZ4HC_CLEVEL_DEFAULT,
              LZ4HC_CLEVEL_MAX = ZLZ4HC_CLEVEL_MAX;

// lz4.compressBound, src/lz4.zig:80-83
inline std::size_t compressBound(std::size_t n) { return zlz4_compress_bound(n); }
// lz4.compressDefault, src/lz4.zig:283-285
inline Result compressDefault(const std::uint8_t *src, std::size_t n, std::uint8_t *dst, std::size_t cap) {
    return wrap(zlz4_compress_default(src, n, dst, cap));
}
// lz4.compressFast, src/lz4.zig:292-447
inline Result compressFast(const std::uint8_t *src, std::size_t n, std::uint8_t *dst, std::size_t cap, std::uint32_t accel) {
    return wrap(zlz4_compress_fast(src, n, dst, cap, accel));
}
// lz4.decompressSafe, src/lz4.zig:257-259
inline Result decompressSafe(const std::uint8_t *src, std::size_t n, std::uint8_t *dst, std::size_t cap) {
    return wrap(zlz4_decompress_safe(src, n, dst, cap));
}
// lz4.decompressSafePartial, src/lz4.zig:619-621
inline Result decompressSafePartial(const std::uint8_t *src, std::size_t n, std::uint8_t *dst, std::size_t cap, std::size_t target) {
    return wrap(zlz4_decompress_safe_partial(src, n, dst, cap, target));
}
// lz4.sizeofState / compressFastExtState / compressDestSize, src/lz4.zig:524-616
inline std::size_t sizeofState() { return zlz4_sizeof_state(); }
inline Result compressFastExtState(void *state, std::size_t state_len, const std::uint8_t *src, std::size_t n,
                                   std::uint8_t *dst, std::size_t cap, std::uint32_t accel) {
    return wrap(zlz4_compress_fast_ext_state(state, state_len, src, n, dst, cap, accel));
}
inline Result compressDestSize(const std::uint8_t *src, std::uint8_t *dst, std::size_t cap, std::size_t *src_size) {
    return wrap(zlz4_compress_dest_size(src, dst, cap, src_size));
}
// lz4hc.compressHC, src/lz4hc.zig:1440-1453
inline Result compressHC(const std::uint8_t *src, std::size_t n, std::uint8_t *dst, std::size_t cap, std::int32_t level) {
    return wrap(zlz4_compress_hc(src, n, dst, cap, level));
}

// lz4hc.sizeofStateHC / compressHCExtState, src/lz4hc.zig:1457-1494 (fresh context, passed as the bytes it occupies)
inline std::size_t sizeofStateHC() { return zlz4_sizeof_state_hc(); }
inline Result compressHCExtState(void *ctx, std::size_t ctx_len, const std::uint8_t *src, std::size_t n, std::uint8_t *dst,
                                 std::size_t cap, std::int32_t level) {
    return wrap(zlz4_compress_hc_ext_state(ctx, ctx_len, src, n, dst, cap, level));
}

// The hot path itself: many independent blocks per call, DEVICE pointers, asynchronous on `stream` (hipStream_t as
// void*).  A host-side slice-of-slices becomes the four descriptor arrays (they live in device memory too).
namespace device {
struct Blocks {
    const std::uint8_t *in; const std::uint64_t *in_off; const std::uint32_t *in_len;
    std::uint8_t *out; const std::uint64_t *out_off; const std::uint32_t *out_cap;
    std::int64_t *result; std::uint32_t nblocks;
};
inline Result compressFastBatch(void *stream, const Blocks &b, std::uint32_t max_in_len, std::uint32_t accel = 1) {
    return wrap(zlz4_batch_compress_fast(stream, b.in, b.in_off, b.in_len, b.out, b.out_off, b.out_cap, b.result, b.nblocks, max_in_len, accel));
}
inline Result decompressSafeBatch(void *stream, const Blocks &b) {
    return wrap(zlz4_batch_decompress_safe(stream, b.in, b.in_off, b.in_len, b.out, b.out_off, b.out_cap, b.result, b.nblocks));
}
inline std::size_t compressHCWorkspace(std::uint32_t nblocks, std::uint32_t max_in_len) {
    return zlz4_batch_compress_hc_workspace(nblocks, max_in_len);
}
inline Result compressHCBatch(void *stream, const Blocks &b, std::uint32_t max_in_len, std::int32_t level, void *ws, std::size_t ws_bytes) {
    return wrap(zlz4_batch_compress_hc(stream, b.in, b.in_off, b.in_len, b.out, b.out_off, b.out_cap, b.result, b.nblocks, max_in_len, level, ws, ws_bytes));
}
}  // namespace device

namespace lz4f {   // src/lz4f.zig
using Preferences = zlz4f_prefs;
constexpr std::uint32_t MAGICNUMBER = ZLZ4F_MAGICNUMBER;
inline std::size_t compressFrameBound(std::size_t n, const Preferences *p = nullptr) { return zlz4f_compress_frame_bound(n, p); }
inline Result compressFrame(const std::uint8_t *src, std::size_t n, std::uint8_t *dst, std::size_t cap, const Preferences *p = nullptr) {
    return wrap(zlz4f_compress_frame(src, n, dst, cap, p));
}
inline Result decompressFrame(const std::uint8_t *src, std::size_t n, std::uint8_t *dst, std::size_t cap) {
    return wrap(zlz4f_decompress_frame(src, n, dst, cap));
}
inline Result headerSize(const std::uint8_t *src, std::size_t n) { return wrap(zlz4f_header_size(src, n)); }
// device-resident frames and per-rank frame segments (BASELINE configs[4])
inline Result compressFrameDevice(void *stream, const std::uint8_t *d_src, std::size_t n, std::uint8_t *d_dst, std::size_t cap, const Preferences *p = nullptr) {
    return wrap(zlz4f_compress_frame_device(stream, d_src, n, d_dst, cap, p));
}
inline Result decompressFrameDevice(void *stream, const std::uint8_t *d_src, std::size_t n, std::uint8_t *d_dst, std::size_t cap) {
    return wrap(zlz4f_decompress_frame_device(stream, d_src, n, d_dst, cap));
}
inline Result compressFrameSegmentDevice(void *stream, const std::uint8_t *d_src, std::size_t n, std::uint8_t *d_dst, std::size_t cap, const Preferences *p, std::uint32_t seg) {
    return wrap(zlz4f_compress_frame_segment_device(stream, d_src, n, d_dst, cap, p, seg));
}
inline Result decompressFrameSegmentDevice(void *stream, const std::uint8_t *d_src, std::size_t n, std::uint8_t *d_dst, std::size_t cap, const Preferences *p, std::uint32_t seg) {
    return wrap(zlz4f_decompress_frame_segment_device(stream, d_src, n, d_dst, cap, p, seg));
}
}  // namespace lz4f

}  // namespace zlz4
