// zlz4_capi.hip -- the C ABI declared in include/zlz4_amd.h.
//
// Host side of the MI355X LZ4 codec: validates arguments the way the reference
// entry points do (src/lz4.zig, src/lz4hc.zig, src/lz4f.zig -- cited per function),
// stages host buffers, and enqueues the gfx950 kernels.  There is no CPU codec in
// here: without a usable HIP device every compute call returns ZLZ4_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/zlz4_amd.h"
#include "zlz4_host.hpp"

// kernel launchers (one per .hip file)
extern "C" int zlz4_launch_decompress_safe(hipStream_t, const uint8_t *, const uint64_t *, const uint32_t *, uint8_t *,
                                           const uint64_t *, const uint32_t *, int64_t *, uint32_t);
extern "C" int zlz4_launch_compress_fast(hipStream_t, const uint8_t *, const uint64_t *, const uint32_t *, uint8_t *,
                                         const uint64_t *, const uint32_t *, int64_t *, uint32_t, uint32_t, uint32_t);
extern "C" int zlz4_launch_compress_hc(hipStream_t, const uint8_t *, const uint64_t *, const uint32_t *, uint8_t *,
                                       const uint64_t *, const uint32_t *, int64_t *, uint32_t, uint32_t, int32_t,
                                       void *, size_t);
extern "C" size_t zlz4_hc_workspace_bytes(uint32_t nblocks, uint32_t max_in_len);

namespace {

// ---------------------------------------------------------------- device context
std::once_flag g_once;
int g_device_ok = ZLZ4_ERR_DEVICE;

void probe_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return;
    if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0) return;   // kernels are built for gfx950 only
    g_device_ok = 0;
}

bool device_ok() {
    std::call_once(g_once, probe_device);
    return g_device_ok == 0;
}

using zlz4host::DevBuf;
using zlz4host::DeviceCall;

enum class Op { Fast, Hc, Decompress };

// One block, host pointers: stage -> kernel -> copy back.
int64_t run_single(Op op, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap, uint32_t accel,
                   int32_t level) {
    if (!device_ok()) return ZLZ4_ERR_DEVICE;
    if (src_len > 0xFFFFFFFFull) return op == Op::Decompress ? ZLZ4_ERR_CORRUPTED_DATA : ZLZ4_ERR_INPUT_TOO_LARGE;
    // the kernels index with 32 bits; a destination larger than 4 GiB-1 is clamped (never reached:
    // compressBound(0x7E000000) and the largest decodable block both fit)
    const uint32_t cap32 = dst_cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)dst_cap;
    const uint32_t len32 = (uint32_t)src_len;

    // scratch from the parked-buffer cache (zlz4_host.hpp): a caller that loops over single blocks does not pay a
    // hipMalloc / hipFree pair per call
    hipStream_t st = nullptr;
    DeviceCall dc(st);
    const size_t ws = op == Op::Hc ? zlz4_hc_workspace_bytes(1, len32) : 0;
    DevBuf d_in(src_len, &dc), d_out(cap32, &dc), d_meta(64, &dc), d_ws(ws, &dc);
    if (!d_in.p || !d_out.p || !d_meta.p || !d_ws.p) return ZLZ4_ERR_ALLOCATION_FAILED;
    struct Meta { uint64_t in_off; uint64_t out_off; int64_t result; uint32_t in_len; uint32_t out_cap; } m;
    m.in_off = 0; m.out_off = 0; m.result = 0; m.in_len = len32; m.out_cap = cap32;
    dc.launched();
    if (src_len && hipMemcpyAsync(d_in.p, src, src_len, hipMemcpyHostToDevice, st) != hipSuccess) return ZLZ4_ERR_DEVICE;
    if (hipMemcpyAsync(d_meta.p, &m, sizeof m, hipMemcpyHostToDevice, st) != hipSuccess) return ZLZ4_ERR_DEVICE;
    auto *dm = d_meta.as<uint8_t>();
    const uint64_t *p_in_off = reinterpret_cast<const uint64_t *>(dm + offsetof(Meta, in_off));
    const uint64_t *p_out_off = reinterpret_cast<const uint64_t *>(dm + offsetof(Meta, out_off));
    int64_t *p_res = reinterpret_cast<int64_t *>(dm + offsetof(Meta, result));
    const uint32_t *p_in_len = reinterpret_cast<const uint32_t *>(dm + offsetof(Meta, in_len));
    const uint32_t *p_out_cap = reinterpret_cast<const uint32_t *>(dm + offsetof(Meta, out_cap));
    int rc;
    if (op == Op::Fast) {
        rc = zlz4_launch_compress_fast(st, d_in.as<uint8_t>(), p_in_off, p_in_len, d_out.as<uint8_t>(), p_out_off,
                                       p_out_cap, p_res, 1, len32, accel);
    } else if (op == Op::Hc) {
        rc = zlz4_launch_compress_hc(st, d_in.as<uint8_t>(), p_in_off, p_in_len, d_out.as<uint8_t>(), p_out_off,
                                     p_out_cap, p_res, 1, len32, level, d_ws.p, ws);
    } else {
        rc = zlz4_launch_decompress_safe(st, d_in.as<uint8_t>(), p_in_off, p_in_len, d_out.as<uint8_t>(), p_out_off,
                                         p_out_cap, p_res, 1);
    }
    if (rc != 0) return rc;
    int64_t result = 0;
    if (hipMemcpyAsync(&result, p_res, sizeof result, hipMemcpyDeviceToHost, st) != hipSuccess) return ZLZ4_ERR_DEVICE;
    if (!dc.sync()) return ZLZ4_ERR_DEVICE;
    if (result > 0) {
        if ((uint64_t)result > dst_cap) return ZLZ4_ERR_DEVICE;   // cannot happen; never overrun the caller
        if (hipMemcpy(dst, d_out.p, (size_t)result, hipMemcpyDeviceToHost) != hipSuccess) return ZLZ4_ERR_DEVICE;
    }
    return result;
}

// src/lz4hc.zig:1445 + :1464-1466 level normalisation, strategy table :72-86
int32_t normalise_hc_level(int32_t level) {
    if (level < ZLZ4HC_CLEVEL_MIN) level = ZLZ4HC_CLEVEL_DEFAULT;
    if (level > ZLZ4HC_CLEVEL_MAX) level = ZLZ4HC_CLEVEL_MAX;
    return level;
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------- introspection
int32_t zlz4_device_check(void) { return device_ok() ? 0 : ZLZ4_ERR_DEVICE; }

const char *zlz4_version_string(void) { return "zlz4-amd 0.1.0 (gfx950)"; }

const char *zlz4_error_name(int64_t code) {
    switch (code) {
        case ZLZ4_ERR_OUTPUT_TOO_SMALL: return "OutputTooSmall";
        case ZLZ4_ERR_INPUT_TOO_LARGE: return "InputTooLarge";
        case ZLZ4_ERR_CORRUPTED_DATA: return "CorruptedData";
        case ZLZ4_ERR_DECOMPRESSION_FAILED: return "DecompressionFailed";
        case ZLZ4_ERR_INVALID_STATE: return "InvalidState";
        case ZLZ4_ERR_ALLOCATION_FAILED: return "AllocationFailed";
        case ZLZ4_ERR_DEVICE: return "DeviceError";
        case ZLZ4_ERR_UNSUPPORTED: return "Unsupported";
        case ZLZ4_ERR_VERIFY: return "VerifyFailed";
        case ZLZ4F_ERR_GENERIC: return "Generic";
        case ZLZ4F_ERR_MAX_BLOCK_SIZE_INVALID: return "MaxBlockSizeInvalid";
        case ZLZ4F_ERR_BLOCK_MODE_INVALID: return "BlockModeInvalid";
        case ZLZ4F_ERR_PARAMETER_INVALID: return "ParameterInvalid";
        case ZLZ4F_ERR_COMPRESSION_LEVEL_INVALID: return "CompressionLevelInvalid";
        case ZLZ4F_ERR_HEADER_VERSION_WRONG: return "HeaderVersionWrong";
        case ZLZ4F_ERR_BLOCK_CHECKSUM_INVALID: return "BlockChecksumInvalid";
        case ZLZ4F_ERR_RESERVED_FLAG_SET: return "ReservedFlagSet";
        case ZLZ4F_ERR_ALLOCATION_FAILED: return "AllocationFailed";
        case ZLZ4F_ERR_SRC_SIZE_TOO_LARGE: return "SrcSizeTooLarge";
        case ZLZ4F_ERR_DST_MAX_SIZE_TOO_SMALL: return "DstMaxSizeTooSmall";
        case ZLZ4F_ERR_FRAME_HEADER_INCOMPLETE: return "FrameHeaderIncomplete";
        case ZLZ4F_ERR_FRAME_TYPE_UNKNOWN: return "FrameTypeUnknown";
        case ZLZ4F_ERR_FRAME_SIZE_WRONG: return "FrameSizeWrong";
        case ZLZ4F_ERR_SRC_PTR_WRONG: return "SrcPtrWrong";
        case ZLZ4F_ERR_DECOMPRESSION_FAILED: return "DecompressionFailed";
        case ZLZ4F_ERR_HEADER_CHECKSUM_INVALID: return "HeaderChecksumInvalid";
        case ZLZ4F_ERR_CONTENT_CHECKSUM_INVALID: return "ContentChecksumInvalid";
        default: return code >= 0 ? "ok" : "unknown";
    }
}

// ---------------------------------------------------------------- single buffer, host pointers
size_t zlz4_compress_bound(size_t n) {                      // src/lz4.zig:80-83
    if (n > ZLZ4_MAX_INPUT_SIZE) return 0;
    return n + (n / 255) + 16;
}

int64_t zlz4_compress_fast(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, uint32_t accel) {
    if (n > ZLZ4_MAX_INPUT_SIZE) return ZLZ4_ERR_INPUT_TOO_LARGE;   // src/lz4.zig:296
    if (n == 0) return 0;                                           // :299
    return run_single(Op::Fast, src, n, dst, cap, accel, 0);
}

int64_t zlz4_compress_default(const uint8_t *src, size_t n, uint8_t *dst, size_t cap) {   // src/lz4.zig:283-285
    return zlz4_compress_fast(src, n, dst, cap, 1);
}

int64_t zlz4_compress_hc(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, int32_t level) {
    if (n > ZLZ4_MAX_INPUT_SIZE) return ZLZ4_ERR_INPUT_TOO_LARGE;   // src/lz4hc.zig:1442
    if (n == 0) return 0;                                           // :1443
    if (cap == 0) return ZLZ4_ERR_OUTPUT_TOO_SMALL;                 // :1461
    level = normalise_hc_level(level);                              // 2 lz4mid, 3-9 lz4hc, 10-12 lz4opt (:72-86)
    return run_single(Op::Hc, src, n, dst, cap, 0, level);
}

int64_t zlz4_decompress_safe(const uint8_t *src, size_t n, uint8_t *dst, size_t cap) {
    if (n == 0) return 0;                                           // src/lz4.zig:97
    if (cap == 0) return 0;                                         // :98
    return run_single(Op::Decompress, src, n, dst, cap, 0, 0);
}

int64_t zlz4_decompress_safe_partial(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t target) {
    if (n == 0) return 0;                                           // src/lz4.zig:97
    if (cap == 0) return 0;                                         // :98
    if (target > cap) return ZLZ4_ERR_OUTPUT_TOO_SMALL;             // :99
    if (target > 0) return run_single(Op::Decompress, src, n, dst, target, 0, 0);   // oend = targetOutputSize (:109)
    // target == 0 with a non-empty dst: no byte can be produced, the result is decided by the first sequence
    // header alone (:111-174) -- nothing to run on the device
    size_t ip = 0;
    const uint8_t token = src[ip++];
    size_t lit = token >> 4;
    if (lit == 15) {
        for (;;) {
            if (ip >= n) return ZLZ4_ERR_CORRUPTED_DATA;            // :125
            const uint8_t s = src[ip++];
            lit += s;
            if (s != 255) break;
        }
    }
    if (lit > 0) {
        if (ip + lit > n) return ZLZ4_ERR_CORRUPTED_DATA;           // :136
        return ZLZ4_ERR_OUTPUT_TOO_SMALL;                           // :137 (op + lit > 0)
    }
    if (ip >= n) return 0;                                          // :146
    if (ip + 2 > n) return ZLZ4_ERR_CORRUPTED_DATA;                 // :149
    if ((src[ip] | (src[ip + 1] << 8)) == 0) return ZLZ4_ERR_CORRUPTED_DATA;   // :154
    ip += 2;
    if ((token & 15) == 15) {
        for (;;) {
            if (ip >= n) return ZLZ4_ERR_CORRUPTED_DATA;            // :162
            if (src[ip++] != 255) break;
        }
    }
    return ZLZ4_ERR_OUTPUT_TOO_SMALL;                               // :174 (op + matchLength > 0)
}

size_t zlz4_sizeof_state(void) { return 4096 * sizeof(uint32_t); }  // src/lz4.zig:524-526, :263-265

// src/lz4hc.zig:1492-1494: @sizeOf(Context) -- hashTable 32768 x u32 + chainTable 65536 x u16 (:391-393) + the scalars
size_t zlz4_sizeof_state_hc(void) { return 32768u * 4u + 65536u * 2u + 3u * 8u + 3u * 4u + 2u + 1u + 1u + 8u; }

// src/lz4hc.zig:1457-1489.  Differences from compressHC: no `< 2 -> 9` clamp (level < 1 -> 9, level 1 takes the
// table's row 1 = lz4mid, :72-97), dst.len == 0 is checked here (:1461).
int64_t zlz4_compress_hc_ext_state(void *state, size_t state_len, const uint8_t *src, size_t n, uint8_t *dst, size_t cap,
                                   int32_t level) {
    if (!state || state_len < zlz4_sizeof_state_hc()) return ZLZ4_ERR_INVALID_STATE;
    if (n > ZLZ4_MAX_INPUT_SIZE) return ZLZ4_ERR_INPUT_TOO_LARGE;   // :1459
    if (n == 0) return 0;                                           // :1460
    if (cap == 0) return ZLZ4_ERR_OUTPUT_TOO_SMALL;                 // :1461
    if (level < 1) level = ZLZ4HC_CLEVEL_DEFAULT;                   // :1464-1466
    if (level > ZLZ4HC_CLEVEL_MAX) level = ZLZ4HC_CLEVEL_MAX;
    if (level == 1) level = 2;                                      // clevelTable[1] == clevelTable[2] (:73-75)
    return run_single(Op::Hc, src, n, dst, cap, 0, level);
}

int64_t zlz4_compress_fast_ext_state(void *state, size_t state_len, const uint8_t *src, size_t n, uint8_t *dst,
                                     size_t cap, uint32_t accel) {
    (void)state;
    if (state_len < zlz4_sizeof_state()) return ZLZ4_ERR_INVALID_STATE;   // src/lz4.zig:532
    return zlz4_compress_fast(src, n, dst, cap, accel);                  // :534-545
}

int64_t zlz4_compress_dest_size(const uint8_t *src, uint8_t *dst, size_t cap, size_t *src_size) {
    const size_t max_src = *src_size;
    if (max_src == 0) { *src_size = 0; return 0; }                  // src/lz4.zig:553-556
    if (cap >= zlz4_compress_bound(max_src)) {                      // :559-564
        const int64_t r = zlz4_compress_default(src, max_src, dst, cap);
        if (r < 0) return r;
        *src_size = max_src;
        return r;
    }
    if (!device_ok()) return ZLZ4_ERR_DEVICE;
    if (max_src > ZLZ4_MAX_INPUT_SIZE) {
        // probes larger than the input limit fail with InputTooLarge in the reference (:594-607 -> `high = mid - 1`);
        // the search below never needs more than this many source bytes on the device
    }
    const size_t stage = max_src > ZLZ4_MAX_INPUT_SIZE ? (size_t)ZLZ4_MAX_INPUT_SIZE : max_src;
    const uint32_t cap32 = cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap;
    DevBuf d_in(stage), d_out(cap32), d_meta(64);
    if (!d_in.p || !d_out.p || !d_meta.p) return ZLZ4_ERR_ALLOCATION_FAILED;
    if (hipMemcpy(d_in.p, src, stage, hipMemcpyHostToDevice) != hipSuccess) return ZLZ4_ERR_DEVICE;
    struct Meta { uint64_t in_off; uint64_t out_off; int64_t result; uint32_t in_len; uint32_t out_cap; };
    auto *dm = d_meta.as<uint8_t>();
    // one probe = compressDefault(src[0..len], dst) on the device; the input stays resident
    auto probe = [&](size_t len) -> int64_t {
        if (len > ZLZ4_MAX_INPUT_SIZE) return ZLZ4_ERR_INPUT_TOO_LARGE;        // src/lz4.zig:296
        if (len == 0) return 0;                                                // :299
        Meta m; m.in_off = 0; m.out_off = 0; m.result = 0; m.in_len = (uint32_t)len; m.out_cap = cap32;
        if (hipMemcpy(d_meta.p, &m, sizeof m, hipMemcpyHostToDevice) != hipSuccess) return ZLZ4_ERR_DEVICE;
        const int rc = zlz4_launch_compress_fast(nullptr, d_in.as<uint8_t>(),
                                                 reinterpret_cast<const uint64_t *>(dm + offsetof(Meta, in_off)),
                                                 reinterpret_cast<const uint32_t *>(dm + offsetof(Meta, in_len)),
                                                 d_out.as<uint8_t>(),
                                                 reinterpret_cast<const uint64_t *>(dm + offsetof(Meta, out_off)),
                                                 reinterpret_cast<const uint32_t *>(dm + offsetof(Meta, out_cap)),
                                                 reinterpret_cast<int64_t *>(dm + offsetof(Meta, result)), 1, (uint32_t)len, 1);
        if (rc != 0) return rc;
        int64_t r = 0;
        if (hipMemcpy(&r, dm + offsetof(Meta, result), sizeof r, hipMemcpyDeviceToHost) != hipSuccess) return ZLZ4_ERR_DEVICE;
        return r;
    };
    size_t low = 1, high = max_src, best = 0, best_c = 0, last_ok_len = (size_t)-1;     // :567-570
    auto attempt = [&](size_t len, bool &fits) -> int64_t {
        const int64_t r = probe(len);
        fits = r >= 0 && (size_t)r <= cap;
        last_ok_len = r >= 0 ? len : (size_t)-1;      // a failed probe leaves a partial stream in d_out
        return r;
    };
    if (cap <= max_src) {                                           // :573-586
        const size_t estimate = cap < max_src ? cap : max_src;
        bool fits;
        const int64_t r = attempt(estimate, fits);
        if (r == ZLZ4_ERR_DEVICE) return r;
        if (fits) { best = estimate; best_c = (size_t)r; low = estimate + 1; }
        else high = estimate - 1;
    }
    while (low <= high) {                                           // :589-612
        const size_t mid = low + (high - low) / 2;
        if (mid == 0 || mid > max_src) break;
        bool fits;
        const int64_t r = attempt(mid, fits);
        if (r == ZLZ4_ERR_DEVICE) return r;
        if (fits) {
            best = mid; best_c = (size_t)r;
            if (mid == max_src) break;
            low = mid + 1;
        } else {
            high = mid - 1;
        }
        if (low > max_src) break;
    }
    if (best > 0) {
        if (last_ok_len != best) { bool f; if (attempt(best, f) < 0) return ZLZ4_ERR_DEVICE; }   // put the best one into d_out
        if (hipMemcpy(dst, d_out.p, best_c, hipMemcpyDeviceToHost) != hipSuccess) return ZLZ4_ERR_DEVICE;
    }
    *src_size = best;                                               // :614-615
    return (int64_t)best_c;
}

// ---------------------------------------------------------------- batch, device pointers
int32_t zlz4_batch_compress_fast(void *stream, const uint8_t *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
                                 uint8_t *d_out, const uint64_t *d_out_off, const uint32_t *d_out_cap,
                                 int64_t *d_result, uint32_t nblocks, uint32_t max_in_len, uint32_t acceleration) {
    if (!device_ok()) return ZLZ4_ERR_DEVICE;
    return zlz4_launch_compress_fast((hipStream_t)stream, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap,
                                     d_result, nblocks, max_in_len, acceleration);
}

int32_t zlz4_batch_decompress_safe(void *stream, const uint8_t *d_in, const uint64_t *d_in_off,
                                   const uint32_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off,
                                   const uint32_t *d_out_cap, int64_t *d_result, uint32_t nblocks) {
    if (!device_ok()) return ZLZ4_ERR_DEVICE;
    return zlz4_launch_decompress_safe((hipStream_t)stream, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap,
                                       d_result, nblocks);
}

size_t zlz4_batch_compress_hc_workspace(uint32_t nblocks, uint32_t max_in_len) {
    return zlz4_hc_workspace_bytes(nblocks, max_in_len);
}

int32_t zlz4_batch_compress_hc(void *stream, const uint8_t *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
                               uint8_t *d_out, const uint64_t *d_out_off, const uint32_t *d_out_cap,
                               int64_t *d_result, uint32_t nblocks, uint32_t max_in_len, int32_t level,
                               void *d_workspace, size_t workspace_bytes) {
    if (!device_ok()) return ZLZ4_ERR_DEVICE;
    level = normalise_hc_level(level);
    if (workspace_bytes < zlz4_hc_workspace_bytes(nblocks, max_in_len) || (!d_workspace && nblocks))
        return ZLZ4_ERR_INVALID_STATE;
    return zlz4_launch_compress_hc((hipStream_t)stream, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap,
                                   d_result, nblocks, max_in_len, level, d_workspace, workspace_bytes);
}

// ---------------------------------------------------------------- opt-in round-trip check (levels 10..12)
}  // extern "C"

namespace {
// decode descriptors from the compress results: length 0 for a block that reported an error, capacity = the input length
__global__ void k_verify_prep(const int64_t *__restrict__ comp_result, const uint32_t *__restrict__ in_len,
                              const uint64_t *__restrict__ in_off, uint64_t base_off, uint32_t *__restrict__ clen,
                              uint64_t *__restrict__ dec_off, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = comp_result[i];
    clen[i] = r > 0 ? (uint32_t)r : 0u;
    dec_off[i] = in_off[i] - base_off;
    (void)in_len;
}
// one wavefront per block: decoded size and bytes against the input
__global__ __launch_bounds__(256) void k_verify_compare(const uint8_t *__restrict__ d_in, const uint64_t *__restrict__ in_off,
                                                        const uint32_t *__restrict__ in_len, const uint8_t *__restrict__ d_dec,
                                                        const uint64_t *__restrict__ dec_off, const int64_t *__restrict__ dec_result,
                                                        const int64_t *__restrict__ comp_result, int64_t *__restrict__ verify,
                                                        unsigned long long *__restrict__ nbad, uint32_t n) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t cr = comp_result[i];
    int64_t out = cr;
    if (cr >= 0) {
        const uint32_t len = in_len[i];
        bool bad = dec_result[i] != (int64_t)len && !(len == 0 && cr == 0);
        if (!bad) {
            const uint8_t *a = d_in + in_off[i], *b = d_dec + dec_off[i];
            bool diff = false;
            for (uint32_t k = lane; k < len; k += 64u) diff |= a[k] != b[k];
            bad = __ballot(diff) != 0;
        }
        if (bad) out = ZLZ4_ERR_VERIFY;
    }
    if (lane == 0) {
        verify[i] = out;
        if (out == ZLZ4_ERR_VERIFY) atomicAdd(nbad, 1ull);
    }
}
}  // namespace

extern "C" int64_t zlz4_batch_verify(void *stream, const uint8_t *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
                                     const uint8_t *d_comp, const uint64_t *d_comp_off, const int64_t *d_comp_result,
                                     int64_t *d_verify, uint32_t nblocks) {
    if (!device_ok()) return ZLZ4_ERR_DEVICE;
    if (nblocks == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    // the extent of the input arena (descriptors come back to the host: this call synchronises anyway)
    std::vector<uint64_t> off(nblocks);
    std::vector<uint32_t> len(nblocks);
    if (hipMemcpyAsync(off.data(), d_in_off, nblocks * sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(len.data(), d_in_len, nblocks * sizeof(uint32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) return ZLZ4_ERR_DEVICE;
    uint64_t lo = ~0ull, hi = 0;
    for (uint32_t i = 0; i < nblocks; i++) {
        if (off[i] < lo) lo = off[i];
        if (off[i] + len[i] > hi) hi = off[i] + len[i];
    }
    DeviceCall call(st);
    DevBuf d_dec((size_t)(hi - lo) + 64, &call), d_clen((size_t)nblocks * 4, &call), d_doff((size_t)nblocks * 8, &call),
        d_dres((size_t)nblocks * 8, &call), d_nbad(8, &call);
    if (!d_dec.p || !d_clen.p || !d_doff.p || !d_dres.p || !d_nbad.p) return ZLZ4_ERR_ALLOCATION_FAILED;
    call.launched();
    if (hipMemsetAsync(d_nbad.p, 0, 8, st) != hipSuccess) return ZLZ4_ERR_DEVICE;
    hipLaunchKernelGGL(k_verify_prep, dim3((nblocks + 255) / 256), dim3(256), 0, st, d_comp_result, d_in_len, d_in_off, lo,
                       d_clen.as<uint32_t>(), d_doff.as<uint64_t>(), nblocks);
    const int rc = zlz4_launch_decompress_safe(st, d_comp, d_comp_off, d_clen.as<uint32_t>(), d_dec.as<uint8_t>(),
                                               d_doff.as<uint64_t>(), d_in_len, d_dres.as<int64_t>(), nblocks);
    if (rc != 0) return rc;
    hipLaunchKernelGGL(k_verify_compare, dim3((nblocks + 3) / 4), dim3(256), 0, st, d_in, d_in_off, d_in_len, d_dec.as<uint8_t>(),
                       d_doff.as<uint64_t>(), d_dres.as<int64_t>(), d_comp_result, d_verify,
                       d_nbad.as<unsigned long long>(), nblocks);
    unsigned long long nbad = 0;
    if (hipMemcpyAsync(&nbad, d_nbad.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || !call.sync() ||
        hipGetLastError() != hipSuccess) return ZLZ4_ERR_DEVICE;
    return (int64_t)nbad;
}
