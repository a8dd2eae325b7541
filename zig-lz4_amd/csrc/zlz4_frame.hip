// zlz4_frame.hip -- LZ4 frame container (reference src/lz4f.zig) around the block kernels.
//
// compressFrame (src/lz4f.zig:354-446): the block loop carries no state from one block to the
// next, so every block is compressed by the batch kernel into a compressBound-sized slot; a
// one-wave plan kernel then does what the serial loop does with `dstPos` (stored-block decision
// :407-417, block header :418, optional block checksum :422-427) as a prefix sum, and a scatter
// kernel moves header + payload (+ checksum) of every block to its final place.  Frame header
// (:304-351), end mark (:433) and content checksum (:437-441) are written by a one-lane kernel.
//
// decompressFrame (src/lz4f.zig:541-638): the block chain is walked by one lane (each block header
// tells where the next one is), a size pass of the block decoder gives every block's decompressed
// size, a plan pass reproduces the serial `dstPos` accumulation including the order in which the
// reference would have hit an error, then all blocks are decoded / copied in parallel.
//
// XXH32 (std.hash.XxHash32 in the reference, standard XXH32 seed 0) is a strictly serial hash: one
// lane per block for block checksums, one lane for the content checksum (both are off by default,
// src/lz4f.zig:109,113).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/zlz4_amd.h"
#include <mutex>
#include <vector>

#include "zlz4_device.hpp"
#include "zlz4_host.hpp"

extern "C" int zlz4_launch_decompress_safe(hipStream_t, const uint8_t *, const uint64_t *, const uint32_t *, uint8_t *,
                                           const uint64_t *, const uint32_t *, int64_t *, uint32_t);
extern "C" int zlz4_launch_decompress_sizes(hipStream_t, const uint8_t *, const uint64_t *, const uint32_t *,
                                            const uint64_t *, const uint32_t *, int64_t *, uint32_t);
extern "C" int zlz4_launch_compress_fast(hipStream_t, const uint8_t *, const uint64_t *, const uint32_t *, uint8_t *,
                                         const uint64_t *, const uint32_t *, int64_t *, uint32_t, uint32_t, uint32_t);
extern "C" int zlz4_launch_compress_hc(hipStream_t, const uint8_t *, const uint64_t *, const uint32_t *, uint8_t *,
                                       const uint64_t *, const uint32_t *, int64_t *, uint32_t, uint32_t, int32_t,
                                       void *, size_t);
extern "C" size_t zlz4_hc_workspace_bytes(uint32_t nblocks, uint32_t max_in_len);

// ------------------------------------------------------------------ device buffers (zlz4_host.hpp)
namespace {
struct ParkedBuf { void *p; size_t n; int dev; };
std::mutex g_park_mutex;
std::vector<ParkedBuf> g_parked;
size_t g_parked_bytes = 0;
constexpr size_t kMaxParked = 12;
constexpr size_t kMaxParkedBytes = 8ull << 30;   // ~3 % of the HBM: one configs[4] slot arena (4 GiB) and its tables
}  // namespace

namespace zlz4host {
void *cache_take(size_t &n, int dev) {
    std::lock_guard<std::mutex> lock(g_park_mutex);
    size_t best = g_parked.size();
    for (size_t i = 0; i < g_parked.size(); i++)     // smallest parked buffer that fits and is not wastefully large
        if (g_parked[i].dev == dev && g_parked[i].n >= n && g_parked[i].n / 2 <= n + (1u << 20) &&
            (best == g_parked.size() || g_parked[i].n < g_parked[best].n))
            best = i;
    if (best == g_parked.size()) return nullptr;
    void *p = g_parked[best].p;
    n = g_parked[best].n;
    g_parked_bytes -= n;
    g_parked.erase(g_parked.begin() + (long)best);
    return p;
}
bool cache_give(void *p, size_t n, int dev) {
    std::lock_guard<std::mutex> lock(g_park_mutex);
    if (g_parked.size() >= kMaxParked || g_parked_bytes + n > kMaxParkedBytes) return false;
    g_parked.push_back({p, n, dev});
    g_parked_bytes += n;
    return true;
}
}  // namespace zlz4host
using zlz4host::DevBuf;
typedef zlz4host::DeviceCall FrameCall;

namespace {

const zlz4f_prefs kDefaultPrefs = {0, 0, 0, 0, 0, 0, 0};   // src/lz4f.zig:106-122 defaults

// BlockSizeID.toBlockSize, src/lz4f.zig:71-78
size_t block_size_of(uint32_t id) {
    switch (id) {
        case 5: return 256u * 1024;
        case 6: return 1024u * 1024;
        case 7: return 4u * 1024 * 1024;
        default: return 64u * 1024;
    }
}

// ------------------------------------------------------------------ XXH32 (host + device)
#define ZX_P1 2654435761u
#define ZX_P2 2246822519u
#define ZX_P3 3266489917u
#define ZX_P4 668265263u
#define ZX_P5 374761393u

__host__ __device__ inline uint32_t zx_rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__host__ __device__ inline uint32_t zx_rd32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
__host__ __device__ inline uint32_t zx_round(uint32_t acc, uint32_t in) { return zx_rotl(acc + in * ZX_P2, 13) * ZX_P1; }

__host__ __device__ inline uint32_t xxh32(const uint8_t *p, uint64_t len, uint32_t seed) {
    const uint8_t *const end = p + len;
    uint32_t h;
    if (len >= 16) {
        uint32_t v1 = seed + ZX_P1 + ZX_P2, v2 = seed + ZX_P2, v3 = seed, v4 = seed - ZX_P1;
        const uint8_t *const limit = end - 16;
        do {
            v1 = zx_round(v1, zx_rd32(p)); v2 = zx_round(v2, zx_rd32(p + 4));
            v3 = zx_round(v3, zx_rd32(p + 8)); v4 = zx_round(v4, zx_rd32(p + 12));
            p += 16;
        } while (p <= limit);
        h = zx_rotl(v1, 1) + zx_rotl(v2, 7) + zx_rotl(v3, 12) + zx_rotl(v4, 18);
    } else {
        h = seed + ZX_P5;
    }
    h += (uint32_t)len;
    while (p + 4 <= end) { h = zx_rotl(h + zx_rd32(p) * ZX_P3, 17) * ZX_P4; p += 4; }
    while (p < end) { h = zx_rotl(h + (*p) * ZX_P5, 11) * ZX_P1; p += 1; }
    h ^= h >> 15; h *= ZX_P2; h ^= h >> 13; h *= ZX_P3; h ^= h >> 16;
    return h;
}

// ------------------------------------------------------------------ frame header (host side, <= 19 bytes)
struct HeaderBytes { uint8_t b[20]; uint32_t n; };

// writeFrameHeader, src/lz4f.zig:304-351 (encodeFLG :152-184, encodeBD :224-232, headerChecksum :138-141)
HeaderBytes encode_header(const zlz4f_prefs &p) {
    HeaderBytes h;
    std::memset(&h, 0, sizeof h);
    uint32_t pos = 0;
    const uint32_t magic = ZLZ4F_MAGICNUMBER;
    std::memcpy(h.b, &magic, 4); pos = 4;
    uint8_t flg = 0x40;
    if (p.block_mode == 1) flg |= 0x20;
    if (p.block_checksum == 1) flg |= 0x10;
    if (p.content_size != 0) flg |= 0x08;
    if (p.content_checksum == 1) flg |= 0x04;
    if (p.dict_id != 0) flg |= 0x01;
    h.b[pos++] = flg;
    uint8_t bd = 4;
    if (p.block_size_id == 5) bd = 5; else if (p.block_size_id == 6) bd = 6; else if (p.block_size_id == 7) bd = 7;
    h.b[pos++] = (uint8_t)(bd << 4);
    if (p.content_size != 0) { std::memcpy(h.b + pos, &p.content_size, 8); pos += 8; }
    if (p.dict_id != 0) { std::memcpy(h.b + pos, &p.dict_id, 4); pos += 4; }
    h.b[pos] = (uint8_t)((xxh32(h.b + 4, pos - 4, 0) >> 8) & 0xFF);
    pos += 1;
    h.n = pos;
    return h;
}

struct ParsedHeader { int64_t size; uint8_t flg; size_t block_size; };

// parseFrameHeader, src/lz4f.zig:483-538 (decodeFLG :187-221, decodeBD :235-249); `have` = bytes available
__host__ __device__ inline ParsedHeader parse_header(const uint8_t *src, size_t have) {
    ParsedHeader r = {0, 0, 0};
    if (have < 7) { r.size = ZLZ4F_ERR_FRAME_HEADER_INCOMPLETE; return r; }
    const uint32_t magic = zx_rd32(src);
    if (magic != ZLZ4F_MAGICNUMBER) { r.size = ZLZ4F_ERR_FRAME_TYPE_UNKNOWN; return r; }
    size_t pos = 4;
    const uint8_t flg = src[pos];
    if (((flg >> 6) & 3) != 1) { r.size = ZLZ4F_ERR_HEADER_VERSION_WRONG; return r; }
    if (flg & 0x02) { r.size = ZLZ4F_ERR_RESERVED_FLAG_SET; return r; }
    pos += 1;
    const uint8_t bd = src[pos];
    if (bd & 0x8F) { r.size = ZLZ4F_ERR_RESERVED_FLAG_SET; return r; }
    switch ((bd >> 4) & 7) {
        case 0: case 4: r.block_size = 64u * 1024; break;
        case 5: r.block_size = 256u * 1024; break;
        case 6: r.block_size = 1024u * 1024; break;
        case 7: r.block_size = 4u * 1024 * 1024; break;
        default: r.size = ZLZ4F_ERR_MAX_BLOCK_SIZE_INVALID; return r;
    }
    pos += 1;
    if (flg & 0x08) { if (have < pos + 8) { r.size = ZLZ4F_ERR_FRAME_HEADER_INCOMPLETE; return r; } pos += 8; }
    if (flg & 0x01) { if (have < pos + 4) { r.size = ZLZ4F_ERR_FRAME_HEADER_INCOMPLETE; return r; } pos += 4; }
    if (have < pos + 1) { r.size = ZLZ4F_ERR_FRAME_HEADER_INCOMPLETE; return r; }
    if (src[pos] != (uint8_t)((xxh32(src + 4, pos - 4, 0) >> 8) & 0xFF)) { r.size = ZLZ4F_ERR_HEADER_CHECKSUM_INVALID; return r; }
    pos += 1;
    r.size = (int64_t)pos;
    r.flg = flg;
    return r;
}

extern "C" void zlz4_release_device_cache(void) {
    std::vector<ParkedBuf> take;
    {
        std::lock_guard<std::mutex> lock(g_park_mutex);
        take.swap(g_parked);
        g_parked_bytes = 0;
    }
    for (const ParkedBuf &b : take) (void)hipFree(b.p);
}

// ------------------------------------------------------------------ compress-side kernels
// block descriptors for the batch kernels: block i = src[i*bs, min(n, (i+1)*bs)) -> slot i
__global__ void k_frame_desc(uint64_t n, uint64_t bs, uint64_t slot, uint32_t nblocks, uint64_t *in_off,
                             uint32_t *in_len, uint64_t *out_off, uint32_t *out_cap) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nblocks; i += gridDim.x * blockDim.x) {
        const uint64_t o = (uint64_t)i * bs;
        in_off[i] = o;
        in_len[i] = (uint32_t)((n - o) < bs ? (n - o) : bs);
        out_off[i] = (uint64_t)i * slot;
        out_cap[i] = (uint32_t)slot;
    }
}

// plan[0] = total frame bytes, plan[1] = first failing block + 1 (0 = none), plan[2] = its error code
// One wavefront; 64 blocks per step with a shuffle prefix sum (the serial dstPos of src/lz4f.zig:379-430).
__global__ __launch_bounds__(64) void k_frame_plan(const int64_t *__restrict__ csize, const uint32_t *__restrict__ in_len,
                                                   uint32_t nblocks, uint32_t block_checksum, uint64_t start,
                                                   uint64_t *__restrict__ dst_off, uint32_t *__restrict__ hdr,
                                                   int64_t *__restrict__ plan) {
    const uint32_t lane = threadIdx.x;
    uint64_t pos = start;
    uint32_t bad = 0;
    int64_t bad_code = 0;
    for (uint32_t base = 0; base < nblocks; base += 64u) {
        const uint32_t i = base + lane;
        uint64_t bytes = 0;
        uint32_t h = 0;
        bool err = false;
        int64_t c = 0;
        if (i < nblocks) {
            c = csize[i];
            const uint32_t len = in_len[i];
            err = c < 0;
            const bool stored = !err && (uint64_t)c >= len;           // :407
            const uint32_t actual = stored ? len : (uint32_t)(err ? 0 : c);
            h = actual | (stored ? 0x80000000u : 0u);                 // :411-414
            bytes = 4u + (uint64_t)actual + (block_checksum ? 4u : 0u);
        }
        const uint64_t em = zlz4::ballot(err);
        if (em && !bad) {
            const uint32_t l = zlz4::first_lane(em);
            bad = base + l + 1u;
            bad_code = (int64_t)(int32_t)zlz4::rdlane((uint32_t)c, l);   // error codes are small negatives
        }
        uint64_t incl = bytes;                                        // inclusive scan over the 64 lanes
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const uint32_t lo = __shfl_up((uint32_t)incl, d), hi = __shfl_up((uint32_t)(incl >> 32), d);
            if (lane >= d) incl += ((uint64_t)hi << 32) | lo;
        }
        if (i < nblocks) { dst_off[i] = pos + incl - bytes; hdr[i] = h; }
        const uint32_t tlo = zlz4::rdlane((uint32_t)incl, 63), thi = zlz4::rdlane((uint32_t)(incl >> 32), 63);
        pos += ((uint64_t)thi << 32) | tlo;
    }
    if (lane == 0) { plan[0] = (int64_t)pos; plan[1] = bad; plan[2] = bad_code; }
}

// one lane per block: XXH32 of the bytes that will be stored for the block (:422-427)
__global__ void k_block_xxh32(const uint8_t *__restrict__ src, const uint64_t *__restrict__ src_off,
                              const uint8_t *__restrict__ slots, const uint64_t *__restrict__ slot_off,
                              const uint32_t *__restrict__ hdr, uint32_t nblocks, uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblocks) return;
    const uint32_t h = hdr[i];
    const uint8_t *p = (h & 0x80000000u) ? src + src_off[i] : slots + slot_off[i];
    out[i] = xxh32(p, h & 0x7FFFFFFFu, 0);
}

// one workgroup per block: header word, payload (compressed slot or raw source), optional checksum
__global__ __launch_bounds__(256) void k_frame_scatter(const uint8_t *__restrict__ src, const uint64_t *__restrict__ src_off,
                                                        const uint8_t *__restrict__ slots,
                                                        const uint64_t *__restrict__ slot_off,
                                                        const uint32_t *__restrict__ hdr,
                                                        const uint64_t *__restrict__ dst_off,
                                                        const uint32_t *__restrict__ cks, uint32_t block_checksum,
                                                        uint8_t *__restrict__ dst) {
    const uint32_t i = blockIdx.x, t = threadIdx.x;
    const uint32_t h = hdr[i];
    const uint32_t n = h & 0x7FFFFFFFu;
    const uint8_t *p = (h & 0x80000000u) ? src + src_off[i] : slots + slot_off[i];
    uint8_t *o = dst + dst_off[i];
    if (t < 4) o[t] = (uint8_t)(h >> (8u * t));                                   // :418
    o += 4;
    for (uint32_t k = t * 16u; k + 16u <= n; k += 256u * 16u) zlz4::st128(o + k, zlz4::ld128(p + k));
    const uint32_t t0 = n & ~15u;
    if (t < 16u && t0 + t < n) o[t0 + t] = p[t0 + t];
    if (block_checksum && t < 4) o[n + t] = (uint8_t)(cks[i] >> (8u * t));        // :425
}

// one lane: frame header at dst[0..) (hb.n = 0: none), then -- for the segment that ends the frame -- the end mark and the
// optional content checksum at dst[plan[0]..)
__global__ void k_frame_head_tail(HeaderBytes hb, uint8_t *dst, const int64_t *plan, const uint8_t *src, uint64_t n,
                                  uint32_t tail, uint32_t content_checksum, int64_t *total_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (uint32_t k = 0; k < hb.n; k++) dst[k] = hb.b[k];
    uint64_t pos = (uint64_t)plan[0];
    if (tail) {
        for (int k = 0; k < 4; k++) dst[pos + k] = 0;                             // :433
        pos += 4;
        if (content_checksum) {
            const uint32_t c = xxh32(src, n, 0);                                  // :384-386, :437-441
            for (int k = 0; k < 4; k++) dst[pos + k] = (uint8_t)(c >> (8 * k));
            pos += 4;
        }
    }
    *total_out = (int64_t)pos;
}

// ------------------------------------------------------------------ decompress-side kernels
// walk[0] = number of data blocks, walk[1] = srcPos after the walk, walk[2] = pending error code (0 = none),
// walk[3] = header size or header error, walk[4] = FLG byte, walk[5] = block size
// The walk of src/lz4f.zig:563-600 without the payload work: block k's header position depends on all earlier block
// sizes, so this is a serial chain of 4-byte reads (one lane).  The frame header is parsed here too (:547), so that
// the host needs one read-back for header, block count and walk status.  Blocks beyond `max_blocks` are counted, not
// recorded (the caller then repeats the walk with a larger table).
// seg: bit 0 = the bytes start with a frame header; otherwise flg_in / bs_in describe the frame (a later rank's segment)
__global__ void k_frame_walk(const uint8_t *__restrict__ src, uint64_t src_len, uint32_t seg, uint32_t flg_in, uint64_t bs_in,
                             uint64_t *__restrict__ data_off, uint32_t *__restrict__ data_len,
                             uint32_t *__restrict__ flags, uint64_t *__restrict__ cks_off, uint64_t max_blocks,
                             int64_t *__restrict__ walk) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint64_t src_pos = 0;
    uint32_t flg = flg_in;
    uint64_t bs = bs_in;
    walk[0] = 0; walk[1] = 0; walk[2] = 0; walk[3] = 0;
    if (seg & 1u) {
        uint8_t head[19];
        const size_t have = src_len < sizeof head ? (size_t)src_len : sizeof head;
        for (size_t k = 0; k < have; k++) head[k] = src[k];
        const ParsedHeader ph = parse_header(head, have);                               // :547
        walk[3] = ph.size;
        if (ph.size < 0) { walk[4] = 0; walk[5] = 0; return; }
        src_pos = (uint64_t)ph.size;
        flg = ph.flg;
        bs = ph.block_size;
    }
    walk[4] = flg; walk[5] = (int64_t)bs;
    const uint32_t block_checksum = (flg & 0x10u) ? 1u : 0u;
    uint64_t nb = 0;
    int64_t err = 0;
    while (src_pos < src_len) {                                       // :563
        if (src_pos + 4 > src_len) { err = ZLZ4F_ERR_FRAME_SIZE_WRONG; break; }        // :565
        const uint32_t h = zx_rd32(src + src_pos);
        src_pos += 4;
        if (h == 0) break;                                            // :573
        const uint32_t sz = h & 0x7FFFFFFFu;
        if (src_pos + sz > src_len) { err = ZLZ4F_ERR_FRAME_SIZE_WRONG; break; }       // :582
        const uint64_t off = src_pos;
        src_pos += sz;
        uint32_t fl = (h >> 31);
        uint64_t co = 0;
        if (block_checksum) {                                         // :590
            if (src_pos + 4 > src_len) fl |= 2u;                      // FrameSizeWrong when this block is reached
            else { co = src_pos; src_pos += 4; }
        }
        if (nb < max_blocks) { data_off[nb] = off; data_len[nb] = sz; flags[nb] = fl; cks_off[nb] = co; }
        nb++;
        if (fl & 2u) break;
    }
    walk[0] = (int64_t)nb; walk[1] = (int64_t)src_pos; walk[2] = err;
}

// one lane per block: verify the stored XXH32 of the block payload (:594-598); ok[i] = 1 / 0
__global__ void k_block_verify(const uint8_t *__restrict__ src, const uint64_t *__restrict__ data_off,
                               const uint32_t *__restrict__ data_len, const uint32_t *__restrict__ flags,
                               const uint64_t *__restrict__ cks_off, uint32_t nblocks, uint32_t *__restrict__ ok) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblocks) return;
    if (flags[i] & 2u) { ok[i] = 2; return; }
    ok[i] = xxh32(src + data_off[i], data_len[i], 0) == zx_rd32(src + cks_off[i]) ? 1u : 0u;
}

// Serial dstPos accumulation of :602-621 with the reference's error order; one lane.
// dplan[0] = total output bytes, dplan[1] = error code (0 = none)
__global__ void k_dframe_plan(const uint32_t *__restrict__ data_len, const uint32_t *__restrict__ flags,
                              const int64_t *__restrict__ sizes, const uint32_t *__restrict__ cks_ok,
                              uint32_t block_checksum, uint32_t nblocks, uint64_t dst_cap, int64_t walk_err,
                              uint64_t *__restrict__ out_off, uint32_t *__restrict__ out_cap, int64_t *__restrict__ dplan) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint64_t pos = 0;
    int64_t err = 0;
    for (uint32_t i = 0; i < nblocks; i++) {
        if (block_checksum) {
            if (cks_ok[i] == 2u) { err = ZLZ4F_ERR_FRAME_SIZE_WRONG; break; }         // :591
            if (cks_ok[i] == 0u) { err = ZLZ4F_ERR_BLOCK_CHECKSUM_INVALID; break; }   // :596
        }
        const uint64_t rem = dst_cap - pos;
        uint64_t sz;
        if (flags[i] & 1u) {                                          // stored block :603-608
            sz = data_len[i];
            if (pos + sz > dst_cap) { err = ZLZ4F_ERR_DST_MAX_SIZE_TOO_SMALL; break; }
        } else {                                                      // :610 decompressSafe(blockData, dst[dstPos..])
            if (data_len[i] == 0 || rem == 0) sz = 0;                 // src/lz4.zig:97-98
            else {
                const int64_t s = sizes[i];
                if (s < 0 || (uint64_t)s > rem) { err = ZLZ4F_ERR_DECOMPRESSION_FAILED; break; }   // :611
                sz = (uint64_t)s;
            }
        }
        out_off[i] = pos;
        out_cap[i] = (uint32_t)sz;      // exact: the block decoders may chunk-write up to their capacity, never beyond
        pos += sz;
    }
    if (!err) err = walk_err;
    dplan[0] = (int64_t)pos; dplan[1] = err;
}

// Speculative layout: every frame compressFrame writes has blocks that decode to exactly block_size bytes, except the
// last one (:372-381), so block i can be decoded straight into dst + i * block_size with an exact capacity, without the
// size pass.  k_dframe_check then proves the guess from the decoder's results; any deviation (a foreign frame with
// short blocks, an error of any kind, a destination that is too small) sends the call to the exact two-pass plan.
__global__ void k_dframe_spec(const uint32_t *__restrict__ data_len, const uint32_t *__restrict__ flags, uint32_t nblocks,
                              uint64_t bs, uint64_t dst_cap, uint64_t *__restrict__ out_off, uint32_t *__restrict__ out_cap,
                              uint32_t *__restrict__ dec_len) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nblocks; i += gridDim.x * blockDim.x) {
        const uint64_t o = (uint64_t)i * bs;
        out_off[i] = o < dst_cap ? o : dst_cap;
        out_cap[i] = o >= dst_cap ? 0u : (uint32_t)((dst_cap - o) < bs ? (dst_cap - o) : bs);
        dec_len[i] = (flags[i] & 1u) ? 0u : data_len[i];
    }
}

// dplan[0] = total output bytes, dplan[1] = error (stays 0 here), dplan[2] = 1 if the speculative layout is proven
__global__ __launch_bounds__(64) void k_dframe_check(const uint32_t *__restrict__ data_len, const uint32_t *__restrict__ flags,
                                                     const int64_t *__restrict__ sizes, const uint32_t *__restrict__ cks_ok,
                                                     const uint32_t *__restrict__ out_cap, uint32_t block_checksum,
                                                     uint32_t nblocks, uint64_t bs, const int64_t *__restrict__ walk,
                                                     int64_t *__restrict__ dplan) {
    const uint32_t lane = threadIdx.x;
    bool ok = walk[2] == 0;
    uint64_t last = 0;
    for (uint32_t i = lane; i < nblocks; i += 64u) {
        if (block_checksum && cks_ok[i] != 1u) ok = false;
        uint64_t sz;
        if (flags[i] & 1u) { sz = data_len[i]; if (sz > out_cap[i]) ok = false; }
        else { const int64_t r = sizes[i]; if (r < 0 || data_len[i] == 0) ok = false; sz = r < 0 ? 0 : (uint64_t)r; }
        if (i + 1u < nblocks) { if (sz != bs) ok = false; }
        else last = sz;
    }
    const bool all_ok = zlz4::ballot(!ok) == 0;
    const uint32_t owner = nblocks ? (nblocks - 1u) & 63u : 0u;      // the lane that saw the last block
    const uint64_t last_sz = ((uint64_t)zlz4::rdlane((uint32_t)(last >> 32), owner) << 32) | zlz4::rdlane((uint32_t)last, owner);
    if (lane == 0) {
        dplan[0] = nblocks ? (int64_t)((uint64_t)(nblocks - 1u) * bs + last_sz) : 0;
        dplan[1] = 0;
        dplan[2] = all_ok ? 1 : 0;
    }
}

// one workgroup per block: raw copy of stored blocks (:607); compressed blocks are skipped here
__global__ __launch_bounds__(256) void k_copy_stored(const uint8_t *__restrict__ src, const uint64_t *__restrict__ data_off,
                                                      const uint32_t *__restrict__ data_len,
                                                      const uint32_t *__restrict__ flags,
                                                      const uint64_t *__restrict__ out_off, uint8_t *__restrict__ dst,
                                                      uint64_t dst_cap) {
    const uint32_t i = blockIdx.x, t = threadIdx.x;
    if (!(flags[i] & 1u)) return;
    const uint32_t n = data_len[i];
    if (out_off[i] + n > dst_cap) return;               // (speculative layout: the check kernel reports it)
    const uint8_t *p = src + data_off[i];
    uint8_t *o = dst + out_off[i];
    for (uint32_t k = t * 16u; k + 16u <= n; k += 256u * 16u) zlz4::st128(o + k, zlz4::ld128(p + k));
    const uint32_t t0 = n & ~15u;
    if (t < 16u && t0 + t < n) o[t0 + t] = p[t0 + t];
}

// for the decoder batch call stored blocks become empty inputs (decoded size 0, nothing written)
__global__ void k_mask_stored(uint32_t *data_len_for_decode, const uint32_t *data_len, const uint32_t *flags,
                              uint32_t nblocks, uint64_t *zero_off, uint32_t *big_cap) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nblocks; i += gridDim.x * blockDim.x) {
        data_len_for_decode[i] = (flags[i] & 1u) ? 0u : data_len[i];
        if (zero_off) { zero_off[i] = 0; big_cap[i] = 0xFFFFFFFFu; }
    }
}

// content checksum over dst[0, total) with total read from the device (total_p[0]); only when no error is pending,
// and -- on the speculative path (need_proven) -- only when the layout guess has been proven: an unproven total can
// be anything a corrupted frame says (and XXH32 of gigabytes on one lane takes seconds)
__global__ void k_content_check(const uint8_t *dst, const int64_t *total_p, const uint8_t *stored, int64_t *dplan,
                                uint32_t need_proven, uint64_t dst_cap) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (dplan[1] != 0) return;
    if (need_proven && dplan[2] != 1) return;
    const uint64_t total = (uint64_t)total_p[0];
    if (total > dst_cap) return;
    if (xxh32(dst, total, 0) != zx_rd32(stored)) dplan[1] = ZLZ4F_ERR_CONTENT_CHECKSUM_INVALID;   // :631
}

int64_t map_block_error(int64_t e) {     // mapCompressionError, src/lz4f.zig:144-149
    if (e == ZLZ4_ERR_OUTPUT_TOO_SMALL) return ZLZ4F_ERR_DST_MAX_SIZE_TOO_SMALL;
    if (e == ZLZ4_ERR_UNSUPPORTED || e == ZLZ4_ERR_DEVICE) return e;
    return ZLZ4F_ERR_GENERIC;
}

bool gfx950_ok() { return zlz4_device_check() == 0; }

}  // namespace

extern "C" {

size_t zlz4f_compress_frame_bound(size_t src_size, const zlz4f_prefs *prefs) {   // src/lz4f.zig:274-301
    const zlz4f_prefs *p = prefs ? prefs : &kDefaultPrefs;
    const size_t block_size = block_size_of(p->block_size_id);
    const size_t num_blocks = (src_size + block_size - 1) / block_size;
    size_t per_block = 4 + zlz4_compress_bound(block_size);
    if (p->block_checksum == 1) per_block += 4;
    size_t result = 19 + num_blocks * per_block + 4;
    if (p->content_checksum == 1) result += 4;
    return result;
}

int64_t zlz4f_header_size(const uint8_t *src, size_t n) {   // src/lz4f.zig:451-480
    if (n < 5) return ZLZ4F_ERR_FRAME_HEADER_INCOMPLETE;
    uint32_t magic;
    std::memcpy(&magic, src, 4);
    if (magic != ZLZ4F_MAGICNUMBER) {
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) return 8;
        return ZLZ4F_ERR_FRAME_TYPE_UNKNOWN;
    }
    const uint8_t flg = src[4];
    int64_t size = 7;
    if (flg & 0x08) size += 8;
    if (flg & 0x01) size += 4;
    return size;
}

}  // extern "C"

namespace {

constexpr uint32_t kSegFirst = 1u, kSegLast = 2u;

// src/lz4f.zig:354-446 on a device-resident source.  seg: kSegFirst = write the frame header (:369), kSegLast = write the
// end mark (:433) and the content checksum (:437-441); a whole frame has both.
int64_t compress_frame_impl(hipStream_t st, const uint8_t *d_src, size_t n, uint8_t *d_dst, size_t cap, const zlz4f_prefs &p,
                            uint32_t seg) {
    if (cap < zlz4f_compress_frame_bound(n, &p)) return ZLZ4F_ERR_DST_MAX_SIZE_TOO_SMALL;   // :363-366
    if (!gfx950_ok()) return ZLZ4_ERR_DEVICE;
    HeaderBytes hb = encode_header(p);                                                     // :369
    if (!(seg & kSegFirst)) hb.n = 0;
    const size_t bs = block_size_of(p.block_size_id);                                      // :372
    const uint64_t nb64 = (n + bs - 1) / bs;
    if (nb64 > 0x7FFFFFFFull) return ZLZ4F_ERR_SRC_SIZE_TOO_LARGE;
    const uint32_t nb = (uint32_t)nb64;
    // level routing :393-404; compressHC then normalises <2 -> 9, >12 -> 12 (src/lz4hc.zig:1445)
    int32_t hc_level = 0;
    if (p.compression_level > 0) {
        hc_level = p.compression_level < 2 ? 9 : (p.compression_level > 12 ? 12 : p.compression_level);
    }
    const uint32_t tail = (seg & kSegLast) ? 1u : 0u;
    const uint32_t content_checksum = (tail && p.content_checksum == 1) ? 1u : 0u;
    const uint64_t slot = (zlz4_compress_bound(bs) + 15) & ~15ull;
    FrameCall fc(st);
    DevBuf d_plan(4 * sizeof(int64_t), &fc);
    if (!d_plan.p) return ZLZ4F_ERR_ALLOCATION_FAILED;
    int64_t total = 0;
    if (nb == 0) {
        int64_t plan0[3] = {(int64_t)hb.n, 0, 0};
        if (hipMemcpyAsync(d_plan.p, plan0, sizeof plan0, hipMemcpyHostToDevice, st) != hipSuccess) return ZLZ4_ERR_DEVICE;
        fc.launched();
        hipLaunchKernelGGL(k_frame_head_tail, dim3(1), dim3(64), 0, st, hb, d_dst, d_plan.as<int64_t>(), d_src,
                           (uint64_t)n, tail, content_checksum, d_plan.as<int64_t>() + 3);
        if (hipMemcpyAsync(&total, d_plan.as<int64_t>() + 3, 8, hipMemcpyDeviceToHost, st) != hipSuccess || !fc.sync())
            return ZLZ4_ERR_DEVICE;
        return total;
    }
    DevBuf d_slots((uint64_t)nb * slot, &fc), d_u64((uint64_t)nb * 3 * sizeof(uint64_t), &fc),
        d_u32((uint64_t)nb * 4 * sizeof(uint32_t), &fc), d_res((uint64_t)nb * sizeof(int64_t), &fc);
    if (!d_slots.p || !d_u64.p || !d_u32.p || !d_res.p) return ZLZ4F_ERR_ALLOCATION_FAILED;
    const size_t wsb = hc_level ? zlz4_hc_workspace_bytes(nb, (uint32_t)bs) : 0;
    DevBuf d_ws(wsb, &fc);
    if (hc_level && !d_ws.p) return ZLZ4F_ERR_ALLOCATION_FAILED;
    uint64_t *in_off = d_u64.as<uint64_t>(), *out_off = in_off + nb, *dst_off = out_off + nb;
    uint32_t *in_len = d_u32.as<uint32_t>(), *out_cap = in_len + nb, *hdr = out_cap + nb, *cks = hdr + nb;
    fc.launched();
    hipLaunchKernelGGL(k_frame_desc, dim3((nb + 255) / 256 > 1024 ? 1024 : (nb + 255) / 256), dim3(256), 0, st, (uint64_t)n,
                       (uint64_t)bs, slot, nb, in_off, in_len, out_off, out_cap);
    int rc;
    if (hc_level == 0) {
        rc = zlz4_launch_compress_fast(st, d_src, in_off, in_len, d_slots.as<uint8_t>(), out_off, out_cap,
                                       d_res.as<int64_t>(), nb, (uint32_t)bs, 1);                        // :400-404
    } else {
        rc = zlz4_launch_compress_hc(st, d_src, in_off, in_len, d_slots.as<uint8_t>(), out_off, out_cap,
                                     d_res.as<int64_t>(), nb, (uint32_t)bs, hc_level, d_ws.p, wsb);      // :394-398
    }
    if (rc != 0) return rc;
    hipLaunchKernelGGL(k_frame_plan, dim3(1), dim3(64), 0, st, d_res.as<int64_t>(), in_len, nb,
                       p.block_checksum == 1 ? 1u : 0u, (uint64_t)hb.n, dst_off, hdr, d_plan.as<int64_t>());
    if (p.block_checksum == 1)
        hipLaunchKernelGGL(k_block_xxh32, dim3((nb + 63) / 64), dim3(64), 0, st, d_src, in_off, d_slots.as<uint8_t>(),
                           out_off, hdr, nb, cks);
    hipLaunchKernelGGL(k_frame_scatter, dim3(nb), dim3(256), 0, st, d_src, in_off, d_slots.as<uint8_t>(), out_off, hdr,
                       dst_off, cks, p.block_checksum == 1 ? 1u : 0u, d_dst);
    hipLaunchKernelGGL(k_frame_head_tail, dim3(1), dim3(64), 0, st, hb, d_dst, d_plan.as<int64_t>(), d_src, (uint64_t)n,
                       tail, content_checksum, d_plan.as<int64_t>() + 3);
    int64_t plan[4];
    if (hipMemcpyAsync(plan, d_plan.p, sizeof plan, hipMemcpyDeviceToHost, st) != hipSuccess || !fc.sync() ||
        hipGetLastError() != hipSuccess)
        return ZLZ4_ERR_DEVICE;
    if (plan[1] != 0) return map_block_error(plan[2]);                                                   // :398, :404
    return plan[3];
}

// src/lz4f.zig:541-638 on device-resident bytes.  seg & kSegFirst: the bytes start with the frame header (:547), else
// `p` describes the frame (block checksum flag, block size).  A whole frame has kSegFirst | kSegLast; without kSegLast
// the bytes are expected to end after a block (no end mark, no content checksum).
int64_t decompress_frame_impl(hipStream_t st, const uint8_t *d_src, size_t n, uint8_t *d_dst, size_t cap, const zlz4f_prefs &p,
                              uint32_t seg) {
    if (!gfx950_ok()) return ZLZ4_ERR_DEVICE;
    FrameCall fc(st);
    DevBuf d_walk(12 * sizeof(int64_t), &fc);
    if (!d_walk.p) return ZLZ4F_ERR_ALLOCATION_FAILED;
    int64_t *walk = d_walk.as<int64_t>(), *dplan = walk + 8;
    const uint32_t flg_in = 0x40u | (p.block_mode == 1 ? 0x20u : 0u) | (p.block_checksum == 1 ? 0x10u : 0u);
    const uint64_t bs_in = block_size_of(p.block_size_id);
    // the block table is sized for the blocks a frame of this length usually has; a frame of tiny blocks repeats the walk
    uint64_t table_cap = n / 1024 + 64;
    if (table_cap > n / 5 + 1) table_cap = n / 5 + 1;
    int64_t w[6];
    for (int attempt = 0;; attempt++) {
        DevBuf d_u64(table_cap * 3 * sizeof(uint64_t), &fc), d_u32(table_cap * 5 * sizeof(uint32_t), &fc),
            d_sz(table_cap * sizeof(int64_t), &fc);
        if (!d_u64.p || !d_u32.p || !d_sz.p) return ZLZ4F_ERR_ALLOCATION_FAILED;
        uint64_t *data_off = d_u64.as<uint64_t>(), *cks_off = data_off + table_cap, *out_off = cks_off + table_cap;
        uint32_t *data_len = d_u32.as<uint32_t>(), *flags = data_len + table_cap, *cks_ok = flags + table_cap,
                 *out_cap = cks_ok + table_cap, *dec_len = out_cap + table_cap;
        fc.launched();
        hipLaunchKernelGGL(k_frame_walk, dim3(1), dim3(64), 0, st, d_src, (uint64_t)n, seg, flg_in, bs_in, data_off, data_len,
                           flags, cks_off, table_cap, walk);
        if (hipMemcpyAsync(w, walk, sizeof w, hipMemcpyDeviceToHost, st) != hipSuccess || !fc.sync()) return ZLZ4_ERR_DEVICE;
        if (w[3] < 0) return w[3];                                                                       // header error :547
        if (w[0] > 0x7FFFFFFFll) return ZLZ4F_ERR_FRAME_SIZE_WRONG;
        if ((uint64_t)w[0] > table_cap) {
            if (attempt) return ZLZ4_ERR_DEVICE;
            table_cap = (uint64_t)w[0];
            continue;
        }
        const uint32_t nb = (uint32_t)w[0];
        const uint64_t src_pos_end = (uint64_t)w[1];
        const int64_t walk_err = w[2];
        const uint32_t flg = (uint32_t)w[4];
        const uint64_t bs = (uint64_t)w[5];
        const uint32_t bc = (flg & 0x10) ? 1u : 0u, cc = ((seg & kSegLast) && (flg & 0x04)) ? 1u : 0u;
        int64_t plan_host[3] = {0, walk_err, 0};
        int64_t tot[2] = {0, 0};          // host source of an async copy: lives until the call's last synchronisation
        const uint32_t gridb = (nb + 255) / 256 > 1024 ? 1024 : (nb + 255) / 256;
        if (nb) {
            // speculative single pass: block i -> dst + i * block_size, proven afterwards
            fc.launched();
            if (bc) hipLaunchKernelGGL(k_block_verify, dim3((nb + 63) / 64), dim3(64), 0, st, d_src, data_off, data_len, flags,
                                       cks_off, nb, cks_ok);
            hipLaunchKernelGGL(k_dframe_spec, dim3(gridb), dim3(256), 0, st, data_len, flags, nb, bs, (uint64_t)cap, out_off,
                               out_cap, dec_len);
            int rc = zlz4_launch_decompress_safe(st, d_src, data_off, dec_len, d_dst, out_off, out_cap, d_sz.as<int64_t>(), nb);
            if (rc != 0) return rc;
            hipLaunchKernelGGL(k_copy_stored, dim3(nb), dim3(256), 0, st, d_src, data_off, data_len, flags, out_off, d_dst,
                               (uint64_t)cap);
            hipLaunchKernelGGL(k_dframe_check, dim3(1), dim3(64), 0, st, data_len, flags, d_sz.as<int64_t>(), cks_ok, out_cap, bc,
                               nb, bs, walk, dplan);
            if (cc && src_pos_end + 4 <= n)     // only meaningful when the guess holds; checked below
                hipLaunchKernelGGL(k_content_check, dim3(1), dim3(64), 0, st, d_dst, dplan, d_src + src_pos_end, dplan, 1u,
                                   (uint64_t)cap);
            if (hipMemcpyAsync(plan_host, dplan, sizeof plan_host, hipMemcpyDeviceToHost, st) != hipSuccess || !fc.sync())
                return ZLZ4_ERR_DEVICE;
            if (plan_host[2] == 1) {
                if (cc && src_pos_end + 4 > n) return ZLZ4F_ERR_FRAME_SIZE_WRONG;                        // :626
                if (hipGetLastError() != hipSuccess) return ZLZ4_ERR_DEVICE;
                return plan_host[1] != 0 ? plan_host[1] : plan_host[0];
            }
            // exact plan: size pass (unlimited capacity), the serial dstPos accumulation with the reference's error
            // order (:602-621), then decode + raw copies at the exact places
            DevBuf d_x64((uint64_t)nb * sizeof(uint64_t), &fc), d_x32((uint64_t)nb * sizeof(uint32_t), &fc);
            if (!d_x64.p || !d_x32.p) return ZLZ4F_ERR_ALLOCATION_FAILED;
            uint64_t *zero_off = d_x64.as<uint64_t>();
            uint32_t *big_cap = d_x32.as<uint32_t>();
            fc.launched();
            hipLaunchKernelGGL(k_mask_stored, dim3(gridb), dim3(256), 0, st, dec_len, data_len, flags, nb, zero_off, big_cap);
            rc = zlz4_launch_decompress_sizes(st, d_src, data_off, dec_len, zero_off, big_cap, d_sz.as<int64_t>(), nb);
            if (rc != 0) return rc;
            hipLaunchKernelGGL(k_dframe_plan, dim3(1), dim3(64), 0, st, data_len, flags, d_sz.as<int64_t>(), cks_ok, bc, nb,
                               (uint64_t)cap, walk_err, out_off, out_cap, dplan);
            if (hipMemcpyAsync(plan_host, dplan, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, st) != hipSuccess || !fc.sync())
                return ZLZ4_ERR_DEVICE;
            if (plan_host[1] != 0) return plan_host[1];
            fc.launched();
            rc = zlz4_launch_decompress_safe(st, d_src, data_off, dec_len, d_dst, out_off, out_cap, d_sz.as<int64_t>(), nb);
            if (rc != 0) return rc;
            hipLaunchKernelGGL(k_copy_stored, dim3(nb), dim3(256), 0, st, d_src, data_off, data_len, flags, out_off, d_dst,
                               (uint64_t)cap);
        } else if (walk_err != 0) {
            return walk_err;
        }
        if (cc) {                                                                                        // :625-635
            if (src_pos_end + 4 > n) { (void)fc.sync(); return ZLZ4F_ERR_FRAME_SIZE_WRONG; }
            tot[0] = plan_host[0];
            if (hipMemcpyAsync(dplan, tot, sizeof tot, hipMemcpyHostToDevice, st) != hipSuccess) return ZLZ4_ERR_DEVICE;
            fc.launched();
            hipLaunchKernelGGL(k_content_check, dim3(1), dim3(64), 0, st, d_dst, dplan, d_src + src_pos_end, dplan, 0u,
                               (uint64_t)cap);
            if (hipMemcpyAsync(plan_host, dplan, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, st) != hipSuccess) return ZLZ4_ERR_DEVICE;
        }
        if (!fc.sync() || hipGetLastError() != hipSuccess) return ZLZ4_ERR_DEVICE;
        if (plan_host[1] != 0) return plan_host[1];
        return plan_host[0];
    }
}

}  // namespace

extern "C" {

int64_t zlz4f_compress_frame_device(void *stream_, const uint8_t *d_src, size_t n, uint8_t *d_dst, size_t cap,
                                    const zlz4f_prefs *prefs) {
    return compress_frame_impl((hipStream_t)stream_, d_src, n, d_dst, cap, prefs ? *prefs : kDefaultPrefs, kSegFirst | kSegLast);
}

// One rank's part of a frame whose blocks are spread over several GPUs (the block loop :379-430 carries no state)
int64_t zlz4f_compress_frame_segment_device(void *stream_, const uint8_t *d_src, size_t n, uint8_t *d_dst, size_t cap,
                                            const zlz4f_prefs *prefs, uint32_t segment_flags) {
    const zlz4f_prefs p = prefs ? *prefs : kDefaultPrefs;
    if (segment_flags & ~(kSegFirst | kSegLast)) return ZLZ4F_ERR_PARAMETER_INVALID;
    // XXH32 of the whole content is one serial chain over every rank's bytes: not available for a split frame
    if (p.content_checksum == 1 && segment_flags != (kSegFirst | kSegLast)) return ZLZ4_ERR_UNSUPPORTED;
    const size_t bs = block_size_of(p.block_size_id);
    if (!(segment_flags & kSegLast) && n % bs != 0) return ZLZ4F_ERR_PARAMETER_INVALID;   // only the last block may be short
    return compress_frame_impl((hipStream_t)stream_, d_src, n, d_dst, cap, p, segment_flags);
}

int64_t zlz4f_decompress_frame_device(void *stream_, const uint8_t *d_src, size_t n, uint8_t *d_dst, size_t cap) {
    return decompress_frame_impl((hipStream_t)stream_, d_src, n, d_dst, cap, kDefaultPrefs, kSegFirst | kSegLast);
}

int64_t zlz4f_decompress_frame_segment_device(void *stream_, const uint8_t *d_src, size_t n, uint8_t *d_dst, size_t cap,
                                              const zlz4f_prefs *prefs, uint32_t segment_flags) {
    const zlz4f_prefs p = prefs ? *prefs : kDefaultPrefs;
    if (segment_flags & ~(kSegFirst | kSegLast)) return ZLZ4F_ERR_PARAMETER_INVALID;
    if (p.content_checksum == 1 && segment_flags != (kSegFirst | kSegLast)) return ZLZ4_ERR_UNSUPPORTED;
    return decompress_frame_impl((hipStream_t)stream_, d_src, n, d_dst, cap, p, segment_flags);
}

// src/lz4f.zig:354-446, host pointers: stage -> device path -> copy the frame back
int64_t zlz4f_compress_frame(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, const zlz4f_prefs *prefs) {
    const zlz4f_prefs p = prefs ? *prefs : kDefaultPrefs;
    const size_t bound = zlz4f_compress_frame_bound(n, &p);
    if (cap < bound) return ZLZ4F_ERR_DST_MAX_SIZE_TOO_SMALL;                                            // :363-366
    if (!gfx950_ok()) return ZLZ4_ERR_DEVICE;
    DevBuf d_src(n), d_dst(bound);
    if (!d_src.p || !d_dst.p) return ZLZ4F_ERR_ALLOCATION_FAILED;
    if (n && hipMemcpy(d_src.p, src, n, hipMemcpyHostToDevice) != hipSuccess) return ZLZ4_ERR_DEVICE;
    const int64_t r = zlz4f_compress_frame_device(nullptr, d_src.as<uint8_t>(), n, d_dst.as<uint8_t>(), bound, &p);
    if (r < 0) return r;
    if ((uint64_t)r > cap) return ZLZ4_ERR_DEVICE;
    if (hipMemcpy(dst, d_dst.p, (size_t)r, hipMemcpyDeviceToHost) != hipSuccess) return ZLZ4_ERR_DEVICE;
    return r;
}

// src/lz4f.zig:541-638, host pointers
int64_t zlz4f_decompress_frame(const uint8_t *src, size_t n, uint8_t *dst, size_t cap) {
    const ParsedHeader ph = parse_header(src, n);      // header errors need no device
    if (ph.size < 0) return ph.size;
    if (!gfx950_ok()) return ZLZ4_ERR_DEVICE;
    DevBuf d_src(n), d_dst(cap);
    if (!d_src.p || !d_dst.p) return ZLZ4F_ERR_ALLOCATION_FAILED;
    if (hipMemcpy(d_src.p, src, n, hipMemcpyHostToDevice) != hipSuccess) return ZLZ4_ERR_DEVICE;
    const int64_t r = zlz4f_decompress_frame_device(nullptr, d_src.as<uint8_t>(), n, d_dst.as<uint8_t>(), cap);
    if (r <= 0) return r;
    if ((uint64_t)r > cap) return ZLZ4_ERR_DEVICE;
    if (hipMemcpy(dst, d_dst.p, (size_t)r, hipMemcpyDeviceToHost) != hipSuccess) return ZLZ4_ERR_DEVICE;
    return r;
}

}  // extern "C"
