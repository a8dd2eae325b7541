// zlz4_decompress.hip -- LZ4 block decoder for gfx950, one wavefront per block.
//
// Replaces lz4.decompressSafe (reference src/lz4.zig:257-259), i.e.
// decompressGeneric (src/lz4.zig:89-251) with lowPrefix == dst.ptr and no
// dictionary.  The decision order and the error returned at every exit follow
// SURVEY.md Appendix C exactly; `dst` contents after an error are unspecified
// (as in the reference).
//
// Layout: the token stream is parsed wave-uniformly out of a 64-byte register
// window (lane i holds src[wbase + i], tokens/lengths/offsets are pulled with
// v_readlane), short literal runs are stored straight from that window, long
// ones and matches are copied 16 B per lane.  Overlapping matches
// (offset < matchLength, src/lz4.zig:235-241) use the periodic-extension
// identity out[op+k] = out[op-offset + (k mod offset)].
#include <cstdlib>

#include "zlz4_device.hpp"

namespace zlz4 {

// kWrite == false: size pass (no stores) -- used by the frame decoder to learn every block's
// decompressed size before placing the blocks (lz4f.decompressFrame accumulates dstPos serially).
template <bool kWrite>
__global__ __launch_bounds__(256) void k_decompress_safe(
    const uint8_t *__restrict__ d_in, const uint64_t *__restrict__ d_in_off,
    const uint32_t *__restrict__ d_in_len, uint8_t *d_out, const uint64_t *__restrict__ d_out_off,
    const uint32_t *__restrict__ d_out_cap, int64_t *__restrict__ d_result, uint32_t nblocks) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t blk = rfl(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (blk >= nblocks) return;

    const uint8_t *src = d_in + d_in_off[blk];
    uint8_t *dst = d_out + d_out_off[blk];
    const uint32_t iend = rfl(d_in_len[blk]);   // src.len
    const uint32_t oend = rfl(d_out_cap[blk]);  // dst.len == targetOutputSize

    int64_t res = 0;
    uint32_t ip = 0, op = 0;

    if (iend != 0 && oend != 0) {               // src/lz4.zig:97-98
        // 64-byte register window over the compressed stream
        uint32_t wbase = 0;
        uint32_t w = (lane < iend) ? src[lane] : 0u;
        auto reload = [&](uint32_t pos) {
            wbase = pos;
            w = (pos + lane < iend) ? src[pos + lane] : 0u;
        };
        auto fetch = [&](uint32_t pos) -> uint32_t {   // caller guarantees pos < iend
            if (pos - wbase >= 64u) reload(pos);
            return rdlane(w, pos - wbase);
        };

        for (;;) {
            if (ip >= iend) break;                                  // :113
            // ---- fast path: the whole sequence header (token, <= 14 literals, offset) sits inside the window,
            //      no length extension bytes.  Same checks in the same order as the general path below,
            //      written with 32-bit "remaining" arithmetic (ip <= iend and op <= oend always hold). ----
            if (ip - wbase > 44u) reload(ip);                       // keep >= 20 window bytes ahead of the token
            {
                const uint32_t i0 = ip - wbase;
                const uint32_t token = rdlane(w, i0);               // :116
                const uint32_t lit = token >> 4, mlc = token & 15u; // :120, :157
                const uint32_t in_rem = iend - ip - 1u;             // bytes after the token
                if (lit != 15u && mlc != 15u && in_rem >= lit + 2u) {
                    // (in_rem >= lit + 2 : literals fit (:136) and the offset is present (:146, :149))
                    if (lit > oend - op) { res = kErrOutputTooSmall; break; }            // :137
                    if (kWrite && lane > i0 && lane <= i0 + lit) dst[op + (lane - i0 - 1u)] = (uint8_t)w;   // :140
                    op += lit;
                    const uint32_t offset = rdlane(w, i0 + 1u + lit) | (rdlane(w, i0 + 2u + lit) << 8);   // :150
                    ip += 3u + lit;
                    if (offset == 0) { res = kErrCorrupted; break; }                     // :154
                    const uint32_t ml = mlc + kMinMatch;                                 // :171 (4..18)
                    if (ml > oend - op) { res = kErrOutputTooSmall; break; }             // :174
                    if (offset > op) { res = kErrCorrupted; break; }                     // :181-186 / :231
                    if (kWrite) {
                        const uint8_t *m = dst + (op - offset);
                        // out[op+k] = out[op-offset + (k mod offset)]; ml <= 18 lanes, one load + one store
                        uint32_t k = lane;
                        if (offset < ml) {                           // overlap (:235-241): k mod offset, k < 18
                            if (offset == 1u) k = 0;
                            else { while (k >= offset) k -= offset; }
                        }
                        if (lane < ml) dst[op + lane] = m[k];
                    }
                    op += ml;
                    continue;
                }
            }
            // ---- general path (length extensions, long literal runs, end of block, malformed input) ----
            const uint32_t token = fetch(ip);                       // :116
            ip += 1;
            uint32_t lit = token >> 4;                              // :120
            if (lit == 15u) {                                       // :123-131
                bool bad = false;
                for (;;) {
                    if (ip >= iend) { bad = true; break; }          // :125
                    const uint32_t s = fetch(ip);
                    ip += 1;
                    lit += s;
                    if (lit > 0xFFFF0000u) lit = 0xFFFF0000u;       // saturate: already larger than any input
                    if (s != 255u) break;
                }
                if (bad) { res = kErrCorrupted; break; }
            }
            if (lit > 0) {                                          // :134
                if (lit > iend - ip) { res = kErrCorrupted; break; }        // :136
                if (lit > oend - op) { res = kErrOutputTooSmall; break; }   // :137
                if (!kWrite) {
                    // size pass: literals are skipped, the window follows lazily
                } else if (lit <= 64u) {
                    if (ip - wbase + lit > 64u) reload(ip);
                    const uint32_t j0 = ip - wbase;
                    if (lane >= j0 && lane < j0 + lit) dst[op + (lane - j0)] = (uint8_t)w;
                } else {
                    copy_bytes(dst + op, src + ip, lit, lane);      // :140
                }
                ip += lit;
                op += lit;
            }
            if (ip >= iend) break;                                  // :146
            if (iend - ip < 2u) { res = kErrCorrupted; break; }     // :149
            const uint32_t offset = fetch(ip) | (fetch(ip + 1u) << 8);   // :150
            ip += 2;
            if (offset == 0) { res = kErrCorrupted; break; }        // :154
            uint32_t ml = token & 15u;                              // :157
            if (ml == 15u) {                                        // :160-168
                bool bad = false;
                for (;;) {
                    if (ip >= iend) { bad = true; break; }          // :162
                    const uint32_t s = fetch(ip);
                    ip += 1;
                    ml += s;
                    if (ml > 0xFFFF0000u) ml = 0xFFFF0000u;         // saturate: can only end in OutputTooSmall / Corrupted
                    if (s != 255u) break;
                }
                if (bad) { res = kErrCorrupted; break; }
            }
            ml += kMinMatch;                                        // :171
            if (ml > oend - op) { res = kErrOutputTooSmall; break; }   // :174
            if (offset > op) { res = kErrCorrupted; break; }        // :181-186 (no dict) / :231
            const uint8_t *m = dst + (op - offset);
            uint8_t *o = dst + op;
            if (!kWrite) {
                // size pass: nothing to copy
            } else if (offset >= ml || offset >= 1024u) {
                // disjoint, or far enough apart that 1 KiB chunks in address order are exact (:244 / :238-240)
                copy_bytes(o, m, ml, lane);
            } else if (offset >= 64u) {
                for (uint32_t k = lane; k < ml; k += 64u) o[k] = m[k];
            } else {
                // RLE-style overlap: every output byte is m[k mod offset]; with a chunk that is a
                // multiple of `offset` each lane's value is loop-invariant.
                const uint32_t cs = 64u - (64u % offset);
                const uint8_t v = m[lane % offset];
                if (lane < cs)
                    for (uint32_t k = lane; k < ml; k += cs) o[k] = v;
            }
            op += ml;
        }
        if (res == 0) res = (int64_t)op;                            // :250
    }
    if (lane == 0) d_result[blk] = res;
}

// ------------------------------------------------------------------------------------------------
// One LANE per block (large batches).  The wave-per-block kernel above spends ~50 scalar instructions per
// sequence on the single scalar unit of a CU; here every instruction serves 64 blocks at once and the copies
// are per-lane 16-byte chunk moves.  Same decision order / error codes as decompressGeneric.
// Chunk copies may write up to 15 bytes past the end of a literal run or match, but never past dst + cap and
// only ahead of the write position, where later output overwrites them (bytes between the returned size and
// the capacity are unspecified, as in every wild-copy LZ4 decoder).  Output regions of different blocks must
// therefore not overlap (the frame decoder passes the exact block size as capacity).
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
__device__ __forceinline__ uint64_t ld64u(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

__global__ __launch_bounds__(64) void k_decompress_lane(
    const uint8_t *__restrict__ d_in, const uint64_t *__restrict__ d_in_off,
    const uint32_t *__restrict__ d_in_len, uint8_t *d_out, const uint64_t *__restrict__ d_out_off,
    const uint32_t *__restrict__ d_out_cap, int64_t *__restrict__ d_result, uint32_t nblocks) {
  // grid-stride over blocks: the launcher may cap the number of concurrently decoded blocks
  for (uint32_t blk = blockIdx.x * blockDim.x + threadIdx.x; blk < nblocks; blk += gridDim.x * blockDim.x) {   // blockDim.x = active lanes per wavefront
    const uint8_t *src = d_in + d_in_off[blk];
    uint8_t *dst = d_out + d_out_off[blk];
    const uint32_t iend = d_in_len[blk], oend = d_out_cap[blk];
    int64_t res = 0;
    uint32_t ip = 0, op = 0;
    if (iend != 0 && oend != 0) {                                   // src/lz4.zig:97-98
        // 16-byte look-ahead at the next token: one load yields token, <= 13 literals and the offset, and it is
        // issued before the copies of the previous sequence so the two round trips overlap
        u32x4 w = {0, 0, 0, 0};
        bool have_w = false;
        if (16u <= iend) { w = ld128(src); have_w = true; }
        for (;;) {
            if (ip >= iend) break;                                  // :113
            if (have_w) {
                const uint32_t tok = w.x & 0xFFu;
                const uint32_t lit = tok >> 4, mlc = tok & 15u;
                // fast path: no length extension, header + literals inside the 16 bytes, room for chunk stores
                if (lit <= 13u && mlc != 15u && (uint64_t)op + lit + 48u <= oend) {
                    // (ip + 16 <= iend holds, so the literals fit (:136) and the offset is present (:146, :149);
                    //  op + lit + 48 <= oend covers :137 and :174 and keeps every chunk store inside dst)
                    const uint32_t oi = 1u + lit;                   // byte index of the offset inside w
                    const uint32_t d0 = oi >> 2, sh = (oi & 3u) * 8u;
                    const uint32_t lo = d0 == 0 ? w.x : (d0 == 1 ? w.y : (d0 == 2 ? w.z : w.w));
                    const uint32_t hi = d0 == 0 ? w.y : (d0 == 1 ? w.z : w.w);     // d0 == 3 only with sh <= 16
                    const uint32_t offset = (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) & 0xFFFFu;   // :150
                    const uint32_t nip = ip + 3u + lit;
                    u32x4 wn = {0, 0, 0, 0};
                    const bool have_n = (uint64_t)nip + 16u <= iend;
                    if (have_n) wn = ld128(src + nip);              // next header: in flight during the copies
                    u32x4 ls;                                       // literals = bytes 1..lit of w (:140), chunk store
                    ls.x = (w.x >> 8) | (w.y << 24); ls.y = (w.y >> 8) | (w.z << 24);
                    ls.z = (w.z >> 8) | (w.w << 24); ls.w = w.w >> 8;
                    st128(dst + op, ls);
                    op += lit;
                    ip = nip;
                    if (offset == 0) { res = kErrCorrupted; break; }            // :154
                    const uint32_t ml = mlc + kMinMatch;                        // :171 (4..18)
                    if (offset > op) { res = kErrCorrupted; break; }            // :181-186 / :231
                    const uint8_t *m = dst + (op - offset);
                    uint8_t *o = dst + op;
                    if (offset >= 16u) {
                        st128(o, ld128(m));
                        if (ml > 16u) st128(o + 16u, ld128(m + 16u));
                    } else if (offset >= 8u) {
                        for (uint32_t k = 0; k < ml; k += 8u) { const uint64_t v = ld64u(m + k); __builtin_memcpy(o + k, &v, 8); }
                    } else {
                        for (uint32_t k = 0; k < ml; k++) o[k] = m[k];          // :238-240
                    }
                    op += ml;
                    w = wn;
                    have_w = have_n;
                    continue;
                }
            }
            // ---- general path (length extensions, long literal runs, end of the stream / output, malformed input) ----
            const uint32_t token = src[ip];                         // :116
            ip += 1;
            uint32_t lit = token >> 4;                              // :120
            if (lit == 15u) {                                       // :123-131
                bool bad = false;
                for (;;) {
                    if (ip >= iend) { bad = true; break; }          // :125
                    const uint32_t s = src[ip];
                    ip += 1;
                    lit += s;
                    if (lit > 0xFFFF0000u) lit = 0xFFFF0000u;       // saturate (can only end in an error)
                    if (s != 255u) break;
                }
                if (bad) { res = kErrCorrupted; break; }
            }
            if (lit > 0) {                                          // :134
                if (lit > iend - ip) { res = kErrCorrupted; break; }        // :136
                if (lit > oend - op) { res = kErrOutputTooSmall; break; }   // :137
                uint32_t k = 0;                                     // :140
                while (k < lit && (uint64_t)ip + k + 16u <= iend && (uint64_t)op + k + 16u <= oend) {
                    st128(dst + op + k, ld128(src + ip + k));
                    k += 16u;
                }
                for (; k < lit; k++) dst[op + k] = src[ip + k];     // exact bytes next to the end of either buffer
                ip += lit;
                op += lit;
            }
            if (ip >= iend) break;                                  // :146
            if (iend - ip < 2u) { res = kErrCorrupted; break; }     // :149
            const uint32_t offset = ld16(src + ip);                 // :150
            ip += 2;
            if (offset == 0) { res = kErrCorrupted; break; }        // :154
            uint32_t ml = token & 15u;                              // :157
            if (ml == 15u) {                                        // :160-168
                bool bad = false;
                for (;;) {
                    if (ip >= iend) { bad = true; break; }          // :162
                    const uint32_t s = src[ip];
                    ip += 1;
                    ml += s;
                    if (ml > 0xFFFF0000u) ml = 0xFFFF0000u;
                    if (s != 255u) break;
                }
                if (bad) { res = kErrCorrupted; break; }
            }
            ml += kMinMatch;                                        // :171
            if (ml > oend - op) { res = kErrOutputTooSmall; break; }   // :174
            if (offset > op) { res = kErrCorrupted; break; }        // :181-186 (no dict) / :231
            const uint8_t *m = dst + (op - offset);
            uint8_t *o = dst + op;
            uint32_t k = 0;
            if (offset >= 16u) {                                    // chunks never read what they are about to write
                while (k < ml && (uint64_t)op + k + 16u <= oend) { st128(o + k, ld128(m + k)); k += 16u; }
            } else if (offset >= 8u) {
                while (k < ml && (uint64_t)op + k + 8u <= oend) { const uint64_t v = ld64u(m + k); __builtin_memcpy(o + k, &v, 8); k += 8u; }
            }
            for (; k < ml; k++) o[k] = m[k];                        // :238-240 byte-serial (overlap, or next to the end)
            op += ml;
            have_w = (uint64_t)ip + 16u <= iend;
            if (have_w) w = ld128(src + ip);
        }
        if (res == 0) res = (int64_t)op;                            // :250
    }
    d_result[blk] = res;
  }
}

}  // namespace zlz4

extern "C" int zlz4_launch_decompress_safe(hipStream_t stream, const uint8_t *d_in, const uint64_t *d_in_off,
                                           const uint32_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off,
                                           const uint32_t *d_out_cap, int64_t *d_result, uint32_t nblocks) {
    if (nblocks == 0) return 0;
    // large batches: one lane per block (64 blocks per wavefront); small batches: one wavefront per block
    static const uint32_t lane_min = [] { const char *e = getenv("ZLZ4_DECOMP_LANE_MIN"); return e ? (uint32_t)atoll(e) : 16384u; }();
    if (nblocks >= lane_min) {
        // the kernel is latency-bound: with few blocks use fewer lanes per wavefront so that ~8192 wavefronts exist
        static const uint32_t lanes_env = [] { const char *e = getenv("ZLZ4_DECOMP_LANES"); return e ? (uint32_t)atoi(e) : 0u; }();
        // measured on MI355X (65 536 blocks): 16 or 32 active lanes per wavefront are ~5 % faster than 64
        uint32_t lanes = nblocks >= 524288u ? 64u : (nblocks >= 262144u ? 32u : 16u);
        if (lanes_env >= 1 && lanes_env <= 64) lanes = lanes_env;
        static const uint32_t max_lanes = [] { const char *e = getenv("ZLZ4_DECOMP_MAXLANES"); return e ? (uint32_t)atoll(e) : 0u; }();
        uint32_t grid = (nblocks + lanes - 1u) / lanes;
        if (max_lanes && (uint64_t)grid * lanes > max_lanes) grid = (max_lanes + lanes - 1u) / lanes;
        hipLaunchKernelGGL(zlz4::k_decompress_lane, dim3(grid), dim3(lanes), 0, stream, d_in, d_in_off,
                           d_in_len, d_out, d_out_off, d_out_cap, d_result, nblocks);
        return hipGetLastError() == hipSuccess ? 0 : -7;
    }
    const uint32_t waves_per_wg = 4;
    const uint32_t grid = (nblocks + waves_per_wg - 1) / waves_per_wg;
    hipLaunchKernelGGL(zlz4::k_decompress_safe<true>, dim3(grid), dim3(64 * waves_per_wg), 0, stream, d_in, d_in_off,
                       d_in_len, d_out, d_out_off, d_out_cap, d_result, nblocks);
    return hipGetLastError() == hipSuccess ? 0 : -7;
}

// size pass: d_out may be null; d_out_off / d_out_cap still give (dummy offset 0, capacity) per block
extern "C" int zlz4_launch_decompress_sizes(hipStream_t stream, const uint8_t *d_in, const uint64_t *d_in_off,
                                            const uint32_t *d_in_len, const uint64_t *d_out_off,
                                            const uint32_t *d_out_cap, int64_t *d_result, uint32_t nblocks) {
    if (nblocks == 0) return 0;
    const uint32_t waves_per_wg = 4;
    const uint32_t grid = (nblocks + waves_per_wg - 1) / waves_per_wg;
    hipLaunchKernelGGL(zlz4::k_decompress_safe<false>, dim3(grid), dim3(64 * waves_per_wg), 0, stream, d_in, d_in_off,
                       d_in_len, (uint8_t *)nullptr, d_out_off, d_out_cap, d_result, nblocks);
    return hipGetLastError() == hipSuccess ? 0 : -7;
}
