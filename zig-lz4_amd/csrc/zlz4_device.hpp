// zlz4_device.hpp -- wave64 helpers shared by the gfx950 LZ4 kernels.
// One wavefront owns one independent LZ4 block; everything "scalar" about the
// block (ip, op, anchor, lengths) is wave-uniform and lives in SGPRs, the 64
// lanes are used for probing 64 hash positions at once, match-length
// extension (ballot + ffs) and byte copies.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

// Experiment / A-B knobs exist only in the diagnostic builds (-DZLZ4_TUNING: `make tuning`,
// `make stamps`); the shipped library reads no environment variable.
#ifdef ZLZ4_TUNING
static inline const char *zlz4_tune_env(const char *name) { return getenv(name); }
#else
static inline const char *zlz4_tune_env(const char *) { return nullptr; }
#endif

namespace zlz4 {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr uint32_t kMinMatch = 4;        // src/lz4.zig:12
constexpr uint32_t kLastLiterals = 5;    // src/lz4.zig:14
constexpr uint32_t kMfLimit = 12;        // src/lz4.zig:15
constexpr uint32_t kMaxInput = 0x7E000000u;  // src/lz4.zig:23
constexpr uint32_t kMaxDist = 65535u;    // src/lz4.zig:24-25
constexpr uint32_t kHashMul = 2654435761u;   // src/lz4.zig:44

constexpr int64_t kErrOutputTooSmall = -1;   // lz4.Error order, src/lz4.zig:48-55
constexpr int64_t kErrInputTooLarge = -2;
constexpr int64_t kErrCorrupted = -3;
constexpr int64_t kErrInvalidState = -5;   // also: a batch block longer than the call's max_in_len (include/zlz4_amd.h)

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ uint32_t first_lane(uint64_t m) { return (uint32_t)__ffsll((long long)m) - 1u; }  // m != 0
// NOTE: call it from wave-uniform control flow only -- ds_bpermute reads 0 from lanes that are masked off.
__device__ __forceinline__ uint32_t shfl(uint32_t v, uint32_t src_lane) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}

// write the wave-uniform `val` into lane `l` (uniform) of `old` (v_cmp + v_cndmask; VALU has headroom here)
__device__ __forceinline__ uint32_t wrlane(uint32_t val, uint32_t l, uint32_t old) {
    return (__lane_id() == l) ? val : old;
}

__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ u32x4 ld128(const uint8_t *p) { u32x4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st128(uint8_t *p, u32x4 v) { __builtin_memcpy(p, &v, 16); }

// index of the first differing byte of two 16-byte vectors, 16 if equal
__device__ __forceinline__ uint32_t first_diff16(u32x4 a, u32x4 b) {
    const uint32_t x0 = a.x ^ b.x, x1 = a.y ^ b.y, x2 = a.z ^ b.z, x3 = a.w ^ b.w;
    if (x0) return (uint32_t)__builtin_ctz(x0) >> 3;
    if (x1) return 4u + ((uint32_t)__builtin_ctz(x1) >> 3);
    if (x2) return 8u + ((uint32_t)__builtin_ctz(x2) >> 3);
    if (x3) return 12u + ((uint32_t)__builtin_ctz(x3) >> 3);
    return 16u;
}

// the same with selects only (no divergent branches: for per-lane loops whose cost is scalar exec-mask handling)
__device__ __forceinline__ uint32_t first_diff16_sel(u32x4 a, u32x4 b) {
    const uint32_t x0 = a.x ^ b.x, x1 = a.y ^ b.y, x2 = a.z ^ b.z, x3 = a.w ^ b.w;
    const uint32_t lo = x0 ? x0 : x1, hi = x2 ? x2 : x3;
    const uint32_t blo = x0 ? 0u : 4u, bhi = x2 ? 8u : 12u;
    const bool in_lo = (x0 | x1) != 0;
    const uint32_t x = in_lo ? lo : hi;
    const uint32_t base = in_lo ? blo : bhi;
    return x ? base + ((uint32_t)__builtin_ctz(x) >> 3) : 16u;
}

// Wave-cooperative forward copy of n bytes (n wave-uniform).  Chunks of 1 KiB
// (16 B per lane, unaligned dwordx4) in increasing address order, then a byte
// tail.  Safe for dst/src in the same buffer when dst - src >= 1024 or the
// ranges do not overlap.
__device__ __forceinline__ void copy_bytes(uint8_t *dst, const uint8_t *src, uint32_t n, uint32_t lane) {
    uint32_t k = lane * 16u;
    for (; k + 16u <= n; k += 1024u) st128(dst + k, ld128(src + k));
    const uint32_t t0 = n & ~15u;
    if (t0 + lane < n && lane < 16u) dst[t0 + lane] = src[t0 + lane];
}

// number of extension bytes the LZ4 length code needs for a value v >= 15 (token nibble saturated)
__device__ __forceinline__ uint32_t ext_len_bytes(uint32_t v) { return v >= 15u ? 1u + (v - 15u) / 255u : 0u; }

// Wave-cooperative write of the 255-run length extension of value v (>= 15) at p:
// (v-15)/255 bytes of 255 followed by (v-15)%255.  src/lz4.zig:368-382, :416-429.
__device__ __forceinline__ void write_ext_len(uint8_t *p, uint32_t v, uint32_t lane) {
    const uint32_t rem = v - 15u;
    const uint32_t n255 = rem / 255u;
    for (uint32_t k = lane; k < n255; k += 64u) p[k] = 255;
    if (lane == 0) p[n255] = (uint8_t)(rem - n255 * 255u);
}

}  // namespace zlz4
