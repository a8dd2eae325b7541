// zlz4_compress_hc.hip -- LZ4-HC hash-chain compressor (levels 3..9) for gfx950.
//
// Replaces lz4hc.compressHC -> compressHCExtState -> compressHashChain
// (reference src/lz4hc.zig:1440-1489, :976-1064) with its helpers insertHC (:491-510),
// insertAndGetWiderMatch (:538-681), lz4Count (:234-264), countPattern /
// reverseCountPattern / isRepetitivePattern (:170-228), encodeSequence (:308-386),
// encodeLiterals (:1394-1425).  Output is byte-identical to the Zig algorithm
// (SURVEY.md Appendix A.2, quirks H1-H8).
//
// Why this is not a line-by-line port
// -----------------------------------
// In the one-shot path the tables start zeroed (Context.init, :405-419) and insertHC
// inserts EVERY position below the one being searched, whatever the parse did.  So the
// table state a search at position p sees is a pure function of the input:
//     hashTable[h(p)]  at the time p is searched = prev(p) = the last q < p with h(q) == h(p)
//     chainTable[q]                                = min(q - prev(q), 65535)        (prev = 0 if none)
// That turns the serial insert/search loop into three data-parallel passes:
//   K1 k_hc_build_links   one block per CU, four wavefronts: insertHC for 256 positions per step as ONE LDS atomic max
//                         with return per 64 positions (128 KiB 32-bit hash table in LDS); writes link[q] to HBM.
//   K2s k_hc_seg_search   (levels 3-9) one 1024-lane workgroup per block, the block's links in LDS.  The greedy parse is
//                         a walk in a functional graph (next(p) = p + len(p) or p + 1), so walks that start every 32
//                         positions run speculatively, mark what they search, and stop where somebody else has been;
//                         the walk from position 0 is the parse.  Only positions on some walk are searched (:571-622,
//                         + the level-9 pattern analysis :626-678), four candidates per trip, matches stored per position.
//                         Long counts are done by the wavefront and their outcome (a counted run) is shared by the
//                         block's walks; the walk from 0 hands the parse over when it merges into another walk.
//   K2  k_hc_search       (levels 10-12, whose price-based parse looks results up everywhere; and blocks > 64 KiB at
//                         every level's A/B switch) one LANE per position walks its chain; a candidate is only counted
//                         if its byte best_len matches, a search ends when its match reaches iHighLimit, counts beyond
//                         64 bytes are done by the wavefront and shared by its lanes.
//   K3  k_hc_parse_emit   one wavefront per block: the greedy walk of :1009-1032 over the stored results,
//                         encodeSequence with its limitedOutput checks, final literals; runs on a side stream beside
//                         K1 / K2s of the next round.
#include <cstdlib>
#include <type_traits>

#include "zlz4_device.hpp"

#ifdef ZLZ4_STAMPS
__device__ unsigned long long g_zlz4_hstamps[16];
extern "C" int zlz4_debug_read_hstamps(unsigned long long *out4) {
    return hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_zlz4_hstamps), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -7;
}
#endif

namespace zlz4 {

constexpr uint32_t kHcHashLog = 15;                 // src/lz4hc.zig:37
constexpr uint32_t kHcTableSize = 1u << kHcHashLog; // :38
__device__ __forceinline__ uint32_t hash_hc(uint32_t seq) { return (seq * kHashMul) >> (32 - kHcHashLog); }   // :129-131

// T = uint16_t: blocks whose positions fit 16 bits; link[q] = q - prev(q)  (exact delta, 0 only for q == 0)
// T = uint32_t: any block;                           link[q] = prev(q)      (0 = none)
template <typename T> struct Links;
template <> struct Links<uint16_t> {
    static __device__ __forceinline__ uint16_t make(uint32_t q, uint32_t prev) { return (uint16_t)(q - prev); }
    static __device__ __forceinline__ uint32_t first(uint32_t p, uint16_t l) { return p - l; }        // hashTable[h(p)]
    static __device__ __forceinline__ uint32_t delta(uint32_t, uint16_t l) { return l; }              // chainTable[m]
};
template <> struct Links<uint32_t> {
    static __device__ __forceinline__ uint32_t make(uint32_t, uint32_t prev) { return prev; }
    static __device__ __forceinline__ uint32_t first(uint32_t, uint32_t l) { return l; }
    static __device__ __forceinline__ uint32_t delta(uint32_t m, uint32_t l) {                        // :502-503 clamp
        const uint32_t d = m - l;
        return d > kMaxDist ? kMaxDist : d;
    }
};

// ------------------------------------------------------------------ K1: chain links
// number of positions that can be searched or inserted: p in [0, n-12]  (:1009 `ip <= mflimit`)
__device__ __forceinline__ uint32_t n_positions(uint32_t n) { return n < kMfLimit + 1u ? 0u : n - kMfLimit + 1u; }

// insertHC for every position, 256 positions per step.  What the reference does one position at a time --
//     prev = hashTable[h]; chainTable[idx] = idx - prev; hashTable[h] = idx          (:499-505)
// -- is ONE LDS instruction per 64 positions here: atomic max with return.  Positions grow with the lane index, so if the
// LDS unit serves the lanes of an instruction that hit the same slot in ascending lane order, every lane gets back the
// largest smaller position with its hash (or the value from before the step) = prev, and the slot ends up holding the
// last one.  That order is not promised by the ISA, so it is checked, not assumed: a lane served before a lower lane of
// its group leaves a value >= that lane's position in the slot, i.e. some lane reads old >= q (every pre-step value is
// smaller than every position of the step).  `old < q` on all lanes therefore PROVES the ascending order; otherwise the
// group is resolved by ballot (prev of a member = the nearest lower member, else the smallest value anyone in the group
// read, which is the pre-step value).  The 32768-entry table is 32-bit for every block size (LDS atomics are): 128 KiB,
// one block per CU.  Four wavefronts share the block: wavefront w takes the steps s = w (mod 4) and does everything of
// a step -- input bytes (requested kDepth of its steps ahead), hashes, order check, link staging -- on its own; only the
// four atomics of a step wait for their turn (an LDS counter: step s goes when the atomics of step s - 1 have RETURNED),
// which is what keeps the insertions in position order.  k_hc_parse_emit of the previous round fills the rest of the
// machine.
constexpr uint32_t kLinkWaves = 4;
template <typename T>
__global__ __launch_bounds__(64 * kLinkWaves) void k_hc_build_links(const uint8_t *__restrict__ d_in,
                                                                     const uint64_t *__restrict__ d_in_off,
                                                                     const uint32_t *__restrict__ d_in_len,
                                                                     T *__restrict__ d_link, uint64_t link_stride,
                                                                     uint32_t blk0, uint32_t nblocks, uint32_t max_in_len) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = rfl(threadIdx.x >> 6);
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const uint8_t *src = d_in + d_in_off[blk0 + b];
    const uint32_t n = rfl(d_in_len[blk0 + b]);
    if (n > max_in_len) return;                         // the workspace stride comes from max_in_len; K3 reports it
    const uint32_t np = n_positions(n);
    T *link = d_link + (uint64_t)b * link_stride;
    typedef __attribute__((address_space(3))) uint32_t lds_slot;
    typedef __attribute__((address_space(3))) T lds_link;
    typedef __attribute__((address_space(3))) volatile uint32_t lds_turn;
    lds_slot *table = (lds_slot *)lds_raw;
    // links are staged in LDS and written out 4096 at a time with 16-byte stores: a global store per step would sit in
    // the same in-order queue as the prefetch loads and make every step wait for HBM writes
    constexpr uint32_t kStage = 4096;
    lds_link *stage = (lds_link *)(lds_raw + kHcTableSize * 4u);
    lds_turn *turn = (lds_turn *)(lds_raw + kHcTableSize * 4u + kStage * sizeof(T));
    {
        u32x4 z = {0, 0, 0, 0};
        u32x4 *t4 = reinterpret_cast<u32x4 *>(lds_raw);
        for (uint32_t k = threadIdx.x; k < kHcTableSize * 4u / 16u; k += 64u * kLinkWaves) t4[k] = z;   // Context.init :405-419
        if (threadIdx.x == 0) *turn = 0;
    }
    __syncthreads();
    const uint64_t lane_bit = 1ull << lane, lanes_below = lane_bit - 1ull;
    constexpr uint32_t kSub = 4;                        // 64-position groups per step
    constexpr uint32_t kStep = 64u * kSub;              // positions per step
    constexpr uint32_t kDepth = 2;                      // steps of THIS wavefront whose bytes are in flight
    constexpr uint32_t kStepsPerChunk = kStage / kStep; // 16 steps fill the staging area: 4 per wavefront
    uint32_t seq_pf[kDepth][kSub];
#pragma unroll
    for (uint32_t d = 0; d < kDepth; d++)
#pragma unroll
        for (uint32_t j = 0; j < kSub; j++) {
            const uint32_t q0 = kStep * (wave + kLinkWaves * d) + 64u * j + lane;
            seq_pf[d][j] = q0 < np ? ld32(src + q0) : 0u;
        }
    const uint32_t nsteps = (np + kStep - 1u) / kStep;
    for (uint32_t c0 = 0; c0 < nsteps; c0 += kStepsPerChunk) {
        // the steps of this chunk that belong to this wavefront: c0 + wave, + 4, + 8, + 12  (kDepth divides their number)
#pragma unroll
        for (uint32_t k = 0; k < kStepsPerChunk / kLinkWaves; k++) {
            const uint32_t d = k % kDepth;
            const uint32_t step = c0 + wave + kLinkWaves * k;
            if (step >= nsteps) break;
            const uint32_t base = step * kStep;
            uint32_t seq[kSub], old[kSub];
#pragma unroll
            for (uint32_t j = 0; j < kSub; j++) seq[j] = seq_pf[d][j];
#pragma unroll
            for (uint32_t j = 0; j < kSub; j++) {       // the bytes of this wavefront's step kDepth steps ahead
                const uint32_t qn = base + kStep * kLinkWaves * kDepth + 64u * j + lane;
                if (qn < np) seq_pf[d][j] = ld32(src + qn);
            }
            uint32_t hsh[kSub];
#pragma unroll
            for (uint32_t j = 0; j < kSub; j++) hsh[j] = hash_hc(seq[j]);
            while (*turn != step) __builtin_amdgcn_s_sleep(1);       // the atomics of step - 1 are done
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (uint32_t j = 0; j < kSub; j++) {       // in position order: LDS operations of a wavefront keep their order
                const uint32_t q = base + 64u * j + lane;
                old[j] = 0;
                if (q < np) old[j] = __hip_atomic_fetch_max(table + hsh[j], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the hand-over stays behind the four atomics
            if (lane == 0) *turn = step + 1u;
#pragma unroll
            for (uint32_t j = 0; j < kSub; j++) {
                const uint32_t q = base + 64u * j + lane;
                const bool active = q < np;
                uint32_t prev = old[j];
                uint64_t viol = ballot(active && prev >= q && q != 0u);   // (position 0 reads the empty slot: prev = 0 is right)
                if (viol) {
                    // the lanes of some group were not served in ascending order: resolve those groups by ballot
                    const uint32_t h = hsh[j];
                    while (viol) {
                        const uint32_t hh = rdlane(h, first_lane(viol));
                        const uint64_t same = ballot(active && h == hh);
                        uint32_t pre = 0xFFFFFFFFu;       // the pre-step value: the smallest value any member read
                        for (uint64_t r = same; r; r &= r - 1ull) {
                            const uint32_t o = rdlane(old[j], first_lane(r));
                            pre = o < pre ? o : pre;
                        }
                        if (active && h == hh) {
                            const uint64_t below = same & lanes_below;
                            prev = below ? base + 64u * j + (63u - (uint32_t)__clzll((long long)below)) : pre;
                        }
                        viol &= ~same;
                    }
                }
                if (active) stage[q & (kStage - 1u)] = Links<T>::make(q, prev);   // :502-504 (clamp applied on read for T = u32)
            }
        }
        __syncthreads();                                 // every step of the chunk is staged
        {
            const uint32_t cb = c0 * kStep;                                     // first position of the staged chunk
            const uint32_t ce = cb + kStage < np ? cb + kStage : np;
            const uint32_t n16 = ((ce - cb) * (uint32_t)sizeof(T) + 15u) >> 4;  // (the link array is padded to 16 entries)
            const u32x4 *s4 = reinterpret_cast<const u32x4 *>(lds_raw + kHcTableSize * 4u);
            u32x4 *g4 = reinterpret_cast<u32x4 *>(link + cb);
            for (uint32_t k = threadIdx.x; k < n16; k += 64u * kLinkWaves) g4[k] = s4[k];
        }
        __syncthreads();                                 // the staging area may be overwritten
    }
}

// ------------------------------------------------------------------ K2: best match per position
// lz4Count (:234-264): common prefix length of a.. and b.. with a < limit
__device__ __forceinline__ uint32_t lz4_count(const uint8_t *src, uint32_t a, uint32_t b, uint32_t limit) {
    uint32_t c = 0;
    while (a + 16u <= limit) {                          // (16 bytes per round trip of this lane's serial chain; b < a)
        const uint32_t d = first_diff16_sel(ld128(src + a), ld128(src + b));
        if (d < 16u) return c + d;
        a += 16; b += 16; c += 16;
    }
    while (a + 4u <= limit) {
        const uint32_t x = ld32(src + a) ^ ld32(src + b);
        if (x) return c + ((uint32_t)__builtin_ctz(x) >> 3);
        a += 4; b += 4; c += 4;
    }
    while (a < limit && src[a] == src[b]) { a++; b++; c++; }
    return c;
}
// countPattern (:170-199) for a pattern that passed isRepetitivePattern (all four bytes equal)
__device__ __forceinline__ uint32_t count_pattern(const uint8_t *src, uint32_t a, uint32_t end, uint32_t pattern) {
    uint32_t c = 0;
    // (this loop is one lane's serial chain: 64 bytes per round trip while the run lasts -- a block of one repeated byte
    //  counts 2 x 64 KiB here)
    while (a + 64u <= end) {
        const u32x4 v0 = ld128(src + a), v1 = ld128(src + a + 16u), v2 = ld128(src + a + 32u), v3 = ld128(src + a + 48u);
        const uint32_t d = (v0.x ^ pattern) | (v0.y ^ pattern) | (v0.z ^ pattern) | (v0.w ^ pattern) |
                           (v1.x ^ pattern) | (v1.y ^ pattern) | (v1.z ^ pattern) | (v1.w ^ pattern) |
                           (v2.x ^ pattern) | (v2.y ^ pattern) | (v2.z ^ pattern) | (v2.w ^ pattern) |
                           (v3.x ^ pattern) | (v3.y ^ pattern) | (v3.z ^ pattern) | (v3.w ^ pattern);
        if (d) break;                                   // the 16-byte loop below finds the byte
        a += 64; c += 64;
    }
    while (a + 16u <= end) {
        const u32x4 v = ld128(src + a);
        const uint32_t x0 = v.x ^ pattern, x1 = v.y ^ pattern, x2 = v.z ^ pattern, x3 = v.w ^ pattern;
        if (x0 | x1 | x2 | x3) {
            if (x0) return c + ((uint32_t)__builtin_ctz(x0) >> 3);
            if (x1) return c + 4u + ((uint32_t)__builtin_ctz(x1) >> 3);
            if (x2) return c + 8u + ((uint32_t)__builtin_ctz(x2) >> 3);
            return c + 12u + ((uint32_t)__builtin_ctz(x3) >> 3);
        }
        a += 16; c += 16;
    }
    while (a + 4u <= end) {
        const uint32_t x = ld32(src + a) ^ pattern;
        if (x) return c + ((uint32_t)__builtin_ctz(x) >> 3);
        a += 4; c += 4;
    }
    const uint8_t pb = (uint8_t)pattern;
    while (a < end && src[a] == pb) { a++; c++; }
    return c;
}
// reverseCountPattern (:202-222), same restriction; counts bytes equal to the pattern byte below `a`, down to 0
__device__ __forceinline__ uint32_t reverse_count_pattern(const uint8_t *src, uint32_t a, uint32_t pattern) {
    uint32_t c = 0;
    const uint8_t pb = (uint8_t)pattern;
    while (a >= 64u) {                                  // (64 bytes per round trip, as in count_pattern)
        const u32x4 v0 = ld128(src + a - 16u), v1 = ld128(src + a - 32u), v2 = ld128(src + a - 48u), v3 = ld128(src + a - 64u);
        if ((v0.x ^ pattern) | (v0.y ^ pattern) | (v0.z ^ pattern) | (v0.w ^ pattern) |
            (v1.x ^ pattern) | (v1.y ^ pattern) | (v1.z ^ pattern) | (v1.w ^ pattern) |
            (v2.x ^ pattern) | (v2.y ^ pattern) | (v2.z ^ pattern) | (v2.w ^ pattern) |
            (v3.x ^ pattern) | (v3.y ^ pattern) | (v3.z ^ pattern) | (v3.w ^ pattern)) break;
        a -= 64; c += 64;
    }
    while (a >= 16u) {
        const u32x4 v = ld128(src + a - 16u);
        if ((v.x ^ pattern) | (v.y ^ pattern) | (v.z ^ pattern) | (v.w ^ pattern)) break;
        a -= 16; c += 16;
    }
    while (a >= 4u && ld32(src + a - 4u) == pattern) { a -= 4; c += 4; }
    while (a > 0 && src[a - 1u] == pb) { a--; c++; }
    return c;
}

// Wave-cooperative lz4Count (:234-264) for one long match: bytes equal at a + k / b + k, k >= from, while a + k < limit;
// 16 bytes per lane and step (1 KiB per step).  All 64 lanes must call it with the same arguments.
__device__ __forceinline__ uint32_t hc_coop_count(const uint8_t *__restrict__ src, uint32_t a, uint32_t b, uint32_t from,
                                                  uint32_t limit, uint32_t n, uint32_t lane) {
    uint32_t k = from;
    for (;;) {
        const uint32_t pa = a + k + lane * 16u;
        uint32_t cmp = 0, d = 0;                          // bytes this lane may compare / equal bytes found
        if (pa < limit) cmp = limit - pa < 16u ? limit - pa : 16u;
        if (cmp) {
            const uint32_t pb = b + k + lane * 16u;
            if (pa + 16u <= n) { d = first_diff16_sel(ld128(src + pa), ld128(src + pb)); d = d < cmp ? d : cmp; }
            else while (d < cmp && src[pa + d] == src[pb + d]) d++;
        }
        const uint64_t stop = ballot(d < 16u);            // mismatch or limit inside this lane's chunk
        if (stop) {
            const uint32_t sl = first_lane(stop);
            return k + sl * 16u + rdlane(d, sl);
        }
        k += 1024u;
    }
}

// One lane per position, each lane runs its own chain walk.  (A per-wave work-queue variant that re-assigns
// finished lanes and tests one candidate per loop iteration with 16-byte compares was measured 15-40 % SLOWER
// on MI355X: this kernel is bound by gather throughput, not by divergence -- see DESIGN.md section 6.)
// the patternAnalysis step that follows the chain walk (:626-676); m = where the walk stopped
template <typename T, typename LinkPtr>
__device__ __forceinline__ void hc_pattern_step(const uint8_t *src, LinkPtr link, uint32_t p, uint32_t m, uint32_t pattern,
                                                uint32_t lowest, uint32_t limit, int32_t &best_len, uint32_t &best_off) {
    {
        // (:626 patternAnalysis is checked by the caller; the register test of :629-631 goes first so that the link is
        //  only loaded for a four-equal-bytes pattern -- both conditions are pure)
        if (best_len > 0 && ((pattern & 0xFFFFu) == (pattern >> 16)) && ((pattern & 0xFFu) == (pattern >> 24))) {
            const uint32_t delta = Links<T>::delta(m, link[m]);  // :627 (m == 0 -> link[0] -> delta 0)
            if (delta == 1) {
                const uint32_t src_pat_len = count_pattern(src, p + 4u, limit, pattern) + 4u;   // :633
                const uint32_t cand = m - 1u;                    // :636
                if (cand >= lowest) {                            // :637 (dictIdx == 0)
                    if (ld32(src + cand) == pattern) {           // :644
                        const uint32_t fwd_len = count_pattern(src, cand + 4u, limit, pattern) + 4u;   // :646
                        const uint32_t back_len = reverse_count_pattern(src, cand, pattern);           // :650
                        uint32_t lo = cand - back_len;           // :653
                        if (lo < lowest) lo = lowest;
                        const uint32_t lim_back = cand - lo;
                        const uint32_t seg_len = lim_back + fwd_len;                                   // :654
                        const int32_t max_ml = (int32_t)(seg_len < src_pat_len ? seg_len : src_pat_len);   // :658
                        uint32_t new_m;
                        if (seg_len >= src_pat_len && fwd_len <= src_pat_len) new_m = cand + fwd_len - src_pat_len;   // :660-662
                        else new_m = cand - lim_back;            // :665
                        if (max_ml > best_len && (p - new_m) <= kMaxDist) {   // :669
                            best_len = max_ml;
                            best_off = p - new_m;
                        }
                    }
                }
            }
        }
        }
}

template <typename T, typename R>   // R = packed result: u32 (len | off << 16) for T = u16, u64 (len | off << 32) otherwise
__global__ __launch_bounds__(256) void k_hc_search(const uint8_t *__restrict__ d_in,
                                                    const uint64_t *__restrict__ d_in_off,
                                                    const uint32_t *__restrict__ d_in_len,
                                                    const T *__restrict__ d_link, uint64_t link_stride,
                                                    R *__restrict__ d_res, uint32_t blk0, uint32_t nblocks,
                                                    int32_t max_attempts, int32_t force_pattern_analysis,
                                                    uint32_t max_in_len) {
    const uint32_t b = blockIdx.y;
    if (b >= nblocks) return;
    const uint32_t n = d_in_len[blk0 + b];
    if (n > max_in_len) return;
    const uint32_t np = n_positions(n);
    if (blockIdx.x * blockDim.x >= np) return;                   // (whole workgroups only: the long counts below take a full wavefront)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t p_raw = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = p_raw < np;
    const uint32_t p = valid ? p_raw : 0u;                       // (lanes past the last position idle along at position 0)
    const uint8_t *src = d_in + d_in_off[blk0 + b];
    const T *link = d_link + (uint64_t)b * link_stride;
    const uint32_t limit = n - kLastLiterals;                    // iHighLimit = matchlimit :989, :1011
    // :983 for the hash-chain levels; compressOptimal always passes patternAnalysis = true (:1123, :1201)
    const bool pattern_analysis = max_attempts > 128 || force_pattern_analysis != 0;

    // insertAndGetWiderMatch with iLowLimit == ip, longest = MINMATCH-1 (:522-534)
    const uint32_t lowest = p < 65536u ? 0u : p - kMaxDist;      // :553-554 (lowLimit == 0)
    const uint32_t pattern = ld32(src + p);                      // :558
    int32_t best_len = (int32_t)kMinMatch - 1;                   // :560
    uint32_t best_off = 0;
    uint32_t m = valid ? Links<T>::first(p, link[p]) : 0u;       // :563  hashTable[hashPtr(ip)]
    const bool searched = m != 0;                                // :566-568
    int32_t nb = max_attempts;
    bool at_limit = false;
    // the 16 bytes at p stay in registers: one 16-byte gather per candidate gives the 4-byte test (:586) and the
    // first 12 bytes of lz4Count (:588) together
    const bool wide = p + 16u <= n;
    u32x4 p16 = {pattern, 0, 0, 0};
    if (wide) p16 = ld128(src + p);
    const uint32_t avail = limit - p;                            // lz4Count stops at iHighLimit
    {
        // Same exits and the same final m as the reference loop: candidate out of range (:573), attempts used up (:571), a
        // match longer than nbAttempts (:613), end of chain (:620).  The loop is wave-uniform (a wavefront goes round until
        // its last lane is done either way): a candidate that has matched 64 bytes is counted by the whole wavefront, 1 KiB
        // per step, and the counted run (distance, from, end) is kept for the lanes that follow -- 64 consecutive positions
        // inside a run of zeros, or in a block with a period, all ask for the same run.
#ifdef ZLZ4_STAMPS
        const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
        unsigned long long t_coop = 0;
#endif
        bool go = searched && wide && m <= p && (p - m) <= kMaxDist;   // :571 (m > 0, nb > 0 here), :573
        uint8_t pb = 0;                                          // src[p + best_len] while best_len >= 16
        [[maybe_unused]] bool in_run = false;                    // the last candidate failed the byte test and its link is 1
        uint32_t run_d = 0, run_from = 0, run_end = 0;           // wave-uniform
#ifdef ZLZ4_STAMPS
        unsigned long long my_steps = 0, wave_iters = 0, coop_counts = 0, coop_steps = 0, long_lanes = 0;
#endif
        T lk = 0;
        auto finish = [&](int32_t mlt) {                         // the candidate at m is done: :579, :607-:621
            mlt = m >= lowest ? mlt : 0;                         // :579
            // back == 0: `ip > iLowLimit` is false (:596)
            const bool better = mlt > best_len;                  // :607 (mlt == 0 when the 4 bytes differ)
            best_len = better ? mlt : best_len;
            best_off = better ? p - m : best_off;
            const uint32_t delta = Links<T>::delta(m, lk);       // :619
            // (bitwise, not short-circuit: no branches)
            const bool stop = (bool)((int)better & (int)(mlt > max_attempts)) | (delta == 0) | (delta > m);   // :613, :620
            m = stop ? m : m - delta;                            // :621
            go = (bool)((int)!stop & (int)(nb > 0) & (int)(m > 0) & (int)((p - m) <= kMaxDist));
            if (better) {
                // the match ends at iHighLimit: nothing can be longer (lz4Count stops there, :234; the pattern step's
                // maxML is at most the pattern's length from p, which ends there too, :658): the result is final
                at_limit = (uint32_t)best_len >= avail;
                go = go && !at_limit;
                if (go && (uint32_t)best_len >= 16u) pb = src[p + (uint32_t)best_len];
            }
        };
        while (ballot(go)) {
            const bool active = go;
            bool need_long = false;
            int32_t mlt = 0;
            if (active) {
#ifdef ZLZ4_STAMPS
                my_steps++;
#endif
                nb -= 1;                                         // :577
                lk = link[m];                                    // chain link, fetched together with the candidate bytes
                // A candidate only matters if it matches MORE than best_len bytes (:607), i.e. if its byte best_len matches
                // too: that byte is fetched with the candidate, and lz4Count (:588, as long as the match) is only run
                // when it can change something.  (Without this a run of one byte value -- every candidate of every
                // position in it matches to the end of the run -- costs run length x attempts byte compares per position;
                // the reference's parser never searches inside a match longer than sufficient_len, this kernel searches
                // every position.)
                const uint32_t bl = (uint32_t)best_len;
                const bool probe = bl >= 16u && bl < avail;
                const uint8_t mb = probe ? src[m + bl] : (uint8_t)0;
                const uint32_t d = first_diff16_sel(p16, ld128(src + m));   // m < p, so m + 16 <= n too
                mlt = d >= kMinMatch ? (int32_t)(d < avail ? d : avail) : 0;   // :586, :588
                in_run = false;
                if (d == 16u && avail > 16u) {
                    if (probe && mb != pb) { mlt = 16; in_run = Links<T>::delta(m, lk) == 1u; }   // <= best_len: not better, whatever its length
                    else {
                        const uint32_t cap = avail > 64u ? p + 64u : limit;
                        const uint32_t t = 16u + lz4_count(src, p + 16u, m + 16u, cap);
                        need_long = t == 64u && avail > 64u;
                        mlt = (int32_t)t;
                    }
                }
            }
#ifdef ZLZ4_STAMPS
            wave_iters++;
#endif
            if (uint64_t lm = ballot(need_long)) {
                uint32_t long_total = 0;
                while (lm) {
                    const uint32_t L = first_lane(lm);
                    lm &= lm - 1ull;
                    const uint32_t P = rdlane(p, L), M = rdlane(m, L), dist = P - M;
                    uint32_t total;
                    if (run_d == dist && run_from <= P + 64u && P + 64u <= run_end) total = run_end - P;
                    else {
#ifdef ZLZ4_STAMPS
                        const unsigned long long tc0 = __builtin_amdgcn_s_memtime();
#endif
                        total = hc_coop_count(src, P, M, 64u, limit, n, lane);
                        run_d = dist; run_from = P + 64u; run_end = P + total;
#ifdef ZLZ4_STAMPS
                        t_coop += __builtin_amdgcn_s_memtime() - tc0;
                        coop_counts += 1; coop_steps += (total - 64u) / 1024u + 1u;
#endif
                    }
#ifdef ZLZ4_STAMPS
                    long_lanes += 1;
#endif
                    if (lane == L) long_total = total;
                }
                if (need_long) mlt = (int32_t)long_total;
            }
            if (active) finish(mlt);
#ifdef ZLZ4_EXPERIMENT_RUN_SKIP   // measured, not shipped: level 12 on D-mixed 11.1 s -> 4.6 s per GiB, on D-text 396 -> 416 ms (profiles/r03_hc_runs.md)
            if constexpr (sizeof(T) == 2) {
                // Inside a run of one byte value the chain is p-1, p-2, p-3 ... (every link is 1) and, once the first
                // candidate has given the length of the run, every further one fails the byte test above: eight of them
                // per round trip here -- their eight links are one 16-byte load, their eight test bytes one 8-byte load.
                // Each counts as an attempt (:577) that changes nothing else, exactly as one by one.  (Off the loop's
                // straight path: on text the flag is never set.)
                if (in_run && go) {
                    while (m >= 9u && nb > 8 && (p - (m - 8u)) <= kMaxDist) {
                        const u32x4 l8 = ld128(reinterpret_cast<const uint8_t *>(link + (m - 7u)));
                        unsigned long long t8;
                        __builtin_memcpy(&t8, src + (m - 7u) + (uint32_t)best_len, 8);
                        const unsigned long long x = t8 ^ (0x0101010101010101ull * pb);      // a zero byte = a test byte that matches
                        const bool none = ((x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull) == 0ull;
                        if (!(none && l8.x == 0x00010001u && l8.y == 0x00010001u && l8.z == 0x00010001u && l8.w == 0x00010001u)) break;
                        nb -= 8; m -= 8u;
#ifdef ZLZ4_STAMPS
                        my_steps++;
#endif
                    }
                }
            }
#endif
        }
#ifdef ZLZ4_STAMPS
        atomicAdd(&g_zlz4_hstamps[0], my_steps);
        if (lane == 0) {
            atomicAdd(&g_zlz4_hstamps[1], wave_iters * 64ull); atomicAdd(&g_zlz4_hstamps[2], coop_counts);
            atomicAdd(&g_zlz4_hstamps[3], coop_steps); atomicAdd(&g_zlz4_hstamps[4], long_lanes);
            atomicAdd(&g_zlz4_hstamps[5], __builtin_amdgcn_s_memtime() - t_begin); atomicAdd(&g_zlz4_hstamps[6], t_coop);
        }
#endif
    }
    if (searched && !wide) {
        while (m > 0 && nb > 0) {                                // :571 (the last <= 4 positions of a block)
            if (m > p || (p - m) > kMaxDist) break;              // :573
            nb -= 1;                                             // :577
            const T lk = link[m];
            if (m >= lowest) {                                   // :579
                int32_t mlt = 0;
                if (ld32(src + m) == pattern) {                  // :586
                    mlt = (int32_t)(kMinMatch + lz4_count(src, p + kMinMatch, m + kMinMatch, limit));
                }
                // back == 0: `ip > iLowLimit` is false (:596)
                if (mlt > best_len) {                            // :607 (mlt == 0 when the 4 bytes differ)
                    best_len = mlt;
                    best_off = p - m;
                    if (mlt > max_attempts) break;               // :613
                    // (the result is final at the limit, as above: in a block with a short period these last positions
                    //  otherwise walk nbAttempts candidates that all match to the limit -- 16 384 dependent round trips
                    //  of one lane at level 12, the straggler of every block)
                    if ((uint32_t)mlt >= avail) { at_limit = true; break; }
                }
            }
            const uint32_t delta = Links<T>::delta(m, lk);       // :619
            if (delta == 0 || delta > m) break;                  // :620
            m -= delta;                                          // :621
        }
    }
    if (searched && pattern_analysis && !at_limit) hc_pattern_step<T, const T *>(src, link, p, m, pattern, lowest, limit, best_len, best_off);
    if (!valid) return;
    R r;
    if (sizeof(R) == 4) r = (R)((uint32_t)best_len | (best_off << 16));
    else r = (R)((uint64_t)(uint32_t)best_len | ((uint64_t)best_off << 32));
    d_res[(uint64_t)b * link_stride + p] = r;
}


// ------------------------------------------------------------------ K2s: parse-aware search (levels 3..9, blocks <= 64 KiB)
// The greedy parse (:1009-1032) is a walk in a functional graph: next(p) = p + len(p) when the search at p finds a match,
// p + 1 otherwise, and len(p) is a pure function of the input (see the file header).  Two walks that meet stay together,
// and on real data walks that start a few bytes apart meet within a sequence or two (both matches end where the repeated
// string ends; inside a literal run both visit every byte).  That makes the parse itself data-parallel:
//   * the block is cut into start points every `seg_len` positions; a LANE takes the next start point from a counter in
//     LDS and runs the reference's serial loop from there, as if the parse arrived at that byte (speculation);
//   * every position a walk visits is marked in an LDS bitmap (atomic or); a walk ends at the first position that is
//     already marked -- whoever marked it continues from there, so nothing is lost -- or at the end of the block.
// The true parse (the walk from 0) is then the union of pieces that have all been searched, and K3 follows it through
// res[] (zeroed by K1; only matches are stored).  Only visited positions are searched: 19 % of the positions and 5 % of
// the chain steps of D-text at level 9 (the long chains belong to positions inside matches, which the parse never looks
// at).  The block's chain links (u16 x 65536 = 128 KiB) stay in LDS for the whole search; one workgroup per block.
// A loop trip is one memory round trip for every lane: either the next candidate of its chain (16-byte compare) or the
// next 16 bytes of a candidate that matched so far.
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;

// Wave-cooperative countPattern (:170-199) / reverseCountPattern (:202-222) for a four-equal-bytes pattern, 1 KiB per step
// (count_pattern / reverse_count_pattern above are one lane's serial chain: a block of one repeated byte is 2 x 64 KiB of
// it, during which the 15 other wavefronts of the block wait for that lane -- level 9 on D-zero, round 3).  Wave-uniform
// arguments.
__device__ __forceinline__ uint32_t hc_coop_count_pattern(const uint8_t *__restrict__ src, uint32_t a, uint32_t end, uint32_t pattern,
                                                          uint32_t n, uint32_t lane) {
    const u32x4 pat4 = {pattern, pattern, pattern, pattern};
    const uint8_t pb = (uint8_t)pattern;
    uint32_t k = 0;
    for (;;) {
        const uint32_t pa = a + k + lane * 16u;
        uint32_t cmp = 0, d = 0;
        if (pa < end) cmp = end - pa < 16u ? end - pa : 16u;
        if (cmp) {
            if (pa + 16u <= n) { d = first_diff16_sel(ld128(src + pa), pat4); d = d < cmp ? d : cmp; }
            else while (d < cmp && src[pa + d] == pb) d++;
        }
        const uint64_t stop = ballot(d < 16u);
        if (stop) {
            const uint32_t sl = first_lane(stop);
            return k + sl * 16u + rdlane(d, sl);
        }
        k += 1024u;
    }
}
__device__ __forceinline__ uint32_t hc_coop_reverse_count_pattern(const uint8_t *__restrict__ src, uint32_t a, uint32_t pattern, uint32_t lane) {
    const uint8_t pb = (uint8_t)pattern;
    uint32_t k = 0;
    for (;;) {
        const uint32_t top = a - k;                              // bytes [0, top) are left; this lane: the 16 below top - lane * 16
        uint32_t d = 0;                                          // bytes equal to the pattern byte, counted downwards
        if (top >= (lane + 1u) * 16u) {
            const u32x4 v = ld128(src + top - (lane + 1u) * 16u);
            const uint32_t x3 = v.w ^ pattern, x2 = v.z ^ pattern, x1 = v.y ^ pattern, x0 = v.x ^ pattern;
            if (x3) d = (uint32_t)__builtin_clz(x3) >> 3;
            else if (x2) d = 4u + ((uint32_t)__builtin_clz(x2) >> 3);
            else if (x1) d = 8u + ((uint32_t)__builtin_clz(x1) >> 3);
            else if (x0) d = 12u + ((uint32_t)__builtin_clz(x0) >> 3);
            else d = 16u;
        } else if (top > lane * 16u) {
            uint32_t q = top - lane * 16u;                       // fewer than 16 bytes left: position 0 ends the count
            while (q > 0 && src[q - 1u] == pb) { q--; d++; }
        }
        const uint64_t stop = ballot(d < 16u);
        if (stop) {
            const uint32_t sl = first_lane(stop);
            return k + sl * 16u + rdlane(d, sl);
        }
        k += 1024u;
    }
}

// kLds = true : blocks <= 64 KiB, links = u16 deltas (q - prev) copied into LDS, results u32 (len | off << 16), bitmap in LDS
// kLds = false: any block size, links = u32 predecessors read from the HBM workspace, results u64 (len | off << 32), bitmap
//               in the workspace (zeroed by the launcher).  Here the walk also meets the two tests that cannot fire in
//               a 64 KiB block: candidates further than 65535 bytes end the walk (:573), candidates below
//               lowestMatchIndex are counted but not compared (:579).
template <int kCands, bool kLds>   // kCands = candidates of a chain examined per loop trip (1, 2, 4 or 8; 4 ships)
__global__ __launch_bounds__(1024) void k_hc_seg_search(const uint8_t *__restrict__ d_in,
                                                         const uint64_t *__restrict__ d_in_off,
                                                         const uint32_t *__restrict__ d_in_len,
                                                         const void *__restrict__ d_link_v, uint64_t link_stride,
                                                         void *__restrict__ d_res_v, uint32_t *__restrict__ d_bitmap,
                                                         uint64_t bitmap_stride, uint32_t blk0, uint32_t nblocks,
                                                         int32_t max_attempts, uint32_t max_in_len, uint32_t lk_bytes,
                                                         uint32_t seg_len, int fetch_rounds) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    typedef typename std::conditional<kLds, uint16_t, uint32_t>::type T;
    typedef typename std::conditional<kLds, uint32_t, uint64_t>::type R;
    typedef typename std::conditional<kLds, const lds_u16 *, const uint32_t *>::type LinkPtr;
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const uint32_t n = d_in_len[blk0 + b];
    if (n > max_in_len) return;                                  // K3 reports it; the workspace is sized by max_in_len
    const uint32_t np = n_positions(n);
    if (np == 0) return;
    const uint8_t *src = d_in + d_in_off[blk0 + b];
    R *res = static_cast<R *>(d_res_v) + (uint64_t)b * link_stride;
    LinkPtr lk;
    lds_u32 *next_seg;                                           // start-point counter
    [[maybe_unused]] lds_u32 *bm_l = nullptr;                    // visited bitmap, one bit per position
    [[maybe_unused]] uint32_t *bm_g = nullptr;
    // Counted runs (blocks <= 16 MiB): the last long counts of this block, as "the bytes at x and x - d are equal for every
    // x in [from, end), and not at end (or end is iHighLimit)", packed d | from << 16 | end << 32, in kRunSlots slots indexed
    // by a hash of d.  Any other walk whose candidate lies d back and has matched up to somewhere inside [from, end] knows
    // its count without reading a byte.  On input with a period every start point's first match is the SAME run (a 64 KiB
    // block of period 256: 1 000 walks x 64 KiB; round 3, tools/cliff_probe.py: 4.4 GiB/s at level 9).
    constexpr uint32_t kRunSlots = 8;                            // slot of a distance: the top bits of d x 2654435761 (periods are often powers of two)
    [[maybe_unused]] lds_u64 *runs = nullptr;
    if constexpr (kLds) {
        lk = (const lds_u16 *)lds_raw;
        runs = (lds_u64 *)(lds_raw + lk_bytes);                  // kRunSlots counted runs, see below
        bm_l = (lds_u32 *)(lds_raw + lk_bytes + kRunSlots * 8u);
        const uint32_t bm_words = (np + 31u) >> 5;
        next_seg = bm_l + bm_words;                              // [0] start-point counter, [1] frontier of the walk from 0, [2] that walk is over
        const u32x4 *g4 = reinterpret_cast<const u32x4 *>(static_cast<const uint16_t *>(d_link_v) + (uint64_t)b * link_stride);
        u32x4 *l4 = reinterpret_cast<u32x4 *>(lds_raw);
        const uint32_t n16 = (np * 2u + 15u) >> 4;
        for (uint32_t k = threadIdx.x; k < n16; k += blockDim.x) l4[k] = g4[k];
        for (uint32_t k = threadIdx.x; k <= bm_words + 2u; k += blockDim.x) bm_l[k] = 0;
        if (threadIdx.x < kRunSlots) runs[threadIdx.x] = 0;
    } else {
        lk = static_cast<const uint32_t *>(d_link_v) + (uint64_t)b * link_stride;
        bm_g = d_bitmap + (uint64_t)b * bitmap_stride;
        next_seg = (lds_u32 *)lds_raw;
        if (threadIdx.x < 3u) next_seg[threadIdx.x] = 0;
        runs = (lds_u64 *)(lds_raw + 16u);
        if (threadIdx.x < kRunSlots) runs[threadIdx.x] = 0;
    }
    // (an entry is d (16 bits) | from | end: 16 + 32 bits for a block <= 64 KiB, 24 + 24 for one <= 16 MiB; none beyond)
    constexpr uint32_t kFromBits = kLds ? 16u : 24u;
    const bool use_runs = kLds || n <= (1u << 24);
    __syncthreads();
    // what a link says: the first candidate of q (0 = none), the distance to the next one (:502-504, :619)
    auto first_of = [](uint32_t q, uint32_t raw) { return kLds ? q - raw : raw; };
    auto delta_of = [](uint32_t mm, uint32_t raw) { return Links<T>::delta(mm, (T)raw); };
    auto mark = [&](uint32_t q) -> bool {                        // true = q was marked already
        const uint32_t bit = 1u << (q & 31u);
        uint32_t old;
        if constexpr (kLds) old = __hip_atomic_fetch_or(bm_l + (q >> 5), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else old = __hip_atomic_fetch_or(bm_g + (q >> 5), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (old & bit) != 0;
    };
    auto pack = [](int32_t len, uint32_t off_) -> R {
        if constexpr (kLds) return (uint32_t)len | (off_ << 16);
        else return (uint64_t)(uint32_t)len | ((uint64_t)off_ << 32);
    };
    const bool pattern_analysis = max_attempts > 128;            // :983
    const uint32_t limit = n - kLastLiterals;                    // iHighLimit = matchlimit :989, :1011
    const uint32_t nseg = (np + seg_len - 1u) / seg_len;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lanes_below = (1ull << lane) - 1ull;

    // (lane state as 0 / 1 in VECTOR registers: as `bool` each of them is a 64-bit lane mask in two scalar registers for the
    //  whole loop, and the scalar registers are what this kernel runs out of)
    uint32_t have = 0, exhausted = 0, in_chain = 0, in_ext = 0, fresh = 0;
    uint32_t is_true = 0;                                        // this lane runs the walk that started at position 0
    uint32_t ext_final = 0;                                      // `off` already is the candidate's full match length
    uint32_t fword_next = 0;                                     // frontier word of the walk from 0, as of the previous trip
    // Hand-over of the true walk (blocks <= 64 KiB).  When the walk from 0 arrives at a position X somebody else has marked,
    // the parse goes on in that walk -- but only the true walk publishes the frontier that retires the others and keeps
    // them from counting huge matches, so without a hand-over a block whose first long match sits where a speculative walk
    // got first (any input with a period: the ramp of the reference's tests) had every walk count its own 64 KiB match
    // (level 9 on D-ramp: 199 ms per GiB).  The frontier word carries the offer: X | bit 30 = "the parse continues in the
    // walk that searched X".  That walk takes it -- it knows the last three positions it marked, and an offer takes two
    // trips to arrive, in which a walk finishes at most two searches -- simply by becoming the true walk: its own frontier
    // store (its current position: X or what the parse reaches after X) clears the bit.  If nobody has after four trips
    // the old rule applies: the walk from 0 is declared over, bit 31.  A walk that searched X and has since merged into a
    // third one (walks that start inside the same stretch without candidates all merge at its end, one after the other)
    // passes the offer on to the position where it merged; its lane remembers that for one more walk.
    constexpr uint32_t kOffer = 1u << 30;
    uint32_t hand_wait = 0, hand_pos = 0;
    uint32_t mark0 = ~0u, mark1 = ~0u, mark2 = ~0u;              // the last three positions this walk has marked (searched), newest first
    uint32_t was0 = ~0u, was1 = ~0u, was2 = ~0u, went_to = ~0u;  // the same of this lane's previous walk, and where that one merged
    uint32_t trips = 0;          // safety net: after 2^18 trips nobody waits for a frontier any more (a block takes 10^2..10^4)
    u32x4 p16 = {0, 0, 0, 0};                                    // the 16 bytes at pos
    u32x4 aw = p16;                                              // the 16 bytes at pos + aw_off (the compare window)
    uint32_t aw_off = 0;
    uint32_t pos = 0, m = 0, off = 0, avail = 0, best_off = 0;
    [[maybe_unused]] uint32_t lowest = 0;                        // lowestMatchIndex :553-554 (0 in a 64 KiB block)
    int32_t nb = 0, best_len = (int32_t)kMinMatch - 1;
#ifdef ZLZ4_STAMPS
    unsigned long long st_trips = 0, st_chain = 0, st_fetch = 0, st_walks = 0, t_assign = 0, t_fetch = 0, t_chain = 0;
    unsigned long long st_counts = 0, st_known = 0, st_pats = 0, st_count_steps = 0, t_long = 0, st_miss_d = 0, st_miss_lo = 0, st_miss_hi = 0;
#define HSTAMP(acc, t0) acc += __builtin_amdgcn_s_memtime() - t0
#define HNOW() __builtin_amdgcn_s_memtime()
#else
#define HSTAMP(acc, t0)
#define HNOW() 0
#endif
    for (;;) {
        [[maybe_unused]] unsigned long long t0 = HNOW();
        // ---- start points for the lanes that have none
        const uint64_t want = ballot(!have && !exhausted);
        if (want) {
            uint32_t base = 0;
            if (lane == first_lane(want)) base = __hip_atomic_fetch_add(next_seg, (uint32_t)__popcll(want), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            base = rdlane(base, first_lane(want));
            if (!have && !exhausted) {
                const uint32_t seg = base + (uint32_t)__popcll(want & lanes_below);
                if (seg < nseg) { pos = seg * seg_len; have = true; in_chain = false; is_true = seg == 0u; mark0 = mark1 = mark2 = ~0u; }
                else exhausted = true;
            }
        }
        // The walk from position 0 IS the parse as long as it has not merged into another walk; everything below its
        // frontier is settled, so start points and walks below it are dropped.  (Without this a block that begins with
        // one huge match -- zeros, a constant prefix -- would have every other lane count the same huge match.)
        // (read one trip ahead: both the frontier and the flag only ever grow, a stale value is merely less helpful)
        const uint32_t fword = fword_next;
        fword_next = __hip_atomic_load(next_seg + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t frontier = fword & (kLds ? 0x3FFFFFFFu : 0x7FFFFFFFu);   // (bit 31: the walk from 0 is over; bit 30: offer)
        trips += 1;
        if constexpr (kLds) {
            if (rfl(fword) & kOffer) {                           // (uniform, rare) the parse continues in my walk?
                if (have && !is_true && (mark0 == frontier || mark1 == frontier || mark2 == frontier)) is_true = 1;
                else if (went_to != ~0u && hand_wait == 0u && (was0 == frontier || was1 == frontier || was2 == frontier)) {
                    // the walk that searched X was this lane's previous one, and it has merged into another at went_to:
                    // the parse continues there.  Pass the offer on (and the duty to close it).
                    __hip_atomic_store(next_seg + 1, went_to | kOffer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    hand_wait = 4; hand_pos = went_to; went_to = ~0u;
                }
            }
            if (ballot(hand_wait != 0u)) {                       // (uniform, rare) my offer: settled?
                if (hand_wait != 0u && --hand_wait == 0u) {
                    // nobody took it: the walk from 0 is over (compare-and-swap: a taker's frontier store may have come since)
                    uint32_t expect = hand_pos | kOffer;
                    (void)__hip_atomic_compare_exchange_strong(next_seg + 1, &expect, hand_pos | 0x80000000u, __ATOMIC_RELAXED,
                                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (have && !is_true && pos < frontier) { have = false; in_chain = false; in_ext = false; ext_final = false; }
        if (have && is_true) __hip_atomic_store(next_seg + 1, pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // (a lane that has offered the hand-over keeps its wavefront in the loop until the offer is settled: walks that wait
        //  for the frontier wait for that)
        if (!ballot((have | (exhausted ^ 1u) | hand_wait) != 0u)) break;
        HSTAMP(t_assign, t0);
#ifdef ZLZ4_STAMPS
        st_trips += 1; st_walks += __popcll(want); st_fetch += __popcll(ballot(have && !in_chain));
        t0 = HNOW();
#endif
        // ---- next position of the walk.  Positions whose hash was never seen before have no candidate (:566-568): the
        //      parse steps over them one by one (:1013-1016) and they cost one LDS read each, four per round trip.  Only
        //      positions WITH candidates are marked and tested: walks inside the same run of candidate-less positions all
        //      reach the position that ends the run, and merge there.
        if (have && !in_chain) {
            for (int it = 0; it < fetch_rounds; ++it) {
                if (pos >= np) {
                    if (is_true) {
                        __hip_atomic_store(next_seg + 1, pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_or(next_seg + 1, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    have = false;
                    break;
                }
                const uint32_t l0 = lk[pos], l1 = lk[pos + 1u], l2 = lk[pos + 2u], l3 = lk[pos + 3u];   // (both arrays are padded)
                const uint32_t k = first_of(pos, l0) != 0u ? 0u : first_of(pos + 1u, l1) != 0u ? 1u
                                 : first_of(pos + 2u, l2) != 0u ? 2u : first_of(pos + 3u, l3) != 0u ? 3u : 4u;
                pos += k;
                if (pos >= np) {                                 // (also ends a run that ran into the garbage past np)
                    if (is_true) {
                        __hip_atomic_store(next_seg + 1, pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_or(next_seg + 1, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    have = false;
                    break;
                }
                if (k == 4u) continue;
                const uint32_t l = k == 0u ? l0 : k == 1u ? l1 : k == 2u ? l2 : l3;
                if (mark(pos)) {                                 // somebody else's walk continues from here
                    if (is_true) {
                        if constexpr (kLds) {                    // ... and with it the parse: offer the hand-over
                            __hip_atomic_store(next_seg + 1, pos | kOffer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            hand_wait = 4; hand_pos = pos;
                        } else {
                            __hip_atomic_fetch_or(next_seg + 1, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    } else { was0 = mark0; was1 = mark1; was2 = mark2; went_to = pos; }   // (for an offer that comes too late)
                    have = false;
                    break;
                }
                mark2 = mark1; mark1 = mark0; mark0 = pos;
                m = first_of(pos, l);                            // :563 hashTable[hashPtr(ip)], != 0 here
                best_len = (int32_t)kMinMatch - 1;               // :560
                best_off = 0;
                nb = max_attempts;
                if constexpr (!kLds) lowest = pos < 65536u ? 0u : pos - kMaxDist;
                if (pos + 16u > n) {
                    // the last <= 4 searchable positions of a block: no room for 16-byte compares
                    const uint32_t pattern = ld32(src + pos);
                    while (m > 0 && nb > 0) {                    // :571
                        if (!kLds && (m > pos || pos - m > kMaxDist)) break;   // :573
                        nb -= 1;                                 // :577
                        const uint32_t delta = delta_of(m, lk[m]);
                        int32_t mlt = 0;
                        if ((kLds || m >= lowest) && ld32(src + m) == pattern)   // :579, :586
                            mlt = (int32_t)(kMinMatch + lz4_count(src, pos + kMinMatch, m + kMinMatch, limit));
                        if (mlt > best_len) {                    // :607
                            best_len = mlt;
                            best_off = pos - m;
                            if (mlt > max_attempts) break;       // :613
                            if ((uint32_t)mlt >= limit - pos) break;   // final: nothing is longer than up to iHighLimit, nor is the pattern step's maxML (:658)
                        }
                        if (delta == 0 || delta > m) break;      // :620
                        m -= delta;                              // :621
                    }
                    if (pattern_analysis && (uint32_t)best_len < limit - pos)
                        hc_pattern_step<T, LinkPtr>(src, lk, pos, m, pattern, lowest, limit, best_len, best_off);
                    const bool found = best_len >= (int32_t)kMinMatch && best_off != 0;
                    if (found) res[pos] = pack(best_len, best_off);
                    pos += found ? (uint32_t)best_len : 1u;      // :1013-1016, :382
                    continue;
                }
                if (!kLds && pos - m > kMaxDist) {               // :573 on the first candidate: the walk ends at once
                    if (pattern_analysis)
                        hc_pattern_step<T, LinkPtr>(src, lk, pos, m, ld32(src + pos), lowest, limit, best_len, best_off);
                    const bool found = best_len >= (int32_t)kMinMatch && best_off != 0;
                    if (found) res[pos] = pack(best_len, best_off);
                    pos += found ? (uint32_t)best_len : 1u;
                    continue;
                }
                avail = limit - pos;                             // lz4Count stops at iHighLimit
                off = 0;
                in_chain = true;
                in_ext = false;
                fresh = true;
                break;
            }
        }
        HSTAMP(t_fetch, t0);
#ifdef ZLZ4_STAMPS
        st_chain += __popcll(ballot(have && in_chain));
        t0 = HNOW();
#endif
        // ---- a candidate that has matched 64 bytes and more is counted by the whole wavefront, 1 KiB per step, lowest
        //      lane first (lane 0 of wave 0 runs the walk from 0, whose frontier then retires the others)
        // While the walk from 0 is still on its own, the other walks do not start a long count: most likely its frontier
        // is about to retire them (a block that is one long run), and if it merges first they go on from where they are.
        // While the walk from 0 is still on its own, the other walks do not start a long count (>= 64 bytes matched): most
        // likely its frontier is about to retire them (a block that is one long run); if it merges first they go on.  A
        // long count is done by the whole wavefront, 1 KiB per step, lowest lane first.
        bool parked = false;
        if (const uint64_t wantm = ballot(have && in_chain && in_ext && !ext_final && off >= 64u)) {
            const bool walk0_over = (fword >> 31) != 0u || trips > (1u << 18);
            const bool want = (wantm >> lane) & 1ull;
            parked = want && !is_true && !walk0_over;
            uint64_t longm = wantm & ~ballot(parked);
            while (longm) {
                const uint32_t L = first_lane(longm);
                longm &= longm - 1ull;
                const uint32_t P = rdlane(pos, L);
                const bool walk0 = rdlane((uint32_t)is_true, L) != 0;
                if (!walk0 && P < (__hip_atomic_load(next_seg + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & (kLds ? 0x3FFFFFFFu : 0x7FFFFFFFu))) {
                    if (lane == L) { have = false; in_chain = false; in_ext = false; }
                    continue;
                }
                const uint32_t M = rdlane(m, L), O = rdlane(off, L);
                uint32_t total = 0;
                [[maybe_unused]] const uint32_t d = P - M;
                bool known = false;                              // a counted run answers it
                [[maybe_unused]] uint32_t run_d = 0, run_from = 0, run_end = 0;
                if (use_runs) {
                    const unsigned long long e = __hip_atomic_load(runs + ((d * 2654435761u) >> 29), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t e_lo = rfl((uint32_t)e), e_hi = rfl((uint32_t)(e >> 32));   // (one address: the same in every lane)
                    const unsigned long long eu = ((unsigned long long)e_hi << 32) | e_lo;
                    run_d = e_lo & 0xFFFFu;
                    run_from = (uint32_t)(eu >> 16) & ((1u << kFromBits) - 1u);
                    run_end = (uint32_t)(eu >> (16u + kFromBits));
                    known = run_d == d && run_from <= P + O && P + O <= run_end;
                    total = run_end - P;
                }
#ifdef ZLZ4_STAMPS
                st_known += known; st_counts += !known;
                if (!known) { if (run_d != d) st_miss_d += 1; else if (P + O < run_from) st_miss_lo += 1; else st_miss_hi += 1; }
#endif
                if (!known) {
                    total = hc_coop_count(src, P, M, O, limit, n, lane);
#ifdef ZLZ4_STAMPS
                    st_count_steps += (total - O) / 1024u + 1u;
#endif
                    if (use_runs) {
                        // (the same run counted from further down -- another wavefront's entry, this walk started below it:
                        //  the entry keeps the lower start.  Not atomic with the load above: whatever lands is a true statement.)
                        uint32_t nf = P + O;
                        const uint32_t ne = P + total;
                        if (run_d == d && run_end == ne && run_from < nf) nf = run_from;
                        if (lane == L && total - O >= 1024u)       // (a count of one step is as cheap as the look-up: it only evicts)
                            __hip_atomic_store(runs + ((d * 2654435761u) >> 29), (unsigned long long)d | ((unsigned long long)nf << 16) | ((unsigned long long)ne << (16u + kFromBits)),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                if (lane == L) { off = total; ext_final = true; }
            }
        }
        HSTAMP(t_long, t0);
        // ---- one round trip of the chain walk :571-622
        bool done = false;                                       // the search at pos is over
        if (have && in_chain && !parked) {
            // a candidate is done: :577, :586-:621.  `total` = its match length (bytes equal to the ones at pos, not yet
            // clamped), or 0 for a candidate that is known not to beat best_len (the reference counts those in full,
            // :588, but only `mlt > longest` is ever used, :607)
            bool changed = false;
            auto complete = [&](uint32_t c, uint32_t delta, uint32_t total) -> bool {
                nb -= 1;                                         // :577
                const int32_t mlt = total >= kMinMatch ? (int32_t)(total < avail ? total : avail) : 0;   // :586, :588
                const bool better = mlt > best_len;              // :607
                changed = changed | better;
                best_len = better ? mlt : best_len;
                best_off = better ? pos - c : best_off;
                const bool stop = (bool)((int)better & (int)(mlt > max_attempts)) | (delta == 0) | (delta > c);   // :613, :620
                m = stop ? c : c - delta;                        // :621
                return (bool)((int)!stop & (int)(nb > 0) & (int)(m > 0) & (int)(kLds || pos - m <= kMaxDist));   // :571, :573
            };
            if (!in_ext) {
                // up to kCands candidates per trip: the links are chased first (the walk itself does not depend on the
                // compares), then the 16-byte loads fly together.  A candidate can only matter if it matches MORE
                // than best_len bytes, so what is compared first is the 16-byte window that ends at byte best_len
                // (bytes 0..15 while best_len < 16): a mismatch there settles it in one load, whatever the candidate's
                // real length -- at level 9 a frequent 4-gram has 256 candidates that match 20-40 bytes each.
                const uint32_t w = best_len >= 16 ? (uint32_t)best_len - 15u : 0u;
                uint32_t c[kCands], l[kCands];                   // candidates of this trip, l[k] = distance to the next one
                c[0] = m; l[0] = delta_of(c[0], lk[c[0]]);
                {
                    bool more_k = true;
#pragma unroll
                    for (int k = 1; k < kCands; k++) {
                        more_k = more_k && !(l[k - 1] == 0 || l[k - 1] > c[k - 1]) && c[k - 1] - l[k - 1] > 0;
                        c[k] = more_k ? c[k - 1] - l[k - 1] : 0; l[k] = delta_of(c[k], lk[c[k]]);
                    }
                }
                if (fresh) { p16 = ld128(src + pos); aw_off = 0; fresh = false; }
                if (aw_off != w) { aw = ld128(src + pos + w); aw_off = w; }      // pos + w + 16 = pos + best_len + 1 <= n - 4
                const u32x4 awin = w == 0u ? p16 : aw;           // (a select after the loads: no wait on p16 alone)
                // (a candidate slot that the walk does not reach holds position 0: a harmless load, never looked at)
                u32x4 bk[kCands];
#pragma unroll
                for (int k = 0; k < kCands; k++) bk[k] = ld128(src + c[k] + w);
                uint32_t fd[kCands];
#pragma unroll
                for (int k = 0; k < kCands; k++) fd[k] = first_diff16_sel(awin, bk[k]);
                // Fast path: every candidate the walk reaches in this trip is rejected by the window test (the usual
                // case on a long chain).  Then nothing but the attempt counter and the walk position change, exactly as
                // :577 / :619-621 would leave them.  att[k] = candidate k is attempted (:571, :573), rejected = it cannot
                // beat best_len (or lies below lowestMatchIndex, :579), end[k] = the chain ends behind it (:620).
                auto rejected = [&](uint32_t cc, uint32_t fdd) {
                    const uint32_t cl = fdd < avail ? fdd : avail;           // lz4Count stops at iHighLimit
                    bool r = w != 0u ? fdd != 16u : (fdd < kMinMatch) | ((fdd < 16u) & ((int32_t)cl <= best_len));
                    r = r | ((int32_t)avail <= best_len);        // the best match ends at iHighLimit: lz4Count cannot return more (:234)
                    return kLds ? r : (r | (cc < lowest));
                };
                auto near_enough = [&](uint32_t cc) { return kLds || pos - cc <= kMaxDist; };
                // (0 / 1 in vector registers, like the lane state: as `bool` these are two scalar registers each)
                uint32_t att[kCands], end[kCands];
                att[0] = 1u; end[0] = (uint32_t)((l[0] == 0) | (l[0] > c[0]));
                uint32_t all_rej = (uint32_t)rejected(c[0], fd[0]);
#pragma unroll
                for (int k = 1; k < kCands; k++) {
                    att[k] = att[k - 1] & (end[k - 1] ^ 1u) & (uint32_t)(c[k - 1] - l[k - 1] > 0) & (uint32_t)(nb > k) & (uint32_t)near_enough(c[k]);
                    end[k] = (uint32_t)((l[k] == 0) | (l[k] > c[k]));
                    all_rej = all_rej & ((att[k] ^ 1u) | (uint32_t)rejected(c[k], fd[k]));
                }
                if (all_rej) {
                    uint32_t cl = c[0], ll = l[0];               // the last candidate attempted
                    uint32_t el = end[0];
                    int32_t natt = 1;
#pragma unroll
                    for (int k = 1; k < kCands; k++) {
                        cl = att[k] ? c[k] : cl; ll = att[k] ? l[k] : ll; el = att[k] ? end[k] : el;
                        natt += (int32_t)att[k];
                    }
                    nb -= natt;                                                           // :577
                    m = el ? cl : cl - ll;                                                // :620-621
                    done = !((bool)((int)!el & (int)(nb > 0) & (int)(m > 0) & (int)near_enough(m)));   // :571, :573
                } else {
                // (after a candidate that raised best_len the rest of the trip is dropped: their window is stale)
#pragma unroll
                for (int k = 0; k < kCands; k++) {
                    if (!done && !in_ext && !changed) {
                        const uint32_t fdk = fd[k];
                        if (!kLds && c[k] < lowest) { if (!complete(c[k], l[k], 0u)) done = true; }    /* :579 */
                        else if (w != 0) {
                            if (fdk == 16u) { m = c[k]; off = 0; in_ext = true; }          /* count it from byte 0 */
                            else if (!complete(c[k], l[k], 0u)) done = true;
                        } else {
                            uint32_t total = fdk;
                            bool more = total == 16u && total < avail;
                            if (more && pos + 32u > n) { total += lz4_count(src, pos + 16u, c[k] + 16u, limit); more = false; }
                            if (more) { m = c[k]; off = 16u; in_ext = true; }
                            else if (!complete(c[k], l[k], total)) done = true;
                        }
                    }
                }
                }
            } else {
                // the candidate at m matched `off` bytes so far: the next 16
                const uint32_t delta = delta_of(m, lk[m]);       // :619 chainTable[matchIndex]
                uint32_t total = off;
                bool more = false;
                if (!ext_final) {
                    const u32x4 a16 = ld128(src + pos + off);
                    const u32x4 b16 = ld128(src + m + off);      // m < pos, so m + off + 16 <= n too
                    total = off + first_diff16_sel(a16, b16);
                    more = total == off + 16u && total < avail;
                    if (more && pos + total + 16u > n) {         // no room for another 16-byte compare: finish by bytes
                        total += lz4_count(src, pos + total, m + total, limit);
                        more = false;
                    }
                }
                ext_final = false;
                if (more) off = total;
                else { in_ext = false; done = !complete(m, delta, total); }
            }
        }
        // patternAnalysis (:626-676), whole wavefront per lane that gets there (rare outside runs of one byte; the
        // three counts are each as long as the run)
        if (pattern_analysis) {
            const uint32_t pt = p16.x;
            bool pat_here = done && best_len > 0 && ((pt & 0xFFFFu) == (pt >> 16)) && ((pt & 0xFFu) == (pt >> 24));   // :629-631
            if (pat_here) pat_here = delta_of(m, lk[m]) == 1u;                                                       // :627
            uint64_t patm = ballot(pat_here);
            while (patm) {
                const uint32_t L = first_lane(patm);
                patm &= patm - 1ull;
#ifdef ZLZ4_STAMPS
                st_pats += 1;
#endif
                const uint32_t P = rdlane(pos, L), M = rdlane(m, L), pattern = rdlane(pt, L);
                const uint32_t low = kLds ? 0u : rdlane(lowest, L);
                const uint32_t src_pat_len = hc_coop_count_pattern(src, P + 4u, limit, pattern, n, lane) + 4u;       // :633
                const uint32_t cand = M - 1u;                                                                        // :636
                if (cand >= low && rfl(ld32(src + cand)) == pattern) {                                               // :637, :644
                    const uint32_t fwd_len = hc_coop_count_pattern(src, cand + 4u, limit, pattern, n, lane) + 4u;    // :646
                    const uint32_t back_len = hc_coop_reverse_count_pattern(src, cand, pattern, lane);               // :650
                    uint32_t lo = cand - back_len;                                                                   // :653
                    if (lo < low) lo = low;
                    const uint32_t lim_back = cand - lo;
                    const uint32_t seg_total = lim_back + fwd_len;                                                   // :654
                    const int32_t max_ml = (int32_t)(seg_total < src_pat_len ? seg_total : src_pat_len);             // :658
                    const uint32_t new_m = (seg_total >= src_pat_len && fwd_len <= src_pat_len) ? cand + fwd_len - src_pat_len   // :660-662
                                                                                                : cand - lim_back;   // :665
                    if (lane == L && max_ml > best_len && (P - new_m) <= kMaxDist) {                                 // :669
                        best_len = max_ml;
                        best_off = P - new_m;
                    }
                }
            }
        }
        if (done) {
            const bool found = best_len >= (int32_t)kMinMatch && best_off != 0;
#ifndef ZLZ4_EXPERIMENT_NOSTORE
            if (found) res[pos] = pack(best_len, best_off);
#endif
            pos += found ? (uint32_t)best_len : 1u;          // :1013-1016, :382
            in_chain = false;
        }
        HSTAMP(t_chain, t0);
        // a wavefront whose walks all wait for the frontier (or that only holds an offer open) leaves the issue slots to the others
        if (!ballot((have && !parked) || (!have && !exhausted))) __builtin_amdgcn_s_sleep(8);
    }
#ifdef ZLZ4_STAMPS
    if (lane == 0) {
        atomicAdd(&g_zlz4_hstamps[0], st_chain); atomicAdd(&g_zlz4_hstamps[1], st_trips); atomicAdd(&g_zlz4_hstamps[2], st_fetch);
        atomicAdd(&g_zlz4_hstamps[3], st_walks); atomicAdd(&g_zlz4_hstamps[4], t_assign); atomicAdd(&g_zlz4_hstamps[5], t_fetch);
        atomicAdd(&g_zlz4_hstamps[6], t_chain);
        atomicAdd(&g_zlz4_hstamps[7], st_counts); atomicAdd(&g_zlz4_hstamps[8], st_known); atomicAdd(&g_zlz4_hstamps[9], st_pats);
        atomicAdd(&g_zlz4_hstamps[10], st_count_steps); atomicAdd(&g_zlz4_hstamps[11], t_long);
        atomicMax(&g_zlz4_hstamps[12], t_assign + t_fetch + t_chain);
        atomicAdd(&g_zlz4_hstamps[13], st_miss_d); atomicAdd(&g_zlz4_hstamps[14], st_miss_lo); atomicAdd(&g_zlz4_hstamps[15], st_miss_hi);
    }
#endif
#undef HSTAMP
#undef HNOW
}

// ------------------------------------------------------------------ K3: greedy parse + emit
template <typename R>
__global__ __launch_bounds__(256) void k_hc_parse_emit(const uint8_t *__restrict__ d_in,
                                                        const uint64_t *__restrict__ d_in_off,
                                                        const uint32_t *__restrict__ d_in_len,
                                                        uint8_t *__restrict__ d_out,
                                                        const uint64_t *__restrict__ d_out_off,
                                                        const uint32_t *__restrict__ d_out_cap,
                                                        int64_t *__restrict__ d_result, const R *__restrict__ d_res,
                                                        uint64_t link_stride, uint32_t blk0, uint32_t nblocks,
                                                        uint32_t max_in_len) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t b = rfl(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (b >= nblocks) return;
    const uint32_t blk = blk0 + b;
    const uint8_t *src = d_in + d_in_off[blk];
    uint8_t *dst = d_out + d_out_off[blk];
    const uint32_t n = rfl(d_in_len[blk]);
    const uint32_t oend = rfl(d_out_cap[blk]);
    const R *res = d_res + (uint64_t)b * link_stride;

    int64_t out;
    if (n > kMaxInput) {                                         // :1442
        out = kErrInputTooLarge;
    } else if (n > max_in_len) {                                 // K1/K2 skipped it: nothing of it is in the workspace
        out = kErrInvalidState;
    } else if (n == 0) {                                         // :1443
        out = 0;
    } else if (oend == 0) {                                      // :1461
        out = kErrOutputTooSmall;
    } else if (n < kMfLimit + 1u) {                              // :995-998 encodeLiterals (:1394-1425)
        if (oend < n + 1u + n / 255u) out = kErrOutputTooSmall;  // :1395
        else {
            if (lane == 0) dst[0] = (uint8_t)(n << 4);           // n < 13 < 15
            if (lane < n) dst[1u + lane] = src[lane];
            out = (int64_t)n + 1;
        }
    } else {
        const uint32_t mflimit = n - kMfLimit;                   // :988
        uint32_t ip = 0, anchor = 0, op = 0;
        bool failed = false;
        // 64-position window of search results held in registers -- and the window's input bytes with it: a sequence's
        // literals are stored straight from them (a byte per lane), so the loop has no load that a store waits for (round 3;
        // copy_bytes' load -> store per sequence made this kernel a chain of ~1 900-cycle steps).  The window starts at the
        // anchor whenever the pending literal run is shorter than 64 bytes, so its literals are inside.
        uint32_t wbase = 0;
        R w = (lane <= mflimit) ? res[lane] : (R)0;
        uint32_t sb = lane < n ? src[lane] : 0u;
        // lanes of the window whose position holds a match; the loop steps over the others one by one (:1013-1016), i.e.
        // it goes to the first such lane at or after ip
        auto is_match = [](R r) {
            if (sizeof(R) == 4) return ((uint32_t)r & 0xFFFFu) >= kMinMatch && ((uint32_t)r >> 16) != 0;
            return (uint32_t)r >= kMinMatch && (uint32_t)((uint64_t)r >> 32) != 0;
        };
        uint64_t wmatch = ballot(is_match(w));
        while (ip <= mflimit) {                                  // :1009
            if (ip - wbase >= 64u) {
                wbase = ip - anchor < 64u ? anchor : ip;
                w = (wbase + lane <= mflimit) ? res[wbase + lane] : (R)0;
                sb = wbase + lane < n ? src[wbase + lane] : 0u;
                wmatch = ballot(is_match(w));
            }
            const uint64_t ahead = wmatch >> (ip - wbase);
            if (ahead == 0) { ip = wbase + 64u; continue; }      // no match in the rest of the window
            ip += first_lane(ahead);
            uint32_t len, off;
            if (sizeof(R) == 4) {
                const uint32_t r = rdlane((uint32_t)w, ip - wbase);
                len = r & 0xFFFFu; off = r >> 16;
            } else {
                len = rdlane((uint32_t)w, ip - wbase);
                off = rdlane((uint32_t)((uint64_t)w >> 32), ip - wbase);
            }
            // encodeSequence (:308-386) with limitedOutput
            const uint32_t lit = ip - anchor;                    // :317
            if ((uint64_t)op + lit / 255u + lit + (2u + 1u + kLastLiterals) > oend) { failed = true; break; }   // :320-325
            const uint32_t ml_code = len - kMinMatch;            // :354
            const uint32_t nle = ext_len_bytes(lit), nme = ext_len_bytes(ml_code);
            const uint32_t op2 = op + 1u + nle + lit + 2u;       // after token, literal length, literals, offset
            if ((uint64_t)op2 + ml_code / 255u + (1u + kLastLiterals) > oend) { failed = true; break; }   // :355-359
            if (lane == 0)
                dst[op] = (uint8_t)(((lit >= 15u ? 15u : lit) << 4) | (ml_code >= 15u ? 15u : ml_code));
            if (lit >= 15u) write_ext_len(dst + op + 1u, lit, lane);
            if (anchor >= wbase) {                               // :346 (ip - wbase < 64 here: the literals are window bytes)
                const uint32_t a0 = anchor - wbase;
                if (lane >= a0 && lane < a0 + lit) dst[op + 1u + nle + (lane - a0)] = (uint8_t)sb;
            } else {
                copy_bytes(dst + op + 1u + nle, src + anchor, lit, lane);
            }
            if (lane < 2u) dst[op2 - 2u + lane] = (uint8_t)(off >> (8u * lane));   // :350
            if (ml_code >= 15u) write_ext_len(dst + op2, ml_code, lane);           // :361-376 (510-steps == 255-run)
            op = op2 + nme;
            ip += len;                                           // :382
            anchor = ip;
        }
        if (failed) {
            out = kErrOutputTooSmall;                            // :1029-1031
        } else {
            const uint32_t fl = n - anchor;                      // :1035
            out = (int64_t)op;
            if (fl > 0) {
                const uint32_t nle = ext_len_bytes(fl);
                // :1037 tests only op + fl + 1; the reference then writes the length-extension bytes
                // unchecked (out of bounds when they do not fit).  We refuse instead of overrunning.
                if ((uint64_t)op + fl + 1u > oend || (uint64_t)op + 1u + nle + fl > oend) out = kErrOutputTooSmall;
                else {
                    if (lane == 0) dst[op] = (uint8_t)((fl >= 15u ? 15u : fl) << 4);
                    if (fl >= 15u) write_ext_len(dst + op + 1u, fl, lane);
                    copy_bytes(dst + op + 1u + nle, src + anchor, fl, lane);   // :1059
                    out = (int64_t)(op + 1u + nle + fl);
                }
            }
        }
    }
    if (lane == 0) d_result[blk] = out;
}

}  // namespace zlz4

extern "C" int zlz4_launch_hc_mid(hipStream_t, const uint8_t *, const uint64_t *, const uint32_t *, uint8_t *, const uint64_t *,
                                  const uint32_t *, int64_t *, uint32_t, void *, uint32_t, uint32_t);
extern "C" int zlz4_launch_hc_opt_parse(hipStream_t, const uint8_t *, const uint64_t *, const uint32_t *, uint8_t *,
                                        const uint64_t *, const uint32_t *, int64_t *, const void *, uint64_t, int, void *,
                                        uint32_t, uint32_t, uint32_t, uint32_t);
extern "C" size_t zlz4_hc_mid_workspace_bytes(uint32_t chunk_blocks);
extern "C" size_t zlz4_hc_opt_workspace_bytes(uint32_t chunk_blocks);

namespace zlz4 {

// A second stream per host thread and device for the emit pass: K3 of one round needs no LDS and is bound by the scalar
// unit, the search of the next round owns the LDS and leaves half the wave slots empty, so the two share the CUs well.
struct HcSideStream {
    hipStream_t st = nullptr;
    hipEvent_t searched[2] = {nullptr, nullptr}, emitted[2] = {nullptr, nullptr};
    bool ok = false;
    bool init() {
        if (ok) return true;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { st = nullptr; return false; }
        for (int k = 0; k < 2; k++)
            if (hipEventCreateWithFlags(&searched[k], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&emitted[k], hipEventDisableTiming) != hipSuccess) { destroy(); return false; }
        ok = true;
        return true;
    }
    void destroy() {       // also after a partial init(): nothing is left behind for the next attempt to leak
        for (int k = 0; k < 2; k++) {
            if (searched[k]) (void)hipEventDestroy(searched[k]);
            if (emitted[k]) (void)hipEventDestroy(emitted[k]);
            searched[k] = emitted[k] = nullptr;
        }
        if (st) (void)hipStreamDestroy(st);
        st = nullptr;
        ok = false;
    }
    // thread exit (the slots are thread_local): a service that calls the HC batch API from short-lived threads gives its
    // streams and events back.  hipStreamDestroy waits for work still queued on the stream.
    ~HcSideStream() { destroy(); }
};
static HcSideStream *hc_side_stream() {
    static thread_local HcSideStream per_device[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    return per_device[dev].init() ? &per_device[dev] : nullptr;
}

// K1 + K2 (+ K3 for the greedy levels, or the price-based parse for levels 10-12) in rounds of `chunk` blocks
// (A per-wavefront work-queue form of the search-every-position K2 -- a lane that finishes its chain takes the next
//  unassigned position -- was bit-exact but ran 2.3x SLOWER in round 1: neighbouring positions walk neighbouring chains, so
//  the lock-step kernel's 64 gathers of a trip fall into a few cache lines, while the queue's lanes drift apart.  With
//  the links in LDS and only the parse's positions searched, k_hc_seg_search is exactly such a queue and wins.)
template <typename T, typename R>
int launch_hc_chunked(hipStream_t stream, const uint8_t *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
                      uint8_t *d_out, const uint64_t *d_out_off, const uint32_t *d_out_cap, int64_t *d_result,
                      uint32_t nblocks, uint32_t max_in_len, int32_t max_attempts, void *ws, uint32_t chunk,
                      bool optimal, uint32_t sufficient_len) {
    const uint64_t stride = ((uint64_t)max_in_len + 15u) & ~15ull;           // entries per block in both arrays
    T *d_link = static_cast<T *>(ws);
    uint8_t *after_link = static_cast<uint8_t *>(ws) + (uint64_t)chunk * stride * sizeof(T);
    R *d_res = reinterpret_cast<R *>(after_link);
    void *d_opt = after_link + (uint64_t)chunk * stride * sizeof(R);
    const uint32_t np_max = max_in_len < 13u ? 1u : max_in_len - 11u;
    // 128 KiB of the CU's 160 KiB LDS for the 32-bit table
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hc_build_links<T>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kHcTableSize * 4u + 4096u * sizeof(T) + 16u));
    static const bool legacy_search = zlz4_tune_env("ZLZ4_HC_LEGACY_SEARCH") != nullptr;   // A/B switch for profiles/
    const bool seg_search = !optimal && !legacy_search;
    if (seg_search) {
        // parse-aware search (k_hc_seg_search).  Rounds of half a chunk, results in the two halves of the result
        // area in turn: K3 of round r runs on the side stream while K1 / K2s of round r + 1 run on `stream`.
        constexpr bool kLds = sizeof(T) == 2;
        static const uint32_t seg_len = [] { const char *e = zlz4_tune_env("ZLZ4_HC_SEG"); return e ? (uint32_t)atoi(e) : 32u; }();   // start points of the speculative walks
        static const uint32_t thr_div = [] { const char *e = zlz4_tune_env("ZLZ4_HC_LPS"); return e ? (uint32_t)atoi(e) : 2u; }();
        static const bool no_overlap = zlz4_tune_env("ZLZ4_HC_NO_OVERLAP") != nullptr;          // A/B switch for profiles/
        static const int fetch_rounds = [] { const char *e = zlz4_tune_env("ZLZ4_HC_FETCH"); const int v = e ? atoi(e) : 1; return v >= 1 && v <= 8 ? v : 1; }();
        const uint32_t nseg_max = (np_max + seg_len - 1u) / seg_len;
        uint32_t threads = (nseg_max / thr_div + 63u) & ~63u;                // ~2 start points per lane
        if (threads > 1024u) threads = 1024u;
        if (threads < 64u) threads = 64u;
        const uint32_t lk_bytes = ((np_max * 2u + 15u) & ~15u) + 16u;        // + padding: the walk reads 3 links ahead
        const uint32_t lds = kLds ? lk_bytes + 64u + ((np_max + 31u) / 32u + 3u) * 4u : 16u + 64u;   // links, counted runs, bitmap, 3 words | 3 words, counted runs
        static const int cands = [] { const char *e = zlz4_tune_env("ZLZ4_HC_CANDS"); return e ? atoi(e) : 4; }();
        auto kern = cands == 8 ? &k_hc_seg_search<8, kLds> : cands == 4 ? &k_hc_seg_search<4, kLds> : cands == 2 ? &k_hc_seg_search<2, kLds> : &k_hc_seg_search<1, kLds>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const uint64_t bm_stride = (stride + 31u) / 32u + 1u;                // bitmap words per block (HBM links only)
        uint32_t *d_bitmap = static_cast<uint32_t *>(d_opt);                 // (the area of the price-based parse is idle here)
        HcSideStream *side = (chunk >= 2u && nblocks > chunk / 2u && !no_overlap) ? hc_side_stream() : nullptr;
        const uint32_t sub = side ? chunk / 2u : chunk;
        uint32_t round = 0;
        // error exit: K3 may still be running on the side stream and the caller parks the workspace as soon as `stream` is
        // idle -- wait for the side stream first
        auto fail = [&]() -> int { if (side) (void)hipStreamSynchronize(side->st); return -7; };
        for (uint32_t b0 = 0; b0 < nblocks; b0 += sub, round++) {
            const uint32_t nb = nblocks - b0 < sub ? nblocks - b0 : sub;
            const uint32_t half = side ? (round & 1u) : 0u;
            R *res = d_res + (uint64_t)half * sub * stride;
            if (side && round >= 2u && hipStreamWaitEvent(stream, side->emitted[half], 0) != hipSuccess) return fail();   // K3 of round - 2 read this half
            // K2s stores matches only: every other position of the parse must read "no match"
            if (hipMemsetAsync(res, 0, (size_t)nb * stride * sizeof(R), stream) != hipSuccess) return fail();
            if (!kLds && hipMemsetAsync(d_bitmap, 0, (size_t)nb * bm_stride * 4u, stream) != hipSuccess) return fail();
            hipLaunchKernelGGL((k_hc_build_links<T>), dim3(nb), dim3(64 * kLinkWaves), kHcTableSize * 4u + 4096u * sizeof(T) + 16u, stream, d_in,
                               d_in_off, d_in_len, d_link, stride, b0, nb, max_in_len);
            hipLaunchKernelGGL(kern, dim3(nb), dim3(threads), lds, stream, d_in, d_in_off, d_in_len,
                               static_cast<const void *>(d_link), stride, static_cast<void *>(res), d_bitmap, bm_stride, b0, nb,
                               max_attempts, max_in_len, lk_bytes, seg_len, fetch_rounds);
            hipStream_t emit_on = stream;
            if (side) {
                if (hipEventRecord(side->searched[half], stream) != hipSuccess ||
                    hipStreamWaitEvent(side->st, side->searched[half], 0) != hipSuccess) return fail();
                emit_on = side->st;
            }
            hipLaunchKernelGGL((k_hc_parse_emit<R>), dim3((nb + 3u) / 4u), dim3(256), 0, emit_on, d_in, d_in_off, d_in_len,
                               d_out, d_out_off, d_out_cap, d_result, static_cast<const R *>(res), stride, b0, nb, max_in_len);
            if (side && hipEventRecord(side->emitted[half], side->st) != hipSuccess) return fail();
        }
        if (side)     // join: everything enqueued here is ordered before whatever the caller enqueues on `stream` next
            for (uint32_t k = 0; k < 2u && k < round; k++)
                if (hipStreamWaitEvent(stream, side->emitted[k], 0) != hipSuccess) return fail();
        return hipGetLastError() == hipSuccess ? 0 : -7;
    }
    // search-every-position pipeline (levels 10-12, and the A/B switch).  The price-based parse of round r runs on the
    // side stream beside K2 of round r + 1: the parse keeps its records in LDS and leaves the vector memory path alone,
    // the search is all gathers.  K1 of round r + 1 goes FIRST (it needs a CU's whole LDS, and a CU that the parse's
    // small workgroups keep refilling never has it free: K1 launched beside the parse waited 30-45 ms per round), so
    // links and results both alternate between two halves of their areas:
    //   stream: K1(0) K2(0) K1(1) | K2(1) K1(2) | K2(2) ...        side: parse(0) | parse(1) | ...
    static const bool opt_no_overlap = zlz4_tune_env("ZLZ4_HC_NO_OVERLAP") != nullptr;
    HcSideStream *side = (optimal && chunk >= 2u && nblocks > chunk / 2u && !opt_no_overlap) ? hc_side_stream() : nullptr;
    const uint32_t sub = side ? chunk / 2u : chunk;
    auto fail = [&](int rc) -> int { if (side) (void)hipStreamSynchronize(side->st); return rc; };
    auto links_of = [&](uint32_t round) { return d_link + (uint64_t)(side ? (round & 1u) : 0u) * sub * stride; };
    auto launch_k1 = [&](uint32_t round, uint32_t b0) {
        const uint32_t nb = nblocks - b0 < sub ? nblocks - b0 : sub;
        hipLaunchKernelGGL((k_hc_build_links<T>), dim3(nb), dim3(64 * kLinkWaves), kHcTableSize * 4u + 4096u * sizeof(T) + 16u, stream, d_in, d_in_off,
                           d_in_len, links_of(round), stride, b0, nb, max_in_len);
    };
    uint32_t round = 0;
    if (nblocks) launch_k1(0, 0);
    for (uint32_t b0 = 0; b0 < nblocks; b0 += sub, round++) {
        const uint32_t nb = nblocks - b0 < sub ? nblocks - b0 : sub;
        const uint32_t half = side ? (round & 1u) : 0u;
        R *res = d_res + (uint64_t)half * sub * stride;
        if (side && round >= 2u && hipStreamWaitEvent(stream, side->emitted[half], 0) != hipSuccess) return fail(-7);   // parse(round - 2) read this half
        // every position (the price-based parse of levels 10-12 looks results up everywhere; blocks > 64 KiB)
        // (one-wave workgroups: 376 / 401 / 416 ms for 64 / 128 / 256 threads on configs[3] -- the wavefronts of a
        //  workgroup finish at very different times and a four-wave workgroup keeps its slots until the last one is done)
        hipLaunchKernelGGL((k_hc_search<T, R>), dim3((np_max + 63u) / 64u, nb), dim3(64), 0, stream, d_in,
                           d_in_off, d_in_len, links_of(round), stride, res, b0, nb, max_attempts, optimal ? 1 : 0, max_in_len);
        if (side && b0 + sub < nblocks) launch_k1(round + 1u, b0 + sub);      // (its half of the links was last read by K2(round - 1))
        if (optimal) {
            hipStream_t parse_on = stream;
            if (side) {
                if (hipEventRecord(side->searched[half], stream) != hipSuccess ||
                    hipStreamWaitEvent(side->st, side->searched[half], 0) != hipSuccess) return fail(-7);
                parse_on = side->st;
            }
            const int rc = zlz4_launch_hc_opt_parse(parse_on, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap, d_result,
                                                    res, stride, sizeof(R) == 8 ? 1 : 0, d_opt, b0, nb, sufficient_len, max_in_len);
            if (rc != 0) return fail(rc);
            if (side && hipEventRecord(side->emitted[half], side->st) != hipSuccess) return fail(-7);
        } else {
            hipLaunchKernelGGL((k_hc_parse_emit<R>), dim3((nb + 3u) / 4u), dim3(256), 0, stream, d_in, d_in_off, d_in_len,
                               d_out, d_out_off, d_out_cap, d_result, res, stride, b0, nb, max_in_len);
        }
        if (!side && b0 + sub < nblocks) launch_k1(round + 1u, b0 + sub);
    }
    if (side)     // join
        for (uint32_t k = 0; k < 2u && k < round; k++)
            if (hipStreamWaitEvent(stream, side->emitted[k], 0) != hipSuccess) return fail(-7);
    return hipGetLastError() == hipSuccess ? 0 : -7;
}

}  // namespace zlz4

namespace {
constexpr uint32_t kHcChunkBlocks = 8192;   // blocks per round (bounds the workspace: 3.5 GiB for 64 KiB blocks; 4096 / 8192 / 14043: 52.3 / 50.5 / 51.5 ms on configs[3])
bool hc_small(uint32_t max_in_len) {
    // ZLZ4_HC_HBM_LINKS (A/B switch for profiles/): blocks <= 64 KiB through the variant that keeps the links in HBM
    static const bool force_hbm = zlz4_tune_env("ZLZ4_HC_HBM_LINKS") != nullptr;
    return max_in_len <= 65536u && !force_hbm;
}
uint64_t hc_per_block_bytes(uint32_t max_in_len) {
    const uint64_t chain = (((uint64_t)max_in_len + 15u) & ~15ull) * (hc_small(max_in_len) ? 6u : 12u);   // links + results
    uint64_t opt = zlz4_hc_opt_workspace_bytes(1);                                                           // levels 10-12
    const uint64_t bitmap = hc_small(max_in_len) ? 0 : ((((uint64_t)max_in_len + 15u) & ~15ull) / 32u + 2u) * 4u;   // visited bits (HBM links)
    if (bitmap > opt) opt = bitmap;
    const uint64_t mid = zlz4_hc_mid_workspace_bytes(1);                                                     // level 2
    return chain + opt > mid ? chain + opt : mid;
}
uint32_t hc_chunk(uint32_t nblocks, uint32_t max_in_len) {
    // keep the workspace around <= 6 GiB for big blocks
    uint64_t c = (6ull << 30) / hc_per_block_bytes(max_in_len);
    if (c < 1) c = 1;
    if (c > kHcChunkBlocks) c = kHcChunkBlocks;
    if (c > nblocks) c = nblocks ? nblocks : 1;
    return (uint32_t)c;
}
}  // namespace

// one workspace size for every level (the caller need not know which strategy a level maps to)
extern "C" size_t zlz4_hc_workspace_bytes(uint32_t nblocks, uint32_t max_in_len) {
    return (size_t)(hc_chunk(nblocks, max_in_len) * hc_per_block_bytes(max_in_len));
}

// src/lz4hc.zig:72-86: level 2 -> lz4mid; 3..9 -> lz4hc, nbSearches 4..256; 10..12 -> lz4opt (96/64, 512/128, 16384/4096)
extern "C" int zlz4_launch_compress_hc(hipStream_t stream, const uint8_t *d_in, const uint64_t *d_in_off,
                                       const uint32_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off,
                                       const uint32_t *d_out_cap, int64_t *d_result, uint32_t nblocks,
                                       uint32_t max_in_len, int32_t level, void *ws, size_t ws_bytes) {
    if (nblocks == 0) return 0;
    if (level < 2 || level > 12) return -8;
    if (ws_bytes < zlz4_hc_workspace_bytes(nblocks, max_in_len)) return -5;
    const uint32_t chunk = hc_chunk(nblocks, max_in_len);
    if (level == 2) {
        // lz4mid needs its two tables only (128 KiB per block): the same workspace holds more blocks per launch, and one
        // block per wavefront wants as many blocks in flight as the chip can hold
        uint64_t mid_chunk = ws_bytes / zlz4_hc_mid_workspace_bytes(1);
        if (mid_chunk > nblocks) mid_chunk = nblocks;
        return zlz4_launch_hc_mid(stream, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap, d_result, nblocks, ws,
                                  (uint32_t)mid_chunk, max_in_len);
    }
    const bool optimal = level >= 10;
    static const int32_t opt_nb[3] = {96, 512, 16384};
    static const uint32_t opt_target[3] = {64, 128, 4096};
    const int32_t max_attempts = optimal ? opt_nb[level - 10] : 1 << (level - 1);   // 3 -> 4 ... 9 -> 256
    const uint32_t sufficient = optimal ? opt_target[level - 10] : 0;
    if (hc_small(max_in_len))
        return zlz4::launch_hc_chunked<uint16_t, uint32_t>(stream, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap,
                                                           d_result, nblocks, max_in_len, max_attempts, ws, chunk, optimal,
                                                           sufficient);
    return zlz4::launch_hc_chunked<uint32_t, uint64_t>(stream, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap,
                                                       d_result, nblocks, max_in_len, max_attempts, ws, chunk, optimal,
                                                       sufficient);
}
