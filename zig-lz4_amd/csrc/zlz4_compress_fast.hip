// zlz4_compress_fast.hip -- LZ4 "fast" block compressor for gfx950, one wavefront per block.
//
// Replaces lz4.compressFast / lz4.compressDefault (reference src/lz4.zig:283-447,
// helpers :449-519).  Output is byte-identical to the Zig algorithm, including
// the quirks listed in SURVEY.md Appendix A.1:
//   Q1 position 0 is never inserted / never matchable (0 == empty)   :317, :345
//   Q2 the skip schedule  step = searchMatchNb >> 6  with searchMatchNb starting at
//      `acceleration`, and the bail-out test on the post-increment forwardIp      :322-338
//   Q3 probe = read table, 4 validity tests, unconditional put                     :342-350
//   Q4 no backward extension, Q5 forward extension up to srcSize-5               :401-413
//   Q6 after a match only the byte right after it is inserted                    :435-442
//
// How the serial probe loop becomes 64-wide without changing a byte
// -----------------------------------------------------------------
// A search that starts at F0 visits a position sequence that depends only on F0 and
// the acceleration a (not on data): with c = max(64, a) and S(x) = sum_{y<x} (y >> 6)
//     U(0) = F0                        bail iff F0 + a > L
//     U(1) = F0 + a                    bail iff U(1) + (a >> 6) > L
//     U(u) = F0 + a + S(c+u-1) - S(c)  bail iff U(u) + ((c+u-1) >> 6) > L      (u >= 2)
// (L = srcSize - 12).  The iterations in which the reference's step is 0 re-probe the
// same position, see match == ip, fail `match < ip` and re-put the same value: they
// change nothing and are skipped.  Lane i of a batch takes probe u = ub + i.  What a
// probe reads from the hash table is either the value from before the batch or the
// position of the nearest earlier lane of the batch with the same hash; those lanes
// are found with one LDS write/read-back (losers of the write race are members of a
// duplicate-hash group) plus one ballot per duplicate group.  The first valid lane
// wins (ballot + ffs); lanes after it put their old table value back, so the table
// ends up exactly as the serial loop would have left it.
#include <cstdlib>

#include "zlz4_device.hpp"

// Diagnostic build only (-DZLZ4_STAMPS): per-phase shader-cycle sums, see profiles/ notes.
#ifdef ZLZ4_STAMPS
__device__ unsigned long long g_zlz4_stamps[24];
#define STAMP_DECL unsigned long long st_acc[24] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}; unsigned long long st_last = __builtin_amdgcn_s_memtime();
#define STAMP(i) do { unsigned long long st_now; __builtin_amdgcn_sched_barrier(0); \
                      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_now) :: "memory"); \
                      __builtin_amdgcn_sched_barrier(0); st_acc[i] += st_now - st_last; st_last = st_now; } while (0)
#define STAMP_COUNT(i) do { st_acc[i] += 1; } while (0)
#define STAMP_FLUSH do { if (lane == 0) for (int st_k = 0; st_k < 24; st_k++) atomicAdd(&g_zlz4_stamps[st_k], st_acc[st_k]); } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_COUNT(i)
#define STAMP_FLUSH
#endif

namespace zlz4 {

__device__ __forceinline__ uint32_t hash4(uint32_t seq) { return (seq * kHashMul) >> 20; }   // src/lz4.zig:75-77

// S(x) = sum_{y < x} (y >> 6)
__device__ __forceinline__ uint32_t skip_sum(uint32_t x) {
    const uint32_t q = x >> 6, r = x & 63u;
    return 32u * q * (q - 1u) + q * r;      // q == 0 -> 0 (32*0*(-1) wraps to 0)
}

// literal-only sequence: compressAsLiterals (:449-482) and finishCompression (:484-519)
__device__ __forceinline__ int64_t emit_last_literals(uint8_t *dst, uint32_t dst_len, uint32_t op,
                                                       const uint8_t *lit_src, uint32_t lit, uint32_t lane) {
    if (lit == 0) return (int64_t)op;                                   // :488
    const uint32_t nle = ext_len_bytes(lit);
    if ((uint64_t)op + 1u + nle + lit > dst_len) return kErrOutputTooSmall;   // :491-514 / :454-477
    if (lane == 0) dst[op] = (uint8_t)((lit >= 15u ? 15u : lit) << 4);
    if (lit >= 15u) write_ext_len(dst + op + 1u, lit, lane);
    copy_bytes(dst + op + 1u + nle, lit_src, lit, lane);
    return (int64_t)(op + 1u + nle + lit);
}

// Wave-cooperative forward extension (:401-413): continues from `mlen` equal bytes after the 4 MINMATCH
// bytes, 16 B per lane (1 KiB per step), first mismatch or the srcSize-5 limit found with ballot + ffs.
__device__ __forceinline__ uint32_t extend_match(const uint8_t *__restrict__ src, uint32_t m_pos, uint32_t m_cand,
                                                 uint32_t mlen, uint32_t match_limit, uint32_t src_size,
                                                 uint32_t lane) {
    const uint32_t ip0 = m_pos + kMinMatch, mt0 = m_cand + kMinMatch;
    for (;;) {
        const uint32_t p = ip0 + mlen + lane * 16u;         // first byte this lane compares
        uint32_t n = 0;                                     // how many bytes it may compare
        if (p < match_limit) n = (match_limit - p) < 16u ? (match_limit - p) : 16u;
        uint32_t d = 0;                                     // equal bytes found
        if (n > 0) {
            const uint32_t q = mt0 + mlen + lane * 16u;
            if (p + 16u <= src_size) {
                d = first_diff16(ld128(src + p), ld128(src + q));
                if (d > n) d = n;
            } else {
                while (d < n && src[p + d] == src[q + d]) d++;
            }
        }
        const uint64_t stop = ballot(d < 16u);              // mismatch or limit inside this lane's chunk
        if (stop) {
            const uint32_t sl = first_lane(stop);
            return mlen + sl * 16u + rdlane(d, sl);
        }
        mlen += 1024u;
    }
}

// T = uint16_t when every stored position fits 16 bits, else uint32_t.
// kTag: every table entry carries 8 more bits of the hashed sequence's product (bits 12..19; the index is bits 20..31),
// so a probe whose candidate cannot pass the 4-byte test of :348 is settled from LDS and never gathers the candidate's
// bytes from memory (on text more than half of all candidates are false: 4096 slots, 64 Ki positions).  A tag mismatch
// implies a 4-byte mismatch, so no decision changes.  1 = a separate u8 array beside the u16 table (12 KiB per
// wavefront), 2 = packed into bits 24..31 of a u32 entry (positions < 2^24), 0 = none.
template <typename T, int kTag>
__global__ __launch_bounds__(256) void k_compress_fast(
    const uint8_t *__restrict__ d_in, const uint64_t *__restrict__ d_in_off,
    const uint32_t *__restrict__ d_in_len, uint8_t *__restrict__ d_out,
    const uint64_t *__restrict__ d_out_off, const uint32_t *__restrict__ d_out_cap,
    int64_t *__restrict__ d_result, uint32_t nblocks, uint32_t acceleration, uint32_t max_in_len) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_in_wg = threadIdx.x >> 6;
    const uint32_t blk = rfl(blockIdx.x * (blockDim.x >> 6) + wave_in_wg);
    if (blk >= nblocks) return;
    // HashTable, src/lz4.zig:263-277.  volatile: lanes communicate through it (write / read-back race
    // detection), so the compiler must neither forward stores to loads nor drop the re-reads.
    // (typed as address space 3: a generic `volatile T *` would compile to FLAT loads/stores with a full
    //  `s_waitcnt vmcnt(0)` each, i.e. every table access would also wait for the global gathers in flight)
    typedef __attribute__((address_space(3))) volatile T lds_entry;
    typedef __attribute__((address_space(3))) volatile uint8_t lds_tag;
    lds_entry *table = (lds_entry *)lds_raw + wave_in_wg * 4096u;
    // kTag == 1: the tag arrays follow the tables of all wavefronts of the workgroup (never initialised: an empty slot
    // is recognised by its position 0 before its tag is looked at)
    lds_tag *tags = (lds_tag *)lds_raw + (blockDim.x >> 6) * 4096u * (uint32_t)sizeof(T) + wave_in_wg * 4096u;
    constexpr uint32_t kPosMask = kTag == 2 ? 0x00FFFFFFu : 0xFFFFFFFFu;

    const uint8_t *src = d_in + d_in_off[blk];
    uint8_t *dst = d_out + d_out_off[blk];
    const uint32_t src_size = rfl(d_in_len[blk]);
    const uint32_t dst_len = rfl(d_out_cap[blk]);

    int64_t res;
    if (src_size > kMaxInput) {                                         // :296
        res = kErrInputTooLarge;
    } else if (src_size > max_in_len) {                                 // the table width was chosen from max_in_len
        res = kErrInvalidState;
    } else if (src_size == 0) {                                         // :299
        res = 0;
    } else if (src_size < kMfLimit + 1u) {                              // :302-304
        res = emit_last_literals(dst, dst_len, 0, src, src_size, lane);
    } else {
        // HashTable.init(): zero fill (:266-268)
        {
            u32x4 z = {0, 0, 0, 0};
            u32x4 *t4 = reinterpret_cast<u32x4 *>(lds_raw) + wave_in_wg * (4096u * sizeof(T) / 16u);
            const uint32_t n16 = 4096u * sizeof(T) / 16u;
            for (uint32_t k = lane; k < n16; k += 64u) t4[k] = z;
        }
        const uint32_t accel = acceleration < 1u ? 1u : (acceleration > 65537u ? 65537u : acceleration);   // :321
        const uint32_t cbase = accel > 64u ? accel : 64u;
        const uint32_t s_cbase = skip_sum(cbase);
        const uint32_t L = src_size - kMfLimit;                         // mflimitPlusOne :313
        const uint32_t match_limit = src_size - kLastLiterals;          // :314
        const uint64_t lane_bit = 1ull << lane;
        const uint64_t lanes_below = lane_bit - 1ull;

        uint32_t anchor = 0, op = 0;
        uint32_t F0 = 1;                                                // :317
        bool has_ins = false;    // pending put(anchor) of :438-441, folded into the next batch as lane 0
        bool failed = false;
        STAMP_DECL
        STAMP(0);   // table init

        // a window keeps going to its last lane (handing over to the next window earlier, once it holds a match, was
        // measured at every lane: 49 -> 55.6 ms, 64 -> 46.8 ms on configs[1]; any choice gives the same bytes)
        uint32_t guard = 0;      // every round of this loop consumes at least one input byte
        // forward bytes of the next window, loaded as soon as its anchor is known (before the emission and the table
        // fix-up of the current window, which hide the load)
        u32x4 fwd_pf = {0, 0, 0, 0};
        uint32_t pf_anchor = 0xFFFFFFFFu;
        while (F0 < L) {                                                // :320
            if (++guard > src_size) { failed = true; break; }           // unreachable; never spin on the GPU
            int32_t ub = -1;      // probe index of lane 0 of the next generic batch (-1 = pending insert pseudo-probe)

            // =====================================================================================
            // Window path (acceleration 1, away from the end of the block): lane i <-> position
            // anchor + i.  Every search that starts inside the window probes consecutive positions
            // (the first 66 probes of a search have stride 1), so all sequences that begin in these
            // 64 positions are resolved from registers: one 16-byte forward load per lane, one table
            // read / speculative put / read-back, one candidate gather -- then one short loop
            // iteration per sequence.  `ins` = lanes the serial loop would have put() so far; a lane's
            // table read is its nearest earlier `ins` lane with the same hash, else the pre-window
            // value.  At the end lanes not in `ins` put their old value back.
            // =====================================================================================
            if (accel == 1u && F0 == anchor + 1u && (has_ins || anchor == 0u) && (uint64_t)anchor + 192u < L) {
              bool moved = false;           // the last window of the run advanced the anchor ...
              bool to_generic = false;      // ... or handed its search over to the generic path
              for (;;) {                    // consecutive windows: the next one starts without re-deriving the entry test
                const uint32_t A = anchor;
                const uint32_t pos = A + lane;
                const bool wr = has_ins || lane > 0;                    // position 0 is never inserted (Q1)
                const u32x4 fwd = (pf_anchor == A) ? fwd_pf : ld128(src + pos);
                const uint32_t prod = fwd.x * kHashMul;
                const uint32_t h = prod >> 20;                          // :341
                const uint32_t tg = (prod >> 12) & 0xFFu;
                const uint32_t mine = kTag == 2 ? (pos | (tg << 24)) : (uint32_t)(T)pos;     // my table entry
                STAMP(1);
                STAMP_COUNT(16);
                uint32_t old_e = 0, rb = 0, told = tg;
                if (wr) {
                    old_e = table[h];                                   // :342
                    if (kTag == 1) told = tags[h];
                }
                if (kTag == 2) told = old_e >> 24;
                const uint32_t old = old_e & kPosMask;
                // pre-window candidates: the old table value passes `match > 0`, `match < ip` (always) and
                // the distance test (:345-347); its bytes are gathered once for the whole window.  The gather is
                // issued right away so that its latency overlaps the speculative put / read-back below.
                const bool old_ok = wr && old > 0 && (old + kMaxDist >= pos) && (kTag == 0 || told == tg);
                u32x4 cold = {0, 0, 0, 0};
                if (old_ok) cold = ld128(src + old);
                // the second compare level (bytes 16..47, below) is a round trip of its own behind the first -- unless it is
                // asked for now: a lane whose candidate and the candidate twelve lanes on are twelve bytes apart sits at the
                // start of >= 16 matching bytes almost certainly, and fetches its 32 more bytes with the gather (a wrong
                // guess costs two loads; a lane that was not guessed fetches them when it knows, as before)
                // (blocks > 64 KiB only: few large blocks leave the chip to one wavefront per SIMD, where the round trip is
                //  what counts -- 1024 x 4 MiB 121.5 -> 118.9 ms; with 20 wavefronts per CU the extra requests cost more
                //  than the round trip, configs[1] 41.3 -> 41.6 ms, D-reptext 58.9 -> 60.3 ms)
                constexpr bool kGuess2 = sizeof(T) == 4;
                const uint32_t old12 = kGuess2 ? shfl(old, (lane + 12u) & 63u) : 0u;
                const bool pred2 = kGuess2 && old_ok && lane < 52u && old12 == old + 12u;
                u32x4 f2 = {0, 0, 0, 0}, c2 = f2, f3 = f2, c3 = f2;
                if (pred2) {
                    f2 = ld128(src + pos + 16u); c2 = ld128(src + old + 16u);
                    f3 = ld128(src + pos + 32u); c3 = ld128(src + old + 32u);
                }
                STAMP(3);
                if (wr) table[h] = (T)mine;                             // :350 (speculative)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if (wr) rb = table[h];
                uint64_t losers = ballot(wr && rb != mine);
                uint64_t grp = lane_bit;
                while (losers) {                                        // one round per duplicate-hash group
                    const uint32_t l = first_lane(losers);
                    const uint32_t hh = rdlane(h, l);
                    const uint64_t same = ballot(wr && h == hh);
                    if (wr && h == hh) grp = same;
                    losers &= ~same;
                }
                STAMP(2);

                // What a probe at lane i finds if its table slot still holds the pre-window value: validity
                // (:345-348), the first 12 bytes of forward extension (:401-413) and the offset.  For a lane
                // whose hash is unique in the window this does not depend on the parse at all.
                const bool vo = old_ok && cold.x == fwd.x;
                uint32_t mlo;
                {
                    // selects only (a nested ?: chain compiles to exec-mask branches, i.e. scalar instructions)
                    const uint32_t x1 = fwd.y ^ cold.y, x2 = fwd.z ^ cold.z, x3 = fwd.w ^ cold.w;
                    const uint32_t xs = x1 ? x1 : (x2 ? x2 : x3);
                    const uint32_t xbase = x1 ? 0u : (x2 ? 4u : 8u);
                    mlo = xs ? xbase + ((uint32_t)__builtin_ctz(xs) >> 3) : 12u;
                }
                // second level: the few lanes whose 16 bytes all match compare 16 more (matches of 16..31 bytes are a
                // fifth of all sequences on text; without this each of them costs an exact step and its own emission)
                const bool need2 = vo && mlo == 12u;
                if (need2 && !pred2) {                        // (all four loads in one round trip)
                    f2 = ld128(src + pos + 16u); c2 = ld128(src + old + 16u);
                    f3 = ld128(src + pos + 32u); c3 = ld128(src + old + 32u);
                }
                if (need2) {
                    const uint32_t d2 = first_diff16_sel(f2, c2);
                    mlo += d2 == 16u ? 16u + first_diff16_sel(f3, c3) : d2;
                }
                // a lane whose hash no EARLIER lane of the window shares reads the pre-window value whatever the parse does --
                // the only lane of its hash, or the first of a duplicate group (round 3: the first of a group used to take
                // an exact step like the others, which found no in-window candidate and fell back to the same registers;
                // 0.55 of the 0.72 exact steps per window on text were on duplicate-hash lanes, half of them first ones)
                const bool single = (grp & lanes_below) == 0;
                const bool oldfast = vo && mlo < 44u;         // result against the pre-window value is complete in registers
                const uint64_t wrmask = ballot(wr);
                const uint64_t cfast = ballot(oldfast);                                       // usable if no in-window put precedes
                const uint64_t slow = ballot(wr && ((!single && !oldfast) || (vo && mlo >= 44u)));   // exact step if reached
                const uint64_t nsing = ballot(wr && !single);
                // per lane i: J = first cfast lane >= i, S = first slow lane >= i (64 = none),
                // E = lane of the new anchor if the search that starts at i ends with the match at J
                uint32_t J, S;
                {
                    const uint64_t mj = cfast >> lane, ms = slow >> lane;
                    J = mj ? lane + (uint32_t)__builtin_ctzll(mj) : 64u;
                    S = ms ? lane + (uint32_t)__builtin_ctzll(ms) : 64u;
                }
                uint32_t v_end = lane + kMinMatch + mlo;                          // anchor lane after a match at this lane
                uint32_t mlo_e = mlo, off_e = pos - old;                          // match length / offset the flush emits for a match lane
                // fast-run step of a search that starts at lane f, precomputed for every f: PK = j | (v_end[j] << 6) when
                // the search ends with the in-register match at j = J[f] (no slow lane first, j's hash unique in the
                // window), else ~0.  (Exact steps only rewrite v_end of lanes the search has already passed.)
                // for the exact step: XP = first lane >= f that can match at all (min(J, S), 64 = none) | its nsing bit << 7;
                // Q = what a probe finds against the pre-window value: valid (:345-348) | matched extension bytes << 1
                uint32_t XP, Q;
                {
                    const uint32_t xm = J < S ? J : S;
                    XP = xm | ((uint32_t)((nsing >> (xm & 63u)) & 1ull) << 7);
                    Q = (vo ? 1u : 0u) | (mlo << 1);
                }
                uint32_t PK;
                {
                    const uint32_t jc = J & 63u;
                    const uint32_t ve_j = shfl(v_end, jc);
                    const bool fastok = J < 64u && S > J && !((nsing >> jc) & 1ull);
                    PK = fastok ? (J | (ve_j << 6)) : 0xFFFFFFFFu;
                }

                STAMP(6);
                uint32_t f = 1;          // next lane to probe
                uint32_t a = 0;          // lane of the current anchor
                uint32_t nseq = 0;
                uint64_t covered_x = 0;  // lanes strictly inside a match that was emitted immediately (Q6: never put())
                uint64_t mm_win = 0;     // match lanes of the runs already flushed
                bool continue_generic = false;
                // with less than 512 bytes of room left every sequence takes the exact step, which checks the capacity
                const bool tight = dst_len - op < 512u;
                uint32_t a0 = a, op0 = op;   // anchor lane / output position at the start of the pending (unflushed) run
                uint64_t mm_run = 0;         // match lanes of the pending run
                // lanes strictly inside a match so far (never put(), Q6): from the match lanes and their end lanes
                auto covered_now = [&]() -> uint64_t {
                    const uint64_t mb = (mm_win | mm_run) & lanes_below;
                    const uint32_t pj = mb ? 63u - (uint32_t)__builtin_clzll(mb) : 0u;
                    const uint32_t pe = shfl(v_end, pj);                // (unconditional: see zlz4_device.hpp)
                    return covered_x | ballot(mb != 0 && lane < pe);
                };
                // flush of the pending run: every offset from popcounts, three stores for all its sequences
                auto flush_run = [&]() {
                    STAMP(9); STAMP_COUNT(21);
                    const uint64_t mb = mm_run & lanes_below;
                    const bool has_prev = mb != 0;
                    const uint32_t pj = has_prev ? 63u - (uint32_t)__builtin_clzll(mb) : 0u;   // previous match lane
                    const uint32_t pend_all = shfl(v_end, pj);          // (unconditional: see zlz4_device.hpp)
                    const uint32_t pend = has_prev ? pend_all : a0;     // first lane of my literal run
                    const bool cov = has_prev && lane < pend_all;       // strictly inside a match of this run
                    const bool is_m = (mm_run & lane_bit) != 0;
                    const uint32_t jlast = 63u - (uint32_t)__builtin_clzll(mm_run);
                    const bool is_lit = lane >= a0 && lane < jlast && !cov && !is_m;
                    const uint64_t litmask = ballot(is_lit);
                    const uint64_t extm = ballot(is_m && mlo_e >= 15u);                 // matches with one length-extension byte (:416-429)
                    // literal runs of 15..63 bytes carry one extension byte too (:368-382): a lane's sequence is the
                    // first match lane at or above it, its literal count that lane minus the start of the run
                    const uint64_t at_or_above = mm_run & ~lanes_below;
                    const uint32_t my_m = at_or_above ? (uint32_t)__builtin_ctzll(at_or_above) : lane;
                    const uint32_t own_l = (my_m - pend >= 15u) ? 1u : 0u;
                    const uint64_t lextm = ballot(is_m && own_l != 0u);
                    const uint32_t k = (uint32_t)__popcll(mb);                          // sequences completed before me
                    const uint32_t lb = (uint32_t)__popcll(litmask & lanes_below);      // literal bytes before me
                    const uint32_t o1 = op0 + 3u * k + lb + 1u + (uint32_t)__popcll(extm & lanes_below) +
                                        (uint32_t)__popcll(lextm & lanes_below) + own_l;
                    if (is_lit) dst[o1] = (uint8_t)fwd.x;               // literals (:390)
                    if (is_m) {
                        const uint32_t lit_k = lane - pend;             // :360
                        uint8_t *tk = dst + (o1 - 1u - lit_k - own_l);
                        tk[0] = (uint8_t)(((lit_k < 15u ? lit_k : 15u) << 4) | (mlo_e < 15u ? mlo_e : 15u));   // token
                        if (own_l) tk[1] = (uint8_t)(lit_k - 15u);
                        const uint16_t off16 = (uint16_t)off_e;                          // :395
                        __builtin_memcpy(dst + o1, &off16, 2);
                        if (mlo_e >= 15u) dst[o1 + 2u] = (uint8_t)(mlo_e - 15u);         // < 255 (mlen < 270 in a run)
                    }
                    const uint32_t nm = (uint32_t)__popcll(mm_run);
                    op = op0 + 3u * nm + (uint32_t)__popcll(litmask) + (uint32_t)__popcll(extm) + (uint32_t)__popcll(lextm);
                    mm_win |= mm_run;
                    mm_run = 0;
                    STAMP(10);
                };
                for (;;) {
                    // ---- fast run: a minimal scalar loop that only collects the match lanes.  A search that
                    //      starts at f ends at J[f] when no slow lane comes first, the lane's hash is unique in the
                    //      window (its probe reads the pre-window value whatever was put before) and the literal
                    //      run fits the token nibble. ----
                    if (mm_run == 0) { a0 = a; op0 = op; }
                    if (!tight) {
                        // hand-scheduled scalar loop (the compiler spends ~25 scalar instructions per trip on the
                        // boolean plumbing; the scalar unit is what bounds this kernel):
                        //   while (f < 64) { pk = PK[f]; j = pk & 63; if (pk == ~0) break;
                        //                    mm_run |= 1 << j; nseq++; a = pk >> 6; f = a + 1; }          (two trips per branch back)
                        uint32_t t_pk, t_j;
#define ZLZ4_FAST_RUN_TRIP                                   \
                            "s_cmp_gt_u32 %[f], 63\n\t"      \
                            "s_cbranch_scc1 3f\n\t"          \
                            "v_readlane_b32 %[pk], %[PK], %[f]\n\t" \
                            "s_cmp_eq_u32 %[pk], -1\n\t"     \
                            "s_cbranch_scc1 3f\n\t"          \
                            "s_lshr_b32 %[a], %[pk], 6\n\t"  \
                            "s_add_u32 %[f], %[a], 1\n\t"    \
                            "s_and_b32 %[j], %[pk], 63\n\t"  \
                            "s_bitset1_b64 %[mm], %[j]\n\t"  \
                            "s_add_u32 %[nseq], %[nseq], 1\n\t"
                        // (five scalar instructions lie between the write of f and the v_readlane that uses it as its lane
                        //  select: the ISA asks for four wait states there)
                        asm volatile(
                            "s_nop 3\n"
                            "1:\n\t"
                            ZLZ4_FAST_RUN_TRIP
                            ZLZ4_FAST_RUN_TRIP
                            "s_branch 1b\n"
                            "3:\n"
                            : [f] "+s"(f), [a] "+s"(a), [nseq] "+s"(nseq), [mm] "+s"(mm_run), [pk] "=&s"(t_pk), [j] "=&s"(t_j)
                            : [PK] "v"(PK)
                            : "scc");
#undef ZLZ4_FAST_RUN_TRIP
                    }
                    STAMP(8);
                    if (f >= 64u) {      // window done (a = last anchor lane, possibly >= 64)
                        if (nseq == 0u) continue_generic = true;    // every lane probed, no match
                        break;
                    }

                    // ---- exact step for the probe at the first lane >= f that can match at all ----
                    const uint32_t xp = rdlane(XP, f);
                    const uint32_t x = xp & 127u;
                    if (x >= 64u) {
                        if (nseq == 0u) continue_generic = true;   // the search goes on past the window -> generic batches
                        // else: restart a fresh window at the current anchor (its lanes > a are re-probed there)
                        break;
                    }
                    // earlier put()s of this window with the same hash (only lanes of duplicate-hash groups can have one):
                    // every lane below x that is not strictly inside a match has been put
                    uint64_t pm = 0;
                    if (xp >> 7) {
                        const uint64_t grp_x = (uint64_t)rdlane((uint32_t)grp, x) | ((uint64_t)rdlane((uint32_t)(grp >> 32), x) << 32);
                        pm = grp_x & wrmask & ~covered_now() & ((1ull << x) - 1ull);
                    }
                    const uint32_t j = x;
                    const uint32_t m_pos = A + j;
                    uint32_t m_cand, mlen;
                    if (pm) {
                        // the probe reads the nearest earlier put of the window
                        const uint32_t pr = 63u - (uint32_t)__builtin_clzll(pm);
                        if (rdlane(fwd.x, pr) != rdlane(fwd.x, x)) { STAMP_COUNT(19); STAMP(9); f = x + 1u; continue; }   // :348
                        STAMP_COUNT(18);
                        m_cand = A + pr;
                        const uint64_t xa = ((uint64_t)(rdlane(fwd.z, j) ^ rdlane(fwd.z, pr)) << 32) | (rdlane(fwd.y, j) ^ rdlane(fwd.y, pr));
                        const uint32_t xb = rdlane(fwd.w, j) ^ rdlane(fwd.w, pr);
                        if (xa) mlen = (uint32_t)__builtin_ctzll(xa) >> 3;
                        else if (xb) mlen = 8u + ((uint32_t)__builtin_ctz(xb) >> 3);
                        else mlen = extend_match(src, m_pos, m_cand, 12u, match_limit, src_size, lane);
                    } else {
                        // the probe reads the pre-window value: the vector code has already compared up to 48 bytes
                        const uint32_t q = rdlane(Q, x);
                        if (!(q & 1u)) { STAMP_COUNT(19); STAMP(9); f = x + 1u; continue; }      // probed, put, no match: next probe
                        STAMP_COUNT(18);
                        mlen = q >> 1;
                        if (mlen < 44u && !tight) {
                            // v_end / mlo_e / off_e of lane j already describe this sequence: it just joins the run
                            mm_run |= 1ull << j;
                            nseq++;
                            a = j + kMinMatch + mlen;
                            STAMP(9);
                            if (a >= 64u) break;                            // the next window inserts it as its lane 0
                            f = a + 1u;
                            continue;
                        }
                        m_cand = rdlane(old, x);
                        if (mlen >= 44u) mlen = extend_match(src, m_pos, m_cand, 44u, match_limit, src_size, lane);
                    }
                    const uint32_t lit = j - a;
                    const uint32_t offset = m_pos - m_cand;
                    const uint32_t e = j + kMinMatch + mlen;                // lane of the new anchor (may be >= 64)
                    if (!tight && mlen < 270u) {
                        // simple sequence: joins the pending run, emitted by the flush
                        v_end = wrlane(e, j, v_end);
                        mlo_e = wrlane(mlen, j, mlo_e);
                        off_e = wrlane(offset, j, off_e);
                        mm_run |= 1ull << j;
                        nseq++;
                        a = e;
                        STAMP(9);
                        if (e >= 64u) break;                                // the next window inserts it as its lane 0
                        f = e + 1u;
                        continue;
                    }
                    if (mm_run) flush_run();
                    // immediate emission (:360-432), literals = low bytes of lanes a..j-1
                    const uint32_t nle = ext_len_bytes(lit), nme = ext_len_bytes(mlen);
                    const uint64_t seq_end = (uint64_t)op + 1u + nle + lit + 2u + nme;
                    if (seq_end > dst_len) { failed = true; break; }
                    if (lane == 0)
                        dst[op] = (uint8_t)(((lit >= 15u ? 15u : lit) << 4) | (mlen >= 15u ? 15u : mlen));
                    if (lit >= 15u) write_ext_len(dst + op + 1u, lit, lane);
                    uint8_t *o = dst + op + 1u + nle;
                    if (lane >= a && lane < j) o[lane - a] = (uint8_t)fwd.x;
                    o += lit;
                    if (lane < 2u) o[lane] = (uint8_t)(offset >> (8u * lane));
                    if (mlen >= 15u) write_ext_len(o + 2u, mlen, lane);
                    op = (uint32_t)seq_end;
                    {
                        const uint64_t upto_e = e >= 64u ? ~0ull : (1ull << e) - 1ull;
                        covered_x |= upto_e & ~((2ull << j) - 1ull);        // lanes j+1 .. e-1
                    }
                    nseq++;
                    a = e;
                    if (e >= 64u) break;                                    // the next window inserts it as its lane 0
                    f = e + 1u;
                }
                // (next_win implies everything the entry test above asks of the next round)
                const bool next_win = !continue_generic && !failed && a != 0u && (uint64_t)A + a + 192u < L;
                if (next_win) {
                    pf_anchor = A + a;
                    fwd_pf = ld128(src + pf_anchor + lane);
                }
                if (mm_run && !failed) flush_run();
                anchor = A + a;
                STAMP(4);
                // lanes the serial loop put(): below the frontier and not strictly inside a match
                const uint32_t f_end = continue_generic ? 64u : (a >= 64u ? 64u : a + 1u);
                const uint64_t ins = wrmask & ~covered_now() & (f_end >= 64u ? ~0ull : (1ull << f_end) - 1ull);
                if (failed) break;
                // ---- leave the table as the serial loop would have ----
                if (wr && !(ins & lane_bit)) table[h] = (T)old_e;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if ((ins & lane_bit) && (grp & ins & ~lanes_below & ~lane_bit) == 0) {
                    table[h] = (T)mine;
                    if (kTag == 1) tags[h] = (uint8_t)tg;               // (the speculative put left the old tag in place)
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                STAMP(5);
                moved = anchor != A;
                to_generic = continue_generic;
                if (next_win) {
                    has_ins = true;
                    F0 = anchor + 1u;
                    if (++guard > src_size) { failed = true; break; }       // unreachable; never spin on the GPU
                    continue;
                }
                break;
              }
                if (failed) break;
                if (!to_generic && moved) {
                    if (anchor < L) { has_ins = true; F0 = anchor + 1u; }
                    else { has_ins = false; F0 = L; }
                    continue;
                }
                // continue_generic: same search, next probe index 63 (lane 0 of the next batch is position F0 + 63).
                // (anchor == A without continue_generic cannot happen; if it ever did, the generic path below restarts
                //  the search from F0 -- the table then holds exactly the anchor's put -- and always makes progress.)
                if (to_generic) ub = 63;
            }

            // =====================================================================================
            // Generic path: any acceleration, block tail, searches longer than one window.
            // 64 probes of ONE search per step.
            // =====================================================================================
            bool found = false, bailed = false;
            uint32_t m_pos = 0, m_cand = 0, m_local = 0;
            bool m_local_done = false;
            for (;;) {
                const int32_t u = ub + (int32_t)lane;
                uint32_t pos, step_next;
                if (u <= 0) {
                    pos = (u < 0) ? F0 - 1u : F0;
                    step_next = accel;                  // u == 0: bail iff F0 + a > L  (:331-335, first iteration)
                } else if (u == 1) {
                    pos = F0 + accel;
                    step_next = accel >> 6;
                } else {
                    const uint32_t x = cbase + (uint32_t)u - 1u;
                    pos = F0 + accel + skip_sum(x) - s_cbase;
                    step_next = x >> 6;
                }
                const bool is_probe = u >= 0;
                const bool bail = is_probe && ((uint64_t)pos + step_next > L);
                const uint64_t bail_mask = ballot(bail);
                // bail is monotone in u: everything from the first bailing lane on is out of the search
                const uint32_t nb = bail_mask ? first_lane(bail_mask) : 64u;
                const bool active = (lane < nb) && (is_probe || has_ins);

                // forward data: 16 B when they are inside the block, else the 4 hashed bytes only
                const bool have16 = active && (pos + 16u <= src_size);
                u32x4 fwd = {0, 0, 0, 0};
                if (have16) fwd = ld128(src + pos);
                else if (active) fwd.x = ld32(src + pos);
                const uint32_t prod = fwd.x * kHashMul;
                const uint32_t h = prod >> 20;                          // :341
                const uint32_t tg = (prod >> 12) & 0xFFu;
                const uint32_t mine = kTag == 2 ? (pos | (tg << 24)) : (uint32_t)(T)pos;

                // table read / speculative put / read-back
                uint32_t old_e = 0, rb = 0, told = tg;
                if (active) {
                    old_e = table[h];                                   // :342
                    if (kTag == 1) told = tags[h];
                    table[h] = (T)mine;                                 // :350 (speculative)
                }
                if (kTag == 2) told = old_e >> 24;
                const uint32_t old = old_e & kPosMask;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if (active) rb = table[h];
                // duplicate-hash groups inside the batch
                uint64_t losers = ballot(active && rb != mine);
                uint64_t grp = lane_bit;
                int32_t pred = -1;
                while (losers) {
                    const uint32_t l = first_lane(losers);
                    const uint32_t hh = rdlane(h, l);
                    const uint64_t same = ballot(active && h == hh);
                    if (active && h == hh) {
                        grp = same;
                        const uint64_t below = same & lanes_below;
                        pred = below ? 63 - (int32_t)__clzll((long long)below) : -1;
                    }
                    losers &= ~same;
                }
                const uint32_t pred_pos = shfl(pos, (uint32_t)(pred < 0 ? 0 : pred));
                const uint32_t cand = pred >= 0 ? pred_pos : old;

                // the four validity tests of :345-348
                bool valid = is_probe && active && cand > 0 && cand < pos && (cand + kMaxDist >= pos) &&
                             (kTag == 0 || pred >= 0 || told == tg);
                u32x4 cnd = {0, 0, 0, 0};
                if (valid) {
                    if (have16) cnd = ld128(src + cand); else cnd.x = ld32(src + cand);
                    valid = cnd.x == fwd.x;
                }
                const uint64_t valid_mask = ballot(valid);

                if (valid_mask) {
                    // ---- match at lane wl (first valid probe, :352) ----
                    const uint32_t wl = first_lane(valid_mask);
                    // lanes after the winner never ran in the serial loop: undo their puts, then
                    // re-commit the last lane <= wl of every duplicate group
                    if (active && lane > wl) table[h] = (T)old_e;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    const uint64_t upto = (2ull << wl) - 1ull;   // lanes 0..wl
                    if (active && lane <= wl && ((grp & upto & ~lanes_below & ~lane_bit) == 0)) {
                        table[h] = (T)mine;
                        if (kTag == 1) tags[h] = (uint8_t)tg;
                    }
                    // local extension: bytes 4..15 of the two 16-byte reads (:401-413)
                    uint32_t loc = 0;
                    bool loc_done = false;
                    if (lane == wl) {
                        const uint32_t lim = match_limit - (pos + kMinMatch);   // bytes that may still be compared
                        if (have16) {
                            u32x4 a = fwd, b = cnd;
                            a.x = 0; b.x = 0;
                            loc = first_diff16(a, b) - 4u;       // 0..12
                            if (loc >= lim) { loc = lim; loc_done = true; }
                            else if (loc < 12u) loc_done = true;
                        }
                    }
                    m_pos = rdlane(pos, wl);
                    m_cand = rdlane(cand, wl);
                    m_local = rdlane(loc, wl);
                    m_local_done = rdlane((uint32_t)loc_done, wl) != 0;
                    found = true;
                    break;
                }
                if (bail_mask) { bailed = true; break; }                // :335-338 -> finishCompression
                // no match in 64 probes: all puts stand; fix duplicate groups so the last lane's position is stored
                if (active && ((grp & ~lanes_below & ~lane_bit) == 0)) {
                    if (grp != lane_bit) table[h] = (T)mine;
                    if (kTag == 1) tags[h] = (uint8_t)tg;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                ub += 64;
            }
            if (bailed || !found) break;
            STAMP_COUNT(20);

            // ---------------- forward extension (:401-413) ----------------
            const uint32_t mlen = m_local_done ? m_local
                                               : extend_match(src, m_pos, m_cand, m_local, match_limit, src_size, lane);

            // ---------------- emit the sequence (:360-432) ----------------
            const uint32_t lit = m_pos - anchor;                        // :360
            const uint32_t nle = ext_len_bytes(lit), nme = ext_len_bytes(mlen);
            const uint64_t seq_end = (uint64_t)op + 1u + nle + lit + 2u + nme;
            if (seq_end > dst_len) { failed = true; break; }            // :365-:427 (any of them)
            if (lane == 0)
                dst[op] = (uint8_t)(((lit >= 15u ? 15u : lit) << 4) | (mlen >= 15u ? 15u : mlen));
            if (lit >= 15u) write_ext_len(dst + op + 1u, lit, lane);
            uint8_t *o = dst + op + 1u + nle;
            copy_bytes(o, src + anchor, lit, lane);                     // :390
            o += lit;
            const uint32_t offset = m_pos - m_cand;                     // :395
            if (lane < 2u) o[lane] = (uint8_t)(offset >> (8u * lane));  // :397
            if (mlen >= 15u) write_ext_len(o + 2u, mlen, lane);
            op = (uint32_t)seq_end;

            // ---------------- after the match (:435-442) ----------------
            const uint32_t end = m_pos + kMinMatch + mlen;
            anchor = end;
            if (end < L) { has_ins = true; F0 = end + 1u; }
            else { has_ins = false; F0 = L; }
        }
        res = failed ? kErrOutputTooSmall
                     : emit_last_literals(dst, dst_len, op, src + anchor, src_size - anchor, lane);   // :337, :446
        STAMP(7);   // tail
        STAMP_FLUSH;
    }
    if (lane == 0) d_result[blk] = res;
}


}  // namespace zlz4

extern "C" int zlz4_launch_compress_fast(hipStream_t stream, const uint8_t *d_in, const uint64_t *d_in_off,
                                         const uint32_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off,
                                         const uint32_t *d_out_cap, int64_t *d_result, uint32_t nblocks,
                                         uint32_t max_in_len, uint32_t acceleration) {
    if (nblocks == 0) return 0;
    // experiment knob: extra dynamic LDS per workgroup lowers the number of resident waves per CU
    static const uint32_t lds_pad = [] { const char *e = zlz4_tune_env("ZLZ4_TUNE_LDS_PAD"); return e ? (uint32_t)atoi(e) : 0u; }();
    // every position stored in the table is < srcSize - 12, so 16-bit entries are exact up to 65547-byte blocks
    static const uint32_t tune_wpw = [] { const char *e = zlz4_tune_env("ZLZ4_TUNE_WPW"); return e ? (uint32_t)atoi(e) : 0u; }();
    // tags (see k_compress_fast): free in a u32 entry (configs[4] shape 132.6 -> 125.9 ms); beside the u16 table they cost
    // LDS, 20 -> 13 wavefronts per CU, and lose (configs[1] 42.8 -> 54.0 ms; profiles/r03_fast_compress_tags.md), so
    // the u16 build carries them only when asked to (tuning build: ZLZ4_TUNE_TAG=1)
    static const int tune_tag = [] { const char *e = zlz4_tune_env("ZLZ4_TUNE_TAG"); return e ? atoi(e) : -1; }();
#define ZLZ4_LAUNCH_FAST(KERN, T, TAG, WPW, LDS)                                                                      \
    hipLaunchKernelGGL((zlz4::KERN<T, TAG>), dim3((nblocks + (WPW) - 1) / (WPW)), dim3(64 * (WPW)), (LDS),              \
                       stream, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap, d_result, nblocks, acceleration,  \
                       max_in_len)
    if (max_in_len <= 65536u + 11u) {
        // 8 KiB of table (+ 4 KiB of tags) per wavefront in LDS -> 20 (13) wavefronts per CU whatever the workgroup size;
        // one-wave workgroups measured 5 % faster than four-wave ones on MI355X (44.1 / 45.0 / 46.6 ms for 1 / 2 / 4 on
        // configs[1]): a finished block frees its slot at once instead of waiting for the slowest of four
        const uint32_t wpw = (tune_wpw == 2 || tune_wpw == 4) ? tune_wpw : 1;
        if (tune_tag > 0) ZLZ4_LAUNCH_FAST(k_compress_fast, uint16_t, 1, wpw, wpw * (4096 * sizeof(uint16_t) + 4096) + lds_pad);
        else ZLZ4_LAUNCH_FAST(k_compress_fast, uint16_t, 0, wpw, wpw * 4096 * sizeof(uint16_t) + lds_pad);
    } else {
        const uint32_t wpw = (tune_wpw == 2) ? 2 : 1;   // 16 KiB of LDS per wavefront -> 10 wavefronts per CU
        // positions below 2^24 leave the top byte of a u32 entry to the tag
        if (tune_tag != 0 && max_in_len <= (1u << 24)) ZLZ4_LAUNCH_FAST(k_compress_fast, uint32_t, 2, wpw, wpw * 4096 * sizeof(uint32_t) + lds_pad);
        else ZLZ4_LAUNCH_FAST(k_compress_fast, uint32_t, 0, wpw, wpw * 4096 * sizeof(uint32_t) + lds_pad);
    }
#undef ZLZ4_LAUNCH_FAST
    return hipGetLastError() == hipSuccess ? 0 : -7;
}

#ifdef ZLZ4_STAMPS
extern "C" int zlz4_debug_read_stamps(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_zlz4_stamps), 24 * sizeof(unsigned long long)) != hipSuccess) return -7;
    if (reset) {
        unsigned long long z[24] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_zlz4_stamps), z, sizeof z) != hipSuccess) return -7;
    }
    return 0;
}
#endif
