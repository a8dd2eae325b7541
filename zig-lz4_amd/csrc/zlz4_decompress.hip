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
#include <type_traits>

#include "zlz4_device.hpp"

// Diagnostic build only (-DZLZ4_STAMPS): per-phase shader-cycle sums of the wave decoder (tools/stamp_decode.py)
#ifdef ZLZ4_STAMPS
__device__ unsigned long long g_zlz4_dstamps[16];
#define DSTAMP_DECL unsigned long long st_acc[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}; unsigned long long st_last = __builtin_amdgcn_s_memtime();
#define DSTAMP(i) do { unsigned long long st_now; __builtin_amdgcn_sched_barrier(0); \
                      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_now) :: "memory"); \
                      __builtin_amdgcn_sched_barrier(0); st_acc[i] += st_now - st_last; st_last = st_now; } while (0)
#define DSTAMP_ADD(i, n) do { st_acc[i] += (n); } while (0)
#define DSTAMP_FLUSH do { if (lane == 0) for (int st_k = 0; st_k < 16; st_k++) atomicAdd(&g_zlz4_dstamps[st_k], st_acc[st_k]); } while (0)
#else
#define DSTAMP_DECL
#define DSTAMP(i)
#define DSTAMP_ADD(i, n)
#define DSTAMP_FLUSH
#endif

namespace zlz4 {

__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
__device__ __forceinline__ uint64_t ld64u(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

// inclusive prefix sum over the 64 lanes (DPP: Hillis-Steele inside each row of 16, then row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);    // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);    // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);    // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);    // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return x;
}

// kWrite == false: size pass (no stores) -- used by the frame decoder to learn every block's
// decompressed size before placing the blocks (lz4f.decompressFrame accumulates dstPos serially).
// kLaneCopy: short matches are moved four bytes per lane (batches that fill the chip); false = 16 bytes per sequence lane
// for every match (few blocks: the shorter latency chain)
// (the decoder is bound by vector issue and lives on occupancy: eight wavefronts per SIMD = at most 64 VGPRs.  The copy
//  phases brought the lane-copy build to 66 and the seventh-of-eight cost D-text 7 %; it is back at 62 since the lane
//  copy keeps two rounds of registers instead of three.)
template <bool kWrite, bool kLaneCopy = false, bool kPhases = kLaneCopy>
__global__ __launch_bounds__(256) void k_decompress_safe(
    const uint8_t *__restrict__ d_in, const uint64_t *__restrict__ d_in_off,
    const uint32_t *__restrict__ d_in_len, uint8_t *d_out, const uint64_t *__restrict__ d_out_off,
    const uint32_t *__restrict__ d_out_cap, int64_t *__restrict__ d_result, uint32_t nblocks, uint32_t min_phase_tokens) {
    constexpr uint32_t short_max = kLaneCopy ? 32u : 0u;
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
        DSTAMP_DECL
        // Match copies of a batch are split: the loads are issued with the batch, the stores are deferred until the
        // next batch has been parsed (or until any other code touches dst), so the load round trip overlaps the
        // parse and the token walk.  Per lane: pml = match length (0 = nothing pending), po = output offset,
        // pof = match offset, pa / pb = first and last piece (16, 8 or 4 bytes each).
        uint32_t pml = 0, po = 0, pof = 0;
        u32x4 pa = {0, 0, 0, 0}, pb = {0, 0, 0, 0};
        // Matches of <= 32 bytes (nearly all of a batch) are moved four bytes per lane, eight lanes per sequence: ONE dword
        // load and ONE dword store serve eight sequences, where a 16-byte move per sequence lane costs the CU's address
        // path four times the cycles per instruction for an eighth of the lanes.  pd = the lane's piece, pdo = where it
        // goes (kNone = nothing pending); two rounds of eight sequences per phase (a phase is cut after its 16th short match).
        constexpr uint32_t kNone = 0xFFFFFFFFu;
        uint32_t pd0 = 0, pd1 = 0, pdo0 = kNone, pdo1 = kNone;
        auto flush_pending = [&]() {
            if (!kWrite) return;
            if constexpr (kLaneCopy) {
                if (pdo0 != kNone) __builtin_memcpy(dst + pdo0, &pd0, 4);
                if (pdo1 != kNone) __builtin_memcpy(dst + pdo1, &pd1, 4);
                pdo0 = pdo1 = kNone;
            }
            if (pml != 0) {
                // (every address is dst + a 32-bit offset: the uniform base stays in scalar registers and no lane
                //  spends vector instructions on 64-bit pointer arithmetic)
                if (pml >= 16u) {
                    st128(dst + po, pa);
                    for (uint32_t k = 16u; k + 16u <= pml; k += 16u) st128(dst + (po + k), ld128(dst + (po - pof + k)));
                    st128(dst + (po + pml - 16u), pb);
                } else if (pml >= 8u) {
                    // exact-size copies without partial-word stores: the last piece overlaps the first
                    const uint32_t sh = pml - 8u;                   // 0..7: bytes [sh, sh + 8) of pa
                    const bool hi4 = sh >= 4u;
                    const uint32_t x = hi4 ? pa.y : pa.x, y = hi4 ? pa.z : pa.y, z = hi4 ? pa.w : pa.z;
                    const u32x2 a = {pa.x, pa.y};
                    const u32x2 b = {__builtin_amdgcn_alignbyte(y, x, sh & 3u), __builtin_amdgcn_alignbyte(z, y, sh & 3u)};
                    __builtin_memcpy(dst + po, &a, 8); __builtin_memcpy(dst + (po + sh), &b, 8);
                } else {
                    const uint32_t sh = pml - 4u;                   // 0..3
                    const uint32_t a = pa.x, b = __builtin_amdgcn_alignbyte(pa.y, pa.x, sh);
                    __builtin_memcpy(dst + po, &a, 4); __builtin_memcpy(dst + (po + sh), &b, 4);
                }
            }
            pml = 0;
        };
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
            // ---- batch path: every lane parses the 64-byte window as if a token started at its byte; a scalar walk
            //      over next(i) picks the real token starts, and all sequences that lie entirely inside the window
            //      (typically 6-9 on text) are copied together: one store writes every literal of the batch straight
            //      from the window, then each sequence's lane moves its own match (<= 31 bytes, exact-size stores).
            //      A sequence is only taken when it passes every check of :136-:186 trivially and its match source
            //      ends before the first match of the batch; anything else (255-extension chains, long runs,
            //      overlapping or very recent matches, the tail of the stream, malformed input) ends the batch in
            //      front of it and goes through the single-sequence paths below, which hold the reference's
            //      error order. ----
            DSTAMP(6);
            if ((uint64_t)ip + 68u <= iend) {
              // The window streams through three registers: X0 / X1 hold the dwords at xbase + lane and
              // xbase + 64 + lane, X2 (xbase + 128 + lane) is a load in flight.  A batch consumes <= 64 bytes, so one
              // shift per batch keeps ip inside [xbase, xbase + 64) and X2 is awaited a whole batch after its issue.
              // X2 is loaded and awaited by hand (one `s_waitcnt vmcnt(0)` per batch, placed after the parse, where the
              // previous batch's match data is needed anyway); the compiler, tracking it, would drain the queue at
              // the top of every iteration.  X2 must not be touched between wload_async() and that wait.
              auto woff = [&](uint32_t p) { const uint32_t a = p + lane; return a + 4u <= iend ? a : iend - 4u; };
              auto wload_async = [&](uint32_t p) {
                  uint32_t r;
                  asm volatile("global_load_dword %0, %1, %2" : "=v"(r) : "v"(woff(p)), "s"(src) : "memory");
                  return r;
              };
              // The compressed stream is read once, front to back: every window load is a miss all the way to HBM, and
              // with one in-order vmcnt queue a wait for the match data is a wait for that miss too.  So one byte of the
              // stream 1 KiB ahead is touched per batch, as the LAST vector-memory operation of the batch: the batch's
              // one wait is `vmcnt(1)` -- everything but the youngest operation, i.e. everything but the touch, which
              // gets a second batch to come back -- and the window loads behind it find their lines in the cache.
              // The match loads are hand-issued for the same reason (the compiler, tracking them, would drain the
              // queue before their first use); pa / pb / X2 / tdummy must not be touched between issue and wait.
              uint32_t xbase = ip, tdummy = 0;
              uint32_t X0 = ld32(src + woff(xbase)), X1 = ld32(src + woff(xbase + 64u));
              uint32_t X2 = wload_async(xbase + 128u);
              asm volatile("s_waitcnt vmcnt(0)" : "+v"(X2), "+v"(X0), "+v"(X1));
              for (;;) {
                const uint32_t xi = ip - xbase + lane;
                const uint32_t xa = shfl(X0, xi & 63u), xb = shfl(X1, xi & 63u);
                const uint32_t w4 = xi < 64u ? xa : xb;             // lane i: src[ip + i .. ip + i + 3]
                const uint32_t b0 = w4 & 0xFFu, b1 = (w4 >> 8) & 0xFFu;
                uint32_t lit = b0 >> 4, hl = 1u;                    // :120
                bool cx = false;
                if (lit == 15u) { cx = (b1 == 255u); lit += b1; hl = 2u; }      // :123-131, one extension byte
                const uint32_t mpos = lane + hl + lit;              // window index of the offset
                const uint32_t mw = shfl(w4, mpos & 63u);           // offset (2 bytes) + first match-length extension byte
                const uint32_t off = mw & 0xFFFFu;                  // :150
                uint32_t mlc = b0 & 15u, slen = hl + lit + 2u;      // :157
                if (mlc == 15u) { const uint32_t e2 = (mw >> 16) & 0xFFu; cx = cx || (e2 == 255u); mlc += e2; slen += 1u; }   // :160-168
                const uint32_t ml = mlc + kMinMatch;                // :171, 4..273 here
                const uint32_t nxt = lane + slen;
                const bool ok = !cx && mpos <= 63u && nxt <= 64u && off != 0u;  // (:154 offset == 0 -> single path)
                const uint32_t ol = lit + ml;                       // output bytes of the sequence (<= 335)
                const uint32_t pkv = ok ? (nxt | (ol << 7)) : 0xFFFFFFFFu;      // sentinel: stops the walk
                DSTAMP(0);
                // :137, :174 -- and 32 bytes of slack for the 16-byte match loads
                const uint32_t orem = oend - op;
                const uint32_t room0 = kWrite ? (orem >= 32u ? orem - 32u : 0u) : orem;
                const uint32_t room = rfl(room0 < 4095u ? room0 : 4095u);             // (relv must fit 16 bits of the literal lookup word)
                // Scalar walk over the token chain: R = mask of real token starts, T = output bytes, pos = window index
                // of the first token not taken.  Hand-scheduled (two hops per trip, the lane select of each
                // v_readlane is written >= 8 instructions before it is used), 10 scalar instructions per hop:
                //   do { pk = pkv[pos]; if (T + (pk >> 7) > room) break; R |= 1 << pos; T += pk >> 7; pos = pk & 127; } while (pos < 64);
                uint32_t pos, T, wa, wpk, wt;
                uint64_t R;
                // Away from the end of the output (a batch never exceeds 21 x 335 bytes) the room test is not needed, and
                // with "is the next token ok" folded into the word read for the current one a hop takes 7 instructions:
                //   do { pk = pk2v[pos]; R |= 1 << pos; T += pk >> 8; next_ok = pk & 0x80; pos = pk & 127; } while (next_ok);
                const uint64_t okm = ballot(ok);
                const uint32_t pk2v = nxt | ((nxt < 64u && ((okm >> (nxt & 63u)) & 1ull)) ? 0x80u : 0u) | (ol << 8);
                if (room0 >= 7100u && (okm & 1ull)) {
                    asm volatile(
                        "s_mov_b64 %[R], 0\n\t"
                        "s_mov_b32 %[T], 0\n\t"
                        "s_mov_b32 %[A], 0\n\t"
                        "s_nop 3\n"
                        "1:\n\t"
                        "v_readlane_b32 %[pk], %[pkv], %[A]\n\t"
                        "s_and_b32 %[B], %[pk], 0x7f\n\t"
                        "s_bitset1_b64 %[R], %[A]\n\t"
                        "s_lshr_b32 %[t], %[pk], 8\n\t"
                        "s_add_u32 %[T], %[T], %[t]\n\t"
                        "s_bitcmp1_b32 %[pk], 7\n\t"
                        "s_cbranch_scc0 4f\n\t"
                        "v_readlane_b32 %[pk], %[pkv], %[B]\n\t"
                        "s_and_b32 %[A], %[pk], 0x7f\n\t"
                        "s_bitset1_b64 %[R], %[B]\n\t"
                        "s_lshr_b32 %[t], %[pk], 8\n\t"
                        "s_add_u32 %[T], %[T], %[t]\n\t"
                        "s_bitcmp1_b32 %[pk], 7\n\t"
                        "s_cbranch_scc1 1b\n\t"
                        "s_mov_b32 %[B], %[A]\n"
                        "4:\n"
                        : [R] "=&s"(R), [T] "=&s"(T), [A] "=&s"(wa), [B] "=&s"(pos), [pk] "=&s"(wpk), [t] "=&s"(wt)
                        : [pkv] "v"(pk2v)
                        : "scc");
                } else
                asm volatile(
                    "s_mov_b64 %[R], 0\n\t"
                    "s_mov_b32 %[T], 0\n\t"
                    "s_mov_b32 %[A], 0\n\t"
                    "s_nop 3\n"
                    "1:\n\t"
                    "v_readlane_b32 %[pk], %[pkv], %[A]\n\t"
                    "s_and_b32 %[B], %[pk], 0x7f\n\t"
                    "s_lshr_b32 %[t], %[pk], 7\n\t"
                    "s_add_u32 %[t], %[t], %[T]\n\t"
                    "s_cmp_gt_u32 %[t], %[room]\n\t"
                    "s_cbranch_scc1 3f\n\t"
                    "s_bitset1_b64 %[R], %[A]\n\t"
                    "s_mov_b32 %[T], %[t]\n\t"
                    "s_cmp_gt_u32 %[B], 63\n\t"
                    "s_cbranch_scc1 4f\n\t"
                    "v_readlane_b32 %[pk], %[pkv], %[B]\n\t"
                    "s_and_b32 %[A], %[pk], 0x7f\n\t"
                    "s_lshr_b32 %[t], %[pk], 7\n\t"
                    "s_add_u32 %[t], %[t], %[T]\n\t"
                    "s_cmp_gt_u32 %[t], %[room]\n\t"
                    "s_cbranch_scc1 4f\n\t"
                    "s_bitset1_b64 %[R], %[B]\n\t"
                    "s_mov_b32 %[T], %[t]\n\t"
                    "s_cmp_lt_u32 %[A], 64\n\t"
                    "s_cbranch_scc1 1b\n"
                    "3:\n\t"
                    "s_mov_b32 %[B], %[A]\n"
                    "4:\n"
                    : [R] "=&s"(R), [T] "=&s"(T), [A] "=&s"(wa), [B] "=&s"(pos), [pk] "=&s"(wpk), [t] "=&s"(wt)
                    : [pkv] "v"(pkv), [room] "s"(room)
                    : "scc");
                DSTAMP(1);
                uint32_t relv = 0;
                const uint64_t R_walk = R;                          // every token the walk took
                const uint32_t T_walk = T, pos_walk = pos;
                bool viol_err = false;                              // :181-186 / :231 offset > op: never copied here
                const uint32_t op0 = op;                            // output position of the batch
                if (R != 0) {
                    const bool real0 = (R >> lane) & 1ull;
                    const uint32_t x = real0 ? ol : 0u;
                    relv = wave_incl_scan(x) - x;                   // output offset of the sequence inside the batch
                    // :181-186 / :231 offset > op, and (copy pass) a match source that reaches into this batch's
                    // matches: end the phase in front of the first such sequence
                    const uint32_t lit0 = rdlane(lit, 0);
                    viol_err = off > op0 + relv + lit;
                    bool viol = viol_err;
                    if (kWrite) viol = viol || (off + lit0 < relv + ol);
                    const uint64_t vm = ballot(real0 && viol);
                    if (vm != 0) {
                        const uint32_t fb = first_lane(vm);
                        R &= (1ull << fb) - 1ull;
                        T = rdlane(relv, fb);
                        pos = fb;
                    }
                }
                DSTAMP(2);
                // The one wait of a batch: the window load issued at the top of this iteration and the match loads of
                // the previous batch have had the whole parse / walk / scan to arrive.
                asm volatile("s_waitcnt vmcnt(1)" : "+v"(X2), "+v"(pa), "+v"(pb), "+v"(pd0), "+v"(pd1), "+v"(tdummy));
                if (R == 0) break;                                  // the single-sequence paths take this one
                DSTAMP_ADD(8, 1); DSTAMP_ADD(9, __builtin_popcountll(R));
                // A batch is copied in PHASES.  The match loads of a phase are issued with it and stored when the next
                // one begins, so a match whose source reaches into the output of an earlier match of the same phase ends
                // the phase in front of it -- but not the batch: everything the parse side found out about the window
                // (160 of a batch's 200 vector instructions) still holds, so the rest of the walk's tokens simply form the
                // next phase, which costs a wait for the previous phase's loads, their stores and the copy side again.
                // (Repetitive text -- a word that comes back within ~100 bytes -- had 3.9 sequences per batch against 8.)
                // the lane-copy registers hold two rounds of eight short matches: a phase with more is cut after the 16th
                // (what is left is the next phase, or the next batch)
                auto cap_short = [&]() {
                    if constexpr (kWrite && kLaneCopy) {
                        if ((uint32_t)__popcll(R) <= 16u) return;    // (scalar: the usual batch has 8 tokens)
                        const bool sh = ((R >> lane) & 1ull) && ml <= short_max;
                        const uint64_t S = ballot(sh);
                        if ((uint32_t)__popcll(S) > 16u) {
                            const uint32_t rk = (uint32_t)__popcll(S & ((1ull << lane) - 1ull));
                            const uint32_t fb = first_lane(ballot(sh && rk == 16u));
                            R &= (1ull << fb) - 1ull;
                            T = rdlane(relv, fb);
                            pos = fb;
                        }
                    }
                };
                auto copy_side = [&](auto later_phase) {
                if (kWrite) {
                    flush_pending();                                // the previous phase's / batch's match stores
                    DSTAMP(7);
                    // literals (:140): window byte x belongs to the last token at or before x
                    const uint64_t below = R & (~0ull >> (63u - lane));
                    const uint32_t kl = 63u - (uint32_t)__builtin_clzll(below | 1ull);
                    const uint32_t q = shfl(relv | (lit << 16), kl);
                    const uint32_t qlit = q >> 16;
                    const uint32_t d = lane - kl - (qlit >= 15u ? 2u : 1u);
                    // (in a later phase the lanes in front of its first token belong to nobody)
                    if (d < qlit && (!decltype(later_phase)::value || below != 0)) dst[op0 + (q & 0xFFFFu) + d] = (uint8_t)b0;
                    DSTAMP(3);
                    // matches (:244): source entirely older than this phase's first match, no overlap.  One or two
                    // 16-byte loads per sequence lane; the data is stored by flush_pending() after the next batch
                    // has been parsed (or when the next phase begins).
                    const bool real = (R >> lane) & 1ull;
                    const uint32_t po_t = op0 + relv + lit, mo_t = po_t - off;      // (off <= 65535, ml <= 273)
                    // short_max = 32 when the chip is full (the address path is what bounds the kernel then), 0 with few
                    // blocks (one wavefront per SIMD: the permutes are four more dependent steps of a latency chain, and
                    // 16 bytes per sequence lane is the shorter chain: 38.3 against 43.4 ms on 1024 x 4 MiB)
                    [[maybe_unused]] const bool shortm = real && ml <= short_max;
                    if (!kLaneCopy || ballot(real && ml > short_max)) {   // long matches: 16-byte pieces per sequence lane
                        if (real && ml > short_max) {
                            asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(pa) : "v"(mo_t), "s"(dst) : "memory");
                            if (ml >= 16u)
                                asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(pb) : "v"(mo_t + ml - 16u), "s"(dst) : "memory");
                            pml = ml; po = po_t; pof = off;
                        }
                    }
                    if constexpr (kLaneCopy) {
                        const uint64_t S = ballot(shortm);
                        const uint32_t ns = (uint32_t)__popcll(S);                       // <= 21 sequences per batch
                        const uint32_t rk = (uint32_t)__popcll(S & ((1ull << lane) - 1ull));
                        const uint32_t lead = (lane & ~7u) << 2, k4 = (lane & 7u) * 4u, grp = lane >> 3;
                        // one round: the sequences of rank g0 .. g0 + 7 send (source, destination, length) to the first lane
                        // of their group of eight (ds_permute; everybody else sends to lane 63, which leads no group), the
                        // group reads them back, lane k of the group takes the piece at min(4k, ml - 4)
                        auto round = [&](uint32_t g0, uint32_t &pd, uint32_t &pdo) {
                            const uint32_t rel = rk - g0;
                            const uint32_t dest = (shortm && rel < 8u ? rel * 8u : 63u) << 2;
                            const uint32_t s_po = (uint32_t)__builtin_amdgcn_ds_permute((int)dest, (int)po_t);
                            const uint32_t s_om = (uint32_t)__builtin_amdgcn_ds_permute((int)dest, (int)(off | (ml << 16)));
                            const uint32_t g_po = (uint32_t)__builtin_amdgcn_ds_bpermute((int)lead, (int)s_po);
                            const uint32_t g_om = (uint32_t)__builtin_amdgcn_ds_bpermute((int)lead, (int)s_om);
                            const uint32_t g_ml = g_om >> 16, g_mo = g_po - (g_om & 0xFFFFu);
                            const bool act = g0 + grp < ns && k4 < g_ml;
                            const uint32_t pk = k4 + 4u <= g_ml ? k4 : g_ml - 4u;       // the last piece overlaps the one before
                            if (act) {
                                asm volatile("global_load_dword %0, %1, %2" : "+v"(pd) : "v"(g_mo + pk), "s"(dst) : "memory");
                                pdo = g_po + pk;
                            }
                        };
                        round(0u, pd0, pdo0);
                        if (ns > 8u) round(8u, pd1, pdo1);
                    }
                    DSTAMP(4);
                }
                };
                cap_short();
                copy_side(std::false_type{});
                // ---- further phases: only the copy pass has them, only if the phase before was cut short, only while
                //      enough tokens are left to pay for a phase (it costs about half a batch; a batch that lost its last
                //      token or two is better off starting the next batch there), and only while the token that cut it can
                //      itself be copied (no :181 error, no overlap with its own output) ----
                if constexpr (kWrite && kPhases) {
                    for (uint32_t phase = 1; phase < 6u && pos != pos_walk; phase++) {
                        const uint32_t first = pos, tbase = T;      // first token (lane) / output offset of the new phase
                        const uint64_t Rn = R_walk & ~((1ull << first) - 1ull);
                        if ((uint32_t)__builtin_popcountll(Rn) < min_phase_tokens) break;
                        const bool realn = (Rn >> lane) & 1ull;
                        const uint32_t lit_f = rdlane(lit, first);
                        const bool violn = viol_err || (off + lit_f < (relv - tbase) + ol);
                        const uint64_t vmn = ballot(realn && violn);
                        if ((vmn >> first) & 1ull) break;           // single-sequence path for that token
                        if (vmn != 0) {
                            const uint32_t fb = first_lane(vmn);
                            R = Rn & ((1ull << fb) - 1ull);
                            T = rdlane(relv, fb);
                            pos = fb;
                        } else {
                            R = Rn; T = T_walk; pos = pos_walk;
                        }
                        DSTAMP_ADD(12, 1); DSTAMP_ADD(9, __builtin_popcountll(R));
                        // the loads of the phase before must have landed before flush_pending() stores them (right in front
                        // of the copy side: every way into it passes this wait -- tools/check_decoder_asm.py)
                        cap_short();
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pa), "+v"(pb), "+v"(pd0), "+v"(pd1), "+v"(tdummy), "+v"(X2));
                        copy_side(std::true_type{});
                    }
                }
                op += T;
                ip += pos;
                if ((uint64_t)ip + 68u > iend) break;
                {   // branch-free shift, one window load per batch, then the touch
                    const bool adv = ip - xbase >= 64u;
                    X0 = adv ? X1 : X0; X1 = adv ? X2 : X1; xbase += adv ? 64u : 0u;
                    X2 = wload_async(xbase + 128u);
                    const uint32_t ta = xbase + 128u + 1024u;
                    const uint32_t toff = ta < iend ? ta : iend - 1u;
                    asm volatile("global_load_ubyte %0, %1, %2" : "+v"(tdummy) : "v"(toff), "s"(src) : "memory");
                }
              }
              asm volatile("s_waitcnt vmcnt(0)" : "+v"(X2), "+v"(pa), "+v"(pb), "+v"(pd0), "+v"(pd1), "+v"(tdummy));
              if (ip >= iend) { flush_pending(); break; }
            }
            flush_pending();
            DSTAMP_ADD(11, 1);
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
                // overlap with a period of 64..1023 bytes (:235-241 copies it byte by byte): the output is the period
                // repeated, so what has been produced so far can be copied again as a whole -- offset bytes, then 2 x, 4 x ...
                // -- every copy disjoint from its source (round 3; one byte per lane and step took 1024 dependent steps for a
                // 64 KiB match: D-ramp, period 256)
                uint32_t made = 0;                                  // o[0, made) is done; m[0, made + offset) holds the pattern
                while (made < ml) {
                    const uint32_t have = made + offset, left = ml - made;
                    const uint32_t n1 = have < left ? have : left;
                    copy_bytes(o + made, m, n1, lane);
                    made += n1;
                }
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
        flush_pending();
        if (res == 0) res = (int64_t)op;                            // :250
        DSTAMP(5);
        DSTAMP_FLUSH;
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

constexpr uint32_t kLaneCopyMinBlocks = 6144;   // (crossover measured between 4096 and 8192 blocks; the chip holds 8192 wavefronts)
extern "C" int zlz4_launch_decompress_safe(hipStream_t stream, const uint8_t *d_in, const uint64_t *d_in_off,
                                           const uint32_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off,
                                           const uint32_t *d_out_cap, int64_t *d_result, uint32_t nblocks) {
    if (nblocks == 0) return 0;
    // one wavefront per block (batch path) is the faster decoder at every batch size measured on MI355X; the
    // one-lane-per-block kernel stays available behind ZLZ4_DECOMP_LANE_MIN (tests/test_gpu_lane_decoder.py)
    static const uint32_t lane_min = [] { const char *e = zlz4_tune_env("ZLZ4_DECOMP_LANE_MIN"); return e ? (uint32_t)atoll(e) : 0xFFFFFFFFu; }();
    if (nblocks >= lane_min) {
        // the kernel is latency-bound: with few blocks use fewer lanes per wavefront so that ~8192 wavefronts exist
        static const uint32_t lanes_env = [] { const char *e = zlz4_tune_env("ZLZ4_DECOMP_LANES"); return e ? (uint32_t)atoi(e) : 0u; }();
        // measured on MI355X (65 536 blocks): 16 or 32 active lanes per wavefront are ~5 % faster than 64
        uint32_t lanes = nblocks >= 524288u ? 64u : (nblocks >= 262144u ? 32u : 16u);
        if (lanes_env >= 1 && lanes_env <= 64) lanes = lanes_env;
        static const uint32_t max_lanes = [] { const char *e = zlz4_tune_env("ZLZ4_DECOMP_MAXLANES"); return e ? (uint32_t)atoll(e) : 0u; }();
        uint32_t grid = (nblocks + lanes - 1u) / lanes;
        if (max_lanes && (uint64_t)grid * lanes > max_lanes) grid = (max_lanes + lanes - 1u) / lanes;
        hipLaunchKernelGGL(zlz4::k_decompress_lane, dim3(grid), dim3(lanes), 0, stream, d_in, d_in_off,
                           d_in_len, d_out, d_out_off, d_out_cap, d_result, nblocks);
        return hipGetLastError() == hipSuccess ? 0 : -7;
    }
    static const uint32_t waves_per_wg = [] { const char *e = zlz4_tune_env("ZLZ4_DECOMP_WPW"); const uint32_t v = e ? (uint32_t)atoi(e) : 4u;
                                              return (v == 1u || v == 2u || v == 4u) ? v : 4u; }();
    const uint32_t grid = (nblocks + waves_per_wg - 1) / waves_per_wg;
    // experiment knob: dynamic LDS per workgroup only to limit the number of resident wavefronts per CU
    static const uint32_t dyn_lds = [] { const char *e = zlz4_tune_env("ZLZ4_DECOMP_LDS"); return e ? (uint32_t)atoll(e) : 0u; }();
    // short matches four bytes per lane once the batch fills the chip (see the kernel); ZLZ4_DECOMP_SHORT forces 0 / 32
    static const int short_env = [] { const char *e = zlz4_tune_env("ZLZ4_DECOMP_SHORT"); return e ? atoi(e) : -1; }();
    const uint32_t short_max = short_env >= 0 ? (uint32_t)short_env : (nblocks >= kLaneCopyMinBlocks ? 32u : 0u);
    // A/B switch for profiles/ (tuning build): the lane-copy decoder without copy phases
    static const bool no_phases = [] { const char *e = zlz4_tune_env("ZLZ4_DECOMP_PHASES"); return e && atoi(e) == 0; }();
    // tokens that must be left for another copy phase to be worth its wait (ZLZ4_DECOMP_PHASE_MIN in the tuning build)
    static const uint32_t min_phase_tokens = [] { const char *e = zlz4_tune_env("ZLZ4_DECOMP_PHASE_MIN"); return e ? (uint32_t)atoi(e) : 3u; }();
    if (short_max && no_phases)
        hipLaunchKernelGGL((zlz4::k_decompress_safe<true, true, false>), dim3(grid), dim3(64 * waves_per_wg), dyn_lds, stream, d_in,
                           d_in_off, d_in_len, d_out, d_out_off, d_out_cap, d_result, nblocks, min_phase_tokens);
    else if (short_max)
        hipLaunchKernelGGL((zlz4::k_decompress_safe<true, true>), dim3(grid), dim3(64 * waves_per_wg), dyn_lds, stream, d_in,
                           d_in_off, d_in_len, d_out, d_out_off, d_out_cap, d_result, nblocks, min_phase_tokens);
    else
        hipLaunchKernelGGL((zlz4::k_decompress_safe<true, false>), dim3(grid), dim3(64 * waves_per_wg), dyn_lds, stream, d_in,
                           d_in_off, d_in_len, d_out, d_out_off, d_out_cap, d_result, nblocks, min_phase_tokens);
    return hipGetLastError() == hipSuccess ? 0 : -7;
}

// size pass: d_out may be null; d_out_off / d_out_cap still give (dummy offset 0, capacity) per block
extern "C" int zlz4_launch_decompress_sizes(hipStream_t stream, const uint8_t *d_in, const uint64_t *d_in_off,
                                            const uint32_t *d_in_len, const uint64_t *d_out_off,
                                            const uint32_t *d_out_cap, int64_t *d_result, uint32_t nblocks) {
    if (nblocks == 0) return 0;
    const uint32_t waves_per_wg = 4;
    const uint32_t grid = (nblocks + waves_per_wg - 1) / waves_per_wg;
    hipLaunchKernelGGL((zlz4::k_decompress_safe<false, false>), dim3(grid), dim3(64 * waves_per_wg), 0, stream, d_in, d_in_off,
                       d_in_len, (uint8_t *)nullptr, d_out_off, d_out_cap, d_result, nblocks, 0u);
    return hipGetLastError() == hipSuccess ? 0 : -7;
}

#ifdef ZLZ4_STAMPS
extern "C" int zlz4_debug_read_dstamps(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_zlz4_dstamps), 16 * sizeof(unsigned long long)) != hipSuccess) return -7;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_zlz4_dstamps), z, sizeof z) != hipSuccess) return -7;
    }
    return 0;
}
#endif
