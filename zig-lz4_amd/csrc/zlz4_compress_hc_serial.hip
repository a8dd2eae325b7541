// zlz4_compress_hc_serial.hip -- the two remaining compressHC strategies (reference src/lz4hc.zig:72-86):
//   level 2      lz4mid  compressMID      (:687-971)   dual 4-byte / 8-byte hash tables, strictly serial
//   level 10-12  lz4opt  compressOptimal  (:1068-1391) price-based parse over the hash-chain matches
//
// Both parses are serial inside a block.
//   level 2:      `compressMID`'s tables depend on the parse (it inserts around every match it takes), a probe is two table
//                 lookups and one 8-byte compare, and 128 KiB of tables do not fit beside a second block in LDS -- so the
//                 batch dimension is the parallel one: ONE LANE PER BLOCK (two blocks per wavefront) runs the serial loop
//                 with its tables in an HBM workspace (k_hc_mid_serial; DESIGN.md 4.3b says what a wavefront per block
//                 would cost).
//   levels 10-12: the expensive part -- the match search (insertAndGetWiderMatch, :538-681) -- does not run here: as for
//                 levels 3-9 the table state a search sees is a pure function of the input, so K1 / K2 of
//                 zlz4_compress_hc.hip produce the best match of every position with full parallelism and the parse only
//                 looks results up.  The parse itself: ONE WAVEFRONT PER BLOCK for blocks <= 64 KiB
//                 (k_hc_opt_parse_wave: price records in LDS, the lanes take the match lengths of a position), one lane
//                 per block with the records in HBM for larger blocks (k_hc_opt_parse).
//
// NOTE (reference behaviour, reproduced on purpose): the "match is good enough -> encode immediately" branch
// of compressOptimal (:1207-1256) walks opt[] forward WITHOUT the reverse traversal of the normal path
// (:1315-1332), i.e. it reads arrival records as if they were forward steps.  With targetLength 64 (level 10)
// this is reached often and the emitted sequences are not always real matches: the reference's level-10
// stream does not always decode.  Parity here means the same bytes as the reference, decodable or not.
#include "zlz4_device.hpp"

namespace zlz4 {

constexpr uint32_t kMidHashLog = 14;                       // src/lz4hc.zig:45
constexpr uint32_t kMidTableSize = 1u << kMidHashLog;      // :46
constexpr uint64_t kHashMul64 = 58295818150454627ull;      // :51
constexpr uint32_t kOptNum = 1u << 12;                     // :42
constexpr uint32_t kTrailingLiterals = 3;                  // :1075
constexpr uint32_t kOptEntries = kOptNum + kTrailingLiterals;

__device__ __forceinline__ uint64_t ld64(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ uint32_t hash_mid4(const uint8_t *p) { return (ld32(p) * kHashMul) >> (32 - kMidHashLog); }   // :139-146
__device__ __forceinline__ uint32_t hash_mid8(const uint8_t *p) {                                                       // :149-157
    return (uint32_t)(((ld64(p) << 8) * kHashMul64) >> (64 - kMidHashLog));
}

// lz4Count (:234-264) from absolute positions a (< limit) and b (< a); one lane
__device__ __forceinline__ uint32_t count_from(const uint8_t *src, uint32_t a, uint32_t b, uint32_t limit) {
    uint32_t c = 0;
    while (a + 8u <= limit) {
        const uint64_t x = ld64(src + a) ^ ld64(src + b);
        if (x) return c + ((uint32_t)__builtin_ctzll(x) >> 3);
        a += 8; b += 8; c += 8;
    }
    while (a < limit && src[a] == src[b]) { a++; b++; c++; }
    return c;
}

// per-lane byte copy, 8 bytes at a time (ranges never overlap here: source is the input block)
__device__ __forceinline__ void lane_copy(uint8_t *d, const uint8_t *s, uint32_t n) {
    uint32_t k = 0;
    for (; k + 8u <= n; k += 8u) { const uint64_t v = ld64(s + k); __builtin_memcpy(d + k, &v, 8); }
    for (; k < n; k++) d[k] = s[k];
}

// encodeSequence (:308-386) with limitedOutput, one lane.  Returns false when the output does not fit.
__device__ __forceinline__ bool encode_sequence_lane(const uint8_t *src, uint8_t *dst, uint32_t oend, uint32_t &ip,
                                                     uint32_t &op, uint32_t &anchor, uint32_t match_len, uint32_t offset) {
    const uint32_t lit = ip - anchor;                                             // :317
    if ((uint64_t)op + lit / 255u + lit + (2u + 1u + kLastLiterals) > oend) return false;       // :320-325
    const uint32_t tok = op++;
    if (lit >= 15u) {                                                             // :331-343
        uint32_t len = lit - 15u;
        dst[tok] = 0xF0;
        while (len >= 255u) { dst[op++] = 255; len -= 255u; }
        dst[op++] = (uint8_t)len;
    } else {
        dst[tok] = (uint8_t)(lit << 4);
    }
    lane_copy(dst + op, src + anchor, lit);                                       // :346
    op += lit;
    dst[op] = (uint8_t)offset; dst[op + 1u] = (uint8_t)(offset >> 8);             // :350
    op += 2;
    const uint32_t ml_code = match_len - kMinMatch;                               // :354
    if ((uint64_t)op + ml_code / 255u + (1u + kLastLiterals) > oend) return false;              // :355-359
    if (ml_code >= 15u) {                                                         // :361-376 (510-steps == 255-run)
        dst[tok] += 15;
        uint32_t rem = ml_code - 15u;
        while (rem >= 255u) { dst[op++] = 255; rem -= 255u; }
        dst[op++] = (uint8_t)rem;
    } else {
        dst[tok] += (uint8_t)ml_code;
    }
    ip += match_len;                                                              // :382
    anchor = ip;
    return true;
}

// trailing literal-only sequence of compressMID / compressOptimal (:942-968, :1362-1388), one lane
__device__ __forceinline__ int64_t final_literals_lane(const uint8_t *src, uint8_t *dst, uint32_t oend, uint32_t n,
                                                       uint32_t anchor, uint32_t op) {
    const uint32_t fl = n - anchor;
    if (fl == 0) return (int64_t)op;
    const uint32_t nle = ext_len_bytes(fl);
    // the reference tests only op + fl + 1 and then writes the length-extension bytes unchecked (out of bounds
    // when they do not fit); refuse instead of overrunning, as in k_hc_parse_emit
    if ((uint64_t)op + fl + 1u > oend || (uint64_t)op + 1u + nle + fl > oend) return kErrOutputTooSmall;
    if (fl >= 15u) {
        uint32_t len = fl - 15u;
        dst[op++] = 0xF0;
        while (len >= 255u) { dst[op++] = 255; len -= 255u; }
        dst[op++] = (uint8_t)len;
    } else {
        dst[op++] = (uint8_t)(fl << 4);
    }
    lane_copy(dst + op, src + anchor, fl);
    return (int64_t)(op + fl);
}

// encodeLiterals (:1394-1425) for inputs shorter than 13 bytes, one lane
__device__ __forceinline__ int64_t tiny_block_lane(const uint8_t *src, uint8_t *dst, uint32_t oend, uint32_t n) {
    if (oend < n + 1u + n / 255u) return kErrOutputTooSmall;                      // :1395
    dst[0] = (uint8_t)(n << 4);
    for (uint32_t k = 0; k < n; k++) dst[1u + k] = src[k];
    return (int64_t)n + 1;
}

// ------------------------------------------------------------------ level 2: compressMID, one lane per block
// "fill table with end of match" (:789-818 == :899-928); e = index of the first byte after the match
__device__ __forceinline__ void mid_fill_end(const uint8_t *src, uint32_t *h4, uint32_t *h8, uint32_t e, uint32_t ilimit) {
    if (e - 2u < ilimit) {                                                        // pos_m2 < ilimitIdx
        if (e > 5u && e - 5u <= ilimit) h8[hash_mid8(src + e - 5u)] = e - 5u;
        // (`@intFromPtr(ip) >= 3` etc. compare an ADDRESS with a small constant: always true)
        if (e - 3u <= ilimit) h8[hash_mid8(src + e - 3u)] = e - 3u;
        if (e - 2u <= ilimit) { h8[hash_mid8(src + e - 2u)] = e - 2u; h4[hash_mid4(src + e - 2u)] = e - 2u; }
        if (e - 1u <= ilimit) h4[hash_mid4(src + e - 1u)] = e - 1u;
    }
}

__global__ __launch_bounds__(64) void k_hc_mid_serial(const uint8_t *__restrict__ d_in, const uint64_t *__restrict__ d_in_off,
                                                       const uint32_t *__restrict__ d_in_len, uint8_t *__restrict__ d_out,
                                                       const uint64_t *__restrict__ d_out_off,
                                                       const uint32_t *__restrict__ d_out_cap, int64_t *__restrict__ d_result,
                                                       uint32_t *__restrict__ d_tables, uint32_t blk0, uint32_t nblocks,
                                                       uint32_t max_in_len) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const uint32_t blk = blk0 + b;
    const uint8_t *src = d_in + d_in_off[blk];
    uint8_t *dst = d_out + d_out_off[blk];
    const uint32_t n = d_in_len[blk], oend = d_out_cap[blk];
    int64_t out;
    if (n > kMaxInput) out = kErrInputTooLarge;                                   // :1442
    else if (n > max_in_len) out = kErrInvalidState;                              // the batch contract (include/zlz4_amd.h)
    else if (n == 0) out = 0;                                                     // :1443
    else if (oend == 0) out = kErrOutputTooSmall;                                 // :1461
    else if (n < kMfLimit + 1u) out = tiny_block_lane(src, dst, oend, n);         // :706-708
    else {
        uint32_t *h4 = d_tables + (uint64_t)b * (2u * kMidTableSize);             // :721-722 (zeroed by the launcher, :725-726)
        uint32_t *h8 = h4 + kMidTableSize;
        const uint32_t mflimit = n - kMfLimit, matchlimit = n - kLastLiterals, ilimit = n - 8u;   // :698-700
        uint32_t ip = 0, anchor = 0, op = 0;
        bool failed = false;
        // One lane walks the block and every probe is a chain of dependent HBM accesses (table entry -> candidate bytes,
        // twice).  The chain is shortened without changing a single table state: both entries of the NEXT position are
        // requested while the current one is examined (a probe without a match only writes h8[hh8] and h4[hh4], which
        // the pre-read values are patched with), and both candidates' bytes are requested together.
        bool have_pf = false;              // pf8 / pf4 are the entries of position ip as the serial loop would read them
        uint32_t pf8 = 0, pf4 = 0, pfh8 = 0, pfh4 = 0;                            // (and pfh8 / pfh4 its two hashes)
        while (ip <= mflimit) {                                                   // :733
            const uint32_t ip_index = ip;
            const uint32_t hh8 = have_pf ? pfh8 : hash_mid8(src + ip), hh4 = have_pf ? pfh4 : hash_mid4(src + ip);
            const uint32_t pos8 = have_pf ? pf8 : h8[hh8];
            const uint32_t pos4 = have_pf ? pf4 : h4[hh4];      // (nothing writes h4 before :828 when there is no long match)
            // entries of the position a probe without a match goes to next (:937-938)
            const uint32_t nip = ip + 1u + ((ip - anchor) >> 9);
            const bool nvalid = nip <= mflimit;
            uint32_t nh8 = 0, nh4 = 0, n8 = 0, n4 = 0;
            if (nvalid) { nh8 = hash_mid8(src + nip); nh4 = hash_mid4(src + nip); n8 = h8[nh8]; n4 = h4[nh4]; }
            const bool ok8 = pos8 > 0 && ip_index - pos8 <= kMaxDist && pos8 < ip;   // :744-748
            const bool ok4 = pos4 > 0 && ip_index - pos4 <= kMaxDist && pos4 < ip;   // :832-836
            // first 8 bytes of both candidates, in flight together (matchlimit >= ip + 7: mflimit = n - 12)
            const uint64_t here = ld64(src + ip);
            const uint64_t c8 = ok8 ? ld64(src + pos8) : ~here, c4 = ok4 ? ld64(src + pos4) : ~here;
            bool taken = false;
            {   // long match, 8-byte hash (:739-824)
                h8[hh8] = ip_index;                                               // :742
                if (ok8) {
                    const uint64_t x = here ^ c8;
                    uint32_t mlt = x ? ((uint32_t)__builtin_ctzll(x) >> 3)
                                     : 8u + count_from(src, ip + 8u, pos8 + 8u, matchlimit);   // :749
                    if (mlt > matchlimit - ip) mlt = matchlimit - ip;             // (ip == mflimit: only 7 bytes may count)
                    if (mlt >= kMinMatch) {
                        if (ip + 1u <= ilimit) h8[hash_mid8(src + ip + 1u)] = ip_index + 1u;     // :767-769
                        if (ip + 2u <= ilimit) h8[hash_mid8(src + ip + 2u)] = ip_index + 2u;     // :770-772
                        if (ip + 1u <= ilimit) h4[hash_mid4(src + ip + 1u)] = ip_index + 1u;     // :773-775
                        if (!encode_sequence_lane(src, dst, oend, ip, op, anchor, mlt, ip_index - pos8)) { failed = true; break; }
                        mid_fill_end(src, h4, h8, ip, ilimit);                    // :789-818
                        taken = true;
                    }
                }
            }
            if (taken) { have_pf = false; continue; }                             // :819
            {   // short match, 4-byte hash (:827-934)
                h4[hh4] = ip_index;                                               // :830
                if (ok4) {
                    const uint64_t x = here ^ c4;
                    uint32_t match_len = x ? ((uint32_t)__builtin_ctzll(x) >> 3)
                                           : 8u + count_from(src, ip + 8u, pos4 + 8u, matchlimit);   // :837
                    if (match_len > matchlimit - ip) match_len = matchlimit - ip;
                    if (match_len >= kMinMatch) {
                        uint32_t match_dist = ip_index - pos4;
                        if (ip < mflimit) {                                       // :842
                            const uint32_t h8n = hash_mid8(src + ip + 1u);
                            const uint32_t pos8n = h8[h8n];
                            const uint32_t m2_dist = ip_index + 1u - pos8n;       // :845
                            if (m2_dist <= kMaxDist && pos8n > 0 && pos8n < ip + 1u) {   // :847-849
                                const uint32_t ml2 = count_from(src, ip + 1u, pos8n, matchlimit);
                                if (ml2 > match_len) {                            // :851-856
                                    h8[h8n] = ip_index + 1u;
                                    ip += 1;
                                    match_len = ml2;
                                    match_dist = m2_dist;
                                }
                            }
                        }
                        // :873 finalIpIndex4 = the ORIGINAL index even when ip has moved by one
                        if (ip + 1u <= ilimit) h8[hash_mid8(src + ip + 1u)] = ip_index + 1u;     // :875-877
                        if (ip + 2u <= ilimit) h8[hash_mid8(src + ip + 2u)] = ip_index + 2u;     // :878-880
                        if (ip + 1u <= ilimit) h4[hash_mid4(src + ip + 1u)] = ip_index + 1u;     // :881-883
                        if (!encode_sequence_lane(src, dst, oend, ip, op, anchor, match_len, match_dist)) { failed = true; break; }
                        mid_fill_end(src, h4, h8, ip, ilimit);                    // :899-928
                        taken = true;
                    }
                }
            }
            if (taken) { have_pf = false; continue; }                             // :929
            // no match: this probe wrote h8[hh8] = h4[hh4] = ip_index and nothing else
            pf8 = nh8 == hh8 ? ip_index : n8;
            pf4 = nh4 == hh4 ? ip_index : n4;
            pfh8 = nh8; pfh4 = nh4;
            have_pf = nvalid;
            ip = nip;                                                             // :937-938
        }
        out = failed ? kErrOutputTooSmall : final_literals_lane(src, dst, oend, n, anchor, op);
    }
    d_result[blk] = out;
}

// ------------------------------------------------------------------ levels 10-12: compressOptimal parse, one lane per block
struct OptEntry { int32_t price, off, mlen, litlen; };                            // :456-461

__device__ __forceinline__ int32_t literals_price(int32_t litlen) {               // :466-472
    int32_t price = litlen;
    if (litlen >= 15) price += 1 + (litlen - 15) / 255;
    return price;
}
__device__ __forceinline__ int32_t sequence_price(int32_t litlen, int32_t mlen) { // :476-486
    int32_t price = 1 + 2 + literals_price(litlen);
    if (mlen >= 19) price += 1 + (mlen - 19) / 255;
    return price;
}

template <typename R>   // R = K2's packed result (len | off << 16, or len | off << 32)
__global__ __launch_bounds__(64) void k_hc_opt_parse(const uint8_t *__restrict__ d_in, const uint64_t *__restrict__ d_in_off,
                                                      const uint32_t *__restrict__ d_in_len, uint8_t *__restrict__ d_out,
                                                      const uint64_t *__restrict__ d_out_off,
                                                      const uint32_t *__restrict__ d_out_cap, int64_t *__restrict__ d_result,
                                                      const R *__restrict__ d_res, uint64_t res_stride,
                                                      OptEntry *__restrict__ d_opt, uint32_t blk0, uint32_t nblocks,
                                                      uint32_t sufficient_len_in, uint32_t max_in_len) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const uint32_t blk = blk0 + b;
    const uint8_t *src = d_in + d_in_off[blk];
    uint8_t *dst = d_out + d_out_off[blk];
    const uint32_t n = d_in_len[blk], oend = d_out_cap[blk];
    int64_t out;
    if (n > kMaxInput) out = kErrInputTooLarge;
    else if (n > max_in_len) out = kErrInvalidState;                              // not in the K1/K2 workspace
    else if (n == 0) out = 0;
    else if (oend == 0) out = kErrOutputTooSmall;
    else if (n < kMfLimit + 1u) out = tiny_block_lane(src, dst, oend, n);         // :1092-1094
    else {
        const R *res = d_res + (uint64_t)b * res_stride;
        OptEntry *opt = d_opt + (uint64_t)b * kOptEntries;                        // :1079
        const uint32_t mflimit = n - kMfLimit;
        uint32_t sufficient_len = sufficient_len_in;                              // :1105-1108
        if (sufficient_len >= kOptNum) sufficient_len = kOptNum - 1u;
        auto best_match = [&](uint32_t p, int32_t &len, int32_t &off) {           // insertHC + insertAndGetWiderMatch result at p
            const R r = res[p];
            if (sizeof(R) == 4) { len = (int32_t)((uint32_t)r & 0xFFFFu); off = (int32_t)((uint32_t)r >> 16); }
            else { len = (int32_t)(uint32_t)r; off = (int32_t)((uint64_t)r >> 32); }
        };
        uint32_t ip = 0, anchor = 0, op = 0;
        bool failed = false;
        while (ip <= mflimit && !failed) {                                        // :1111
            const int32_t llen = (int32_t)(ip - anchor);
            int32_t first_len, first_off;
            best_match(ip, first_len, first_off);                                 // :1115-1125
            if (first_len == 0) { ip += 1; continue; }                            // :1127 (never true: the result starts at 3)
            if ((uint32_t)first_len > sufficient_len) {                           // :1133-1147
                if (!encode_sequence_lane(src, dst, oend, ip, op, anchor, (uint32_t)first_len, (uint32_t)first_off)) failed = true;
                continue;
            }
            for (uint32_t r = 0; r < kMinMatch; r++) {                            // :1150-1157
                opt[r].mlen = 1; opt[r].off = 0; opt[r].litlen = llen + (int32_t)r; opt[r].price = literals_price(llen + (int32_t)r);
            }
            const uint32_t match_ml = (uint32_t)first_len;                        // :1160
            for (uint32_t ml = kMinMatch; ml <= match_ml; ml++) {                 // :1162-1169
                opt[ml].mlen = (int32_t)ml; opt[ml].off = first_off; opt[ml].litlen = llen; opt[ml].price = sequence_price(llen, (int32_t)ml);
            }
            uint32_t last_match_pos = match_ml;                                   // :1171
            for (uint32_t al = 1; al <= kTrailingLiterals; al++) {                // :1174-1180
                OptEntry &e = opt[last_match_pos + al];
                e.mlen = 1; e.off = 0; e.litlen = (int32_t)al; e.price = opt[last_match_pos].price + literals_price((int32_t)al);
            }
            bool encoded_early = false;
            for (uint32_t cur = 1; cur < last_match_pos; cur++) {                 // :1183-1312
                const uint32_t cur_pos = ip + cur;
                if (cur_pos > mflimit) break;                                     // :1187
                if (opt[cur + 1u].price <= opt[cur].price) continue;              // :1190
                int32_t new_len, new_off;
                best_match(cur_pos, new_len, new_off);                            // :1193-1203
                if (new_len == 0) continue;                                       // :1205
                if (((uint32_t)new_len > sufficient_len) || (new_len + (int32_t)cur >= (int32_t)kOptNum)) {   // :1208
                    uint32_t rp = 0;                                              // :1216-1238 (forward walk, no reverse traversal)
                    while (rp < cur) {
                        const int32_t ml = opt[rp].mlen, off = opt[rp].off;
                        if (ml == 1) { ip += 1; rp += 1; continue; }
                        rp += (uint32_t)ml;
                        if (!encode_sequence_lane(src, dst, oend, ip, op, anchor, (uint32_t)ml, (uint32_t)off)) { failed = true; break; }
                    }
                    if (!failed && !encode_sequence_lane(src, dst, oend, ip, op, anchor, (uint32_t)new_len, (uint32_t)new_off)) failed = true;   // :1241-1252
                    encoded_early = true;                                         // :1255
                    break;
                }
                const int32_t base_litlen = opt[cur].litlen;                      // :1259
                for (uint32_t ll = 1; ll < kMinMatch; ll++) {                     // :1260-1270
                    const int32_t price = opt[cur].price - literals_price(base_litlen) + literals_price(base_litlen + (int32_t)ll);
                    OptEntry &e = opt[cur + ll];
                    if (price < e.price) { e.mlen = 1; e.off = 0; e.litlen = base_litlen + (int32_t)ll; e.price = price; }
                }
                const uint32_t new_ml = (uint32_t)new_len;                        // :1273
                const OptEntry oc = opt[cur];
                for (uint32_t ml = kMinMatch; ml <= new_ml; ml++) {               // :1274-1302
                    const uint32_t pos = cur + ml;
                    int32_t price, ll;
                    if (oc.mlen == 1) {
                        ll = oc.litlen;
                        price = (cur > (uint32_t)ll) ? opt[cur - (uint32_t)ll].price : 0;
                        price += sequence_price(ll, (int32_t)ml);
                    } else {
                        ll = 0;
                        price = oc.price + sequence_price(0, (int32_t)ml);
                    }
                    if (pos > last_match_pos + kTrailingLiterals || price <= opt[pos].price) {   // :1293
                        if (ml == new_ml && last_match_pos < pos) last_match_pos = pos;
                        OptEntry &e = opt[pos];
                        e.mlen = (int32_t)ml; e.off = new_off; e.litlen = ll; e.price = price;
                    }
                }
                for (uint32_t al = 1; al <= kTrailingLiterals; al++) {            // :1305-1311
                    OptEntry &e = opt[last_match_pos + al];
                    e.mlen = 1; e.off = 0; e.litlen = (int32_t)al; e.price = opt[last_match_pos].price + literals_price((int32_t)al);
                }
            }
            if (encoded_early || failed) continue;
            {   // reverse traversal (:1315-1332)
                int32_t sel_ml = opt[last_match_pos].mlen, sel_off = opt[last_match_pos].off;
                uint32_t cand = last_match_pos - (uint32_t)sel_ml;
                for (;;) {
                    const int32_t next_ml = opt[cand].mlen, next_off = opt[cand].off;
                    opt[cand].mlen = sel_ml; opt[cand].off = sel_off;
                    sel_ml = next_ml; sel_off = next_off;
                    if (next_ml > (int32_t)cand) break;
                    cand -= (uint32_t)next_ml;
                }
            }
            uint32_t rp = 0;                                                      // :1335-1358
            while (rp < last_match_pos) {
                const int32_t ml = opt[rp].mlen, off = opt[rp].off;
                if (ml == 1) { ip += 1; rp += 1; continue; }
                rp += (uint32_t)ml;
                if (!encode_sequence_lane(src, dst, oend, ip, op, anchor, (uint32_t)ml, (uint32_t)off)) { failed = true; break; }
            }
        }
        out = failed ? kErrOutputTooSmall : final_literals_lane(src, dst, oend, n, anchor, op);
    }
    d_result[blk] = out;
}


// ------------------------------------------------------------------ levels 10-12, blocks <= 64 KiB: one WAVEFRONT per block
// The price-based parse (compressOptimal, :1068-1391) is serial over the positions of its window, but what it does at a
// position is not: the literal updates (:1260-1270), the price update over all match lengths (:1274-1302) and the
// trailing literals (:1305-1311) touch up to 4095 independent records.  Here the wavefront walks the positions in
// lock step (every scalar of the reference's loop is wave-uniform) and the lanes take the records: one LDS read and
// one LDS write serve 64 match lengths.  opt[] lives in LDS as 8-byte records -- price 18 bits (<= 65 796 + the window's
// own growth), offset 16, match length 13 (<= 4095), literal length 17 (<= 65 539): exact for blocks <= 64 KiB --
// and only its first kOptLds records: the window of an ordinary block stays far below that, so 8 KiB per wavefront keep
// 20 blocks per CU in flight where the whole array (32 KiB) would allow 4; records beyond go to the HBM array the
// one-lane kernel uses (same values, 16-byte records).  The search results of 64 consecutive positions are held in
// registers.  Same decisions in the same order as k_hc_opt_parse, including the reference's early-encode walk.
constexpr uint32_t kOptLds = 1000;
typedef __attribute__((address_space(3))) volatile uint64_t lds_opt;

__device__ __forceinline__ uint64_t opt_pack(uint32_t price, uint32_t off, uint32_t mlen, uint32_t litlen) {
    return (uint64_t)(price & 0x3FFFFu) | ((uint64_t)(off & 0xFFFFu) << 18) | ((uint64_t)(mlen & 0x1FFFu) << 34) |
           ((uint64_t)(litlen & 0x1FFFFu) << 47);
}
__device__ __forceinline__ uint32_t opt_price(uint64_t e) { return (uint32_t)e & 0x3FFFFu; }
__device__ __forceinline__ uint32_t opt_off(uint64_t e) { return (uint32_t)(e >> 18) & 0xFFFFu; }
__device__ __forceinline__ uint32_t opt_mlen(uint64_t e) { return (uint32_t)(e >> 34) & 0x1FFFu; }
__device__ __forceinline__ uint32_t opt_litlen(uint64_t e) { return (uint32_t)(e >> 47); }

// encodeSequence (:308-386) with limitedOutput by the whole wavefront (as in k_hc_parse_emit); false = does not fit
__device__ __forceinline__ bool encode_sequence_wave(const uint8_t *src, uint8_t *dst, uint32_t oend, uint32_t &ip, uint32_t &op,
                                                     uint32_t &anchor, uint32_t len, uint32_t off, uint32_t lane) {
    const uint32_t lit = ip - anchor;                            // :317 (wraps like the reference's usize when ip < anchor)
    if ((uint64_t)op + lit / 255u + lit + (2u + 1u + kLastLiterals) > oend) return false;       // :320-325
    const uint32_t ml_code = len - kMinMatch;                    // :354
    const uint32_t nle = ext_len_bytes(lit), nme = ext_len_bytes(ml_code);
    const uint32_t op2 = op + 1u + nle + lit + 2u;
    if ((uint64_t)op2 + ml_code / 255u + (1u + kLastLiterals) > oend) return false;             // :355-359
    if (lane == 0) dst[op] = (uint8_t)(((lit >= 15u ? 15u : lit) << 4) | (ml_code >= 15u ? 15u : ml_code));
    if (lit >= 15u) write_ext_len(dst + op + 1u, lit, lane);
    copy_bytes(dst + op + 1u + nle, src + anchor, lit, lane);    // :346
    if (lane < 2u) dst[op2 - 2u + lane] = (uint8_t)(off >> (8u * lane));   // :350
    if (ml_code >= 15u) write_ext_len(dst + op2, ml_code, lane);           // :361-376
    op = op2 + nme;
    ip += len;                                                   // :382
    anchor = ip;
    return true;
}

__global__ __launch_bounds__(64) void k_hc_opt_parse_wave(const uint8_t *__restrict__ d_in, const uint64_t *__restrict__ d_in_off,
                                                          const uint32_t *__restrict__ d_in_len, uint8_t *__restrict__ d_out,
                                                          const uint64_t *__restrict__ d_out_off,
                                                          const uint32_t *__restrict__ d_out_cap, int64_t *__restrict__ d_result,
                                                          const uint32_t *__restrict__ d_res, uint64_t res_stride,
                                                          OptEntry *__restrict__ d_opt, uint32_t blk0, uint32_t nblocks,
                                                          uint32_t sufficient_len_in, uint32_t max_in_len) {
    __shared__ uint64_t opt_lds_raw[kOptLds];
    lds_opt *opt_l = (lds_opt *)opt_lds_raw;
    const uint32_t lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const uint32_t blk = blk0 + b;
    const uint8_t *src = d_in + d_in_off[blk];
    uint8_t *dst = d_out + d_out_off[blk];
    const uint32_t n = rfl(d_in_len[blk]), oend = rfl(d_out_cap[blk]);
    int64_t out;
    if (n > kMaxInput) out = kErrInputTooLarge;
    else if (n > max_in_len) out = kErrInvalidState;
    else if (n == 0) out = 0;
    else if (oend == 0) out = kErrOutputTooSmall;
    else if (n < kMfLimit + 1u) {                                                 // :1092-1094
        if (oend < n + 1u + n / 255u) out = kErrOutputTooSmall;
        else {
            if (lane == 0) dst[0] = (uint8_t)(n << 4);
            if (lane < n) dst[1u + lane] = src[lane];
            out = (int64_t)n + 1;
        }
    } else {
        const uint32_t *res = d_res + (uint64_t)b * res_stride;
        OptEntry *opt_g = d_opt + (uint64_t)b * kOptEntries;
        const uint32_t mflimit = n - kMfLimit;
        uint32_t sufficient_len = sufficient_len_in;                              // :1105-1108
        if (sufficient_len >= kOptNum) sufficient_len = kOptNum - 1u;
        // records: per-lane index (may differ between lanes)
        auto get = [&](uint32_t idx) -> uint64_t {
            if (idx < kOptLds) return opt_l[idx];
            const OptEntry e = opt_g[idx];
            return opt_pack((uint32_t)e.price, (uint32_t)e.off, (uint32_t)e.mlen, (uint32_t)e.litlen);
        };
        auto put = [&](uint32_t idx, uint64_t e) {
            if (idx < kOptLds) opt_l[idx] = e;
            else { OptEntry g; g.price = (int32_t)opt_price(e); g.off = (int32_t)opt_off(e); g.mlen = (int32_t)opt_mlen(e); g.litlen = (int32_t)opt_litlen(e); opt_g[idx] = g; }
        };
        // wave-uniform index: every lane reads the same record
        auto get_u = [&](uint32_t idx) -> uint64_t {
            const uint64_t e = get(idx);
            return (uint64_t)rfl((uint32_t)e) | ((uint64_t)rfl((uint32_t)(e >> 32)) << 32);
        };
        // search results of 64 consecutive positions in registers
        uint32_t wbase = 0;
        uint32_t rw = res[lane];                                  // (the result array is padded past n)
        auto best_match = [&](uint32_t p, uint32_t &len, uint32_t &off) {         // insertHC + insertAndGetWiderMatch result at p
            if (p - wbase >= 64u) { wbase = p; rw = res[p + lane]; }
            const uint32_t r = rdlane(rw, p - wbase);
            len = r & 0xFFFFu; off = r >> 16;
        };
        uint32_t ip = 0, anchor = 0, op = 0;
        bool failed = false;
        uint32_t guard = 0;
        while (ip <= mflimit && !failed) {                                        // :1111
            if (++guard > 2u * n + 16u) { failed = true; break; }                 // unreachable; never spin on the GPU
            const uint32_t llen = ip - anchor;
            uint32_t first_len, first_off;
            best_match(ip, first_len, first_off);                                 // :1115-1125
            if (first_len == 0) { ip += 1; continue; }                            // :1127
            if (first_len > sufficient_len) {                                     // :1133-1147
                if (!encode_sequence_wave(src, dst, oend, ip, op, anchor, first_len, first_off, lane)) failed = true;
                continue;
            }
            const uint32_t match_ml = first_len;                                  // :1160
            for (uint32_t k = lane; k <= match_ml; k += 64u) {                    // :1150-1169
                const uint64_t e = k < kMinMatch ? opt_pack((uint32_t)literals_price((int32_t)(llen + k)), 0u, 1u, llen + k)
                                                 : opt_pack((uint32_t)sequence_price((int32_t)llen, (int32_t)k), first_off, k, llen);
                put(k, e);
            }
            uint32_t last_match_pos = match_ml;                                   // :1171
            {   // :1174-1180 (opt[last_match_pos].price is the price just written for ml == match_ml, or of a literal record)
                const uint32_t base = match_ml < kMinMatch ? (uint32_t)literals_price((int32_t)(llen + match_ml))
                                                           : (uint32_t)sequence_price((int32_t)llen, (int32_t)match_ml);
                if (lane >= 1u && lane <= kTrailingLiterals)
                    put(last_match_pos + lane, opt_pack(base + (uint32_t)literals_price((int32_t)lane), 0u, 1u, lane));
            }
            bool encoded_early = false;
            for (uint32_t cur = 1; cur < last_match_pos; cur++) {                 // :1183-1312
                const uint32_t cur_pos = ip + cur;
                if (cur_pos > mflimit) break;                                     // :1187
                const uint64_t ec = get_u(cur), en = get_u(cur + 1u);
                if (opt_price(en) <= opt_price(ec)) continue;                     // :1190
                uint32_t new_len, new_off;
                best_match(cur_pos, new_len, new_off);                            // :1193-1203
                if (new_len == 0) continue;                                       // :1205
                if ((new_len > sufficient_len) || (new_len + cur >= kOptNum)) {   // :1208
                    uint32_t rp = 0;                                              // :1216-1238 (forward walk, no reverse traversal)
                    while (rp < cur) {
                        const uint64_t e = get_u(rp);
                        const uint32_t ml = opt_mlen(e), off = opt_off(e);
                        if (ml == 1u) { ip += 1; rp += 1; continue; }
                        rp += ml;
                        if (!encode_sequence_wave(src, dst, oend, ip, op, anchor, ml, off, lane)) { failed = true; break; }
                    }
                    if (!failed && !encode_sequence_wave(src, dst, oend, ip, op, anchor, new_len, new_off, lane)) failed = true;   // :1241-1252
                    encoded_early = true;                                         // :1255
                    break;
                }
                const uint32_t base_litlen = opt_litlen(ec);                      // :1259
                const uint32_t pc = opt_price(ec);
                // :1276-1286 what a match that starts here costs before its own sequence price
                uint32_t ll, pb;
                if (opt_mlen(ec) == 1u) {
                    ll = base_litlen;
                    pb = cur > ll ? opt_price(get_u(cur - ll)) : 0u;
                } else {
                    ll = 0; pb = pc;
                }
                const uint32_t lit_base = pc - (uint32_t)literals_price((int32_t)base_litlen);
                // :1260-1302 in one pass: record cur + k for k = 1 .. max(3, new_len); k < 4 are the literal updates
                const uint32_t kmax = new_len > 3u ? new_len : 3u;
                bool grew = false;
                for (uint32_t k0 = 1; k0 <= kmax; k0 += 64u) {
                    const uint32_t k = k0 + lane;
                    const bool act = k <= kmax;
                    const uint32_t pos = cur + k;
                    const uint64_t old = act ? get(pos) : 0ull;
                    if (act) {
                        if (k < kMinMatch) {
                            const uint32_t price = lit_base + (uint32_t)literals_price((int32_t)(base_litlen + k));
                            if ((int32_t)price < (int32_t)opt_price(old)) put(pos, opt_pack(price, 0u, 1u, base_litlen + k));
                        } else {
                            const uint32_t price = pb + (uint32_t)sequence_price((int32_t)ll, (int32_t)k);
                            if (pos > last_match_pos + kTrailingLiterals || (int32_t)price <= (int32_t)opt_price(old)) {   // :1293
                                put(pos, opt_pack(price, new_off, k, ll));
                                if (k == new_len) grew = true;
                            }
                        }
                    }
                }
                if (ballot(grew) && last_match_pos < cur + new_len) last_match_pos = cur + new_len;   // :1294
                {   // :1305-1311
                    const uint32_t base = opt_price(get_u(last_match_pos));
                    if (lane >= 1u && lane <= kTrailingLiterals)
                        put(last_match_pos + lane, opt_pack(base + (uint32_t)literals_price((int32_t)lane), 0u, 1u, lane));
                }
            }
            if (encoded_early || failed) continue;
            {   // reverse traversal (:1315-1332)
                const uint64_t el = get_u(last_match_pos);
                uint32_t sel_ml = opt_mlen(el), sel_off = opt_off(el);
                uint32_t cand = last_match_pos - sel_ml;
                for (;;) {
                    const uint64_t e = get_u(cand);
                    const uint32_t next_ml = opt_mlen(e), next_off = opt_off(e);
                    if (lane == 0) put(cand, opt_pack(opt_price(e), sel_off, sel_ml, opt_litlen(e)));
                    sel_ml = next_ml; sel_off = next_off;
                    if (next_ml > cand) break;
                    cand -= next_ml;
                }
            }
            uint32_t rp = 0;                                                      // :1335-1358
            while (rp < last_match_pos) {
                const uint64_t e = get_u(rp);
                const uint32_t ml = opt_mlen(e), off = opt_off(e);
                if (ml == 1u) { ip += 1; rp += 1; continue; }
                rp += ml;
                if (!encode_sequence_wave(src, dst, oend, ip, op, anchor, ml, off, lane)) { failed = true; break; }
            }
        }
        if (failed) out = kErrOutputTooSmall;
        else {
            const uint32_t fl = n - anchor;                                       // :1362-1388
            out = (int64_t)op;
            if (fl > 0) {
                const uint32_t nle = ext_len_bytes(fl);
                if ((uint64_t)op + fl + 1u > oend || (uint64_t)op + 1u + nle + fl > oend) out = kErrOutputTooSmall;
                else {
                    if (lane == 0) dst[op] = (uint8_t)((fl >= 15u ? 15u : fl) << 4);
                    if (fl >= 15u) write_ext_len(dst + op + 1u, fl, lane);
                    copy_bytes(dst + op + 1u + nle, src + anchor, fl, lane);
                    out = (int64_t)(op + 1u + nle + fl);
                }
            }
        }
    }
    if (lane == 0) d_result[blk] = out;
}

}  // namespace zlz4

extern "C" size_t zlz4_hc_mid_workspace_bytes(uint32_t chunk_blocks) {
    return (size_t)chunk_blocks * 2u * zlz4::kMidTableSize * sizeof(uint32_t);
}
extern "C" size_t zlz4_hc_opt_workspace_bytes(uint32_t chunk_blocks) {
    return (size_t)chunk_blocks * zlz4::kOptEntries * sizeof(zlz4::OptEntry);
}

extern "C" int zlz4_launch_hc_mid(hipStream_t stream, const uint8_t *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
                                  uint8_t *d_out, const uint64_t *d_out_off, const uint32_t *d_out_cap, int64_t *d_result,
                                  uint32_t nblocks, void *ws, uint32_t chunk, uint32_t max_in_len) {
    for (uint32_t b0 = 0; b0 < nblocks; b0 += chunk) {
        const uint32_t nb = nblocks - b0 < chunk ? nblocks - b0 : chunk;
        if (hipMemsetAsync(ws, 0, zlz4_hc_mid_workspace_bytes(nb), stream) != hipSuccess) return -7;   // :725-726
        // blocks per wavefront: two walk in lock step (the probe loop re-converges every iteration, a match is handled by
        // the lanes that have one); 1 / 2 / 4 / 8: 124.7 / 106.2 / 133.6 / 153.7 ms on 16 384 blocks, 516 / 474 / 498 / 508 on 65 536
        static const uint32_t lanes = [] { const char *e = zlz4_tune_env("ZLZ4_MID_LANES"); const uint32_t v = e ? (uint32_t)atoi(e) : 2u; return v >= 1u && v <= 64u ? v : 2u; }();
        hipLaunchKernelGGL(zlz4::k_hc_mid_serial, dim3((nb + lanes - 1u) / lanes), dim3(lanes), 0, stream, d_in, d_in_off, d_in_len, d_out,
                           d_out_off, d_out_cap, d_result, static_cast<uint32_t *>(ws), b0, nb, max_in_len);
    }
    return hipGetLastError() == hipSuccess ? 0 : -7;
}

// parse of one chunk whose K2 results are already in d_res (called from zlz4_compress_hc.hip's chunk loop)
extern "C" int zlz4_launch_hc_opt_parse(hipStream_t stream, const uint8_t *d_in, const uint64_t *d_in_off,
                                        const uint32_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off,
                                        const uint32_t *d_out_cap, int64_t *d_result, const void *d_res, uint64_t res_stride,
                                        int wide, void *d_opt, uint32_t b0, uint32_t nb, uint32_t sufficient_len,
                                        uint32_t max_in_len) {
    static const uint32_t lanes = [] { const char *e = zlz4_tune_env("ZLZ4_OPT_LANES"); const uint32_t v = e ? (uint32_t)atoi(e) : 1u; return v >= 1u && v <= 64u ? v : 1u; }();
    // blocks <= 64 KiB: one wavefront per block, the lanes take the records of a position's price update (A/B switch for
    // profiles/: ZLZ4_OPT_WAVE=0 in the tuning build = the one-lane kernel)
    static const bool wave = [] { const char *e = zlz4_tune_env("ZLZ4_OPT_WAVE"); return !(e && atoi(e) == 0); }();
    if (!wide && wave) {
        hipLaunchKernelGGL(zlz4::k_hc_opt_parse_wave, dim3(nb), dim3(64), 0, stream, d_in, d_in_off, d_in_len, d_out, d_out_off,
                           d_out_cap, d_result, static_cast<const uint32_t *>(d_res), res_stride,
                           static_cast<zlz4::OptEntry *>(d_opt), b0, nb, sufficient_len, max_in_len);
        return hipGetLastError() == hipSuccess ? 0 : -7;
    }
    if (wide)
        hipLaunchKernelGGL(zlz4::k_hc_opt_parse<uint64_t>, dim3((nb + lanes - 1u) / lanes), dim3(lanes), 0, stream, d_in, d_in_off, d_in_len,
                           d_out, d_out_off, d_out_cap, d_result, static_cast<const uint64_t *>(d_res), res_stride,
                           static_cast<zlz4::OptEntry *>(d_opt), b0, nb, sufficient_len, max_in_len);
    else
        hipLaunchKernelGGL(zlz4::k_hc_opt_parse<uint32_t>, dim3((nb + lanes - 1u) / lanes), dim3(lanes), 0, stream, d_in, d_in_off, d_in_len,
                           d_out, d_out_off, d_out_cap, d_result, static_cast<const uint32_t *>(d_res), res_stride,
                           static_cast<zlz4::OptEntry *>(d_opt), b0, nb, sufficient_len, max_in_len);
    return hipGetLastError() == hipSuccess ? 0 : -7;
}
