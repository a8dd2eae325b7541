/*
 * oracle/lz4_oracle.c -- TEST INFRASTRUCTURE ONLY (see lz4_oracle.h).
 *
 * Statement-by-statement CPU restatement of jedisct1/zig-lz4's
 *   src/lz4.zig    compressFast / decompressGeneric (no-dict)   :89-259, :263-519
 *   src/lz4hc.zig  compressHC levels 3-9 (hash chain)           :129-136, :162-297,
 *                  :308-386, :391-446, :491-681, :976-1064, :1394-1489
 *   src/lz4f.zig   compressFrame / decompressFrame              :138-351, :354-638
 * Every quirk of the Zig source is kept on purpose:
 * do NOT "fix" this file towards C liblz4 -- its output bytes differ.
 * "parity unpinned" beyond the KATs in tests/golden (see header).
 */
#include "lz4_oracle.h"

/* number of times compressHC hit the reference's `matchIndex - 1` underflow (see insertAndGetWiderMatch) */
int64_t zo_hc_reference_ub_count = 0;
int64_t zo_hc_reference_ub(int reset) { const int64_t v = zo_hc_reference_ub_count; if (reset) zo_hc_reference_ub_count = 0; return v; }

#include <stdlib.h>
#include <string.h>

/* ---- constants, src/lz4.zig:12-44 ---- */
#define MINMATCH 4
#define LASTLITERALS 5
#define MFLIMIT 12
#define ML_BITS 4
#define ML_MASK 15u
#define RUN_MASK 15u
#define LZ4_MAX_INPUT_SIZE 0x7E000000u
#define LZ4_DISTANCE_MAX 65535u
#define LZ4_HASHLOG 12
#define LZ4_HASH_SIZE_U32 (1u << 12)
#define ACCELERATION_MAX 65537u
#define HASH_MULTIPLIER 2654435761u

/* src/lz4.zig:60-72 */
static inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t rd32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static inline uint64_t rd64(const uint8_t *p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }
static inline void wr16(uint8_t *p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static inline void wr32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}
static inline void wr64(uint8_t *p, uint64_t v) { wr32(p, (uint32_t)v); wr32(p + 4, (uint32_t)(v >> 32)); }

/* src/lz4.zig:75-77 */
static inline uint32_t hash4(uint32_t sequence) {
    return (uint32_t)(sequence * HASH_MULTIPLIER) >> ((MINMATCH * 8) - LZ4_HASHLOG);
}

/* src/lz4.zig:80-83 */
size_t zo_compress_bound(size_t inputSize) {
    if (inputSize > LZ4_MAX_INPUT_SIZE) return 0;
    return inputSize + (inputSize / 255) + 16;
}

/* ===================== decompression, src/lz4.zig:89-251 ===================== */
/* decompressSafe = decompressGeneric(src, dst, dst.len, null, null) (:257-259):
 * lowPrefix == dst.ptr, dictEnd == null. */
static int64_t decompress_generic(const uint8_t *src, size_t srcLen, uint8_t *dst, size_t dstLen, size_t targetOutputSize) {
    if (srcLen == 0) return 0;                      /* :97 */
    if (dstLen == 0) return 0;                      /* :98 */
    if (targetOutputSize > dstLen) return ZO_ERR_OUTPUT_TOO_SMALL;   /* :99 */
    size_t ip = 0, op = 0;                          /* :106-107 */
    const size_t iend = srcLen, oend = targetOutputSize;   /* :108-109 */

    for (;;) {                                      /* :111 */
        if (ip >= iend) break;                      /* :113 */
        const uint8_t token = src[ip];              /* :116 */
        ip += 1;
        size_t literalLength = token >> ML_BITS;    /* :120 */
        if (literalLength == RUN_MASK) {            /* :123 */
            for (;;) {
                if (ip >= iend) return ZO_ERR_CORRUPTED_DATA;      /* :125 */
                const uint8_t s = src[ip];
                ip += 1;
                literalLength += s;
                if (s != 255) break;
            }
        }
        if (literalLength > 0) {                    /* :134 */
            if (ip + literalLength > iend) return ZO_ERR_CORRUPTED_DATA;    /* :136 */
            if (op + literalLength > oend) return ZO_ERR_OUTPUT_TOO_SMALL;  /* :137 */
            memcpy(dst + op, src + ip, literalLength);                      /* :140 */
            ip += literalLength;
            op += literalLength;
        }
        if (ip >= iend) break;                      /* :146 */
        if (ip + 2 > iend) return ZO_ERR_CORRUPTED_DATA;           /* :149 */
        const size_t offset = rd16(src + ip);       /* :150 */
        ip += 2;
        if (offset == 0) return ZO_ERR_CORRUPTED_DATA;             /* :154 */
        size_t matchLength = token & ML_MASK;       /* :157 */
        if (matchLength == ML_MASK) {               /* :160 */
            for (;;) {
                if (ip >= iend) return ZO_ERR_CORRUPTED_DATA;      /* :162 */
                const uint8_t s = src[ip];
                ip += 1;
                matchLength += s;
                if (s != 255) break;
            }
        }
        matchLength += MINMATCH;                    /* :171 */
        if (op + matchLength > oend) return ZO_ERR_OUTPUT_TOO_SMALL;        /* :174 */
        /* :177-186: matchPtr < lowPrefix(=dst.ptr) with dictEnd == null -> CorruptedData.
         * (currentPtr - offset < dstPtr  <=>  offset > op)                          */
        if (offset > op) return ZO_ERR_CORRUPTED_DATA;             /* :181-186 / :231 */
        const size_t matchPos = op - offset;        /* :232 */
        if (offset < matchLength) {                 /* :235 */
            for (size_t i = 0; i < matchLength; i++) dst[op + i] = dst[matchPos + i];  /* :238-240 */
            op += matchLength;
        } else {
            memcpy(dst + op, dst + matchPos, matchLength);         /* :244 */
            op += matchLength;
        }
    }
    return (int64_t)op;                             /* :250 */
}

/* src/lz4.zig:257-259 */
int64_t zo_decompress_safe(const uint8_t *src, size_t n, uint8_t *dst, size_t cap) {
    return decompress_generic(src, n, dst, cap, cap);
}
/* src/lz4.zig:619-621 */
int64_t zo_decompress_safe_partial(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t targetOutputSize) {
    return decompress_generic(src, n, dst, cap, targetOutputSize);
}

/* ===================== fast compression, src/lz4.zig:263-519 ===================== */

/* src/lz4.zig:449-482 */
static int64_t compress_as_literals(const uint8_t *src, size_t srcLen, uint8_t *dst, size_t dstLen) {
    const size_t literalLength = srcLen;
    size_t op = 0;
    if (dstLen < 1) return ZO_ERR_OUTPUT_TOO_SMALL;               /* :454 */
    if (literalLength >= RUN_MASK) {
        dst[op] = RUN_MASK << ML_BITS;
        op += 1;
        size_t len = literalLength - RUN_MASK;
        while (len >= 255) {
            if (op >= dstLen) return ZO_ERR_OUTPUT_TOO_SMALL;     /* :462 */
            dst[op] = 255;
            op += 1;
            len -= 255;
        }
        if (op >= dstLen) return ZO_ERR_OUTPUT_TOO_SMALL;         /* :468 */
        dst[op] = (uint8_t)len;
        op += 1;
    } else {
        dst[op] = (uint8_t)(literalLength << ML_BITS);
        op += 1;
    }
    if (op + literalLength > dstLen) return ZO_ERR_OUTPUT_TOO_SMALL;   /* :477 */
    memcpy(dst + op, src, literalLength);
    op += literalLength;
    return (int64_t)op;
}

/* src/lz4.zig:484-519 */
static int64_t finish_compression(const uint8_t *src, size_t srcLen, uint8_t *dst, size_t dstLen,
                                  size_t anchor, size_t op) {
    const size_t literalLength = srcLen - anchor;
    size_t outPos = op;
    if (literalLength == 0) return (int64_t)outPos;               /* :488 */
    if (outPos >= dstLen) return ZO_ERR_OUTPUT_TOO_SMALL;         /* :491 */
    if (literalLength >= RUN_MASK) {
        dst[outPos] = RUN_MASK << ML_BITS;
        outPos += 1;
        size_t len = literalLength - RUN_MASK;
        while (len >= 255) {
            if (outPos >= dstLen) return ZO_ERR_OUTPUT_TOO_SMALL; /* :499 */
            dst[outPos] = 255;
            outPos += 1;
            len -= 255;
        }
        if (outPos >= dstLen) return ZO_ERR_OUTPUT_TOO_SMALL;     /* :505 */
        dst[outPos] = (uint8_t)len;
        outPos += 1;
    } else {
        dst[outPos] = (uint8_t)(literalLength << ML_BITS);
        outPos += 1;
    }
    if (outPos + literalLength > dstLen) return ZO_ERR_OUTPUT_TOO_SMALL;   /* :514 */
    memcpy(dst + outPos, src + anchor, literalLength);
    outPos += literalLength;
    return (int64_t)outPos;
}

/* src/lz4.zig:292-447 */
int64_t zo_compress_fast(const uint8_t *src, size_t srcSize, uint8_t *dst, size_t dstLen,
                         uint32_t acceleration) {
    if (srcSize > LZ4_MAX_INPUT_SIZE) return ZO_ERR_INPUT_TOO_LARGE;     /* :296 */
    if (srcSize == 0) return 0;                                          /* :299 */
    if (srcSize < MFLIMIT + 1) return compress_as_literals(src, srcSize, dst, dstLen);  /* :302-304 */

    uint32_t hashTable[LZ4_HASH_SIZE_U32];                               /* :307, :263-268 */
    memset(hashTable, 0, sizeof hashTable);

    size_t ip = 0, op = 0, anchor = 0;                                   /* :309-311 */
    const size_t mflimitPlusOne = srcSize - MFLIMIT;                     /* :313 */
    const size_t matchLimit = srcSize - LASTLITERALS;                    /* :314 */

    ip += 1;                                                             /* :317 */

    while (ip < mflimitPlusOne) {                                        /* :320 */
        uint32_t accel = acceleration;                                   /* :321 clamp */
        if (accel < 1) accel = 1;
        if (accel > ACCELERATION_MAX) accel = ACCELERATION_MAX;
        size_t step = accel;                                             /* :322 */
        size_t searchMatchNb = accel;                                    /* :323 */
        size_t match;
        size_t forwardIp = ip;                                           /* :327 */

        for (;;) {                                                       /* :329 */
            ip = forwardIp;                                              /* :330 */
            forwardIp += step;                                           /* :331 */
            step = searchMatchNb >> 6;                                   /* :332 */
            searchMatchNb += 1;                                          /* :333 */
            if (forwardIp > mflimitPlusOne)                              /* :335 */
                return finish_compression(src, srcSize, dst, dstLen, anchor, op);
            const uint32_t h = hash4(rd32(src + ip));                    /* :341 */
            match = hashTable[h];                                        /* :342 */
            const int is_valid_match = match > 0 &&                      /* :345-348 */
                match < ip &&
                match + LZ4_DISTANCE_MAX >= ip &&
                rd32(src + match) == rd32(src + ip);
            hashTable[h] = (uint32_t)ip;                                 /* :350 */
            if (is_valid_match) break;                                   /* :352 */
        }

        const size_t literalLength = ip - anchor;                        /* :360 */
        const size_t tokenPos = op;                                      /* :363 */
        op += 1;
        if (op >= dstLen) return ZO_ERR_OUTPUT_TOO_SMALL;                /* :365 */

        if (literalLength >= RUN_MASK) {                                 /* :368 */
            dst[tokenPos] = RUN_MASK << ML_BITS;
            size_t len = literalLength - RUN_MASK;
            while (len >= 255) {
                if (op >= dstLen) return ZO_ERR_OUTPUT_TOO_SMALL;        /* :374 */
                dst[op] = 255;
                op += 1;
                len -= 255;
            }
            if (op >= dstLen) return ZO_ERR_OUTPUT_TOO_SMALL;            /* :380 */
            dst[op] = (uint8_t)len;
            op += 1;
        } else {
            dst[tokenPos] = (uint8_t)(literalLength << ML_BITS);         /* :384 */
        }

        if (op + literalLength > dstLen) return ZO_ERR_OUTPUT_TOO_SMALL; /* :388 */
        if (literalLength > 0) {
            memcpy(dst + op, src + anchor, literalLength);               /* :390 */
            op += literalLength;
        }

        const uint16_t offset = (uint16_t)(ip - match);                  /* :395 */
        if (op + 2 > dstLen) return ZO_ERR_OUTPUT_TOO_SMALL;             /* :396 */
        wr16(dst + op, offset);
        op += 2;

        ip += MINMATCH;                                                  /* :401 */
        match += MINMATCH;
        size_t matchLength = 0;
        while (ip < matchLimit) {                                        /* :405 */
            if (src[ip] == src[match]) {
                ip += 1;
                match += 1;
                matchLength += 1;
            } else {
                break;
            }
        }

        if (matchLength >= ML_MASK) {                                    /* :416 */
            dst[tokenPos] |= ML_MASK;
            size_t len = matchLength - ML_MASK;
            while (len >= 255) {
                if (op >= dstLen) return ZO_ERR_OUTPUT_TOO_SMALL;        /* :421 */
                dst[op] = 255;
                op += 1;
                len -= 255;
            }
            if (op >= dstLen) return ZO_ERR_OUTPUT_TOO_SMALL;            /* :427 */
            dst[op] = (uint8_t)len;
            op += 1;
        } else {
            dst[tokenPos] |= (uint8_t)matchLength;                       /* :431 */
        }

        anchor = ip;                                                     /* :435 */
        if (ip < mflimitPlusOne) {                                       /* :438 */
            const uint32_t h = hash4(rd32(src + ip));
            hashTable[h] = (uint32_t)ip;
            ip += 1;
        }
    }
    return finish_compression(src, srcSize, dst, dstLen, anchor, op);    /* :446 */
}

/* src/lz4.zig:283-285 */
int64_t zo_compress_default(const uint8_t *src, size_t n, uint8_t *dst, size_t cap) {
    return zo_compress_fast(src, n, dst, cap, 1);
}

/* src/lz4.zig:524-526: @sizeOf(HashTable) */
size_t zo_sizeof_state(void) { return LZ4_HASH_SIZE_U32 * sizeof(uint32_t); }

/* src/lz4.zig:531-546; compressFastWithHashTable (:624-740) is a verbatim copy of compressFast's loop */
int64_t zo_compress_fast_ext_state(size_t stateLen, const uint8_t *src, size_t n, uint8_t *dst, size_t cap, uint32_t accel) {
    if (stateLen < zo_sizeof_state()) return ZO_ERR_INVALID_STATE;      /* :532 */
    return zo_compress_fast(src, n, dst, cap, accel);                  /* :534-545 */
}

/* src/lz4.zig:551-616.  NOTE: like the reference, `dst` holds the output of the LAST attempt, which is not
 * necessarily the best one; only the returned size and *srcSizePtr are specified here. */
int64_t zo_compress_dest_size(const uint8_t *src, uint8_t *dst, size_t dstLen, size_t *srcSizePtr) {
    const size_t maxSrcSize = *srcSizePtr;
    if (maxSrcSize == 0) { *srcSizePtr = 0; return 0; }                /* :553-556 */
    const size_t maxCompressed = zo_compress_bound(maxSrcSize);        /* :559 */
    if (dstLen >= maxCompressed) {                                     /* :560-564 */
        const int64_t r = zo_compress_default(src, maxSrcSize, dst, dstLen);
        if (r < 0) return r;
        *srcSizePtr = maxSrcSize;
        return r;
    }
    size_t low = 1, high = maxSrcSize, bestSize = 0, bestCompressedSize = 0;   /* :567-570 */
    if (dstLen <= maxSrcSize) {                                        /* :573-586 */
        const size_t estimate = dstLen < maxSrcSize ? dstLen : maxSrcSize;
        const int64_t r = zo_compress_default(src, estimate, dst, dstLen);
        if (r >= 0) {
            if ((size_t)r <= dstLen) { bestSize = estimate; bestCompressedSize = (size_t)r; low = estimate + 1; }
            else high = estimate - 1;
        } else {
            high = estimate - 1;
        }
    }
    while (low <= high) {                                              /* :589-612 */
        const size_t mid = low + (high - low) / 2;
        if (mid == 0 || mid > maxSrcSize) break;
        const int64_t r = zo_compress_default(src, mid, dst, dstLen);
        if (r >= 0) {
            if ((size_t)r <= dstLen) {
                bestSize = mid; bestCompressedSize = (size_t)r;
                if (mid == maxSrcSize) break;
                low = mid + 1;
            } else {
                high = mid - 1;
            }
        } else {
            high = mid - 1;
        }
        if (low > maxSrcSize) break;
    }
    *srcSizePtr = bestSize;                                            /* :614-615 */
    return (int64_t)bestCompressedSize;
}

/* ===================== HC, src/lz4hc.zig ===================== */
#define LZ4HC_CLEVEL_MIN 2
#define LZ4HC_CLEVEL_DEFAULT 9
#define LZ4HC_CLEVEL_MAX 12
#define LZ4HC_MAXD (1u << 16)
#define LZ4HC_MAXD_MASK (LZ4HC_MAXD - 1)
#define LZ4HC_HASH_LOG 15
#define LZ4HC_HASHTABLESIZE (1u << LZ4HC_HASH_LOG)

enum { STRAT_MID, STRAT_HC, STRAT_OPT };
typedef struct { int strat; int32_t nbSearches; uint32_t targetLength; } clevel_params;
/* src/lz4hc.zig:72-86 */
static const clevel_params clevelTable[13] = {
    {STRAT_MID, 2, 16}, {STRAT_MID, 2, 16}, {STRAT_MID, 2, 16},
    {STRAT_HC, 4, 16}, {STRAT_HC, 8, 16}, {STRAT_HC, 16, 16}, {STRAT_HC, 32, 16},
    {STRAT_HC, 64, 16}, {STRAT_HC, 128, 16}, {STRAT_HC, 256, 16},
    {STRAT_OPT, 96, 64}, {STRAT_OPT, 512, 128}, {STRAT_OPT, 16384, 4096},
};

/* src/lz4hc.zig:391-404 (only the fields the one-shot hash-chain path touches) */
typedef struct {
    uint32_t hashTable[LZ4HC_HASHTABLESIZE];
    uint16_t chainTable[LZ4HC_MAXD];
    const uint8_t *prefixStart;
    const uint8_t *dictStart;
    uint32_t dictLimit, lowLimit, nextToUpdate;
} hc_ctx;

typedef struct { int32_t off, len, back; } hc_match;      /* :449-453 */

static inline uint32_t hashHC(uint32_t sequence) {         /* :129-131 */
    return (uint32_t)(sequence * HASH_MULTIPLIER) >> ((MINMATCH * 8) - LZ4HC_HASH_LOG);
}
static inline uint32_t hashPtr(const uint8_t *p) { return hashHC(rd32(p)); }   /* :134-136 */

/* src/lz4hc.zig:170-199 */
static size_t countPattern(const uint8_t *ip, const uint8_t *iEnd, uint32_t pattern32) {
    const uint8_t *const iStart = ip;
    const uint8_t *ptr = ip;
    const uint64_t pattern64 = (uint64_t)pattern32 | ((uint64_t)pattern32 << 32);
    while ((uintptr_t)ptr + 7 < (uintptr_t)iEnd) {
        const uint64_t diff = rd64(ptr) ^ pattern64;
        if (diff == 0) {
            ptr += 8;
        } else {
            const size_t nbCommon = (size_t)__builtin_ctzll(diff) >> 3;
            return (size_t)(ptr - iStart) + nbCommon;
        }
    }
    uint32_t patternByte = pattern32;
    while (ptr < iEnd) {
        if (ptr[0] != (uint8_t)patternByte) break;
        ptr += 1;
        patternByte >>= 8;
        if (patternByte == 0) patternByte = pattern32;
    }
    return (size_t)(ptr - iStart);
}

/* src/lz4hc.zig:202-222 */
static size_t reverseCountPattern(const uint8_t *ip, const uint8_t *iLow, uint32_t pattern) {
    const uint8_t *const iStart = ip;
    const uint8_t *ptr = ip;
    while ((uintptr_t)ptr >= (uintptr_t)iLow + 4) {
        if (rd32(ptr - 4) != pattern) break;
        ptr -= 4;
    }
    uint8_t patternBytes[4];
    wr32(patternBytes, pattern);                 /* little-endian host in the reference */
    size_t byteIdx = 3;
    while (ptr > iLow) {
        if ((ptr - 1)[0] != patternBytes[byteIdx]) break;
        ptr -= 1;
        if (byteIdx == 0) byteIdx = 3; else byteIdx -= 1;
    }
    return (size_t)(iStart - ptr);
}

/* src/lz4hc.zig:225-228 */
static inline int isRepetitivePattern(uint32_t pattern) {
    return ((pattern & 0xFFFF) == (pattern >> 16)) && ((pattern & 0xFF) == (pattern >> 24));
}

/* src/lz4hc.zig:234-264 */
static size_t lz4Count(const uint8_t *pIn, const uint8_t *pMatch, const uint8_t *pInLimit) {
    const uint8_t *ip = pIn, *match = pMatch;
    size_t counted = 0;
    while ((uintptr_t)ip + 8 <= (uintptr_t)pInLimit) {
        const uint64_t diff = rd64(ip) ^ rd64(match);
        if (diff == 0) {
            ip += 8; match += 8; counted += 8;
        } else {
            return counted + ((size_t)__builtin_ctzll(diff) >> 3);
        }
    }
    while (ip < pInLimit) {
        if (ip[0] != match[0]) break;
        ip += 1; match += 1; counted += 1;
    }
    return counted;
}

/* src/lz4hc.zig:267-297 */
static int32_t countBack(const uint8_t *ip, const uint8_t *match, const uint8_t *iMin, const uint8_t *mMin) {
    int32_t back = 0;
    const size_t d1 = (size_t)(ip - iMin), d2 = (size_t)(match - mMin);
    const int32_t min = -(int32_t)(d1 < d2 ? d1 : d2);
    while ((back - min) > 3) {
        const uint32_t v = rd32(ip + (back - 4)) ^ rd32(match + (back - 4));
        if (v != 0) {
            /* :279-280 kept literally (counts from the low end of the word) */
            const int32_t nbCommon = (int32_t)(__builtin_ctz(v) >> 3);
            return back - nbCommon;
        }
        back -= 4;
    }
    while (back > min) {
        if (ip[back - 1] != match[back - 1]) break;
        back -= 1;
    }
    return back;
}

/* src/lz4hc.zig:308-386; limit is always .limitedOutput on the compressHashChain path (:1025) */
static int encodeSequence(const uint8_t **ip, uint8_t **op, const uint8_t **anchor,
                          int32_t matchLength, int32_t offset, int limitedOutput, uint8_t *oend) {
    /* Reference defect #1 can hand this function an ip below the anchor: `ip - anchor` then
     * underflows in the reference (panic in Zig's safe modes, a wild copy otherwise).  No defined output exists; the
     * restatement reports OutputTooSmall, which is what the HIP path's 32-bit arithmetic makes of the same state. */
    if (*ip < *anchor) { zo_hc_reference_ub_count += 1; return 1; }
    const size_t litLen = (size_t)(*ip - *anchor);                       /* :317 */
    if (limitedOutput) {                                                 /* :320-325 */
        const size_t needed = (litLen / 255) + litLen + (2 + 1 + LASTLITERALS);
        if ((uintptr_t)*op + needed > (uintptr_t)oend) return 1;
    }
    uint8_t *token = *op;                                                /* :328 */
    *op += 1;
    if (litLen >= RUN_MASK) {                                            /* :331 */
        size_t len = litLen - RUN_MASK;
        token[0] = RUN_MASK << ML_BITS;
        while (len >= 255) { (*op)[0] = 255; *op += 1; len -= 255; }
        (*op)[0] = (uint8_t)len;
        *op += 1;
    } else {
        token[0] = (uint8_t)(litLen << ML_BITS);
    }
    memcpy(*op, *anchor, litLen);                                        /* :346 */
    *op += litLen;
    wr16(*op, (uint16_t)offset);                                         /* :350 */
    *op += 2;
    const size_t mlCode = (size_t)(matchLength - MINMATCH);              /* :354 */
    if (limitedOutput) {                                                 /* :355-359 */
        if ((uintptr_t)*op + (mlCode / 255) + (1 + LASTLITERALS) > (uintptr_t)oend) return 1;
    }
    if (mlCode >= ML_MASK) {                                             /* :361 */
        token[0] += ML_MASK;
        size_t remaining = mlCode - ML_MASK;
        while (remaining >= 510) { (*op)[0] = 255; (*op)[1] = 255; *op += 2; remaining -= 510; }
        if (remaining >= 255) { (*op)[0] = 255; *op += 1; remaining -= 255; }
        (*op)[0] = (uint8_t)remaining;
        *op += 1;
    } else {
        token[0] += (uint8_t)mlCode;
    }
    *ip += matchLength;                                                  /* :382 */
    *anchor = *ip;
    return 0;
}

/* src/lz4hc.zig:491-510 */
static void insertHC(hc_ctx *ctx, const uint8_t *ip) {
    const uint8_t *prefixPtr = ctx->prefixStart;
    const uint32_t prefixIdx = ctx->dictLimit;
    const uint32_t target = (uint32_t)((size_t)(ip - prefixPtr) + prefixIdx);
    uint32_t idx = ctx->nextToUpdate;
    while (idx < target) {
        const size_t offset = idx - prefixIdx;
        const uint32_t h = hashPtr(prefixPtr + offset);
        const uint32_t prevIdx = ctx->hashTable[h];
        const uint32_t delta = (prevIdx > idx) ? LZ4_DISTANCE_MAX + 1 : idx - prevIdx;
        const uint16_t deltaClamped = (delta > LZ4_DISTANCE_MAX) ? (uint16_t)LZ4_DISTANCE_MAX : (uint16_t)delta;
        ctx->chainTable[idx & LZ4HC_MAXD_MASK] = deltaClamped;
        ctx->hashTable[h] = idx;
        idx += 1;
    }
    ctx->nextToUpdate = target;
}

/* src/lz4hc.zig:538-681 */
static hc_match insertAndGetWiderMatch(hc_ctx *ctx, const uint8_t *ip, const uint8_t *iLowLimit,
                                       const uint8_t *iHighLimit, int32_t longest,
                                       int32_t maxNbAttempts, int patternAnalysis) {
    const uint8_t *prefixPtr = ctx->prefixStart;
    const uint32_t prefixIdx = ctx->dictLimit;
    const uint32_t ipIndex = (uint32_t)((size_t)(ip - prefixPtr) + prefixIdx);          /* :552 */
    const int withinStartDistance = (ctx->lowLimit + (LZ4_DISTANCE_MAX + 1) > ipIndex); /* :553 */
    const uint32_t lowestMatchIndex = withinStartDistance ? ctx->lowLimit : ipIndex - LZ4_DISTANCE_MAX;
    const uint8_t *dictStart = ctx->dictStart;
    const uint32_t dictIdx = ctx->lowLimit;
    int32_t nbAttempts = maxNbAttempts;
    const uint32_t pattern = rd32(ip);                                                   /* :558 */

    hc_match result = { 0, longest, 0 };                                                 /* :560 */
    uint32_t matchIndex = ctx->hashTable[hashPtr(ip)];                                   /* :563 */
    if (matchIndex == 0) return result;                                                  /* :566-568 */

    while ((matchIndex > 0) && (nbAttempts > 0)) {                                       /* :571 */
        if (matchIndex > ipIndex || (ipIndex - matchIndex) > LZ4_DISTANCE_MAX) break;    /* :573 */
        nbAttempts -= 1;                                                                 /* :577 */
        if (matchIndex >= lowestMatchIndex) {                                            /* :579 */
            const uint8_t *matchPtr = (matchIndex >= dictIdx)
                ? prefixPtr + (matchIndex - prefixIdx)
                : dictStart + (matchIndex - ctx->lowLimit);
            if (rd32(matchPtr) == pattern) {                                             /* :586 */
                const int32_t mlt = (int32_t)(MINMATCH + lz4Count(ip + MINMATCH, matchPtr + MINMATCH, iHighLimit));
                int32_t back = 0;
                if (ip > iLowLimit) {                                                    /* :596 */
                    const uint8_t *mMin = (matchIndex >= dictIdx) ? prefixPtr : dictStart;
                    back = countBack(ip, matchPtr, iLowLimit, mMin);
                }
                const int32_t totalLength = mlt - back;                                  /* :604 */
                if (totalLength > result.len) {                                          /* :607 */
                    result.len = totalLength;
                    result.off = (int32_t)(ipIndex - matchIndex);
                    result.back = back;
                    if (totalLength > maxNbAttempts) break;                              /* :613 */
                }
            }
        }
        const uint16_t delta = ctx->chainTable[matchIndex & LZ4HC_MAXD_MASK];            /* :619 */
        if (delta == 0 || delta > matchIndex) break;                                     /* :620 */
        matchIndex -= delta;                                                             /* :621 */
    }

    if (patternAnalysis && result.len > 0) {                                             /* :626 */
        const uint16_t delta = ctx->chainTable[matchIndex & LZ4HC_MAXD_MASK];            /* :627 */
        /* Reference defect (found by tools/fuzz_parity.py): when the walk ended at matchIndex == 0 -- the normal end
         * of a chain -- chainTable[0] is the slot of position 65536 as well (index & 0xFFFF), so on inputs longer than
         * 64 KiB it can hold 1; with a repetitive pattern :636 then computes `matchIndex - 1` on a u32 zero: a
         * panic in Zig's safe build modes, a wild read (this restatement used to SEGV here) in ReleaseFast.  The
         * reference has no defined output for such an input.  The restatement records the event and skips the
         * branch, which is what the HIP path does by construction (its link of position 0 is 0). */
        if (delta == 1 && matchIndex == 0 && isRepetitivePattern(pattern)) {
            zo_hc_reference_ub_count += 1;
        } else
        if (delta == 1) {                                                                /* :629 */
            if (isRepetitivePattern(pattern)) {                                          /* :631 */
                const size_t srcPatternLength = countPattern(ip + 4, iHighLimit, pattern) + 4;   /* :633 */
                const uint32_t matchCandidateIdx = matchIndex - 1;                       /* :636 */
                if (matchCandidateIdx >= lowestMatchIndex && matchCandidateIdx >= dictIdx) {     /* :637 */
                    const uint8_t *matchPtr = (matchCandidateIdx >= dictIdx)
                        ? prefixPtr + (matchCandidateIdx - prefixIdx)
                        : dictStart + (matchCandidateIdx - ctx->lowLimit);
                    if (rd32(matchPtr) == pattern) {                                     /* :644 */
                        const size_t forwardPatternLength = countPattern(matchPtr + 4, iHighLimit, pattern) + 4;
                        const uint8_t *lowestMatchPtr = (matchCandidateIdx >= dictIdx) ? prefixPtr : dictStart;
                        const size_t backLength = reverseCountPattern(matchPtr, lowestMatchPtr, pattern);
                        uint32_t lo = matchCandidateIdx - (uint32_t)backLength;          /* :653 */
                        if (lo < lowestMatchIndex) lo = lowestMatchIndex;
                        const uint32_t limitedBackLength = matchCandidateIdx - lo;
                        const size_t currentSegmentLength = (size_t)limitedBackLength + forwardPatternLength;
                        uint32_t newMatchIndex = matchCandidateIdx;                      /* :657 */
                        const size_t mn = currentSegmentLength < srcPatternLength ? currentSegmentLength : srcPatternLength;
                        const int32_t maxML = (int32_t)mn;                               /* :658 */
                        if (currentSegmentLength >= srcPatternLength && forwardPatternLength <= srcPatternLength) {
                            newMatchIndex = matchCandidateIdx + (uint32_t)forwardPatternLength - (uint32_t)srcPatternLength;
                        } else {
                            newMatchIndex = matchCandidateIdx - limitedBackLength;       /* :665 */
                        }
                        if (maxML > result.len && (ipIndex - newMatchIndex) <= LZ4_DISTANCE_MAX) {  /* :669 */
                            result.len = maxML;
                            result.off = (int32_t)(ipIndex - newMatchIndex);
                            result.back = 0;
                        }
                    }
                }
            }
        }
    }
    return result;
}

/* src/lz4hc.zig:514-535 */
static hc_match insertAndFindBestMatch(hc_ctx *ctx, const uint8_t *ip, const uint8_t *iLimit,
                                       int32_t maxNbAttempts, int patternAnalysis) {
    insertHC(ctx, ip);
    return insertAndGetWiderMatch(ctx, ip, ip, iLimit, MINMATCH - 1, maxNbAttempts, patternAnalysis);
}

/* src/lz4hc.zig:1394-1425 */
static int64_t encodeLiterals(const uint8_t *src, size_t srcLen, uint8_t *dst, size_t dstLen) {
    if (dstLen < srcLen + 1 + (srcLen / 255)) return ZO_ERR_OUTPUT_TOO_SMALL;   /* :1395 */
    uint8_t *op = dst;
    const size_t litLen = srcLen;
    if (litLen >= RUN_MASK) {
        size_t len = litLen - RUN_MASK;
        op[0] = RUN_MASK << ML_BITS; op += 1;
        while (len >= 255) { op[0] = 255; op += 1; len -= 255; }
        op[0] = (uint8_t)len; op += 1;
    } else {
        op[0] = (uint8_t)(litLen << ML_BITS); op += 1;
    }
    memcpy(op, src, litLen);
    op += litLen;
    return (int64_t)(op - dst);
}

/* src/lz4hc.zig:976-1064 */
static int64_t compressHashChain(hc_ctx *ctx, const uint8_t *src, size_t inputSize, uint8_t *dst,
                                 size_t dstLen, int32_t maxNbAttempts) {
    const int patternAnalysis = (maxNbAttempts > 128);                   /* :983 */
    const uint8_t *ip = src;
    const uint8_t *anchor = ip;
    const uint8_t *const iend = ip + inputSize;
    uint8_t *op = dst;
    uint8_t *const oend = op + dstLen;

    if (inputSize < MFLIMIT + 1) return encodeLiterals(src, inputSize, dst, dstLen);   /* :995-998 */
    const uint8_t *const mflimit = iend - MFLIMIT;                       /* :988 */
    const uint8_t *const matchlimit = iend - LASTLITERALS;               /* :989 */

    ctx->nextToUpdate = 0;                                               /* :1001-1006 */
    ctx->prefixStart = src;
    ctx->dictStart = src;
    ctx->dictLimit = 0;
    ctx->lowLimit = 0;

    while (ip <= mflimit) {                                              /* :1009 */
        const hc_match match = insertAndFindBestMatch(ctx, ip, matchlimit, maxNbAttempts, patternAnalysis);
        if (match.len < MINMATCH || match.off == 0) {                    /* :1013 */
            ip += 1;
            continue;
        }
        if (encodeSequence(&ip, &op, &anchor, match.len, match.off, 1, oend) != 0)   /* :1019-1031 */
            return ZO_ERR_OUTPUT_TOO_SMALL;
    }

    const size_t finalLiterals = (size_t)(iend - anchor);                /* :1035 */
    if (finalLiterals > 0) {
        if ((uintptr_t)op + finalLiterals + 1 > (uintptr_t)oend) return ZO_ERR_OUTPUT_TOO_SMALL;   /* :1037 */
        if (finalLiterals >= RUN_MASK) {
            size_t len = finalLiterals - RUN_MASK;
            op[0] = RUN_MASK << ML_BITS; op += 1;
            while (len >= 255) { op[0] = 255; op += 1; len -= 255; }
            op[0] = (uint8_t)len; op += 1;
        } else {
            op[0] = (uint8_t)(finalLiterals << ML_BITS); op += 1;
        }
        memcpy(op, anchor, finalLiterals);
        op += finalLiterals;
    }
    return (int64_t)(op - dst);
}

/* ---------------- LZ4MID, level 2: src/lz4hc.zig:687-971 ---------------- */
#define LZ4MID_HASHLOG 14
#define LZ4MID_HASHTABLESIZE (1u << LZ4MID_HASHLOG)
#define LZ4MID_HASHSIZE 8
#define HASH_MULTIPLIER_64 58295818150454627ULL                         /* :51 */
static inline uint32_t hashMid4Ptr(const uint8_t *p) {                  /* :139-146 */
    return (uint32_t)(rd32(p) * HASH_MULTIPLIER) >> (32 - LZ4MID_HASHLOG);
}
static inline uint32_t hashMid8Ptr(const uint8_t *p) {                  /* :149-157 (hashes the low 56 bits) */
    const uint64_t masked = rd64(p) << (64 - 56);
    return (uint32_t)((masked * HASH_MULTIPLIER_64) >> (64 - LZ4MID_HASHLOG));
}

/* :789-818 / :899-928 "fill table with end of match" (identical in both branches) */
static void mid_fill_end(uint32_t *hash4Table, uint32_t *hash8Table, const uint8_t *ip, const uint8_t *prefixPtr,
                         uint32_t prefixIdx, const uint8_t *ilimit, uint32_t ilimitIdx) {
    const uint32_t endMatchIdx = (uint32_t)((size_t)(ip - prefixPtr) + prefixIdx);
    const uint32_t pos_m2 = endMatchIdx - 2;
    if (pos_m2 < ilimitIdx) {
        if ((size_t)(ip - prefixPtr) > 5) {
            const uint8_t *ip5 = ip - 5;
            if (ip5 <= ilimit) hash8Table[hashMid8Ptr(ip5)] = endMatchIdx - 5;
        }
        /* `@intFromPtr(ip) >= 3` etc. compare the ADDRESS with a small constant: always true */
        { const uint8_t *ip3 = ip - 3; if (ip3 <= ilimit) hash8Table[hashMid8Ptr(ip3)] = endMatchIdx - 3; }
        { const uint8_t *ip2 = ip - 2;
          if (ip2 <= ilimit) { hash8Table[hashMid8Ptr(ip2)] = endMatchIdx - 2; hash4Table[hashMid4Ptr(ip2)] = endMatchIdx - 2; } }
        { const uint8_t *ip1 = ip - 1; if (ip1 <= ilimit) hash4Table[hashMid4Ptr(ip1)] = endMatchIdx - 1; }
    }
}

static int64_t compressMID(hc_ctx *ctx, const uint8_t *src, size_t inputSize, uint8_t *dst, size_t dstLen) {
    const uint8_t *ip = src;
    const uint8_t *anchor = ip;
    const uint8_t *const iend = ip + inputSize;
    uint8_t *op = dst;
    uint8_t *const oend = op + dstLen;
    if (inputSize < MFLIMIT + 1) return encodeLiterals(src, inputSize, dst, dstLen);     /* :706-708 */
    const uint8_t *const mflimit = iend - MFLIMIT;                       /* :698 */
    const uint8_t *const matchlimit = iend - LASTLITERALS;               /* :699 */
    const uint8_t *const ilimit = iend - LZ4MID_HASHSIZE;                /* :700 */
    ctx->nextToUpdate = 0; ctx->prefixStart = src; ctx->dictStart = src; ctx->dictLimit = 0; ctx->lowLimit = 0;   /* :711-716 */
    uint32_t *hash4Table = ctx->hashTable;                               /* :721 */
    uint32_t *hash8Table = ctx->hashTable + LZ4MID_HASHTABLESIZE;        /* :722 */
    memset(hash4Table, 0, LZ4MID_HASHTABLESIZE * sizeof(uint32_t));      /* :725-726 */
    memset(hash8Table, 0, LZ4MID_HASHTABLESIZE * sizeof(uint32_t));
    const uint8_t *prefixPtr = ctx->prefixStart;
    const uint32_t prefixIdx = ctx->dictLimit;
    const uint32_t ilimitIdx = (uint32_t)((size_t)(ilimit - prefixPtr) + prefixIdx);     /* :730 */

    while (ip <= mflimit) {                                              /* :733 */
        const uint32_t ipIndex = (uint32_t)((size_t)(ip - prefixPtr) + prefixIdx);
        uint32_t matchLength = 0, matchDistance = 0;
        {   /* long match, 8-byte hash :739-824 */
            const uint32_t h8 = hashMid8Ptr(ip);
            const uint32_t pos8 = hash8Table[h8];
            hash8Table[h8] = ipIndex;                                    /* :742 */
            if (pos8 > 0 && ipIndex - pos8 <= LZ4_DISTANCE_MAX) {        /* :744 */
                if (pos8 >= prefixIdx) {
                    const uint8_t *matchPtr = prefixPtr + (pos8 - prefixIdx);
                    if (matchPtr < ip) {                                 /* :748 */
                        const size_t mlt = lz4Count(ip, matchPtr, matchlimit);          /* :749 (counts from ip) */
                        if (mlt >= MINMATCH) {
                            matchLength = (uint32_t)mlt;
                            matchDistance = ipIndex - pos8;
                            const uint32_t finalIpIndex = ipIndex;       /* :765 */
                            if (ip + 1 <= ilimit) hash8Table[hashMid8Ptr(ip + 1)] = finalIpIndex + 1;   /* :767-769 */
                            if (ip + 2 <= ilimit) hash8Table[hashMid8Ptr(ip + 2)] = finalIpIndex + 2;   /* :770-772 */
                            if (ip + 1 <= ilimit) hash4Table[hashMid4Ptr(ip + 1)] = finalIpIndex + 1;   /* :773-775 */
                            if (encodeSequence(&ip, &op, &anchor, (int32_t)matchLength, (int32_t)matchDistance, 1, oend) != 0)
                                return ZO_ERR_OUTPUT_TOO_SMALL;          /* :777-788 */
                            mid_fill_end(hash4Table, hash8Table, ip, prefixPtr, prefixIdx, ilimit, ilimitIdx);   /* :789-818 */
                            continue;                                    /* :819 */
                        }
                    }
                }
            }
        }
        {   /* short match, 4-byte hash :827-934 */
            const uint32_t h4 = hashMid4Ptr(ip);
            const uint32_t pos4 = hash4Table[h4];
            hash4Table[h4] = ipIndex;                                    /* :830 */
            if (pos4 > 0 && ipIndex - pos4 <= LZ4_DISTANCE_MAX) {        /* :832 */
                if (pos4 >= prefixIdx) {
                    const uint8_t *matchPtr = prefixPtr + (pos4 - prefixIdx);
                    if (matchPtr < ip) {
                        matchLength = (uint32_t)lz4Count(ip, matchPtr, matchlimit);      /* :837 */
                        if (matchLength >= MINMATCH) {
                            matchDistance = ipIndex - pos4;
                            if (ip < mflimit) {                          /* :842 */
                                const uint32_t h8_next = hashMid8Ptr(ip + 1);
                                const uint32_t pos8_next = hash8Table[h8_next];
                                const uint32_t m2Distance = ipIndex + 1 - pos8_next;     /* :845 */
                                if (m2Distance <= LZ4_DISTANCE_MAX && pos8_next >= prefixIdx && pos8_next > 0) {   /* :847 */
                                    const uint8_t *m2Ptr = prefixPtr + (pos8_next - prefixIdx);
                                    if (m2Ptr < ip + 1) {
                                        const size_t ml2 = lz4Count(ip + 1, m2Ptr, matchlimit);
                                        if (ml2 > matchLength) {         /* :851-856 */
                                            hash8Table[h8_next] = ipIndex + 1;
                                            ip += 1;
                                            matchLength = (uint32_t)ml2;
                                            matchDistance = m2Distance;
                                        }
                                    }
                                }
                            }
                            const uint32_t finalIpIndex4 = ipIndex;      /* :873 -- the ORIGINAL index even if ip moved */
                            if (ip + 1 <= ilimit) hash8Table[hashMid8Ptr(ip + 1)] = finalIpIndex4 + 1;   /* :875-877 */
                            if (ip + 2 <= ilimit) hash8Table[hashMid8Ptr(ip + 2)] = finalIpIndex4 + 2;   /* :878-880 */
                            if (ip + 1 <= ilimit) hash4Table[hashMid4Ptr(ip + 1)] = finalIpIndex4 + 1;   /* :881-883 */
                            if (encodeSequence(&ip, &op, &anchor, (int32_t)matchLength, (int32_t)matchDistance, 1, oend) != 0)
                                return ZO_ERR_OUTPUT_TOO_SMALL;          /* :886-897 */
                            mid_fill_end(hash4Table, hash8Table, ip, prefixPtr, prefixIdx, ilimit, ilimitIdx);   /* :899-928 */
                            continue;                                    /* :929 */
                        }
                    }
                }
            }
        }
        const size_t skipAmount = 1 + ((size_t)(ip - anchor) >> 9);      /* :937 */
        ip += skipAmount;
    }
    const size_t finalLiterals = (size_t)(iend - anchor);                /* :942 */
    if (finalLiterals > 0) {
        if ((uintptr_t)op + finalLiterals + 1 > (uintptr_t)oend) return ZO_ERR_OUTPUT_TOO_SMALL;   /* :944 */
        if (finalLiterals >= RUN_MASK) {
            size_t len = finalLiterals - RUN_MASK;
            op[0] = RUN_MASK << ML_BITS; op += 1;
            while (len >= 255) { op[0] = 255; op += 1; len -= 255; }
            op[0] = (uint8_t)len; op += 1;
        } else {
            op[0] = (uint8_t)(finalLiterals << ML_BITS); op += 1;
        }
        memcpy(op, anchor, finalLiterals);
        op += finalLiterals;
    }
    return (int64_t)(op - dst);
}

/* ---------------- optimal parser, levels 10-12: src/lz4hc.zig:455-486, :1068-1391 ---------------- */
#define LZ4_OPT_NUM (1 << 12)
#define TRAILING_LITERALS 3
typedef struct { int32_t price, off, mlen, litlen; } opt_match;          /* :456-461 */
static inline int32_t literalsPrice(int32_t litlen) {                    /* :466-472 */
    int32_t price = litlen;
    if (litlen >= (int32_t)RUN_MASK) price += 1 + (litlen - (int32_t)RUN_MASK) / 255;
    return price;
}
static inline int32_t sequencePrice(int32_t litlen, int32_t mlen) {      /* :476-486 */
    int32_t price = 1 + 2;
    price += literalsPrice(litlen);
    if (mlen >= (int32_t)(ML_MASK + MINMATCH)) price += 1 + (mlen - (int32_t)(ML_MASK + MINMATCH)) / 255;
    return price;
}

static int64_t compressOptimal(hc_ctx *ctx, const uint8_t *src, size_t inputSize, uint8_t *dst, size_t dstLen,
                               int32_t nbSearches, size_t sufficientLen) {
    opt_match *opt = (opt_match *)malloc((LZ4_OPT_NUM + TRAILING_LITERALS) * sizeof(opt_match));   /* :1079 */
    if (!opt) return ZO_ERR_ALLOCATION_FAILED;
    /* `opt` is `undefined` in the reference; the algorithm may read entries it has not written in this round
     * (stale values of an earlier round).  Start from zero like a fresh stack page would most likely be. */
    memset(opt, 0, (LZ4_OPT_NUM + TRAILING_LITERALS) * sizeof(opt_match));
    const uint8_t *ip = src;
    const uint8_t *anchor = ip;
    const uint8_t *const iend = ip + inputSize;
    uint8_t *op = dst;
    uint8_t *const oend = op + dstLen;
    int64_t ret;
    if (inputSize < MFLIMIT + 1) { ret = encodeLiterals(src, inputSize, dst, dstLen); free(opt); return ret; }   /* :1092-1094 */
    const uint8_t *const mflimit = iend - MFLIMIT;
    const uint8_t *const matchlimit = iend - LASTLITERALS;
    ctx->nextToUpdate = 0; ctx->prefixStart = src; ctx->dictStart = src; ctx->dictLimit = 0; ctx->lowLimit = 0;   /* :1097-1102 */
    size_t sufficient_len = sufficientLen;                               /* :1105-1108 */
    if (sufficient_len >= LZ4_OPT_NUM) sufficient_len = LZ4_OPT_NUM - 1;

    while (ip <= mflimit) {                                              /* :1111 outer */
        const int32_t llen = (int32_t)(ip - anchor);
        insertHC(ctx, ip);                                               /* :1115 */
        const hc_match firstMatch = insertAndGetWiderMatch(ctx, ip, ip, matchlimit, MINMATCH - 1, nbSearches, 1);
        if (firstMatch.len == 0) { ip += 1; continue; }                  /* :1127 (never true: len starts at 3) */
        if ((size_t)firstMatch.len > sufficient_len) {                   /* :1133 */
            if (encodeSequence(&ip, &op, &anchor, firstMatch.len, firstMatch.off, 1, oend) != 0) { free(opt); return ZO_ERR_OUTPUT_TOO_SMALL; }
            continue;
        }
        for (size_t rPos = 0; rPos < MINMATCH; rPos++) {                 /* :1150-1157 */
            const int32_t cost = literalsPrice(llen + (int32_t)rPos);
            opt[rPos].mlen = 1; opt[rPos].off = 0; opt[rPos].litlen = llen + (int32_t)rPos; opt[rPos].price = cost;
        }
        const size_t matchML = (size_t)firstMatch.len;                   /* :1160 */
        const int32_t offset = firstMatch.off;
        for (size_t mlen = MINMATCH; mlen <= matchML; mlen++) {          /* :1162-1169 */
            const int32_t cost = sequencePrice(llen, (int32_t)mlen);
            opt[mlen].mlen = (int32_t)mlen; opt[mlen].off = offset; opt[mlen].litlen = llen; opt[mlen].price = cost;
        }
        size_t last_match_pos = matchML;                                 /* :1171 */
        for (size_t addLit = 1; addLit <= TRAILING_LITERALS; addLit++) { /* :1174-1180 */
            opt[last_match_pos + addLit].mlen = 1; opt[last_match_pos + addLit].off = 0;
            opt[last_match_pos + addLit].litlen = (int32_t)addLit;
            opt[last_match_pos + addLit].price = opt[last_match_pos].price + literalsPrice((int32_t)addLit);
        }
        int encoded_early = 0;
        size_t cur;
        for (cur = 1; cur < last_match_pos; cur++) {                     /* :1183-1312 */
            const uint8_t *curPtr = ip + cur;
            if (curPtr > mflimit) break;                                 /* :1187 */
            if (opt[cur + 1].price <= opt[cur].price) continue;          /* :1190 */
            insertHC(ctx, curPtr);                                       /* :1193 */
            const hc_match newMatch = insertAndGetWiderMatch(ctx, curPtr, curPtr, matchlimit, MINMATCH - 1, nbSearches, 1);
            if (newMatch.len == 0) continue;                             /* :1205 */
            if (((size_t)newMatch.len > sufficient_len) || (newMatch.len + (int32_t)cur >= (int32_t)LZ4_OPT_NUM)) {   /* :1208 */
                const int32_t best_mlen = newMatch.len, best_off = newMatch.off;
                size_t rp = 0;
                while (rp < cur) {                                       /* :1217-1238 */
                    const int32_t ml = opt[rp].mlen, off = opt[rp].off;
                    if (ml == 1) { ip += 1; rp += 1; continue; }
                    rp += (size_t)ml;
                    if (encodeSequence(&ip, &op, &anchor, ml, off, 1, oend) != 0) { free(opt); return ZO_ERR_OUTPUT_TOO_SMALL; }
                }
                if (encodeSequence(&ip, &op, &anchor, best_mlen, best_off, 1, oend) != 0) { free(opt); return ZO_ERR_OUTPUT_TOO_SMALL; }
                encoded_early = 1;                                       /* :1255 continue :outer */
                break;
            }
            const int32_t baseLitlen = opt[cur].litlen;                  /* :1259 */
            for (size_t litlen = 1; litlen < MINMATCH; litlen++) {       /* :1260-1270 */
                const int32_t price = opt[cur].price - literalsPrice(baseLitlen) + literalsPrice(baseLitlen + (int32_t)litlen);
                const size_t pos = cur + litlen;
                if (price < opt[pos].price) {
                    opt[pos].mlen = 1; opt[pos].off = 0; opt[pos].litlen = baseLitlen + (int32_t)litlen; opt[pos].price = price;
                }
            }
            const size_t newMatchML = (size_t)newMatch.len;              /* :1273 */
            for (size_t ml = MINMATCH; ml <= newMatchML; ml++) {         /* :1274-1302 */
                const size_t pos = cur + ml;
                const int32_t newOffset = newMatch.off;
                int32_t price, ll;
                if (opt[cur].mlen == 1) {
                    ll = opt[cur].litlen;
                    price = (cur > (size_t)ll) ? opt[cur - (size_t)ll].price : 0;
                    price += sequencePrice(ll, (int32_t)ml);
                } else {
                    ll = 0;
                    price = opt[cur].price + sequencePrice(0, (int32_t)ml);
                }
                if (pos > last_match_pos + TRAILING_LITERALS || price <= opt[pos].price) {   /* :1293 */
                    if ((ml == newMatchML) && (last_match_pos < pos)) last_match_pos = pos;
                    opt[pos].mlen = (int32_t)ml; opt[pos].off = newOffset; opt[pos].litlen = ll; opt[pos].price = price;
                }
            }
            for (size_t addLit = 1; addLit <= TRAILING_LITERALS; addLit++) {     /* :1305-1311 */
                opt[last_match_pos + addLit].mlen = 1; opt[last_match_pos + addLit].off = 0;
                opt[last_match_pos + addLit].litlen = (int32_t)addLit;
                opt[last_match_pos + addLit].price = opt[last_match_pos].price + literalsPrice((int32_t)addLit);
            }
        }
        if (encoded_early) continue;
        {   /* backtrack :1315-1332 */
            const int32_t best_mlen = opt[last_match_pos].mlen, best_off = opt[last_match_pos].off;
            cur = last_match_pos - (size_t)best_mlen;
            size_t candidate_pos = cur;
            int32_t selected_matchLength = best_mlen, selected_offset = best_off;
            for (;;) {
                const int32_t next_matchLength = opt[candidate_pos].mlen, next_offset = opt[candidate_pos].off;
                opt[candidate_pos].mlen = selected_matchLength;
                opt[candidate_pos].off = selected_offset;
                selected_matchLength = next_matchLength;
                selected_offset = next_offset;
                if (next_matchLength > (int32_t)candidate_pos) break;
                candidate_pos -= (size_t)next_matchLength;
            }
        }
        {   /* encode :1335-1358 */
            size_t rPos = 0;
            while (rPos < last_match_pos) {
                const int32_t ml = opt[rPos].mlen, off = opt[rPos].off;
                if (ml == 1) { ip += 1; rPos += 1; continue; }
                rPos += (size_t)ml;
                if (encodeSequence(&ip, &op, &anchor, ml, off, 1, oend) != 0) { free(opt); return ZO_ERR_OUTPUT_TOO_SMALL; }
            }
        }
    }
    free(opt);
    if (anchor > iend) { zo_hc_reference_ub_count += 1; return ZO_ERR_OUTPUT_TOO_SMALL; }   /* same defect: iend - anchor underflows */
    const size_t finalLiterals = (size_t)(iend - anchor);                /* :1362 */
    if (finalLiterals > 0) {
        if ((uintptr_t)op + finalLiterals + 1 > (uintptr_t)oend) return ZO_ERR_OUTPUT_TOO_SMALL;   /* :1364 */
        if (finalLiterals >= RUN_MASK) {
            size_t len = finalLiterals - RUN_MASK;
            op[0] = RUN_MASK << ML_BITS; op += 1;
            while (len >= 255) { op[0] = 255; op += 1; len -= 255; }
            op[0] = (uint8_t)len; op += 1;
        } else {
            op[0] = (uint8_t)(finalLiterals << ML_BITS); op += 1;
        }
        memcpy(op, anchor, finalLiterals);
        op += finalLiterals;
    }
    return (int64_t)(op - dst);
}

/* src/lz4hc.zig:1440-1453 + :1457-1489 */
#define ZO_UNSUPPORTED_LEVEL (-1005)
int64_t zo_compress_hc(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, int32_t compressionLevel) {
    if (n > LZ4_MAX_INPUT_SIZE) return ZO_ERR_INPUT_TOO_LARGE;           /* :1442 */
    if (n == 0) return 0;                                                /* :1443 */
    int32_t level = compressionLevel < LZ4HC_CLEVEL_MIN ? LZ4HC_CLEVEL_DEFAULT
                  : compressionLevel > LZ4HC_CLEVEL_MAX ? LZ4HC_CLEVEL_MAX : compressionLevel;   /* :1445 */
    /* compressHCExtState :1457-1489 */
    if (cap == 0) return ZO_ERR_OUTPUT_TOO_SMALL;                        /* :1461 */
    if (level < 1) level = LZ4HC_CLEVEL_DEFAULT;                         /* :1465 */
    if (level > LZ4HC_CLEVEL_MAX) level = LZ4HC_CLEVEL_MAX;
    const clevel_params params = clevelTable[level];                     /* :1469, :88-97 */
    hc_ctx *ctx = (hc_ctx *)calloc(1, sizeof(hc_ctx));                   /* Context.init() :405-419: zero tables */
    if (!ctx) return ZO_ERR_ALLOCATION_FAILED;
    int64_t r;
    switch (params.strat) {                                              /* :1475-1488 */
        case STRAT_HC:  r = compressHashChain(ctx, src, n, dst, cap, params.nbSearches); break;
        case STRAT_MID: r = compressMID(ctx, src, n, dst, cap); break;
        default:        r = compressOptimal(ctx, src, n, dst, cap, params.nbSearches, params.targetLength); break;
    }
    free(ctx);
    return r;
}

/* ===================== XXH32 (Zig std.hash.XxHash32 == standard XXH32) ===================== */
#define XP1 2654435761u
#define XP2 2246822519u
#define XP3 3266489917u
#define XP4 668265263u
#define XP5 374761393u
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static inline uint32_t xround(uint32_t acc, uint32_t in) { return rotl32(acc + in * XP2, 13) * XP1; }

uint32_t zo_xxh32(const uint8_t *p, size_t len, uint32_t seed) {
    const uint8_t *const end = p + len;
    uint32_t h;
    if (len >= 16) {
        uint32_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        const uint8_t *const limit = end - 16;
        do {
            v1 = xround(v1, rd32(p)); v2 = xround(v2, rd32(p + 4));
            v3 = xround(v3, rd32(p + 8)); v4 = xround(v4, rd32(p + 12));
            p += 16;
        } while (p <= limit);
        h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else {
        h = seed + XP5;
    }
    h += (uint32_t)len;
    while (p + 4 <= end) { h = rotl32(h + rd32(p) * XP3, 17) * XP4; p += 4; }
    while (p < end) { h = rotl32(h + (*p) * XP5, 11) * XP1; p += 1; }
    h ^= h >> 15; h *= XP2; h ^= h >> 13; h *= XP3; h ^= h >> 16;
    return h;
}

/* ===================== frame, src/lz4f.zig ===================== */
#define MAGICNUMBER 0x184D2204u
#define MAGIC_SKIPPABLE_START 0x184D2A50u
#define MAGIC_SKIPPABLE_MASK 0xFFFFFFF0u
#define HEADER_SIZE_MIN 7
#define HEADER_SIZE_MAX 19

static const zo_prefs default_prefs = { 0, 0, 0, 0, 0, 0, 0 };   /* lz4f.zig:106-122 defaults */

/* lz4f.zig:71-78; an id outside {0,4..7} cannot be built in Zig (enum) -> treat as 64 KiB like the `catch` at :276 */
static size_t block_size_of(uint32_t id) {
    switch (id) {
        case 5: return 256u * 1024;
        case 6: return 1024u * 1024;
        case 7: return 4u * 1024 * 1024;
        default: return 64u * 1024;
    }
}
/* lz4f.zig:138-141 */
static uint8_t headerChecksum(const uint8_t *p, size_t n) { return (uint8_t)((zo_xxh32(p, n, 0) >> 8) & 0xFF); }
/* lz4f.zig:152-184 */
static uint8_t encodeFLG(const zo_prefs *p) {
    uint8_t flg = 0x40;
    if (p->block_mode == 1) flg |= 0x20;
    if (p->block_checksum == 1) flg |= 0x10;
    if (p->content_size != 0) flg |= 0x08;
    if (p->content_checksum == 1) flg |= 0x04;
    if (p->dict_id != 0) flg |= 0x01;
    return flg;
}
/* lz4f.zig:224-232 */
static uint8_t encodeBD(uint32_t id) {
    uint8_t v = 4;
    if (id == 5) v = 5; else if (id == 6) v = 6; else if (id == 7) v = 7;
    return (uint8_t)(v << 4);
}

/* lz4f.zig:274-301 */
size_t zo_compress_frame_bound(size_t srcSize, const zo_prefs *prefs) {
    const zo_prefs *p = prefs ? prefs : &default_prefs;
    const size_t blockSize = block_size_of(p->block_size_id);
    size_t result = HEADER_SIZE_MAX;
    const size_t numBlocks = (srcSize + blockSize - 1) / blockSize;
    for (size_t i = 0; i < numBlocks; i++) {
        result += 4;
        result += zo_compress_bound(blockSize);
        if (p->block_checksum == 1) result += 4;
    }
    result += 4;
    if (p->content_checksum == 1) result += 4;
    return result;
}

/* lz4f.zig:304-351 */
static int64_t writeFrameHeader(uint8_t *dst, size_t dstLen, const zo_prefs *p) {
    if (dstLen < HEADER_SIZE_MIN) return ZOF_ERR_DST_MAX_SIZE_TOO_SMALL;
    size_t pos = 0;
    wr32(dst + pos, MAGICNUMBER); pos += 4;
    dst[pos] = encodeFLG(p); pos += 1;
    dst[pos] = encodeBD(p->block_size_id); pos += 1;
    const size_t headerStart = 4;
    if (p->content_size != 0) {
        if (dstLen < pos + 8) return ZOF_ERR_DST_MAX_SIZE_TOO_SMALL;
        wr64(dst + pos, p->content_size); pos += 8;
    }
    if (p->dict_id != 0) {
        if (dstLen < pos + 4) return ZOF_ERR_DST_MAX_SIZE_TOO_SMALL;
        wr32(dst + pos, p->dict_id); pos += 4;
    }
    dst[pos] = headerChecksum(dst + headerStart, pos - headerStart);
    pos += 1;
    return (int64_t)pos;
}

/* streaming XXH32 is value-identical to one-shot XXH32 over the concatenation, so the
 * content checksum (lz4f.zig:375, :385, :438) is computed one-shot over src here. */

/* lz4f.zig:144-149 */
static int64_t mapCompressionError(int64_t e) {
    return e == ZO_ERR_OUTPUT_TOO_SMALL ? ZOF_ERR_DST_MAX_SIZE_TOO_SMALL : ZOF_ERR_GENERIC;
}

/* lz4f.zig:354-446 */
int64_t zo_compress_frame(const uint8_t *src, size_t srcLen, uint8_t *dst, size_t dstLen, const zo_prefs *prefs) {
    const zo_prefs *p = prefs ? prefs : &default_prefs;
    const size_t requiredSize = zo_compress_frame_bound(srcLen, p);      /* :363 */
    if (dstLen < requiredSize) return ZOF_ERR_DST_MAX_SIZE_TOO_SMALL;
    int64_t h = writeFrameHeader(dst, dstLen, p);                        /* :369 */
    if (h < 0) return h;
    size_t dstPos = (size_t)h;
    const size_t blockSize = block_size_of(p->block_size_id);            /* :372 */
    size_t srcPos = 0;
    while (srcPos < srcLen) {                                            /* :379 */
        const size_t blockLen = (srcLen - srcPos) < blockSize ? (srcLen - srcPos) : blockSize;
        const uint8_t *srcBlock = src + srcPos;
        const size_t blockStart = dstPos + 4;                            /* :389 */
        uint8_t *dstBlock = dst + blockStart;
        const size_t dstBlockLen = dstLen - blockStart;
        int64_t compressedSize;
        if (p->compression_level > 0)                                    /* :393 */
            compressedSize = zo_compress_hc(srcBlock, blockLen, dstBlock, dstBlockLen, p->compression_level);
        else
            compressedSize = zo_compress_fast(srcBlock, blockLen, dstBlock, dstBlockLen, 1);
        if (compressedSize < 0) return mapCompressionError(compressedSize);
        const int storeUncompressed = (size_t)compressedSize >= blockLen;   /* :407 */
        const size_t actualSize = storeUncompressed ? blockLen : (size_t)compressedSize;
        uint32_t blockHeader = (uint32_t)actualSize;
        if (storeUncompressed) {
            blockHeader |= 0x80000000u;
            memcpy(dst + blockStart, srcBlock, blockLen);                /* :416 */
        }
        wr32(dst + dstPos, blockHeader);                                 /* :418 */
        dstPos = blockStart + actualSize;
        if (p->block_checksum == 1) {                                    /* :422 */
            wr32(dst + dstPos, zo_xxh32(dst + blockStart, actualSize, 0));
            dstPos += 4;
        }
        srcPos += blockLen;
    }
    wr32(dst + dstPos, 0);                                               /* :433 */
    dstPos += 4;
    if (p->content_checksum == 1) {                                      /* :437 */
        wr32(dst + dstPos, zo_xxh32(src, srcLen, 0));
        dstPos += 4;
    }
    return (int64_t)dstPos;
}

/* lz4f.zig:451-480 */
int64_t zo_header_size(const uint8_t *src, size_t n) {
    if (n < 5) return ZOF_ERR_FRAME_HEADER_INCOMPLETE;
    const uint32_t magic = rd32(src);
    if (magic != MAGICNUMBER) {
        if ((magic & MAGIC_SKIPPABLE_MASK) == MAGIC_SKIPPABLE_START) return 8;
        return ZOF_ERR_FRAME_TYPE_UNKNOWN;
    }
    const uint8_t flg = src[4];
    int64_t size = 7;
    if (flg & 0x08) size += 8;
    if (flg & 0x01) size += 4;
    return size;
}

/* lz4f.zig:483-538; returns header size or error; fills *flg_out / *block_size_out */
static int64_t parseFrameHeader(const uint8_t *src, size_t srcLen, uint8_t *flg_out, size_t *block_size_out) {
    if (srcLen < HEADER_SIZE_MIN) return ZOF_ERR_FRAME_HEADER_INCOMPLETE;
    if (rd32(src) != MAGICNUMBER) return ZOF_ERR_FRAME_TYPE_UNKNOWN;
    size_t pos = 4;
    const uint8_t flg = src[pos];
    if (((flg >> 6) & 3) != 1) return ZOF_ERR_HEADER_VERSION_WRONG;      /* decodeFLG :190-194 */
    if (flg & 0x02) return ZOF_ERR_RESERVED_FLAG_SET;                    /* :197-199 */
    pos += 1;
    const uint8_t bd = src[pos];
    if (bd & 0x8F) return ZOF_ERR_RESERVED_FLAG_SET;                     /* decodeBD :237-239 */
    const uint32_t bsv = (bd >> 4) & 7;
    size_t blockSize;
    switch (bsv) {                                                       /* :242-248 */
        case 0: case 4: blockSize = 64u * 1024; break;
        case 5: blockSize = 256u * 1024; break;
        case 6: blockSize = 1024u * 1024; break;
        case 7: blockSize = 4u * 1024 * 1024; break;
        default: return ZOF_ERR_MAX_BLOCK_SIZE_INVALID;
    }
    pos += 1;
    const size_t headerStart = 4;
    if (flg & 0x08) { if (srcLen < pos + 8) return ZOF_ERR_FRAME_HEADER_INCOMPLETE; pos += 8; }
    if (flg & 0x01) { if (srcLen < pos + 4) return ZOF_ERR_FRAME_HEADER_INCOMPLETE; pos += 4; }
    if (srcLen < pos + 1) return ZOF_ERR_FRAME_HEADER_INCOMPLETE;
    if (src[pos] != headerChecksum(src + headerStart, pos - headerStart)) return ZOF_ERR_HEADER_CHECKSUM_INVALID;
    pos += 1;
    *flg_out = flg;
    *block_size_out = blockSize;
    return (int64_t)pos;
}

/* lz4f.zig:541-638 */
int64_t zo_decompress_frame(const uint8_t *src, size_t srcLen, uint8_t *dst, size_t dstLen) {
    uint8_t flg; size_t blockSize;
    const int64_t hs = parseFrameHeader(src, srcLen, &flg, &blockSize);
    if (hs < 0) return hs;
    size_t srcPos = (size_t)hs, dstPos = 0;
    (void)blockSize;                              /* :556-557 allocates and frees an unused buffer */
    const int blockChecksum = (flg & 0x10) != 0, contentChecksum = (flg & 0x04) != 0;

    while (srcPos < srcLen) {                                            /* :563 */
        if (srcPos + 4 > srcLen) return ZOF_ERR_FRAME_SIZE_WRONG;
        const uint32_t blockHeader = rd32(src + srcPos);
        srcPos += 4;
        if (blockHeader == 0) break;                                     /* :573 */
        const int isUncompressed = (blockHeader & 0x80000000u) != 0;
        const size_t blockDataSize = blockHeader & 0x7FFFFFFFu;
        if (srcPos + blockDataSize > srcLen) return ZOF_ERR_FRAME_SIZE_WRONG;   /* :582 */
        const uint8_t *blockData = src + srcPos;
        srcPos += blockDataSize;
        if (blockChecksum) {                                             /* :590 */
            if (srcPos + 4 > srcLen) return ZOF_ERR_FRAME_SIZE_WRONG;
            if (rd32(src + srcPos) != zo_xxh32(blockData, blockDataSize, 0)) return ZOF_ERR_BLOCK_CHECKSUM_INVALID;
            srcPos += 4;
        }
        size_t decompressedSize;
        if (isUncompressed) {                                            /* :603 */
            if (dstPos + blockDataSize > dstLen) return ZOF_ERR_DST_MAX_SIZE_TOO_SMALL;
            memcpy(dst + dstPos, blockData, blockDataSize);
            decompressedSize = blockDataSize;
        } else {
            const int64_t r = zo_decompress_safe(blockData, blockDataSize, dst + dstPos, dstLen - dstPos);   /* :610 */
            if (r < 0) return ZOF_ERR_DECOMPRESSION_FAILED;
            decompressedSize = (size_t)r;
        }
        dstPos += decompressedSize;                                      /* :621 */
    }
    if (contentChecksum) {                                               /* :625 */
        if (srcPos + 4 > srcLen) return ZOF_ERR_FRAME_SIZE_WRONG;
        if (rd32(src + srcPos) != zo_xxh32(dst, dstPos, 0)) return ZOF_ERR_CONTENT_CHECKSUM_INVALID;
        srcPos += 4;
    }
    return (int64_t)dstPos;
}

/* ===================== batch helpers (cpu_baseline leg of bench.py) ===================== */
int64_t zo_batch_compress_default(const uint8_t *in, size_t blk, size_t nblk, uint8_t *out, size_t slot, int64_t *sizes) {
    int64_t total = 0;
    for (size_t i = 0; i < nblk; i++) {
        sizes[i] = zo_compress_fast(in + i * blk, blk, out + i * slot, slot, 1);
        if (sizes[i] > 0) total += sizes[i];
    }
    return total;
}
int64_t zo_batch_compress_hc(const uint8_t *in, size_t blk, size_t nblk, uint8_t *out, size_t slot, int64_t *sizes, int32_t level) {
    int64_t total = 0;
    for (size_t i = 0; i < nblk; i++) {
        sizes[i] = zo_compress_hc(in + i * blk, blk, out + i * slot, slot, level);
        if (sizes[i] > 0) total += sizes[i];
    }
    return total;
}
int64_t zo_batch_decompress_safe(const uint8_t *in, size_t slot, const int64_t *csizes, size_t nblk, uint8_t *out, size_t blk, int64_t *sizes) {
    int64_t total = 0;
    for (size_t i = 0; i < nblk; i++) {
        sizes[i] = zo_decompress_safe(in + i * slot, (size_t)csizes[i], out + i * blk, blk);
        if (sizes[i] > 0) total += sizes[i];
    }
    return total;
}
