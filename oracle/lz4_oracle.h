/*
 * oracle/lz4_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the jedisct1/zig-lz4 block codec and frame
 * wrapper, used as the parity checker for the HIP path and as the
 * "cpu_baseline" leg of bench.py.  Nothing in the product path
 * (zig-lz4_amd/, include/) may include, link or call this.
 *
 * Parity status: the reference is Zig and no zig toolchain exists in the
 * build container, so oracle/_ref cannot be built.  The
 * reference's tests hold NO golden compressed vectors (only round-trip,
 * interop and size-inequality assertions).  The oracle is therefore pinned by
 *   (1) every round-trip / inequality / interop assertion of the reference's
 *       own tests for this path (tests/test_oracle_reference_cases.py),
 *   (2) the hand-traced known-answer vectors of SURVEY.md Appendix B,
 *   (3) liblz4 / lz4 CLI decoding every oracle output back to its input.
 * Compressed-BYTE parity beyond (2) is "parity unpinned": it rests on this
 * file following the Zig source statement by statement.
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef LZ4_ORACLE_H
#define LZ4_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* lz4.Error, src/lz4.zig:48-55, as negative codes in declaration order */
#define ZO_ERR_OUTPUT_TOO_SMALL      (-1)
#define ZO_ERR_INPUT_TOO_LARGE       (-2)
#define ZO_ERR_CORRUPTED_DATA        (-3)
#define ZO_ERR_DECOMPRESSION_FAILED  (-4)
#define ZO_ERR_INVALID_STATE         (-5)
#define ZO_ERR_ALLOCATION_FAILED     (-6)

/* lz4f.Error, src/lz4f.zig:31-55, as -(100 + 1-based declaration index) */
#define ZOF_ERR_GENERIC                  (-101)
#define ZOF_ERR_MAX_BLOCK_SIZE_INVALID   (-102)
#define ZOF_ERR_BLOCK_MODE_INVALID       (-103)
#define ZOF_ERR_PARAMETER_INVALID        (-104)
#define ZOF_ERR_COMPRESSION_LEVEL_INVALID (-105)
#define ZOF_ERR_HEADER_VERSION_WRONG     (-106)
#define ZOF_ERR_BLOCK_CHECKSUM_INVALID   (-107)
#define ZOF_ERR_RESERVED_FLAG_SET        (-108)
#define ZOF_ERR_ALLOCATION_FAILED        (-109)
#define ZOF_ERR_SRC_SIZE_TOO_LARGE       (-110)
#define ZOF_ERR_DST_MAX_SIZE_TOO_SMALL   (-111)
#define ZOF_ERR_FRAME_HEADER_INCOMPLETE  (-112)
#define ZOF_ERR_FRAME_TYPE_UNKNOWN       (-113)
#define ZOF_ERR_FRAME_SIZE_WRONG         (-114)
#define ZOF_ERR_SRC_PTR_WRONG            (-115)
#define ZOF_ERR_DECOMPRESSION_FAILED     (-116)
#define ZOF_ERR_HEADER_CHECKSUM_INVALID  (-117)
#define ZOF_ERR_CONTENT_CHECKSUM_INVALID (-118)

/* src/lz4f.zig:106-122 (FrameInfo + Preferences), flattened */
typedef struct zo_prefs {
    uint32_t block_size_id;     /* 0 (default), 4, 5, 6, 7   lz4f.zig:64-69 */
    uint32_t block_mode;        /* 0 linked, 1 independent   lz4f.zig:82-85 */
    uint32_t content_checksum;  /* 0/1                       lz4f.zig:88-91 */
    uint32_t block_checksum;    /* 0/1                       lz4f.zig:94-97 */
    uint64_t content_size;      /* 0 = unknown               lz4f.zig:111   */
    uint32_t dict_id;           /* 0 = none                  lz4f.zig:112   */
    int32_t  compression_level; /* 0 = fast                  lz4f.zig:119   */
} zo_prefs;

size_t  zo_compress_bound(size_t n);                                   /* lz4.zig:80-83   */
int64_t zo_compress_fast(const uint8_t *src, size_t n, uint8_t *dst,
                         size_t cap, uint32_t accel);                  /* lz4.zig:292-447 */
int64_t zo_compress_default(const uint8_t *src, size_t n, uint8_t *dst,
                            size_t cap);                               /* lz4.zig:283-285 */
int64_t zo_decompress_safe(const uint8_t *src, size_t n, uint8_t *dst,
                           size_t cap);                                /* lz4.zig:257-259 */
size_t  zo_sizeof_state(void);                                         /* lz4.zig:524-526 */
int64_t zo_compress_fast_ext_state(size_t state_len, const uint8_t *src, size_t n, uint8_t *dst,
                                   size_t cap, uint32_t accel);        /* lz4.zig:531-546 */
int64_t zo_compress_dest_size(const uint8_t *src, uint8_t *dst, size_t cap,
                              size_t *src_size_inout);                 /* lz4.zig:551-616 */
int64_t zo_decompress_safe_partial(const uint8_t *src, size_t n, uint8_t *dst, size_t cap,
                                   size_t target_output_size);         /* lz4.zig:619-621 */
int64_t zo_compress_hc(const uint8_t *src, size_t n, uint8_t *dst,
                       size_t cap, int32_t level);                     /* lz4hc.zig:1440-1453 */

/* how often compressHC met the reference's u32 underflow at lz4hc.zig:636 since the last reset (inputs > 64 KiB,
 * levels 9-12; the reference has no defined output there, see lz4_oracle.c) */
int64_t zo_hc_reference_ub(int reset);

uint32_t zo_xxh32(const uint8_t *p, size_t n, uint32_t seed);          /* std.hash.XxHash32 */

size_t  zo_compress_frame_bound(size_t n, const zo_prefs *prefs);      /* lz4f.zig:274-301 */
int64_t zo_compress_frame(const uint8_t *src, size_t n, uint8_t *dst,
                          size_t cap, const zo_prefs *prefs);          /* lz4f.zig:354-446 */
int64_t zo_decompress_frame(const uint8_t *src, size_t n, uint8_t *dst,
                            size_t cap);                               /* lz4f.zig:541-638 */
int64_t zo_header_size(const uint8_t *src, size_t n);                  /* lz4f.zig:451-480 */

/* batch helpers for the cpu_baseline leg: blocks of `blk` bytes, slots of `slot` bytes */
int64_t zo_batch_compress_default(const uint8_t *in, size_t blk, size_t nblk,
                                  uint8_t *out, size_t slot, int64_t *sizes);
int64_t zo_batch_compress_hc(const uint8_t *in, size_t blk, size_t nblk,
                             uint8_t *out, size_t slot, int64_t *sizes, int32_t level);
int64_t zo_batch_decompress_safe(const uint8_t *in, size_t slot, const int64_t *csizes,
                                 size_t nblk, uint8_t *out, size_t blk, int64_t *sizes);

#ifdef __cplusplus
}
#endif
#endif
