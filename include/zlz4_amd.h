/*
 * zlz4_amd.h -- C ABI of the MI355X-native LZ4 block codec (drop-in for the
 * hot path of jedisct1/zig-lz4).  Every entry point below replaces one name
 * that the reference re-exports from src/root.zig; the reference-side binding
 * (Zig `extern "c"` declarations) is shown in INTEGRATION.md.
 *
 * All citations are relative to the reference tree (/root/reference/).
 * Plain pointers and sizes only: no torch / HIP types in the signatures
 * (a HIP stream is passed as `void*`, NULL = the default stream).
 *
 * Result convention (reference: Zig error unions, src/lz4.zig:48-55):
 *   >= 0  number of bytes written
 *   <  0  error, mirroring lz4.Error / lz4f.Error in declaration order.
 * The library is HIP-only: there is NO CPU fallback.  If no gfx950 device is
 * usable every compute entry point returns ZLZ4_ERR_DEVICE.
 */
#ifndef ZLZ4_AMD_H
#define ZLZ4_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- lz4.Error (src/lz4.zig:48-55) ---- */
#define ZLZ4_ERR_OUTPUT_TOO_SMALL      (-1)
#define ZLZ4_ERR_INPUT_TOO_LARGE       (-2)
#define ZLZ4_ERR_CORRUPTED_DATA        (-3)
#define ZLZ4_ERR_DECOMPRESSION_FAILED  (-4)
#define ZLZ4_ERR_INVALID_STATE         (-5)
#define ZLZ4_ERR_ALLOCATION_FAILED     (-6)
/* ---- new in this library ---- */
#define ZLZ4_ERR_DEVICE                (-7)   /* HIP runtime / no gfx950 device / launch failure */
#define ZLZ4_ERR_UNSUPPORTED           (-8)   /* a request the device path cannot serve (content checksum of a split frame) */
#define ZLZ4_ERR_VERIFY                (-9)   /* zlz4_batch_verify: the compressed block does not decode back to its input */

/* ---- lz4f.Error (src/lz4f.zig:31-55): -(100 + 1-based declaration index) ---- */
#define ZLZ4F_ERR_GENERIC                   (-101)
#define ZLZ4F_ERR_MAX_BLOCK_SIZE_INVALID    (-102)
#define ZLZ4F_ERR_BLOCK_MODE_INVALID        (-103)
#define ZLZ4F_ERR_PARAMETER_INVALID         (-104)
#define ZLZ4F_ERR_COMPRESSION_LEVEL_INVALID (-105)
#define ZLZ4F_ERR_HEADER_VERSION_WRONG      (-106)
#define ZLZ4F_ERR_BLOCK_CHECKSUM_INVALID    (-107)
#define ZLZ4F_ERR_RESERVED_FLAG_SET         (-108)
#define ZLZ4F_ERR_ALLOCATION_FAILED         (-109)
#define ZLZ4F_ERR_SRC_SIZE_TOO_LARGE        (-110)
#define ZLZ4F_ERR_DST_MAX_SIZE_TOO_SMALL    (-111)
#define ZLZ4F_ERR_FRAME_HEADER_INCOMPLETE   (-112)
#define ZLZ4F_ERR_FRAME_TYPE_UNKNOWN        (-113)
#define ZLZ4F_ERR_FRAME_SIZE_WRONG          (-114)
#define ZLZ4F_ERR_SRC_PTR_WRONG             (-115)
#define ZLZ4F_ERR_DECOMPRESSION_FAILED      (-116)
#define ZLZ4F_ERR_HEADER_CHECKSUM_INVALID   (-117)
#define ZLZ4F_ERR_CONTENT_CHECKSUM_INVALID  (-118)

/* ---- constants re-exported by src/root.zig:46-49 / src/lz4.zig:12-25 / src/lz4hc.zig:28-31 ---- */
#define ZLZ4_MINMATCH            4
#define ZLZ4_MAX_INPUT_SIZE      0x7E000000u
#define ZLZ4_DISTANCE_MAX        65535u
#define ZLZ4HC_CLEVEL_MIN        2
#define ZLZ4HC_CLEVEL_DEFAULT    9
#define ZLZ4HC_CLEVEL_MAX        12
#define ZLZ4F_MAGICNUMBER        0x184D2204u     /* src/lz4f.zig:12 */

/* ======================================================================
 * 1. Single-buffer entry points, HOST pointers -- the names root.zig binds.
 *    Each stages the buffer to the current HIP device, runs the batch kernel
 *    on one block and copies the result back (PCIe-inclusive).  Re-entrant
 *    like the reference (no global state besides the lazily created device
 *    context).
 *
 *    CORRECT, NOT FAST -- USE THE BATCH CALLS (section 2) OR THE FRAME CALLS
 *    (section 3) FOR THROUGHPUT.  One block is one wavefront's serial chain:
 *    a 64 KiB zlz4_compress_default call takes ~1.9 ms (912 windows of ~2 us,
 *    measured on MI355X; the one-thread host port of the reference needs
 *    0.2 ms), zlz4_decompress_safe ~1.0 ms, whatever the staging costs -- the
 *    greedy parse of ONE block cannot be spread over the chip without changing
 *    its output.  A program that loops over blocks through these calls gets
 *    slower, not faster; hand the whole batch to zlz4_batch_* instead (65 536
 *    blocks per call run at ~94 GiB/s compress / ~410 GiB/s decompress).
 * ====================================================================== */

/* replaces lz4.compressBound, src/lz4.zig:80-83 (pure arithmetic, no device) */
size_t  zlz4_compress_bound(size_t input_size);

/* replaces lz4.compressDefault, src/lz4.zig:283-285 */
int64_t zlz4_compress_default(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap);

/* replaces lz4.compressFast, src/lz4.zig:292-447 (acceleration clamped to [1,65537] as :321) */
int64_t zlz4_compress_fast(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                           uint32_t acceleration);

/* replaces lz4hc.compressHC, src/lz4hc.zig:1440-1453.  Levels <2 -> 9, >12 -> 12 (:1445).
 * All strategies of the level table (:72-86) run on the device: 2 lz4mid, 3..9 lz4hc, 10..12 lz4opt.
 *
 * HAZARD, levels 10..12: the output is the reference's output byte for byte, and the reference's lz4opt has a defect
 * (its "good enough -> encode now" branch, src/lz4hc.zig:1207-1256, reads arrival records as forward steps): the
 * stream it emits does NOT always decode back to the input.  Level 10 loses ordinary text blocks (every 64 KiB block
 * of the benchmark's text), levels 11 and 12 lose some inputs; a few inputs return OutputTooSmall where the reference
 * itself underflows.  Bit-parity with the reference is the contract here, so nothing is "fixed" silently: callers that
 * need their data back should use levels 2..9, or verify by decoding (zlz4_decompress_safe) before they drop the
 * source.  The same applies to zlz4f_compress_frame(_device) with compression_level >= 10 -- add a content checksum
 * there and the decoder will at least report the damage. */
int64_t zlz4_compress_hc(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                         int32_t compression_level);

/* replaces lz4hc.sizeofStateHC, src/lz4hc.zig:1492-1494 (= @sizeOf(Context), tables + scalars) */
size_t  zlz4_sizeof_state_hc(void);

/* replaces lz4hc.compressHCExtState, src/lz4hc.zig:1457-1489, for a context in its initial state (Context.init(),
 * :405-419 -- what compressHC itself passes, :1450): level < 1 -> 9, level 1 -> lz4mid, > 12 -> 12; dst_cap == 0 ->
 * OutputTooSmall.  The device keeps its tables in LDS / its own workspace: `state` is only validated (non-null,
 * >= zlz4_sizeof_state_hc() bytes, else InvalidState) and never read or written, so a context that still holds the
 * tables of an earlier call (the reference would search them, :1001-1006 resets only the index base) is treated as
 * fresh -- carrying history from call to call is the streaming API, which is out of scope. */
int64_t zlz4_compress_hc_ext_state(void *state, size_t state_len, const uint8_t *src, size_t src_len,
                                   uint8_t *dst, size_t dst_cap, int32_t compression_level);

/* replaces lz4.decompressSafe, src/lz4.zig:257-259 (decompressGeneric :89-251, no dict) */
int64_t zlz4_decompress_safe(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap);

/* replaces lz4.decompressSafePartial, src/lz4.zig:619-621: decompressGeneric with targetOutputSize as the
 * output limit (:99, :109) -- i.e. OutputTooSmall as soon as a sequence would pass `target_output_size`. */
int64_t zlz4_decompress_safe_partial(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                                     size_t target_output_size);

/* replaces lz4.sizeofState, src/lz4.zig:524-526 (= @sizeOf(HashTable) = 16384) */
size_t  zlz4_sizeof_state(void);

/* replaces lz4.compressFastExtState, src/lz4.zig:531-546: InvalidState when the caller's state buffer is
 * smaller than sizeofState(), otherwise the output of compressFast (the device keeps its tables in LDS;
 * the state buffer is only validated, never written). */
int64_t zlz4_compress_fast_ext_state(void *state, size_t state_len, const uint8_t *src, size_t src_len,
                                     uint8_t *dst, size_t dst_cap, uint32_t acceleration);

/* replaces lz4.compressDestSize, src/lz4.zig:551-616: largest prefix of src whose compressDefault output fits
 * dst (the reference's binary search, same probes in the same order).  *src_size: in = bytes available,
 * out = bytes consumed.  Returns the compressed size.  dst holds the compression of the consumed prefix (the
 * reference leaves the output of its LAST probe there, which is not always that one). */
int64_t zlz4_compress_dest_size(const uint8_t *src, uint8_t *dst, size_t dst_cap, size_t *src_size);

/* ======================================================================
 * 2. Batch entry points, DEVICE pointers -- the data-parallel hot path.
 *    Block i reads  d_in  + d_in_off[i]  (d_in_len[i] bytes) and writes
 *    d_out + d_out_off[i] (capacity d_out_cap[i]); d_result[i] receives what
 *    the single-buffer call would have returned for that block.  All arrays
 *    live in device memory.  Kernels are enqueued on `stream` (hipStream_t)
 *    and the call returns without synchronising.  Return: 0 or ZLZ4_ERR_*.
 *
 *    PRECONDITION of the compress calls: d_in_len[i] <= max_in_len for every
 *    block.  max_in_len selects the table width (16-bit positions up to
 *    64 KiB blocks) and sizes the HC workspace; a block that is longer gets
 *    d_result[i] = ZLZ4_ERR_INVALID_STATE and is not compressed (nothing is
 *    written outside its own output slot, the other blocks are unaffected).
 * ====================================================================== */
int32_t zlz4_batch_compress_fast(void *stream,
                                 const uint8_t *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
                                 uint8_t *d_out, const uint64_t *d_out_off, const uint32_t *d_out_cap,
                                 int64_t *d_result, uint32_t nblocks, uint32_t max_in_len,
                                 uint32_t acceleration);

int32_t zlz4_batch_decompress_safe(void *stream,
                                   const uint8_t *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
                                   uint8_t *d_out, const uint64_t *d_out_off, const uint32_t *d_out_cap,
                                   int64_t *d_result, uint32_t nblocks);

/* workspace for the HC path: bytes needed for `nblocks` blocks of at most `max_in_len` bytes (448 KiB per 64 KiB block
 * up to 8192 blocks = 3.5 GiB, about 6 GiB at most for large blocks; longer batches are processed in rounds) */
size_t  zlz4_batch_compress_hc_workspace(uint32_t nblocks, uint32_t max_in_len);
int32_t zlz4_batch_compress_hc(void *stream,
                               const uint8_t *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
                               uint8_t *d_out, const uint64_t *d_out_off, const uint32_t *d_out_cap,
                               int64_t *d_result, uint32_t nblocks, uint32_t max_in_len,
                               int32_t compression_level, void *d_workspace, size_t workspace_bytes);

/* Opt-in check for the levels whose output the reference itself does not always get right (10..12, see the HAZARD note
 * at zlz4_compress_hc): decodes every compressed block on the device and compares it with its input.
 *   d_comp_result[i] : what the compress call wrote for block i (size, or a negative code, which is passed through)
 *   d_verify[i]      : receives d_comp_result[i] if the block decodes back to its d_in_len[i] input bytes,
 *                      ZLZ4_ERR_VERIFY if it does not
 * Returns the number of blocks that failed the check (0 = all good), or ZLZ4_ERR_*.  Unlike the calls above this one
 * allocates its scratch memory itself (a decode arena as large as the input) and synchronises `stream` before returning;
 * it is a safety net, not part of the hot path.  No counterpart in the reference. */
int64_t zlz4_batch_verify(void *stream,
                          const uint8_t *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
                          const uint8_t *d_comp, const uint64_t *d_comp_off, const int64_t *d_comp_result,
                          int64_t *d_verify, uint32_t nblocks);

/* ======================================================================
 * 3. Frame container (src/lz4f.zig), HOST pointers.
 * ====================================================================== */
/* src/lz4f.zig:106-122 (FrameInfo + Preferences flattened; NULL = defaults) */
typedef struct zlz4f_prefs {
    uint32_t block_size_id;     /* 0 default, 4 = 64 KiB, 5 = 256 KiB, 6 = 1 MiB, 7 = 4 MiB  (:64-79) */
    uint32_t block_mode;        /* 0 linked, 1 independent (header bit only, :159-161)         */
    uint32_t content_checksum;  /* 0 / 1                                                      */
    uint32_t block_checksum;    /* 0 / 1                                                      */
    uint64_t content_size;      /* 0 = unknown                                                */
    uint32_t dict_id;           /* 0 = none                                                   */
    int32_t  compression_level; /* <= 0 fast (accel 1); > 0 -> compressHC(level)  (:393-404)   */
} zlz4f_prefs;

/* replaces lz4f.compressFrameBound, src/lz4f.zig:274-301 (pure arithmetic) */
size_t  zlz4f_compress_frame_bound(size_t src_size, const zlz4f_prefs *prefs);
/* replaces lz4f.compressFrame, src/lz4f.zig:354-446 (the unused allocator argument is dropped) */
int64_t zlz4f_compress_frame(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                             const zlz4f_prefs *prefs);
/* replaces lz4f.decompressFrame, src/lz4f.zig:541-638 */
int64_t zlz4f_decompress_frame(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap);
/* replaces lz4f.headerSize, src/lz4f.zig:451-480 (pure arithmetic) */
int64_t zlz4f_header_size(const uint8_t *src, size_t src_len);

/* Device-resident frame variants (config 5: per-GPU shard already in HBM).
 * src/dst are DEVICE pointers; the frame header/end-mark bytes and the size
 * prefix sum are produced on the device too.  Synchronises `stream` before
 * returning the frame size. */
int64_t zlz4f_compress_frame_device(void *stream, const uint8_t *d_src, size_t src_len,
                                    uint8_t *d_dst, size_t dst_cap, const zlz4f_prefs *prefs);
int64_t zlz4f_decompress_frame_device(void *stream, const uint8_t *d_src, size_t src_len,
                                      uint8_t *d_dst, size_t dst_cap);

/* One frame over several GPUs (BASELINE configs[4]).  compressFrame's block loop (src/lz4f.zig:379-430) carries no
 * state from block to block, so rank r of G compresses its contiguous range of blocks and the frame is the plain
 * concatenation of the ranks' segments in rank order:
 *     segment = [frame header, if ZLZ4F_SEG_FIRST] [block header, data, block checksum]* [end mark, if ZLZ4F_SEG_LAST]
 * d_src is the rank's byte range of the input; every range but the last must be a multiple of the block size.  The
 * only cross-rank step is the prefix sum of the returned segment sizes (where each segment goes).  A content checksum
 * is one serial XXH32 chain over the whole input (:384-386) and is refused here (ZLZ4_ERR_UNSUPPORTED) unless the
 * segment is the whole frame.  With both flags the call is zlz4f_compress_frame_device. */
#define ZLZ4F_SEG_FIRST 1u
#define ZLZ4F_SEG_LAST  2u
int64_t zlz4f_compress_frame_segment_device(void *stream, const uint8_t *d_src, size_t src_len,
                                            uint8_t *d_dst, size_t dst_cap, const zlz4f_prefs *prefs,
                                            uint32_t segment_flags);
/* The inverse for one rank: decodes the blocks of a segment (src/lz4f.zig:563-621).  A segment without
 * ZLZ4F_SEG_FIRST has no frame header, so `prefs` must carry the frame's block_checksum flag and block_size_id
 * (what rank 0 read from the header); without ZLZ4F_SEG_LAST no end mark is expected. */
int64_t zlz4f_decompress_frame_segment_device(void *stream, const uint8_t *d_src, size_t src_len,
                                              uint8_t *d_dst, size_t dst_cap, const zlz4f_prefs *prefs,
                                              uint32_t segment_flags);

/* ======================================================================
 * 4. Introspection
 * ====================================================================== */
/* 0 if a gfx950 device is usable by this process, else ZLZ4_ERR_DEVICE */
int32_t     zlz4_device_check(void);
const char *zlz4_version_string(void);
/* human-readable name of a ZLZ4_ERR_* / ZLZ4F_ERR_* code (mirrors the Zig error names) */
const char *zlz4_error_name(int64_t code);
/* The frame calls park their device scratch buffers (block slots, descriptors) in a small per-process cache instead of
 * hipFree-ing them; this gives that memory back.  (No counterpart in the reference, which allocates nothing.) */
void        zlz4_release_device_cache(void);

#ifdef __cplusplus
}
#endif
#endif /* ZLZ4_AMD_H */
