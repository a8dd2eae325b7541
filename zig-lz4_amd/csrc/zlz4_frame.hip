// zlz4_frame.hip -- LZ4 frame container (reference src/lz4f.zig) around the block kernels.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "../../include/zlz4_amd.h"

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

int64_t zlz4f_compress_frame(const uint8_t *, size_t, uint8_t *, size_t, const zlz4f_prefs *) { return ZLZ4_ERR_UNSUPPORTED; }
int64_t zlz4f_decompress_frame(const uint8_t *, size_t, uint8_t *, size_t) { return ZLZ4_ERR_UNSUPPORTED; }
int64_t zlz4f_compress_frame_device(void *, const uint8_t *, size_t, uint8_t *, size_t, const zlz4f_prefs *) { return ZLZ4_ERR_UNSUPPORTED; }
int64_t zlz4f_decompress_frame_device(void *, const uint8_t *, size_t, uint8_t *, size_t) { return ZLZ4_ERR_UNSUPPORTED; }

}  // extern "C"
