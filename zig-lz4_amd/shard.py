"""Sharding of independent LZ4 blocks over the GPUs of one node.

Every block is compressed / decompressed with no state shared with any other block
(reference: fresh HashTable per call src/lz4.zig:307, fresh Context src/lz4hc.zig:1450, the
frame block loop carries no history src/lz4f.zig:379-430), so GPU g of G simply takes the
contiguous block range [g*B/G, (g+1)*B/G).  There is no collective on the data path; the only
cross-rank step when ONE contiguous frame is wanted is an exclusive prefix sum over the per-rank
segment sizes (a handful of integers).
"""


def shard_range(nblocks, rank, world):
    """Contiguous block range [lo, hi) of `rank`."""
    return rank * nblocks // world, (rank + 1) * nblocks // world


def segment_offsets(segment_sizes, header_size):
    """Byte offset of every rank's block segment inside the assembled frame, and the offset of the end mark."""
    offs, pos = [], header_size
    for s in segment_sizes:
        offs.append(pos)
        pos += s
    return offs, pos


def block_segment(block_payloads, block_lens, block_checksum, xxh32):
    """Frame bytes of a run of blocks (src/lz4f.zig:406-427): u32 header (bit 31 = stored raw when the
    compressed size is not smaller than the block), payload, optional XXH32 of the stored bytes.
    block_payloads[i] = (compressed_bytes, raw_bytes)."""
    out = bytearray()
    for (comp, raw), n in zip(block_payloads, block_lens):
        stored = len(comp) >= n
        data = raw if stored else comp
        out += (len(data) | (0x80000000 if stored else 0)).to_bytes(4, "little")
        out += data
        if block_checksum:
            out += xxh32(data).to_bytes(4, "little")
    return bytes(out)
