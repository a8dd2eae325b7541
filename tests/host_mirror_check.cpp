// Compile-and-link check of the C++ host mirror (zig-lz4_amd/csrc/host/zlz4.hpp) against libzlz4_amd.so.
// Without a GPU it runs the pure-arithmetic entry points only; with one it does a single-block round trip and a
// BATCH round trip through the C ABI's device-pointer calls (the hot path), built with plain g++ (the four HIP
// runtime calls it needs are declared by hand, so no HIP headers / hipcc are involved on the host side).
#include <cstdio>
#include <cstring>
#include <vector>

#include "../zig-lz4_amd/csrc/host/zlz4.hpp"

extern "C" {   // libamdhip64: hipError_t is an int, 0 = success; hipMemcpyKind 1 = H2D, 2 = D2H
int hipMalloc(void **p, size_t n);
int hipFree(void *p);
int hipMemcpy(void *dst, const void *src, size_t n, int kind);
int hipDeviceSynchronize(void);
}

static int batch_round_trip() {
    const uint32_t nblocks = 24, bs = 65536;
    const uint32_t slot = (uint32_t)((zlz4::compressBound(bs) + 15) / 16 * 16);
    std::vector<unsigned char> in((size_t)nblocks * bs), out(in.size());
    for (size_t i = 0; i < in.size(); i++) in[i] = (unsigned char)((i * 2654435761u >> 13) % 23 < 17 ? 'a' + (i >> 3) % 19 : i * 7);
    std::vector<uint64_t> in_off(nblocks), slot_off(nblocks);
    std::vector<uint32_t> in_len(nblocks, bs), slot_cap(nblocks, slot), clen(nblocks);
    for (uint32_t i = 0; i < nblocks; i++) { in_off[i] = (uint64_t)i * bs; slot_off[i] = (uint64_t)i * slot; }
    in_len[nblocks - 1] = 777;                                       // one ragged block
    void *d_in, *d_comp, *d_out, *d_u64a, *d_u64b, *d_u32a, *d_u32b, *d_u32c, *d_res;
    if (hipMalloc(&d_in, in.size()) || hipMalloc(&d_comp, (size_t)nblocks * slot) || hipMalloc(&d_out, out.size()) ||
        hipMalloc(&d_u64a, nblocks * 8) || hipMalloc(&d_u64b, nblocks * 8) || hipMalloc(&d_u32a, nblocks * 4) ||
        hipMalloc(&d_u32b, nblocks * 4) || hipMalloc(&d_u32c, nblocks * 4) || hipMalloc(&d_res, nblocks * 8)) return 20;
    hipMemcpy(d_in, in.data(), in.size(), 1);
    hipMemcpy(d_u64a, in_off.data(), nblocks * 8, 1); hipMemcpy(d_u64b, slot_off.data(), nblocks * 8, 1);
    hipMemcpy(d_u32a, in_len.data(), nblocks * 4, 1); hipMemcpy(d_u32b, slot_cap.data(), nblocks * 4, 1);
    zlz4::device::Blocks c{(const uint8_t *)d_in, (const uint64_t *)d_u64a, (const uint32_t *)d_u32a, (uint8_t *)d_comp,
                           (const uint64_t *)d_u64b, (const uint32_t *)d_u32b, (int64_t *)d_res, nblocks};
    if (!zlz4::device::compressFastBatch(nullptr, c, bs).ok()) return 21;
    hipDeviceSynchronize();
    std::vector<int64_t> res(nblocks);
    hipMemcpy(res.data(), d_res, nblocks * 8, 2);
    size_t total = 0;
    for (uint32_t i = 0; i < nblocks; i++) { if (res[i] <= 0) return 22; clen[i] = (uint32_t)res[i]; total += clen[i]; }
    hipMemcpy(d_u32c, clen.data(), nblocks * 4, 1);
    zlz4::device::Blocks d{(const uint8_t *)d_comp, (const uint64_t *)d_u64b, (const uint32_t *)d_u32c, (uint8_t *)d_out,
                           (const uint64_t *)d_u64a, (const uint32_t *)d_u32a, (int64_t *)d_res, nblocks};
    if (!zlz4::device::decompressSafeBatch(nullptr, d).ok()) return 23;
    hipDeviceSynchronize();
    hipMemcpy(res.data(), d_res, nblocks * 8, 2);
    hipMemcpy(out.data(), d_out, out.size(), 2);
    for (uint32_t i = 0; i < nblocks; i++)
        if (res[i] != (int64_t)in_len[i] || std::memcmp(in.data() + in_off[i], out.data() + in_off[i], in_len[i])) return 24;
    // the same blocks through compressHC level 9 (workspace from the caller)
    const size_t wsb = zlz4::device::compressHCWorkspace(nblocks, bs);
    void *d_ws;
    if (hipMalloc(&d_ws, wsb)) return 25;
    if (!zlz4::device::compressHCBatch(nullptr, c, bs, 9, d_ws, wsb).ok()) return 26;
    hipDeviceSynchronize();
    hipMemcpy(res.data(), d_res, nblocks * 8, 2);
    size_t total_hc = 0;
    for (uint32_t i = 0; i < nblocks; i++) { if (res[i] <= 0) return 27; total_hc += (size_t)res[i]; }
    if (total_hc > total) return 28;
    std::printf("gpu batch round trip ok: %u blocks, %zu -> %zu (fast) / %zu (hc 9)\n", nblocks, in.size(), total, total_hc);
    for (void *p : {d_in, d_comp, d_out, d_u64a, d_u64b, d_u32a, d_u32b, d_u32c, d_res, d_ws}) hipFree(p);
    return 0;
}

int main() {
    if (zlz4::compressBound(65536) != 65809) return 1;
    if (zlz4::lz4f::compressFrameBound(0) != 19 + 4) return 2;
    const unsigned char hdr[] = {0x04, 0x22, 0x4D, 0x18, 0x40, 0x40, 0xC0};
    if (zlz4::lz4f::headerSize(hdr, sizeof hdr).value != 7) return 3;
    if (zlz4::sizeofStateHC() < 262144) return 7;
    if (zlz4_device_check() == 0) {
        std::vector<unsigned char> in(100000), c(zlz4::compressBound(in.size())), out(in.size());
        for (size_t i = 0; i < in.size(); i++) in[i] = (unsigned char)(i % 251 < 200 ? 'a' + i % 7 : i);
        auto r = zlz4::compressDefault(in.data(), in.size(), c.data(), c.size());
        if (!r.ok()) return 4;
        auto d = zlz4::decompressSafe(c.data(), r.value, out.data(), out.size());
        if (!d.ok() || d.value != in.size() || std::memcmp(in.data(), out.data(), in.size())) return 5;
        std::printf("gpu round trip ok: %zu -> %zu\n", in.size(), r.value);
        std::vector<unsigned char> ctx(zlz4::sizeofStateHC());
        auto h1 = zlz4::compressHC(in.data(), in.size(), c.data(), c.size(), 9);
        std::vector<unsigned char> c2(c.size());
        auto h2 = zlz4::compressHCExtState(ctx.data(), ctx.size(), in.data(), in.size(), c2.data(), c2.size(), 9);
        if (!h1.ok() || !h2.ok() || h1.value != h2.value || std::memcmp(c.data(), c2.data(), h1.value)) return 8;
        if (zlz4::compressHCExtState(ctx.data(), 100, in.data(), in.size(), c2.data(), c2.size(), 9).error_name() != "InvalidState") return 9;
        const int b = batch_round_trip();
        if (b) return b;
    } else {
        auto r = zlz4::compressDefault(hdr, sizeof hdr, nullptr, 0);
        if (r.ok() || r.error_name() != "DeviceError") return 6;   // must fail loudly without a device
    }
    std::printf("host mirror ok\n");
    return 0;
}
