// Compile-and-link check of the C++ host mirror (zig-lz4_amd/csrc/host/zlz4.hpp) against libzlz4_amd.so.
// Runs the pure-arithmetic entry points only (no GPU needed); with a GPU it also does one round trip.
#include <cstdio>
#include <cstring>
#include <vector>

#include "../zig-lz4_amd/csrc/host/zlz4.hpp"

int main() {
    if (zlz4::compressBound(65536) != 65809) return 1;
    if (zlz4::lz4f::compressFrameBound(0) != 19 + 4) return 2;
    const unsigned char hdr[] = {0x04, 0x22, 0x4D, 0x18, 0x40, 0x40, 0xC0};
    if (zlz4::lz4f::headerSize(hdr, sizeof hdr).value != 7) return 3;
    if (zlz4_device_check() == 0) {
        std::vector<unsigned char> in(100000), c(zlz4::compressBound(in.size())), out(in.size());
        for (size_t i = 0; i < in.size(); i++) in[i] = (unsigned char)(i % 251 < 200 ? 'a' + i % 7 : i);
        auto r = zlz4::compressDefault(in.data(), in.size(), c.data(), c.size());
        if (!r.ok()) return 4;
        auto d = zlz4::decompressSafe(c.data(), r.value, out.data(), out.size());
        if (!d.ok() || d.value != in.size() || std::memcmp(in.data(), out.data(), in.size())) return 5;
        std::printf("gpu round trip ok: %zu -> %zu\n", in.size(), r.value);
    } else {
        auto r = zlz4::compressDefault(hdr, sizeof hdr, nullptr, 0);
        if (r.ok() || r.error_name() != "DeviceError") return 6;   // must fail loudly without a device
    }
    std::printf("host mirror ok\n");
    return 0;
}
