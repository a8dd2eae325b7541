// gather_probe.hip -- calibration of rocprofv3's FETCH_SIZE for the access pattern of k_compress_fast's candidate
// gather: one 16-byte load per lane, every lane in a DIFFERENT 128-byte line of a buffer far larger than the
// Infinity Cache (1 GiB), every line touched exactly once per launch.  The known byte count is then
//     lines x 64 B   if an L2 miss of a 16-byte gather is a 64-byte fabric request, or
//     lines x 128 B  if it is a full 128-byte line fill.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/gather_probe.bin tools/gather_probe.hip
// Run:   rocprofv3 --kernel-trace --stats -d gpurun_out/gp_stats -- ./tools/gather_probe.bin
//        rocprofv3 --pmc FETCH_SIZE -d gpurun_out/gp_fetch -- ./tools/gather_probe.bin
//        rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_REQ_sum -d gpurun_out/gp_rdreq -- ./tools/gather_probe.bin
// The program also prints the event-timed rate of every pattern in lines/s (the ceiling for line-granular gathers).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// kOff: byte offset of the 16-byte load inside its 128-byte line
template <int kOff>
__global__ __launch_bounds__(256) void k_gather16(const uint8_t *__restrict__ buf, uint32_t line_mask, uint32_t *__restrict__ sink) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t line = (gid * 2654435761u) & line_mask;              // odd multiplier mod 2^k: a permutation of the lines
    u32x4 v;
    __builtin_memcpy(&v, buf + (size_t)line * 128u + kOff, 16);
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) sink[0] = gid;            // never true for the fill pattern
}

// 4-byte gather, same line permutation
__global__ __launch_bounds__(256) void k_gather4(const uint8_t *__restrict__ buf, uint32_t line_mask, uint32_t *__restrict__ sink) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t line = (gid * 2654435761u) & line_mask;
    uint32_t v;
    __builtin_memcpy(&v, buf + (size_t)line * 128u + 20u, 4);
    if (v == 0x12345u) sink[0] = gid;
}

// the guide's calibration case: coalesced streaming read, 16 B per lane
__global__ __launch_bounds__(256) void k_stream16(const uint8_t *__restrict__ buf, uint32_t *__restrict__ sink) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 v;
    __builtin_memcpy(&v, buf + gid * 16u, 16);
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) sink[0] = (uint32_t)gid;
}

__global__ void k_fill(uint32_t *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (uint32_t)(i * 2246822519u) | 1u;
}

template <typename F>
static void timed(const char *name, uint64_t lines, F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch();                                                          // warm-up
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CHECK(hipEventRecord(e0, 0));
        launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-14s lines %llu  %.3f ms  %.2f G lines/s  (64 B/line: %.2f TB/s, 128 B/line: %.2f TB/s)\n", name,
           (unsigned long long)lines, best, lines / (best * 1e6), lines * 64.0 / (best * 1e9), lines * 128.0 / (best * 1e9));
}

int main() {
    const size_t bytes = 1ull << 30;                                   // 1 GiB = 8 Mi lines of 128 B
    const uint32_t nlines = (uint32_t)(bytes / 128u);
    uint8_t *buf;
    uint32_t *sink;
    CHECK(hipMalloc(&buf, bytes + 256));
    CHECK(hipMalloc(&sink, 64));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t *)buf, bytes / 4);
    CHECK(hipDeviceSynchronize());
    const uint32_t threads = nlines;                                   // every line exactly once
    const dim3 grid(threads / 256), block(256);
    timed("gather16@0", nlines, [&] { hipLaunchKernelGGL(k_gather16<0>, grid, block, 0, 0, buf, nlines - 1, sink); });
    timed("gather16@80", nlines, [&] { hipLaunchKernelGGL(k_gather16<80>, grid, block, 0, 0, buf, nlines - 1, sink); });
    timed("gather16@56", nlines, [&] { hipLaunchKernelGGL(k_gather16<56>, grid, block, 0, 0, buf, nlines - 1, sink); });
    timed("gather16@37", nlines, [&] { hipLaunchKernelGGL(k_gather16<37>, grid, block, 0, 0, buf, nlines - 1, sink); });
    timed("gather4@20", nlines, [&] { hipLaunchKernelGGL(k_gather4, grid, block, 0, 0, buf, nlines - 1, sink); });
    timed("stream16", bytes / 128, [&] { hipLaunchKernelGGL(k_stream16, dim3((uint32_t)(bytes / 16 / 256)), block, 0, 0, buf, sink); });
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
