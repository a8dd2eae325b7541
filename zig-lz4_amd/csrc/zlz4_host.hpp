// zlz4_host.hpp -- host-side scratch memory shared by the C-ABI entry points (zlz4_capi.hip, zlz4_frame.hip).
//
// hipMalloc / hipFree of a multi-GiB slot arena cost milliseconds and hipFree synchronises the device, so freed
// buffers are parked in a small per-process cache and handed out again.  A buffer may only be parked once the work
// that uses it has finished: every DevBuf belongs to a DeviceCall, and the first DevBuf that dies while the call's
// stream may still be busy (an early error return after kernels were enqueued) synchronises the stream first.  The
// cache is bounded by count AND by bytes; zlz4_release_device_cache() gives the memory back.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace zlz4host {

// implemented in zlz4_frame.hip (one cache per process)
void *cache_take(size_t &n, int dev);              // a parked buffer of >= n bytes on `dev` (n := its size), or nullptr
bool cache_give(void *p, size_t n, int dev);       // park it; false = cache full (caller frees)

struct DeviceCall {
    hipStream_t st;
    bool idle = true;                 // false between the first launch and the stream synchronisation that follows it
    explicit DeviceCall(hipStream_t s) : st(s) {}
    void launched() { idle = false; }
    bool sync() { idle = true; return hipStreamSynchronize(st) == hipSuccess; }
};

struct DevBuf {
    void *p = nullptr;
    size_t n = 0;
    int dev = 0;
    DeviceCall *call = nullptr;
    explicit DevBuf(size_t want, DeviceCall *dc = nullptr) : call(dc) {
        n = want ? want : 1;
        if (hipGetDevice(&dev) != hipSuccess) return;
        p = cache_take(n, dev);
        if (!p && hipMalloc(&p, n) != hipSuccess) p = nullptr;
    }
    ~DevBuf() {
        if (!p) return;
        if (call && !call->idle) (void)call->sync();         // error exit with work in flight: wait before anyone reuses p
        if (!cache_give(p, n, dev)) (void)hipFree(p);
    }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

}  // namespace zlz4host
