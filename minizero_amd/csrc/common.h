// Shared declarations of libmzgpu (product).  HIP/gfx950 only; there is no CPU fallback anywhere.
#pragma once
#include "common_host.h"
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>

namespace mz {

#define MZ_HIP(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            mz::setError("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);          \
            return MZ_ERR_DEVICE;                                                                             \
        }                                                                                                     \
    } while (0)

// Kernels with more than 48 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize once PER DEVICE (one process may
// drive several GPUs from several host threads: one worker per device, ref actor/actor_group.cpp:168-187)
struct PerDeviceOnce {
    std::mutex mu;
    unsigned long long done = 0;
    template <class F>
    hipError_t run(F&& f)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) { return e; }
        std::lock_guard<std::mutex> l(mu);
        const unsigned long long bit = 1ull << (dev & 63);
        if (done & bit) { return hipSuccess; }
        e = f();
        if (e == hipSuccess) { done |= bit; }
        return e;
    }
};
#define MZ_LDS_ATTR(kernel, lds)                                                                                              \
    do {                                                                                                                      \
        static mz::PerDeviceOnce once_;                                                                                       \
        if ((lds) > 48 * 1024) {                                                                                              \
            MZ_HIP(once_.run([&]() { return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)); })); \
        }                                                                                                                     \
    } while (0)

// device buffer with size bookkeeping (no exceptions across the ABI: alloc returns false on failure)
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    bool alloc(size_t count)
    {
        free();
        if (count == 0) { return true; }
        if (hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)) != hipSuccess) { p = nullptr; return false; }
        n = count;
        return true;
    }
    bool ensure(size_t count) { return count <= n ? true : alloc(count); }
    void free()
    {
        if (p) { (void)hipFree(p); }
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { free(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// pinned host buffer
template <class T>
struct PinBuf {
    T* p = nullptr;
    size_t n = 0;
    bool alloc(size_t count)
    {
        free();
        if (count == 0) { return true; }
        if (hipHostMalloc(reinterpret_cast<void**>(&p), count * sizeof(T), hipHostMallocDefault) != hipSuccess) { p = nullptr; return false; }
        n = count;
        return true;
    }
    bool ensure(size_t count) { return count <= n ? true : alloc(count); }
    void free()
    {
        if (p) { (void)hipHostFree(p); }
        p = nullptr;
        n = 0;
    }
    ~PinBuf() { free(); }
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
};

} // namespace mz
