// Shared host/device helpers for libsamx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define SAMX_OK 0
#define SAMX_ERR_ARG (-1)
#define SAMX_ERR_HIP (-2)
#define SAMX_ERR_WORKSPACE (-3)
#define SAMX_ERR_UNSUPPORTED (-4)

namespace samx {

// Thread-local last-error text, read back through samx_last_error().
char* err_buf();
int fail(int code, const char* fmt, ...);

#define SAMX_CHECK_HIP(expr)                                                              \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess)                                                            \
            return samx::fail(SAMX_ERR_HIP, "%s failed: %s (%s:%d)", #expr,               \
                              hipGetErrorString(e__), __FILE__, __LINE__);                \
    } while (0)

#define SAMX_REQUIRE(cond, ...)                                                           \
    do {                                                                                  \
        if (!(cond)) return samx::fail(SAMX_ERR_ARG, __VA_ARGS__);                        \
    } while (0)

#define SAMX_LAUNCH_CHECK() SAMX_CHECK_HIP(hipGetLastError())

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over a caller-provided workspace (device memory, 256-byte aligned).
struct Arena {
    char* base;
    size_t cap;
    size_t off;
    bool dry;  // dry run: only measure
    Arena(void* p, size_t n) : base((char*)p), cap(n), off(0), dry(p == nullptr) {}
    template <typename T>
    T* take(size_t count) {
        size_t o = align_up(off, 256);
        off = o + count * sizeof(T) + 16;   // 16 readable bytes behind every carve (GemmArgs::overread_ok)
        if (dry) return nullptr;
        return (T*)(base + o);
    }
    bool ok() const { return dry || off <= cap; }
};

// Dynamic LDS above 64 KB is an opt-in per kernel AND per device: remember the largest size granted to a
// kernel on each device (one slot array per call site).
struct LdsGrant {
    size_t granted[64] = {0};
    int ensure(const void* kernel, size_t bytes, size_t floor_bytes = 0) {
        int dev = 0;
        SAMX_CHECK_HIP(hipGetDevice(&dev));
        SAMX_REQUIRE(dev >= 0 && dev < 64, "device ordinal %d out of range", dev);
        if (bytes > granted[dev] && bytes > floor_bytes) {
            SAMX_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            granted[dev] = bytes;
        }
        return SAMX_OK;
    }
};

}  // namespace samx
