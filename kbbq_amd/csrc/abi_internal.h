// abi_internal.h -- what the translation units of libkbbq_engine.so share behind the C ABI: the error text of
// kbbq_last_error() and the guard that leaves the caller's current HIP device as it was.
#pragma once
#include <hip/hip_runtime.h>

// records the text (printf style) for kbbq_last_error() on this thread and returns `code` (engine.hip)
int kbbq_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Every ABI entry works on its object's device and leaves the calling thread's current device as it found it
// (a process that drives several GPUs, or torch with another current device, is not disturbed).
struct KbbqDeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit KbbqDeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) err = hipSetDevice(device);
        else if (err == hipSuccess) prev = -1;      // nothing to restore
    }
    ~KbbqDeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    KbbqDeviceGuard(const KbbqDeviceGuard &) = delete;
    KbbqDeviceGuard &operator=(const KbbqDeviceGuard &) = delete;
};
