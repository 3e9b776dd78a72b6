// tests/fake_hip/hip/hip_runtime.h — a HOST-ONLY stand-in for the HIP runtime calls that csrc/sdrk_api.hip makes, so that
// the library's host side (plans, the four-slot pinned pipeline of sdrk_exec_host, the pinned-range table, the waterfall
// ring's two-stream read-out, the placement probes) can be built with g++ and run under ThreadSanitizer and
// AddressSanitizer + UBSan in the CPU suite (tests/test_host_sanitizers.py; GPU-side sanitizers do not exist on this pool).
//
// Test infrastructure only: nothing under sdr-iq-visualizer_amd/ includes or links it, and it computes no spectrum — the
// kernels are replaced by tests/fake_hip/fake_kernels.cpp, whose "transform" is a checkable function of its input.
//
// What is modelled faithfully, because the host code's correctness depends on it:
//   * every stream is its own thread with a FIFO of operations, so asynchronous copies and "kernels" really do run
//     concurrently with the calling thread and with each other (TSan sees a missing event or a slot reused too early);
//   * events: hipEventRecord takes a ticket that the stream completes in order; hipStreamWaitEvent / hipEventSynchronize
//     wait for the ticket current at the time of THEIR call (HIP's semantics);
//   * device and pinned memory are plain malloc blocks, so ASan checks every "device" access of the host pipeline's
//     bookkeeping (sizes, offsets, chunk tails).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)

struct float2 { float x, y; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyHostToHost = 0 };
enum : unsigned { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocPortable = 1,
                  hipHostMallocMapped = 2, hipHostRegisterPortable = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    int multiProcessorCount;
    size_t totalGlobalMem;
    int clockRate;
    int pciDomainID, pciBusID, pciDeviceID;
};

namespace fakehip {

struct Stream {
    std::mutex m;
    std::condition_variable cv, idle;
    std::deque<std::function<void()>> q;
    bool stop = false, busy = false;
    std::thread th;
    Stream() : th([this] { run(); }) {}
    ~Stream() {
        {
            std::lock_guard<std::mutex> g(m);
            stop = true;
        }
        cv.notify_all();
        th.join();
    }
    void run() {
        for (;;) {
            std::function<void()> op;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [this] { return stop || !q.empty(); });
                if (q.empty()) return;
                op = std::move(q.front());
                q.pop_front();
                busy = true;
            }
            op();
            {
                std::lock_guard<std::mutex> g(m);
                busy = false;
                idle.notify_all();
            }
        }
    }
    void push(std::function<void()> op) {
        {
            std::lock_guard<std::mutex> g(m);
            q.push_back(std::move(op));
        }
        cv.notify_one();
    }
    void drain() {
        std::unique_lock<std::mutex> lk(m);
        idle.wait(lk, [this] { return q.empty() && !busy; });
    }
};

struct Event {
    std::mutex m;
    std::condition_variable cv;
    uint64_t recorded = 0, completed = 0;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void wait_for(uint64_t ticket) {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return completed >= ticket; });
    }
};

inline Stream& null_stream() {
    static Stream s;
    return s;
}
inline Stream& of(void* s) { return s ? *static_cast<Stream*>(s) : null_stream(); }

}  // namespace fakehip

typedef fakehip::Stream* hipStream_t;
typedef fakehip::Event* hipEvent_t;

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorOutOfMemory ? "out of memory" : "fake HIP error"); }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
namespace fakehip {
inline int& cus() { static int c = 8; return c; }   // the stand-in device's CU count (a test may raise it: 32 = one whole XCD)
}
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof *p);
    strcpy(p->name, "fake MI355X (host threads)");
    strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = fakehip::cus();
    p->totalGlobalMem = (size_t)64 << 30;
    p->clockRate = 2400000;
    return hipSuccess;
}
inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)48 << 30; *t = (size_t)64 << 30; return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

inline hipError_t hipMalloc(void** p, size_t n) {
    *p = malloc(n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }

inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new fakehip::Stream(); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t s) { fakehip::of(s).drain(); return hipSuccess; }

inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new fakehip::Event(); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    uint64_t ticket;
    {
        std::lock_guard<std::mutex> g(e->m);
        ticket = ++e->recorded;
    }
    fakehip::of(s).push([e, ticket] {
        std::lock_guard<std::mutex> g(e->m);      // notified UNDER the lock: a waiter that returns may destroy the event at once
        e->completed = ticket;
        e->t = std::chrono::steady_clock::now();
        e->cv.notify_all();
    });
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t e) {
    uint64_t ticket;
    {
        std::lock_guard<std::mutex> g(e->m);
        ticket = e->recorded;
    }
    e->wait_for(ticket);
    return hipSuccess;
}
inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    uint64_t ticket;
    {
        std::lock_guard<std::mutex> g(e->m);
        ticket = e->recorded;
    }
    fakehip::of(s).push([e, ticket] { e->wait_for(ticket); });
    return hipSuccess;
}
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    std::chrono::steady_clock::time_point ta, tb;
    {
        std::lock_guard<std::mutex> g(a->m);
        ta = a->t;
    }
    {
        std::lock_guard<std::mutex> g(b->m);
        tb = b->t;
    }
    *ms = std::chrono::duration<float, std::milli>(tb - ta).count();
    if (!(*ms > 0.0f)) *ms = 1e-3f;
    return hipSuccess;
}

inline hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t s) {
    fakehip::of(s).push([dst, src, n] { memcpy(dst, src, n); });
    return hipSuccess;
}
inline hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind) {
    fakehip::null_stream().drain();      // the legacy stream's order; blocking for the host
    memcpy(dst, src, n);
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t s) {
    fakehip::of(s).push([dst, v, n] { memset(dst, v, n); });
    return hipSuccess;
}
inline hipError_t hipMemsetD32(void* dst, int v, size_t count) {
    fakehip::null_stream().drain();
    for (size_t i = 0; i < count; ++i) static_cast<int*>(dst)[i] = v;
    return hipSuccess;
}
inline hipError_t hipStreamWriteValue32(hipStream_t s, void* ptr, uint32_t value, unsigned) {
    fakehip::of(s).push([ptr, value] { __atomic_store_n(static_cast<uint32_t*>(ptr), value, __ATOMIC_RELEASE); });
    return hipSuccess;
}
