// Host-side helpers for the C ABI: error plumbing, stream cast, device buffers.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <mutex>
#include <stdexcept>
#include <cstdio>
#include "../../include/cosyvoice_amd.h"

namespace cv {

void set_last_error(const std::string& s);

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

#define CV_CHECK(cond, msg) do { if (!(cond)) throw cv::Error(std::string(msg) + " [" #cond "] at " __FILE__ ":" + std::to_string(__LINE__)); } while (0)
#define CV_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) throw cv::Error(std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

// wraps an entry point body: C++ exceptions never cross the C ABI
template <typename F>
static inline int guarded(F&& f) {
    try { f(); return 0; }
    catch (const std::exception& e) { set_last_error(e.what()); return 1; }
    catch (...) { set_last_error("unknown error"); return 1; }
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Process-wide lock serialising the two runtime operations that must never overlap across host threads: stream capture /
// graph instantiation, and device (re)allocation (hipFree implies a device-wide synchronisation).  CosyVoice2Model.tts runs the
// LLM on its own thread + stream while the caller's thread runs token2wav, so both can happen at the same moment.
std::recursive_mutex& runtime_lock();

// Captures `body()` on `s` into a graph.  If the body throws (argument checks inside the launchers), the capture is ended and the
// partial graph dropped before the exception travels on: a stream left in capture mode would poison every later call on it.
template <typename F>
static inline hipGraph_t capture_graph(hipStream_t s, F&& body) {
    hipGraph_t g = nullptr;
    CV_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    try { body(); }
    catch (...) { (void)hipStreamEndCapture(s, &g); if (g) (void)hipGraphDestroy(g); throw; }
    CV_HIP(hipStreamEndCapture(s, &g));
    return g;
}

// device buffer owned by a handle (workspaces, KV cache)
struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    void ensure(size_t n) {
        if (n <= bytes) return;
        std::lock_guard<std::recursive_mutex> lk(runtime_lock());
        if (p) CV_HIP(hipFree(p));
        p = nullptr; bytes = 0;
        n += n / 2;                                   // geometric growth: streaming requests grow T chunk by chunk
        CV_HIP(hipMalloc(&p, n)); bytes = n;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

}  // namespace cv
