// common.h -- shared declarations of the HIP product library (libvsg_hip.so).
//
// Written for gfx950 (MI355X / CDNA4) only.  All product translation units are compiled with
// -ffp-contract=off and -fhip-fp32-correctly-rounded-divide-sqrt: bit-exact parity with the
// reference's SSE2 scalar arithmetic needs IEEE add/mul/div/sqrt and no FMA contraction
// (SURVEY.md H3).
#ifndef VSG_COMMON_H_
#define VSG_COMMON_H_

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "device_cache.h"

namespace vsg {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void Throw(int code, const std::string& msg) { throw Error(code, msg); }

#define VSG_HIP(expr)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      ::vsg::Throw(-2, std::string(#expr) + ": " + hipGetErrorString(e_) + " at " +      \
                           __FILE__ + ":" + std::to_string(__LINE__));                   \
    }                                                                                    \
  } while (0)

#define VSG_REQUIRE(cond, code, msg)                                                     \
  do {                                                                                   \
    if (!(cond)) ::vsg::Throw((code), std::string(msg) + " [" #cond "]");                \
  } while (0)

// Upper bound on the host threads of one parallel section of the chunk boundary (region tables, tube
// analysis, rasters): VSG_HOST_THREADS, or a share of the machine that leaves room for one process per
// GPU of an 8-GPU node -- 16 at least, 48 at most.
inline int HostThreadCap() {
  static const int cap = [] {
    if (const char* e = getenv("VSG_HOST_THREADS")) return std::max(1, atoi(e));
    const int hw = (int)std::thread::hardware_concurrency();
    return std::max(16, std::min(hw / 8, 48));
  }();
  return cap;
}

// Simple owning device buffer.  The block comes from / goes back to the process-wide cache
// (device_cache.h): a closed handle's memory is adopted by the next one instead of being unmapped.
template <class T>
class DevBuf {
 public:
  DevBuf() = default;
  explicit DevBuf(size_t n) { alloc(n); }
  ~DevBuf() { release(); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) {
      release();
      p_ = o.p_;
      n_ = o.n_;
      o.p_ = nullptr;
      o.n_ = 0;
    }
    return *this;
  }
  void alloc(size_t n) {
    release();
    if (n == 0) return;
    p_ = static_cast<T*>(CacheAlloc(n * sizeof(T), kCacheDevice));
    n_ = n;
  }
  // Grows with slack: sizes that drift from chunk to chunk (intervals, pairs, runs) must not cost
  // a new block -- and a device synchronisation for the old one -- every time they tick up.
  void ensure(size_t n) {
    if (n > n_) alloc(n + n / 8 + 256);
  }
  void release() {
    if (p_) CacheFree(p_);
    p_ = nullptr;
    n_ = 0;
  }
  T* get() const { return p_; }
  size_t size() const { return n_; }

 private:
  T* p_ = nullptr;
  size_t n_ = 0;
};

// Pinned host buffer for small device->host hand-offs.
template <class T>
class PinnedBuf {
 public:
  PinnedBuf() = default;
  ~PinnedBuf() { release(); }
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  void ensure(size_t n) {
    if (n <= n_) return;
    release();
    n = n + n / 4 + 256;   // (with slack: sizes that drift from chunk to chunk)
    p_ = static_cast<T*>(CacheAlloc(n * sizeof(T), kCachePinned));
    n_ = n;
  }
  void release() {
    if (p_) CacheFree(p_);
    p_ = nullptr;
    n_ = 0;
  }
  T* get() const { return p_; }
  size_t size() const { return n_; }

 private:
  T* p_ = nullptr;
  size_t n_ = 0;
};

// Elapsed device time between two points of a stream.
class StageTimer {
 public:
  StageTimer() {
    VSG_HIP(hipEventCreate(&a_));
    VSG_HIP(hipEventCreate(&b_));
  }
  ~StageTimer() {
    (void)hipEventDestroy(a_);
    (void)hipEventDestroy(b_);
  }
  void start(hipStream_t s) { VSG_HIP(hipEventRecord(a_, s)); }
  // Records the end event, waits for it and returns the elapsed milliseconds.
  float stop(hipStream_t s) {
    VSG_HIP(hipEventRecord(b_, s));
    VSG_HIP(hipEventSynchronize(b_));
    float ms = 0;
    VSG_HIP(hipEventElapsedTime(&ms, a_, b_));
    return ms;
  }

 private:
  hipEvent_t a_, b_;
};

constexpr int kNumBuckets = 2048;          // segmentation_graph.h ctor: 2K buckets
constexpr int kBucketSlots = kNumBuckets + 2;   // 0..2047 real, 2048 virtual, 2049 end sentinel
constexpr uint16_t kInvalidKey = 0x0FFF;   // sorts after every real bucket (12 sort bits)

}  // namespace vsg

#endif  // VSG_COMMON_H_
