// device_cache.cpp -- see device_cache.h.
#include "device_cache.h"

#include <execinfo.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace vsg {
namespace {

constexpr int kMaxDevices = 64;
constexpr int kKinds = 3;

struct Block {
  void* p = nullptr;
  size_t bytes = 0;
  int kind = 0;
  int device = 0;
  unsigned long long stamp = 0;   // order of the frees (eviction: oldest first)
};

struct DeviceState {
  std::multimap<size_t, Block> free_blocks[kKinds];
  long long limit = -1;   // device kind; -1: not resolved yet
  CacheStats stats;
  long long pinned_cached = 0;
};

struct Cache {
  std::mutex mu;
  DeviceState dev[kMaxDevices];
  std::unordered_map<void*, Block> live;
  unsigned long long stamp = 0;
};

Cache& TheCache() {
  static Cache* c = new Cache;   // (never destroyed: handles may be closed from atexit handlers)
  return *c;
}

thread_local ThreadAllocCounters t_counters;
thread_local int t_quiesced = 0;

double NowMs() {
  using clk = std::chrono::steady_clock;
  return std::chrono::duration<double, std::milli>(clk::now().time_since_epoch()).count();
}

size_t RoundUp(size_t bytes) {
  if (bytes == 0) bytes = 1;
  const size_t g = bytes < ((size_t)1 << 20) ? (size_t)4096 : ((size_t)2 << 20);
  return (bytes + g - 1) / g * g;
}

// (Cached pinned host bytes per device.  At 1 GiB the streams of a bench run left just under the limit
// behind, and the close of a small seam-3 window a few legs later paid for evicting them: 80 ms of
// hipHostFree in one window of BASELINE configs[1], `slowest_window.phases.close` in r6_c.)
constexpr long long kPinnedLimit = 4ll << 30;

long long DefaultLimit() {
  if (const char* e = getenv("VSG_DEVICE_CACHE_MB")) return std::max(0ll, atoll(e)) << 20;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 8ll << 30;
  return (long long)(total_b / 10 * 4);
}

hipError_t RuntimeAlloc(void** p, size_t bytes, int kind) {
  switch (kind) {
    case kCacheDevice:
      return hipMalloc(p, bytes);
    case kCachePinned:
      return hipHostMalloc(p, bytes, hipHostMallocDefault);
    default:
      return hipHostMalloc(p, bytes, hipHostMallocMapped | hipHostMallocCoherent);
  }
}

void RuntimeFree(const Block& b) {
  const double t0 = NowMs();
  if (b.kind == kCacheDevice) {
    (void)hipFree(b.p);
  } else {
    (void)hipHostFree(b.p);
  }
  const double dt = NowMs() - t0;
  t_counters.runtime_frees += 1;
  t_counters.runtime_free_ms += dt;
  Cache& c = TheCache();
  std::lock_guard<std::mutex> lk(c.mu);
  c.dev[b.device].stats.runtime_frees += 1;
  c.dev[b.device].stats.runtime_free_ms += dt;
}

// (lock held) Takes the blocks that have to go so that the cached bytes of the device respect the
// limit; the caller releases them outside the lock.
void CollectEvictions(DeviceState& d, std::vector<Block>* out) {
  auto over = [&d]() {
    return d.stats.bytes_cached > d.limit || d.pinned_cached > kPinnedLimit;
  };
  while (over()) {
    const bool dev_over = d.stats.bytes_cached > d.limit;
    std::multimap<size_t, Block>::iterator best;
    int best_kind = -1;
    for (int k = 0; k < kKinds; ++k) {
      if ((k == kCacheDevice) != dev_over) continue;
      for (auto it = d.free_blocks[k].begin(); it != d.free_blocks[k].end(); ++it) {
        if (best_kind < 0 || it->second.stamp < best->second.stamp) {
          best = it;
          best_kind = k;
        }
      }
    }
    if (best_kind < 0) break;
    if (best_kind == kCacheDevice) {
      d.stats.bytes_cached -= (long long)best->second.bytes;
    } else {
      d.pinned_cached -= (long long)best->second.bytes;
    }
    out->push_back(best->second);
    d.free_blocks[best_kind].erase(best);
  }
}

int CurrentDevice() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 0;
  return dev;
}

}  // namespace

void* CacheAlloc(size_t bytes, CacheKind kind) {
  const size_t need = RoundUp(bytes);
  const int device = CurrentDevice();
  Cache& c = TheCache();
  Block b;
  {
    std::lock_guard<std::mutex> lk(c.mu);
    DeviceState& d = c.dev[device];
    if (d.limit < 0) d.limit = DefaultLimit();
    auto& fb = d.free_blocks[kind];
    auto it = fb.lower_bound(need);
    if (it != fb.end() && (it->first - need <= need / 4 || it->first - need <= ((size_t)64 << 10))) {
      b = it->second;
      fb.erase(it);
      if (kind == kCacheDevice) {
        d.stats.bytes_cached -= (long long)b.bytes;
        d.stats.bytes_in_use += (long long)b.bytes;
        d.stats.bytes_in_use_peak = std::max(d.stats.bytes_in_use_peak, d.stats.bytes_in_use);
      } else {
        d.pinned_cached -= (long long)b.bytes;
      }
      d.stats.cache_hits += 1;
      c.live[b.p] = b;
    }
  }
  if (b.p) {
    t_counters.cache_hits += 1;
  } else {
    const double t0 = NowMs();
    void* p = nullptr;
    hipError_t e = RuntimeAlloc(&p, need, kind);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      CacheTrim(device);
      e = RuntimeAlloc(&p, need, kind);
    }
    const double dt = NowMs() - t0;
    if (e != hipSuccess) {
      (void)hipGetLastError();
      Throw(-2, std::string(kind == kCacheDevice ? "hipMalloc" : "hipHostMalloc") + " of " +
                    std::to_string(need) + " bytes: " + hipGetErrorString(e));
    }
    t_counters.runtime_mallocs += 1;
    t_counters.runtime_malloc_ms += dt;
    static const bool log_mallocs = getenv("VSG_DEVICE_CACHE_LOG") != nullptr;
    if (log_mallocs) {   // (debugging aid: who still reaches the runtime allocator in the steady state?)
      std::fprintf(stderr, "[vsg] runtime allocation: %zu bytes, kind %d, %.2f ms\n", need, (int)kind, dt);
      void* frames[16];
      const int nfr = backtrace(frames, 16);
      backtrace_symbols_fd(frames, nfr, 2);
    }
    b.p = p;
    b.bytes = need;
    b.kind = kind;
    b.device = device;
    std::lock_guard<std::mutex> lk(c.mu);
    DeviceState& d = c.dev[device];
    d.stats.runtime_mallocs += 1;
    d.stats.runtime_malloc_ms += dt;
    if (kind == kCacheDevice) {
      d.stats.bytes_in_use += (long long)need;
      d.stats.bytes_in_use_peak = std::max(d.stats.bytes_in_use_peak, d.stats.bytes_in_use);
    }
    c.live[p] = b;
  }
  // (test hooks: VSG_DEVICE_CACHE_POISON=1 fills every block handed out; _POISON_FROM / _POISON_TO
  // restrict that to the allocations with these serial numbers and _TRACE prints the call stack of
  // one -- tools/poison_bisect.py finds the buffer whose first reader expects zeros)
  static const bool poison_on = getenv("VSG_DEVICE_CACHE_POISON") && atoi(getenv("VSG_DEVICE_CACHE_POISON")) != 0;
  static const long long poison_from = getenv("VSG_DEVICE_CACHE_POISON_FROM") ? atoll(getenv("VSG_DEVICE_CACHE_POISON_FROM")) : 0;
  static const long long poison_to = getenv("VSG_DEVICE_CACHE_POISON_TO") ? atoll(getenv("VSG_DEVICE_CACHE_POISON_TO")) : (1ll << 62);
  static const long long trace_at = getenv("VSG_DEVICE_CACHE_TRACE") ? atoll(getenv("VSG_DEVICE_CACHE_TRACE")) : -1;
  static std::atomic<long long> serial{0};
  const long long my_serial = serial.fetch_add(1);
  if (my_serial == trace_at) {
    std::fprintf(stderr, "[vsg] allocation %lld: %zu bytes (requested %zu), kind %d\n", my_serial, b.bytes, bytes, (int)kind);
    void* frames[32];
    const int nfr = backtrace(frames, 32);
    backtrace_symbols_fd(frames, nfr, 2);
  }
  const bool poison = poison_on && my_serial >= poison_from && my_serial < poison_to;
  if (poison) {
    if (kind == kCacheDevice) {
      // (the handles' streams are non-blocking: the fill on the null stream has to be complete
      // before the owner's first kernel may write the block)
      (void)hipMemset(b.p, 0xA5, b.bytes);
      (void)hipStreamSynchronize(nullptr);
    } else {
      std::memset(b.p, 0xA5, b.bytes);
    }
  }
  return b.p;
}

void CacheFree(void* p) noexcept {
  if (!p) return;
  Cache& c = TheCache();
  Block b;
  {
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.live.find(p);
    if (it == c.live.end()) return;   // (not ours: nothing sensible to do)
    b = it->second;
    c.live.erase(it);
    if (b.kind == kCacheDevice) c.dev[b.device].stats.bytes_in_use -= (long long)b.bytes;
  }
  long long limit;
  {
    std::lock_guard<std::mutex> lk(c.mu);
    limit = c.dev[b.device].limit;
  }
  if (limit == 0) {   // cache switched off: the old behaviour
    RuntimeFree(b);
    return;
  }
  // Whatever still reads or writes the block has to be complete before another stream may get it
  // (hipFree waited for the device as well).
  if (t_quiesced == 0) {
    const double t0 = NowMs();
    int prev = -1;
    const bool other = hipGetDevice(&prev) == hipSuccess && prev != b.device;
    if (other) (void)hipSetDevice(b.device);
    (void)hipDeviceSynchronize();
    if (other) (void)hipSetDevice(prev);
    const double dt = NowMs() - t0;
    t_counters.device_syncs += 1;
    t_counters.device_sync_ms += dt;
    std::lock_guard<std::mutex> lk(c.mu);
    c.dev[b.device].stats.device_syncs += 1;
    c.dev[b.device].stats.device_sync_ms += dt;
  }
  std::vector<Block> evict;
  {
    std::lock_guard<std::mutex> lk(c.mu);
    DeviceState& d = c.dev[b.device];
    b.stamp = ++c.stamp;
    d.free_blocks[b.kind].emplace(b.bytes, b);
    if (b.kind == kCacheDevice) {
      d.stats.bytes_cached += (long long)b.bytes;
    } else {
      d.pinned_cached += (long long)b.bytes;
    }
    CollectEvictions(d, &evict);
  }
  for (const Block& e : evict) RuntimeFree(e);
}

void CacheTrim(int device) {
  Cache& c = TheCache();
  std::vector<Block> all;
  {
    std::lock_guard<std::mutex> lk(c.mu);
    for (int dv = 0; dv < kMaxDevices; ++dv) {
      if (device >= 0 && dv != device) continue;
      DeviceState& d = c.dev[dv];
      for (int k = 0; k < kKinds; ++k) {
        for (auto& kv : d.free_blocks[k]) all.push_back(kv.second);
        d.free_blocks[k].clear();
      }
      d.stats.bytes_cached = 0;
      d.pinned_cached = 0;
    }
  }
  for (const Block& b : all) RuntimeFree(b);
}

void CacheSetLimit(int device, long long bytes) {
  if (device < 0 || device >= kMaxDevices) return;
  Cache& c = TheCache();
  std::vector<Block> evict;
  {
    std::lock_guard<std::mutex> lk(c.mu);
    DeviceState& d = c.dev[device];
    d.limit = bytes < 0 ? DefaultLimit() : bytes;
    CollectEvictions(d, &evict);
  }
  for (const Block& e : evict) RuntimeFree(e);
}

CacheStats CacheGetStats(int device) {
  CacheStats s;
  if (device < 0 || device >= kMaxDevices) return s;
  Cache& c = TheCache();
  std::lock_guard<std::mutex> lk(c.mu);
  s = c.dev[device].stats;
  s.limit_bytes = c.dev[device].limit;
  return s;
}

ThreadAllocCounters ThreadAllocSnapshot() { return t_counters; }

void QuiesceGuard::Begin() {
  if (active_) return;
  const double t0 = NowMs();
  (void)hipDeviceSynchronize();
  t_counters.device_syncs += 1;
  t_counters.device_sync_ms += NowMs() - t0;
  active_ = true;
  ++t_quiesced;
}

QuiesceGuard::~QuiesceGuard() {
  if (active_) --t_quiesced;
}

}  // namespace vsg
