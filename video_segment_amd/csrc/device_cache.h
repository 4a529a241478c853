// device_cache.h -- process-wide, per-device cache of the blocks behind DevBuf / PinnedBuf.
//
// Why: hipMalloc and hipFree synchronise the whole device and map / unmap gigabytes; a graph handle
// of the seam-3 interface lives for ONE window (dense_seg_graph_interface.h:58-98: create, add the
// frames, segment, read out, delete), so a caller that segments window after window paid ~3 GB of
// hipMalloc inside SegmentFullGraph and thirty hipFree at the end of every window -- and on some
// boxes ten times that (BENCH_r05: 278 ms per window in `segment` against 21 ms).  A closed handle
// now leaves its blocks here and the next handle of the process adopts them: after the first window
// no hipMalloc / hipFree is issued at all.
//
// Rules:
//  * A block goes back to the cache only after the device is idle for it: CacheFree synchronises
//    the device (what hipFree did implicitly) unless the calling thread has declared its handle
//    quiesced (QuiesceGuard: the destructor of a handle synchronises once, then releases sixty
//    buffers without sixty synchronisations).
//  * A request is served by the smallest cached block of its kind and device that is large enough
//    and wastes at most a quarter (or 64 KiB); sizes are rounded to 4 KiB below 1 MiB, to 2 MiB above.
//  * The cached (unused) bytes per device are bounded (default: 40 % of the device's memory,
//    VSG_DEVICE_CACHE_MB overrides, 0 switches the cache off); the blocks freed longest ago go
//    first.  A hipMalloc that fails empties the cache and is tried again.
//  * Contents are unspecified, as those of hipMalloc are.  VSG_DEVICE_CACHE_POISON=1 fills every
//    block handed out with 0xA5 (test hook: nothing may rely on zeroed fresh memory).
#ifndef VSG_DEVICE_CACHE_H_
#define VSG_DEVICE_CACHE_H_

#include <cstddef>
#include <cstdint>

namespace vsg {

enum CacheKind {
  kCacheDevice = 0,        // hipMalloc
  kCachePinned = 1,        // hipHostMalloc(hipHostMallocDefault)
  kCacheMappedCoherent = 2 // hipHostMalloc(hipHostMallocMapped | hipHostMallocCoherent): mailboxes
};

// Throws vsg::Error(-2) when the runtime cannot provide the block even with an empty cache.
void* CacheAlloc(size_t bytes, CacheKind kind);
void CacheFree(void* p) noexcept;
// Returns every cached block of the device (-1: all devices) to the runtime.
void CacheTrim(int device);
// Upper bound of the cached (unused) device bytes; negative restores the default.
void CacheSetLimit(int device, long long bytes);

struct CacheStats {
  long long bytes_in_use = 0;      // device blocks handed out and not returned
  long long bytes_cached = 0;      // device blocks waiting for reuse
  long long bytes_in_use_peak = 0;
  long long limit_bytes = 0;
  long long runtime_mallocs = 0;   // hipMalloc / hipHostMalloc calls actually issued
  long long runtime_frees = 0;     // hipFree / hipHostFree calls actually issued
  long long cache_hits = 0;
  long long device_syncs = 0;      // hipDeviceSynchronize calls issued by CacheFree
  double runtime_malloc_ms = 0, runtime_free_ms = 0, device_sync_ms = 0;
};
CacheStats CacheGetStats(int device);

// What the calling thread's allocations cost since the counters were last taken (a handle is driven
// by one thread at a time: the diagnostics of a segment call are the difference around it).
struct ThreadAllocCounters {
  long long runtime_mallocs = 0, runtime_frees = 0, cache_hits = 0, device_syncs = 0;
  double runtime_malloc_ms = 0, runtime_free_ms = 0, device_sync_ms = 0;
};
ThreadAllocCounters ThreadAllocSnapshot();

// While one is active on a thread, CacheFree on that thread does not synchronise the device: the
// owner has just done so (Begin) and launches nothing more.  Declared as the FIRST member of a handle
// class and begun in its destructor body, it spans the destruction of all later members.
class QuiesceGuard {
 public:
  QuiesceGuard() = default;
  ~QuiesceGuard();
  QuiesceGuard(const QuiesceGuard&) = delete;
  QuiesceGuard& operator=(const QuiesceGuard&) = delete;
  void Begin();   // hipDeviceSynchronize once, then frees of this thread skip it

 private:
  bool active_ = false;
};

}  // namespace vsg

#endif  // VSG_DEVICE_CACHE_H_
