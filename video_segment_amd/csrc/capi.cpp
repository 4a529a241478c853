// capi.cpp -- extern "C" entry points declared in include/vsg.h.
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <fstream>
#include <thread>

#include "../../include/vsg.h"
#include "region_segmentation.h"
#include "stream.h"

namespace {

thread_local std::string g_last_error;

template <class F>
int Guard(F&& f) {
  try {
    f();
    return VSG_OK;
  } catch (const vsg::Error& e) {
    g_last_error = e.what();
    return e.code;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return VSG_ERR_INTERNAL;
  }
}

// The HIP current device is a per-thread setting that defaults to 0, and a handle may be driven
// from any thread (the reference's pipeline calls OpenStreams and ProcessFrame on different
// threads).  Every entry point therefore binds the calling thread to the handle's device for the
// duration of the call and restores the caller's device afterwards.
class DeviceGuard {
 public:
  explicit DeviceGuard(int device) {
    if (device < 0) return;
    if (hipGetDevice(&prev_) != hipSuccess) return;
    if (prev_ != device) {
      VSG_HIP(hipSetDevice(device));
      changed_ = true;
    }
  }
  ~DeviceGuard() {
    if (changed_) (void)hipSetDevice(prev_);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;

 private:
  int prev_ = -1;
  bool changed_ = false;
};

// Resolves -1 (= the caller's current device) to an ordinal at creation time.
int ResolveDevice(int device) {
  if (device >= 0) return device;
  int cur = 0;
  VSG_HIP(hipGetDevice(&cur));
  return cur;
}

void RequireDevice(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    vsg::Throw(VSG_ERR_DEVICE,
               "no usable HIP device (libvsg_hip has no CPU fallback): " +
                   std::string(e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
  }
  if (device >= n) vsg::Throw(VSG_ERR_DEVICE, "device ordinal out of range");
}

}  // namespace

struct vsg_stream {
  int device = 0;
  std::unique_ptr<vsg::DenseSegmentationHip> impl;
  std::vector<int32_t> id_image;
};

struct vsg_regionseg {
  std::unique_ptr<vsg::RegionSegmentationHost> impl;
  int W = 0, H = 0;
};

struct vsg_graph {
  int device = 0;
  int W = 0, H = 0;
  size_t wh = 0;
  hipStream_t stream = nullptr;
  std::unique_ptr<vsg::DenseGraphHip> g;
  std::unique_ptr<vsg::Preprocessor> pre;
  std::vector<std::shared_ptr<vsg::DevBuf<float>>> feats;     // per slice (null for virtual)
  std::vector<std::shared_ptr<vsg::DevBuf<float>>> flows_dev;   // per slice (null: no flow)
  vsg::DevBuf<uint8_t> staging_bgr;
  vsg::DevBuf<float> staging_f32;
  vsg::DevBuf<int32_t> staging_ids;
  vsg_timings timings;
  // backing store of vsg_graph_get_regions / vsg_graph_get_intervals
  std::vector<vsg_region> out_regions;
  std::vector<int32_t> out_nbr_ptr, out_nbr_idx;
  std::vector<vsg_interval> out_intervals;
  ~vsg_graph() {
    if (stream) (void)hipStreamSynchronize(stream);
    g.reset();
    pre.reset();
    if (stream) (void)hipStreamDestroy(stream);
  }
};

static void FillDiagnostics(const vsg::GraphTimings& gt, vsg_diagnostics* d) {
  std::memset(d, 0, sizeof(*d));
  d->segment_wall_ms = gt.segment_wall_ms;
  d->prepare_ms = gt.prepare_ms;
  d->constrained_merge_ms = gt.constrained_merge_ms;
  d->stages = gt.stages;
  d->optimistic_stages = gt.optimistic_stages;
  d->rollbacks = gt.rollbacks;
  d->slab_growths = gt.slab_growths;
  d->slab_growth_ms = gt.slab_growth_ms;
  d->spine_pool_growths = gt.spine_growths;
  d->spine_pool_growth_ms = gt.spine_growth_ms;
  d->runtime_mallocs = gt.runtime_mallocs;
  d->runtime_malloc_ms = gt.runtime_malloc_ms;
  d->runtime_frees = gt.runtime_frees;
  d->runtime_free_ms = gt.runtime_free_ms;
  d->cache_hits = gt.cache_hits;
  d->device_syncs = gt.device_syncs;
  d->device_sync_ms = gt.device_sync_ms;
  d->mail_waits = gt.mail_waits;
  d->mail_wait_ms = gt.mail_wait_ms;
  d->mail_wait_longest_ms = gt.mail_wait_longest_ms;
  d->mail_mode = gt.mail_mode;
}

extern "C" {

const char* vsg_last_error(void) { return g_last_error.c_str(); }

int vsg_device_memory_stats(int device, vsg_memory_stats* out) {
  return Guard([&] {
    VSG_REQUIRE(out, VSG_ERR_INVALID, "null argument");
    const int dev = ResolveDevice(device);
    const vsg::CacheStats s = vsg::CacheGetStats(dev);
    out->bytes_in_use = s.bytes_in_use;
    out->bytes_in_use_peak = s.bytes_in_use_peak;
    out->bytes_cached = s.bytes_cached;
    out->limit_bytes = s.limit_bytes;
    out->runtime_mallocs = s.runtime_mallocs;
    out->runtime_frees = s.runtime_frees;
    out->cache_hits = s.cache_hits;
    out->device_syncs = s.device_syncs;
    out->runtime_malloc_ms = s.runtime_malloc_ms;
    out->runtime_free_ms = s.runtime_free_ms;
    out->device_sync_ms = s.device_sync_ms;
  });
}

int vsg_device_memory_trim(int device) {
  return Guard([&] {
    const int dev = ResolveDevice(device);
    DeviceGuard dg(dev);
    vsg::CacheTrim(dev);
  });
}

int vsg_device_memory_limit(int device, int64_t bytes) {
  return Guard([&] {
    const int dev = ResolveDevice(device);
    DeviceGuard dg(dev);
    vsg::CacheSetLimit(dev, (long long)bytes);
  });
}
int vsg_version(void) { return 100; }

void vsg_default_options(vsg_options* o) {
  o->presmoothing = 2;
  o->frac_min_region_size = 0.01f;
  o->chunk_size = 20;
  o->chunk_overlap_ratio = 0.2f;
  o->num_constraint_frames = 1;
  o->enforce_n4_connectivity = 1;
  o->enforce_spatial_connectedness = 1;
  o->color_distance = 1;
  o->device = -1;
  o->two_stage_oversegment = 0;
  o->compute_vectorization = 0;
}

int vsg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ---- host-only parity hook ---------------------------------------------------------------
int vsg_vectorize_id_image(const int32_t* ids, int width, int height, const uint8_t** data,
                           size_t* len) {
  return Guard([&] {
    VSG_REQUIRE(ids && data && len && width >= 1 && height >= 1, VSG_ERR_INVALID, "bad argument");
    thread_local std::string wire;
    // Region2D list sorted by id with scan-order rasterizations, as RetrieveSegmentation3D emits.
    std::vector<std::pair<int32_t, vsg::Interval>> runs;
    for (int y = 0; y < height; ++y) {
      const int32_t* row = ids + (size_t)y * width;
      for (int x = 0; x < width;) {
        int x2 = x;
        while (x2 + 1 < width && row[x2 + 1] == row[x]) ++x2;
        VSG_REQUIRE(row[x] >= 0, VSG_ERR_INVALID, "negative region id");
        runs.emplace_back(row[x], vsg::Interval{y, x, x2});
        x = x2 + 1;
      }
    }
    std::stable_sort(runs.begin(), runs.end(),
                     [](const std::pair<int32_t, vsg::Interval>& a,
                        const std::pair<int32_t, vsg::Interval>& b) { return a.first < b.first; });
    vsg::SegDesc d;
    d.frame_width = width;
    d.frame_height = height;
    for (const auto& r : runs) {
      if (d.regions.empty() || d.regions.back().id != r.first) {
        d.regions.emplace_back();
        d.regions.back().id = r.first;
      }
      d.regions.back().raster.push_back(r.second);
    }
    for (auto& r : d.regions) vsg::MomentsFromRaster(r.raster, &r.moments);
    vsg::ComputeFrameVectorization(&d);
    wire = vsg::EncodeSegDesc(d);
    *data = reinterpret_cast<const uint8_t*>(wire.data());
    *len = wire.size();
  });
}

// ---- stream ------------------------------------------------------------------------------
int vsg_stream_create(const vsg_options* o, int width, int height, vsg_stream** out) {
  return Guard([&] {
    VSG_REQUIRE(o && out, VSG_ERR_INVALID, "null argument");
    RequireDevice(o->device);
    VSG_REQUIRE(width >= 2 && height >= 1 && width <= 65535 && height <= 65535, VSG_ERR_INVALID,
                "unsupported frame size");
    std::unique_ptr<vsg_stream> s(new vsg_stream);
    s->device = ResolveDevice(o->device);
    DeviceGuard dg(s->device);
    vsg_options opts = *o;
    opts.device = s->device;
    s->impl.reset(new vsg::DenseSegmentationHip(opts, width, height));
    *out = s.release();
  });
}

void vsg_stream_destroy(vsg_stream* s) {
  if (!s) return;
  (void)Guard([&] {
    DeviceGuard dg(s->device);
    s->impl.reset();
  });
  delete s;
}

int vsg_stream_process_frame(vsg_stream* s, int flush, const uint8_t* bgr, size_t stride,
                             const float* flow, int has_flow_stream, int mem, int* num_results) {
  return Guard([&] {
    VSG_REQUIRE(s && num_results, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(s->device);
    *num_results = s->impl->ProcessFrame(flush != 0, bgr, stride, flow, has_flow_stream != 0, mem);
  });
}

int vsg_stream_chunk_size(const vsg_stream* s) { return s ? s->impl->ChunkSize() : VSG_ERR_INVALID; }

int vsg_stream_result_bytes(vsg_stream* s, int i, const uint8_t** data, size_t* len) {
  return Guard([&] {
    VSG_REQUIRE(s && data && len, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(s->device);
    VSG_REQUIRE(i >= 0 && i < s->impl->num_results(), VSG_ERR_INVALID, "result index");
    const std::string& b = s->impl->result_bytes(i);
    *data = reinterpret_cast<const uint8_t*>(b.data());
    *len = b.size();
  });
}

int vsg_stream_result_id_image(vsg_stream* s, int i, int32_t* out) {
  return Guard([&] {
    VSG_REQUIRE(s && out, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(s->device);
    VSG_REQUIRE(i >= 0 && i < s->impl->num_results(), VSG_ERR_INVALID, "result index");
    const size_t n = (size_t)s->impl->W() * s->impl->H();
    for (size_t k = 0; k < n; ++k) out[k] = -1;
    vsg::RenderIdImage(s->impl->result(i), s->impl->W(), out);
  });
}

int vsg_stream_last_merge_stats(const vsg_stream* s, int64_t* st) {
  return Guard([&] {
    VSG_REQUIRE(s && st, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(s->device);
    s->impl->last_merge_stats(st);
  });
}

int vsg_stream_last_timings(const vsg_stream* s, vsg_timings* t) {
  return Guard([&] {
    VSG_REQUIRE(s && t, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(s->device);
    *t = s->impl->last_timings();
  });
}

int vsg_stream_last_diagnostics(const vsg_stream* s, vsg_diagnostics* d) {
  return Guard([&] {
    VSG_REQUIRE(s && d, VSG_ERR_INVALID, "null argument");
    FillDiagnostics(s->impl->last_graph_timings(), d);
  });
}

int vsg_stream_last_smoothed(vsg_stream* s, float* out) {
  return Guard([&] {
    VSG_REQUIRE(s && out, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(s->device);
    s->impl->CopyLastSmoothed(out);
  });
}

int vsg_stream_export_halo(vsg_stream* s, const int32_t** virt, const int32_t** cons,
                           int64_t scalars[4]) {
  return Guard([&] {
    VSG_REQUIRE(s && virt && cons && scalars, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(s->device);
    s->impl->ExportHalo(virt, cons, scalars);
  });
}

int vsg_stream_expect_halo(vsg_stream* s) {
  return Guard([&] {
    VSG_REQUIRE(s, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(s->device);
    s->impl->ExpectHalo();
  });
}

int vsg_stream_restart(vsg_stream* s) {
  return Guard([&] {
    VSG_REQUIRE(s, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(s->device);
    s->impl->Restart();
  });
}

int vsg_stream_import_halo(vsg_stream* s, const int32_t* virt, const int32_t* cons, int mem,
                           const int64_t scalars[4]) {
  return Guard([&] {
    VSG_REQUIRE(s && virt && cons && scalars, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(s->device);
    s->impl->ImportHalo(virt, cons, mem, scalars);
  });
}

// ---- chunk chain over RCCL ------------------------------------------------------------------
}  // extern "C" (reopened below)

#define VSG_NCCL(call)                                                                      \
  do {                                                                                      \
    ncclResult_t r_ = (call);                                                               \
    if (r_ != ncclSuccess) {                                                                \
      vsg::Throw(VSG_ERR_DEVICE, std::string("RCCL: ") + ncclGetErrorString(r_) + " in " #call); \
    }                                                                                       \
  } while (0)

struct vsg_chain {
  int rank = 0, world = 1, device = 0;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  // [2 * W*H label planes | 4 int64 scalars as 8 int32], one buffer per direction
  vsg::DevBuf<int32_t> send_staging, recv_staging;
};

namespace {

// What rank 0 publishes for the others: the id of THIS run's communicator.  The nonce is chosen by
// the caller (the same on every rank of a run, different from earlier runs); a reader that finds
// a file with another nonce -- left behind by an earlier run, or by a crashed one -- keeps polling
// instead of joining a communicator nobody else will ever join.
struct ChainIdFile {
  char magic[8];
  uint64_t nonce;
  ncclUniqueId id;
};
const char kChainMagic[8] = {'V', 'S', 'G', 'C', 'H', 'A', 'I', 'N'};

}  // namespace

extern "C" {

int vsg_chain_create(int rank, int world, const char* id_file, uint64_t nonce, int device,
                     vsg_chain** out) {
  return Guard([&] {
    VSG_REQUIRE(out && id_file && world >= 1 && rank >= 0 && rank < world, VSG_ERR_INVALID,
                "bad argument");
    RequireDevice(device);
    std::unique_ptr<vsg_chain> c(new vsg_chain);
    c->rank = rank;
    c->world = world;
    c->device = ResolveDevice(device);
    DeviceGuard dg(c->device);
    ChainIdFile rec;
    const std::string path(id_file);
    if (rank == 0) {
      // Whatever an earlier run left under this name goes first; the record appears atomically
      // (temporary name + rename), so a reader never sees half of it.
      (void)std::remove(path.c_str());
      std::memcpy(rec.magic, kChainMagic, sizeof(rec.magic));
      rec.nonce = nonce;
      VSG_NCCL(ncclGetUniqueId(&rec.id));
      const std::string tmp = path + ".tmp";
      {
        std::ofstream f(tmp.c_str(), std::ios::binary | std::ios::trunc);
        f.write(reinterpret_cast<const char*>(&rec), sizeof(rec));
        VSG_REQUIRE(f.good(), VSG_ERR_INVALID, "cannot write the communicator id file");
      }
      VSG_REQUIRE(std::rename(tmp.c_str(), path.c_str()) == 0, VSG_ERR_INVALID,
                  "cannot publish the communicator id file");
    } else {
      bool got = false;
      for (int attempt = 0; attempt < 1200 && !got; ++attempt) {   // up to two minutes
        std::ifstream f(path.c_str(), std::ios::binary);
        if (f.good()) {
          f.read(reinterpret_cast<char*>(&rec), sizeof(rec));
          got = f.gcount() == (std::streamsize)sizeof(rec) &&
                std::memcmp(rec.magic, kChainMagic, sizeof(rec.magic)) == 0 && rec.nonce == nonce;
        }
        if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
      }
      VSG_REQUIRE(got, VSG_ERR_STATE,
                  "timed out waiting for the communicator id file of this run (nonce mismatch or "
                  "rank 0 never started)");
    }
    VSG_NCCL(ncclCommInitRank(&c->comm, world, rec.id, rank));
    // ncclCommInitRank is collective: once it has returned here every rank has read the record.
    if (rank == 0) (void)std::remove(path.c_str());
    VSG_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    *out = c.release();
  });
}

void vsg_chain_destroy(vsg_chain* c) {
  if (!c) return;
  (void)Guard([&] {
    DeviceGuard dg(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    c->send_staging.release();
    c->recv_staging.release();
  });
  delete c;
}

int vsg_chain_info(const vsg_chain* c, int* rank, int* world) {
  return Guard([&] {
    VSG_REQUIRE(c && rank && world, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(c->device);
    VSG_NCCL(ncclCommUserRank(c->comm, rank));
    VSG_NCCL(ncclCommCount(c->comm, world));
  });
}

int vsg_chain_exchange_halo(vsg_chain* c, vsg_stream* from, int dst, vsg_stream* into, int src) {
  return Guard([&] {
    VSG_REQUIRE(c && (from || into), VSG_ERR_INVALID, "bad argument");
    VSG_REQUIRE(!from || (dst >= 0 && dst < c->world), VSG_ERR_INVALID, "destination rank");
    VSG_REQUIRE(!into || (src >= 0 && src < c->world), VSG_ERR_INVALID, "source rank");
    // A rank may hand a halo to itself (two handles on one GPU), but only as one exchange: a lone
    // send to oneself would never find its receive.
    VSG_REQUIRE(!from || dst != c->rank || (into && src == c->rank), VSG_ERR_INVALID,
                "a send to the own rank needs the matching receive in the same call");
    VSG_REQUIRE(!into || src != c->rank || (from && dst == c->rank), VSG_ERR_INVALID,
                "a receive from the own rank needs the matching send in the same call");
    VSG_REQUIRE(from != into, VSG_ERR_INVALID, "a stream cannot hand the halo to itself");
    // The staging buffers, the communicator's stream and the streams' halo planes all have to
    // live on the chain's device (no peer copies, no ImportHalo on a foreign current device).
    VSG_REQUIRE(!from || from->device == c->device, VSG_ERR_INVALID,
                "the sending stream lives on another device than the chain");
    VSG_REQUIRE(!into || into->device == c->device, VSG_ERR_INVALID,
                "the receiving stream lives on another device than the chain");
    DeviceGuard dg(c->device);
    size_t wh_send = 0, wh_recv = 0;
    if (from) {
      wh_send = (size_t)from->impl->W() * from->impl->H();
      const int32_t *virt = nullptr, *cons = nullptr;
      int64_t scalars[4];
      from->impl->ExportHalo(&virt, &cons, scalars);
      c->send_staging.ensure(2 * wh_send + 8);
      int32_t* st = c->send_staging.get();
      VSG_HIP(hipMemcpyAsync(st, virt, wh_send * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
      VSG_HIP(hipMemcpyAsync(st + wh_send, cons, wh_send * sizeof(int32_t), hipMemcpyDeviceToDevice,
                             c->stream));
      VSG_HIP(hipMemcpyAsync(st + 2 * wh_send, scalars, sizeof(scalars), hipMemcpyHostToDevice, c->stream));
      // `scalars` is a stack buffer: the copy has to be done before it goes out of scope
      VSG_HIP(hipStreamSynchronize(c->stream));
    }
    if (into) {
      wh_recv = (size_t)into->impl->W() * into->impl->H();
      c->recv_staging.ensure(2 * wh_recv + 8);
    }
    VSG_NCCL(ncclGroupStart());
    if (from) VSG_NCCL(ncclSend(c->send_staging.get(), 2 * wh_send + 8, ncclInt32, dst, c->comm, c->stream));
    if (into) VSG_NCCL(ncclRecv(c->recv_staging.get(), 2 * wh_recv + 8, ncclInt32, src, c->comm, c->stream));
    VSG_NCCL(ncclGroupEnd());
    int64_t scalars[4] = {0, 0, 0, 0};
    if (into) {
      VSG_HIP(hipMemcpyAsync(scalars, c->recv_staging.get() + 2 * wh_recv, sizeof(scalars),
                             hipMemcpyDeviceToHost, c->stream));
    }
    VSG_HIP(hipStreamSynchronize(c->stream));
    if (into) {
      into->impl->ImportHalo(c->recv_staging.get(), c->recv_staging.get() + wh_recv, VSG_MEM_DEVICE, scalars);
    }
  });
}

int vsg_chain_send_halo(vsg_chain* c, vsg_stream* from, int dst) {
  if (!from) {
    g_last_error = "null argument";
    return VSG_ERR_INVALID;
  }
  return vsg_chain_exchange_halo(c, from, dst, nullptr, -1);
}

int vsg_chain_recv_halo(vsg_chain* c, vsg_stream* into, int src) {
  if (!into) {
    g_last_error = "null argument";
    return VSG_ERR_INVALID;
  }
  return vsg_chain_exchange_halo(c, nullptr, -1, into, src);
}

// ---- hierarchical region segmentation (host) -------------------------------------------------
void vsg_regionseg_default_options(vsg_regionseg_options* o) {
  const vsg::RegionSegOptions d;
  o->min_region_num = d.min_region_num;
  o->max_region_num = d.max_region_num;
  o->level_cutoff_fraction = d.level_cutoff_fraction;
  o->small_region_penalizer = d.small_region_penalizer;
  o->luminance_bins = d.luminance_bins;
  o->color_bins = d.color_bins;
  o->flow_bins = d.flow_bins;
  o->chunk_set_size = d.chunk_set_size;
  o->chunk_set_overlap = d.chunk_set_overlap;
  o->constraint_chunks = d.constraint_chunks;
  o->use_appearance = d.use_appearance;
  o->use_flow = d.use_flow;
  o->use_size_penalizer = d.use_size_penalizer;
  o->compute_vectorization = d.compute_vectorization;
  o->save_descriptors = d.save_descriptors;
}

int vsg_regionseg_create(const vsg_regionseg_options* o, int width, int height, vsg_regionseg** out) {
  return Guard([&] {
    VSG_REQUIRE(o && out, VSG_ERR_INVALID, "null argument");
    vsg::RegionSegOptions d;
    d.min_region_num = o->min_region_num;
    d.max_region_num = o->max_region_num;
    d.level_cutoff_fraction = o->level_cutoff_fraction;
    d.small_region_penalizer = o->small_region_penalizer;
    d.luminance_bins = o->luminance_bins;
    d.color_bins = o->color_bins;
    d.flow_bins = o->flow_bins;
    d.chunk_set_size = o->chunk_set_size;
    d.chunk_set_overlap = o->chunk_set_overlap;
    d.constraint_chunks = o->constraint_chunks;
    d.use_appearance = o->use_appearance != 0;
    d.use_flow = o->use_flow != 0;
    d.use_size_penalizer = o->use_size_penalizer != 0;
    d.compute_vectorization = o->compute_vectorization != 0;
    d.save_descriptors = o->save_descriptors != 0;
    std::unique_ptr<vsg_regionseg> r(new vsg_regionseg);
    r->impl.reset(new vsg::RegionSegmentationHost(d, width, height));
    r->W = width;
    r->H = height;
    *out = r.release();
  });
}

void vsg_regionseg_destroy(vsg_regionseg* r) { delete r; }

int vsg_regionseg_process_frame(vsg_regionseg* r, int flush, const uint8_t* seg_desc, size_t seg_len,
                             const uint8_t* bgr, size_t stride, const float* flow, int* num_results) {
  return Guard([&] {
    VSG_REQUIRE(r && num_results, VSG_ERR_INVALID, "null argument");
    if (seg_desc) {
      vsg::SegDesc d;
      VSG_REQUIRE(vsg::DecodeSegDesc(seg_desc, seg_len, &d), VSG_ERR_INVALID, "malformed SegmentationDesc");
      // The rasters index the frame, the flow field and the id image of the vectorisation: a message
      // from a dense unit of another size (or a corrupt one) must not get that far.
      VSG_REQUIRE(d.frame_width == r->W && d.frame_height == r->H, VSG_ERR_INVALID,
                  "SegmentationDesc of another frame size");
      for (const vsg::Region2DOut& reg : d.regions) {
        for (const vsg::Interval& iv : reg.raster) {
          VSG_REQUIRE(iv.y >= 0 && iv.y < r->H && iv.lx >= 0 && iv.lx <= iv.rx && iv.rx < r->W, VSG_ERR_INVALID,
                      "malformed SegmentationDesc: scan interval outside the frame");
        }
      }
      VSG_REQUIRE(bgr != nullptr && stride >= (size_t)r->W * 3, VSG_ERR_INVALID, "frame missing or stride too small");
      *num_results = r->impl->ProcessFrame(flush != 0, &d, bgr, stride, flow);
    } else {
      *num_results = r->impl->ProcessFrame(flush != 0, nullptr, bgr, stride, flow);
    }
  });
}

int vsg_regionseg_result_bytes(vsg_regionseg* r, int i, const uint8_t** data, size_t* len) {
  return Guard([&] {
    VSG_REQUIRE(r && data && len, VSG_ERR_INVALID, "null argument");
    VSG_REQUIRE(i >= 0 && i < r->impl->num_results(), VSG_ERR_INVALID, "result index");
    const std::string& b = r->impl->result_bytes(i);
    *data = reinterpret_cast<const uint8_t*>(b.data());
    *len = b.size();
  });
}

int vsg_bgr_to_lab(const uint8_t* bgr, size_t stride, int width, int height, uint8_t* lab) {
  return Guard([&] {
    VSG_REQUIRE(bgr && lab && width >= 1 && height >= 1 && stride >= (size_t)width * 3, VSG_ERR_INVALID,
                "bad argument");
    vsg::BgrToLab8(bgr, stride, width, height, lab);
  });
}

int vsg_debug_sort_pairs_timed(const uint32_t* keys, const uint32_t* values, int n, int end_bit,
                               uint32_t* keys_out, uint32_t* values_out, int device, int impl, int reps,
                               double* avg_us) {
  return Guard([&] {
    VSG_REQUIRE(keys && values && keys_out && values_out && n >= 0 && end_bit >= 1 && end_bit <= 32 && reps >= 1 &&
                    impl >= 0 && impl <= 2,
                VSG_ERR_INVALID, "bad argument");
    RequireDevice(device);
    DeviceGuard dg(ResolveDevice(device));
    if (avg_us) *avg_us = 0;
    if (n == 0) return;
    vsg::DevBuf<uint32_t> k((size_t)n), v((size_t)n), ko((size_t)n), vo((size_t)n);
    vsg::DevBuf<uint8_t> temp(vsg::SortPairsU32TempBytes(n));
    hipStream_t s = nullptr;
    VSG_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    struct Owner {
      hipStream_t& s;
      hipEvent_t &e0, &e1;
      ~Owner() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipStreamDestroy(s);
      }
    } owner{s, e0, e1};
    VSG_HIP(hipEventCreate(&e0));
    VSG_HIP(hipEventCreate(&e1));
    VSG_HIP(hipMemcpyAsync(k.get(), keys, (size_t)n * 4, hipMemcpyHostToDevice, s));
    VSG_HIP(hipMemcpyAsync(v.get(), values, (size_t)n * 4, hipMemcpyHostToDevice, s));
    auto sort = impl == 1 ? vsg::SortPairsU32Hand : impl == 2 ? vsg::SortPairsU32Lib : vsg::SortPairsU32;
    sort(temp.get(), temp.size(), k.get(), ko.get(), v.get(), vo.get(), n, end_bit, s);   // (warm-up)
    VSG_HIP(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) sort(temp.get(), temp.size(), k.get(), ko.get(), v.get(), vo.get(), n, end_bit, s);
    VSG_HIP(hipEventRecord(e1, s));
    VSG_HIP(hipMemcpyAsync(keys_out, ko.get(), (size_t)n * 4, hipMemcpyDeviceToHost, s));
    VSG_HIP(hipMemcpyAsync(values_out, vo.get(), (size_t)n * 4, hipMemcpyDeviceToHost, s));
    VSG_HIP(hipStreamSynchronize(s));
    float ms = 0;
    VSG_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (avg_us) *avg_us = (double)ms * 1e3 / reps;
  });
}

int vsg_debug_sort_pairs(const uint32_t* keys, const uint32_t* values, int n, int end_bit,
                         uint32_t* keys_out, uint32_t* values_out, int device) {
  return vsg_debug_sort_pairs_timed(keys, values, n, end_bit, keys_out, values_out, device, 0, 1, nullptr);
}

// ---- graph -------------------------------------------------------------------------------
int vsg_graph_create(int width, int height, int max_frames, int l1, int device, vsg_graph** out) {
  return Guard([&] {
    VSG_REQUIRE(out, VSG_ERR_INVALID, "null argument");
    RequireDevice(device);
    VSG_REQUIRE(width >= 2 && height >= 1 && width <= 65535 && height <= 65535, VSG_ERR_INVALID,
                "unsupported frame size");
    std::unique_ptr<vsg_graph> g(new vsg_graph);
    g->device = ResolveDevice(device);
    DeviceGuard dg(g->device);
    g->W = width;
    g->H = height;
    g->wh = (size_t)width * height;
    VSG_HIP(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    g->g.reset(new vsg::DenseGraphHip(width, height, max_frames, l1 != 0, g->stream));
    g->pre.reset(new vsg::Preprocessor(width, height, g->stream));
    std::memset(&g->timings, 0, sizeof(g->timings));
    *out = g.release();
  });
}

void vsg_graph_destroy(vsg_graph* g) {
  if (!g) return;
  (void)Guard([&] {
    DeviceGuard dg(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    vsg::QuiesceGuard quiesce;   // one device synchronisation for everything released below
    quiesce.Begin();
    g->g.reset();
    g->pre.reset();
    g->feats.clear();
    g->staging_bgr.release();
    g->staging_f32.release();
    g->staging_ids.release();
    g->flows_dev.clear();
    if (g->stream) (void)hipStreamDestroy(g->stream);
    g->stream = nullptr;
  });
  delete g;
}

static const int32_t* StageIds(vsg_graph* g, const int32_t* ids, int mem) {
  if (!ids) return nullptr;
  if (mem == VSG_MEM_DEVICE) return ids;
  g->staging_ids.ensure(g->wh);
  VSG_HIP(hipMemcpyAsync(g->staging_ids.get(), ids, g->wh * sizeof(int32_t), hipMemcpyHostToDevice,
                         g->stream));
  return g->staging_ids.get();
}

int vsg_graph_add_frame_bgr(vsg_graph* g, const uint8_t* bgr, size_t stride, int presmoothing,
                            const int32_t* constraint_ids, int mem) {
  return Guard([&] {
    VSG_REQUIRE(g && bgr, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(g->device);
    VSG_REQUIRE(stride >= (size_t)g->W * 3, VSG_ERR_INVALID, "stride smaller than a row");
    const uint8_t* dev = bgr;
    if (mem == VSG_MEM_HOST) {
      g->staging_bgr.ensure(stride * (size_t)g->H);
      VSG_HIP(hipMemcpyAsync(g->staging_bgr.get(), bgr,
                             stride * (size_t)(g->H - 1) + (size_t)g->W * 3, hipMemcpyHostToDevice,
                             g->stream));
      dev = g->staging_bgr.get();
    }
    auto feat = std::make_shared<vsg::DevBuf<float>>(3 * g->wh);
    g->pre->Run(dev, stride, presmoothing, feat->get());
    g->timings.preprocess_ms += g->pre->last_ms();
    g->timings.preprocess_launches += 1;
    const int32_t* ids = StageIds(g, constraint_ids, mem);
    g->g->AddFrame(feat->get(), ids);
    VSG_HIP(hipStreamSynchronize(g->stream));
    g->feats.push_back(feat);
    g->flows_dev.push_back(nullptr);
  });
}

int vsg_graph_add_frame_features(vsg_graph* g, const float* feat_in, const int32_t* constraint_ids,
                                 int mem) {
  return Guard([&] {
    VSG_REQUIRE(g && feat_in, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(g->device);
    const float* src = feat_in;
    if (mem == VSG_MEM_HOST) {
      g->staging_f32.ensure(3 * g->wh);
      VSG_HIP(hipMemcpyAsync(g->staging_f32.get(), feat_in, 3 * g->wh * sizeof(float),
                             hipMemcpyHostToDevice, g->stream));
      src = g->staging_f32.get();
    }
    auto feat = std::make_shared<vsg::DevBuf<float>>(3 * g->wh);
    vsg::LaunchInterleavedToPlanar(src, g->wh, feat->get(), g->stream);
    const int32_t* ids = StageIds(g, constraint_ids, mem);
    g->g->AddFrame(feat->get(), ids);
    VSG_HIP(hipStreamSynchronize(g->stream));
    g->feats.push_back(feat);
    g->flows_dev.push_back(nullptr);
  });
}

int vsg_graph_add_virtual_frame(vsg_graph* g, const int32_t* constraint_ids, int mem) {
  return Guard([&] {
    VSG_REQUIRE(g && constraint_ids, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(g->device);
    // max label: scan on the host (ids come from a SegmentationDesc, i.e. host data, or are small)
    std::vector<int32_t> host(g->wh);
    if (mem == VSG_MEM_HOST) {
      std::memcpy(host.data(), constraint_ids, g->wh * sizeof(int32_t));
    } else {
      VSG_HIP(hipMemcpy(host.data(), constraint_ids, g->wh * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    int max_label = 0;
    for (int32_t v : host) max_label = std::max(max_label, v + 1);
    const int32_t* ids = StageIds(g, constraint_ids, mem);
    g->g->AddVirtualFrame(ids, std::max(max_label, 1));
    VSG_HIP(hipStreamSynchronize(g->stream));
    g->feats.push_back(nullptr);
    g->flows_dev.push_back(nullptr);
  });
}

int vsg_graph_add_temporal(vsg_graph* g, const float* flow, int is_virtual, int mem) {
  return Guard([&] {
    VSG_REQUIRE(g, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(g->device);
    const int nf = g->g->num_frames();
    VSG_REQUIRE(nf >= 2, VSG_ERR_STATE, "temporal edges need two slices");
    const float* fdev = nullptr;
    if (flow) {
      auto fd = std::make_shared<vsg::DevBuf<float>>(2 * g->wh);
      VSG_HIP(hipMemcpyAsync(fd->get(), flow, 2 * g->wh * sizeof(float),
                             mem == VSG_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                             g->stream));
      g->flows_dev[nf - 1] = fd;
      fdev = fd->get();
    }
    const float* cur = g->feats[nf - 1] ? g->feats[nf - 1]->get() : nullptr;
    const float* prev = g->feats[nf - 2] ? g->feats[nf - 2]->get() : nullptr;
    VSG_REQUIRE(is_virtual || (cur && prev), VSG_ERR_STATE, "real temporal edges need two real slices");
    g->g->AddTemporal(cur, prev, fdev, is_virtual != 0);
    VSG_HIP(hipStreamSynchronize(g->stream));
  });
}

int vsg_graph_finish_building(vsg_graph* g) {
  return Guard([&] {
    VSG_REQUIRE(g, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(g->device);
    g->g->FinishBuilding();
  });
}

int vsg_graph_segment_spatially(vsg_graph* g) {
  return Guard([&] {
    VSG_REQUIRE(g, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(g->device);
    g->g->SegmentSpatially();
  });
}

int vsg_graph_segment(vsg_graph* g, int min_region_size, int force_constraints) {
  return Guard([&] {
    VSG_REQUIRE(g, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(g->device);
    g->g->Segment(min_region_size, force_constraints != 0);
  });
}

int vsg_graph_obtain_results(vsg_graph* g, int use_flows, int enforce_n4,
                             int enforce_spatial_connectedness) {
  return Guard([&] {
    VSG_REQUIRE(g, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(g->device);
    std::vector<const float*> flows;
    if (use_flows) {
      for (auto& f : g->flows_dev) flows.push_back(f ? f->get() : nullptr);
    }
    g->g->ObtainResults(use_flows ? &flows : nullptr, enforce_n4 != 0,
                        enforce_spatial_connectedness != 0);
  });
}

int vsg_graph_num_frames(const vsg_graph* g) { return g ? g->g->num_frames() : VSG_ERR_INVALID; }
int vsg_graph_num_regions(const vsg_graph* g) {
  return g ? (int)g->g->regions().size() : VSG_ERR_INVALID;
}
int64_t vsg_graph_num_neighbor_links(const vsg_graph* g) {
  if (!g) return VSG_ERR_INVALID;
  int64_t n = 0;
  for (const auto& r : g->g->regions()) n += (int64_t)r.neighbors.size();
  return n;
}

int vsg_graph_region_sizes(const vsg_graph* g, int32_t* sizes, int32_t* constrained_ids) {
  return Guard([&] {
    VSG_REQUIRE(g && sizes && constrained_ids, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(g->device);
    const auto& regs = g->g->regions();
    for (size_t i = 0; i < regs.size(); ++i) {
      sizes[i] = regs[i].size;
      constrained_ids[i] = regs[i].constrained_id;
    }
  });
}

int vsg_graph_index_image(const vsg_graph* g, int t, int32_t* out) {
  return Guard([&] {
    VSG_REQUIRE(g && out, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(g->device);
    for (size_t k = 0; k < g->wh; ++k) out[k] = -1;
    for (const auto& r : g->g->regions()) {
      if (!r.has_raster) continue;
      for (const auto& sl : r.raster) {
        if (sl.frame != t) continue;
        for (const auto& iv : sl.raster) {
          for (int x = iv.lx; x <= iv.rx; ++x) out[(size_t)iv.y * g->W + x] = r.index;
        }
      }
    }
  });
}

int vsg_graph_get_regions(vsg_graph* g, const vsg_region** regions, size_t* num_regions,
                          const int32_t** nbr_csr_ptr, const int32_t** nbr_csr_idx) {
  return Guard([&] {
    VSG_REQUIRE(g && regions && num_regions && nbr_csr_ptr && nbr_csr_idx, VSG_ERR_INVALID,
                "null argument");
    const auto& regs = g->g->regions();
    g->out_regions.resize(regs.size());
    g->out_nbr_ptr.assign(regs.size() + 1, 0);
    g->out_nbr_idx.clear();
    for (size_t i = 0; i < regs.size(); ++i) {
      const vsg::RegionInfo& r = regs[i];
      VSG_REQUIRE(r.index == (int)i, VSG_ERR_INTERNAL, "region table out of order");
      vsg_region& o = g->out_regions[i];
      o.index = r.index;
      o.size = r.size;
      o.constrained_id = r.constrained_id;
      const bool has = r.has_raster && !r.raster.empty();
      o.first_frame = has ? r.raster.front().frame : -1;
      o.last_frame = has ? r.raster.back().frame : -1;
      g->out_nbr_idx.insert(g->out_nbr_idx.end(), r.neighbors.begin(), r.neighbors.end());
      g->out_nbr_ptr[i + 1] = (int32_t)g->out_nbr_idx.size();
    }
    if (g->out_nbr_idx.empty()) g->out_nbr_idx.push_back(0);   // a valid pointer for an empty list
    *regions = g->out_regions.data();
    *num_regions = regs.size();
    *nbr_csr_ptr = g->out_nbr_ptr.data();
    *nbr_csr_idx = g->out_nbr_idx.data();
  });
}

int vsg_graph_get_intervals(vsg_graph* g, int frame, const vsg_interval** intervals, size_t* n) {
  return Guard([&] {
    VSG_REQUIRE(g && intervals && n, VSG_ERR_INVALID, "null argument");
    VSG_REQUIRE(frame >= 0 && frame < g->g->num_frames(), VSG_ERR_INVALID, "slice index");
    g->out_intervals.clear();
    for (const vsg::RegionInfo& r : g->g->regions()) {
      if (!r.has_raster) continue;
      auto it = std::lower_bound(r.raster.begin(), r.raster.end(), frame,
                                 [](const vsg::RasterSlice& s, int f) { return s.frame < f; });
      if (it == r.raster.end() || it->frame != frame) continue;
      for (const vsg::Interval& iv : it->raster) {
        g->out_intervals.push_back(vsg_interval{r.index, iv.y, iv.lx, iv.rx});
      }
    }
    *n = g->out_intervals.size();
    if (g->out_intervals.empty()) g->out_intervals.push_back(vsg_interval{-1, 0, 0, 0});
    *intervals = g->out_intervals.data();
  });
}

int vsg_graph_smoothed(vsg_graph* g, int t, float* out) {
  return Guard([&] {
    VSG_REQUIRE(g && out, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(g->device);
    VSG_REQUIRE(t >= 0 && t < (int)g->feats.size() && g->feats[t], VSG_ERR_INVALID, "slice index");
    vsg::DevBuf<float> tmp(3 * g->wh);
    vsg::LaunchPlanarToInterleaved(g->feats[t]->get(), g->wh, tmp.get(), g->stream);
    VSG_HIP(hipMemcpyAsync(out, tmp.get(), 3 * g->wh * sizeof(float), hipMemcpyDeviceToHost,
                           g->stream));
    VSG_HIP(hipStreamSynchronize(g->stream));
  });
}

int vsg_graph_spatial_buckets(vsg_graph* g, int t, uint16_t* out) {
  return Guard([&] {
    VSG_REQUIRE(g && out, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(g->device);
    g->g->CopySpatialBuckets(t, out);
  });
}

int vsg_graph_temporal_buckets(vsg_graph* g, int t, uint16_t* out, int32_t* prev_idx) {
  return Guard([&] {
    VSG_REQUIRE(g && out && prev_idx, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(g->device);
    g->g->CopyTemporalBuckets(t, out, prev_idx);
  });
}

int vsg_graph_node_roots(vsg_graph* g, int32_t* out) {
  return Guard([&] {
    VSG_REQUIRE(g && out, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(g->device);
    g->g->CopyNodeRoots(out);
  });
}

int vsg_graph_merge_stats(const vsg_graph* g, int64_t* s3) {
  return Guard([&] {
    VSG_REQUIRE(g && s3, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(g->device);
    const auto& t = g->g->timings();
    s3[0] = t.merges[0];
    s3[1] = t.merges[1];
    s3[2] = t.merges[2];
  });
}

int vsg_graph_timings(const vsg_graph* g, vsg_timings* t) {
  return Guard([&] {
    VSG_REQUIRE(g && t, VSG_ERR_INVALID, "null argument");
    DeviceGuard dg(g->device);
    *t = g->timings;
    const auto& gt = g->g->timings();
    t->merge_ms = gt.merge_ms;
    t->readout_ms = gt.readout_ms;
    t->host_post_ms = gt.host_post_ms;
    t->edges_total = gt.edges_total;
    t->edges_active = gt.edges_active;
    t->merges = gt.merges[0] + gt.merges[1] + gt.merges[2];
    t->wave_kernel_ms = gt.wave_ms;
    t->wave_kernel_launches = gt.wave_launches;
    t->wave_kernel_edges = gt.wave_edges;
    t->filter_kernel_ms = gt.filter_ms;
    t->filter_kernel_launches = gt.filter_launches;
    t->spine_kernel_ms = gt.spine_ms;
    t->spine_kernel_launches = gt.spine_launches;
    t->spine_kernel_edges = gt.spine_edges;
  });
}

int vsg_graph_diagnostics(const vsg_graph* g, vsg_diagnostics* d) {
  return Guard([&] {
    VSG_REQUIRE(g && d, VSG_ERR_INVALID, "null argument");
    FillDiagnostics(g->g->timings(), d);
  });
}

}  // extern "C"
