// stream.h -- DenseSegmentationHip: MI355X drop-in for segmentation::DenseSegmentation
// (segmentation/dense_segmentation.{h,cpp}): chunked streaming over-segmentation with a two frame
// overlap, the first overlap frame re-entering the next chunk as a *virtual* slice and the second
// as a *constrained* slice.
#ifndef VSG_STREAM_H_
#define VSG_STREAM_H_

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vsg.h"
#include "dense_graph.h"

namespace vsg {

// Frame preprocessing shared by the stream and the graph level C entry points
// (DenseSegmentation::PreprocessFeatures, dense_segmentation.cpp:164-198).
class Preprocessor {
 public:
  Preprocessor(int W, int H, hipStream_t stream);
  // bgr_dev: device pointer to H rows of `stride` bytes.  out: 3 planes of W*H f32 (device).
  void Run(const uint8_t* bgr_dev, size_t stride, int presmoothing, float* out_planar_dev);
  float last_ms() const { return last_ms_; }

 private:
  struct Lut {
    DevBuf<float> table;
    float scale = 0;
  };
  const Lut& GetLut(int umin, int umax);
  int W_, H_;
  hipStream_t stream_;
  DevBuf<int> minmax_dev_;
  std::map<std::pair<int, int>, std::unique_ptr<Lut>> luts_;
  bool space_uploaded_ = false;
  float last_ms_ = 0;
};

// Recycles the per-frame device planes (features: 3 W*H f32, flow: 2 W*H f32).  hipMalloc and
// above all hipFree synchronise the device; a stream frees chunk_size-1 feature and flow planes at
// every chunk boundary (measured: 8 ms per 1080p chunk), so planes go back to the pool instead and
// are only released with the handle.
class PlanePool : public std::enable_shared_from_this<PlanePool> {
 public:
  typedef std::shared_ptr<DevBuf<float>> Plane;
  Plane Take(size_t n) {
    DevBuf<float>* b = nullptr;
    {
      std::lock_guard<std::mutex> lock(mu_);
      auto it = free_.find(n);
      if (it != free_.end()) {
        b = it->second;
        free_.erase(it);
      }
    }
    if (!b) b = new DevBuf<float>(n);
    std::weak_ptr<PlanePool> self = shared_from_this();
    return Plane(b, [self](DevBuf<float>* p) {
      if (auto pool = self.lock()) {
        std::lock_guard<std::mutex> lock(pool->mu_);
        pool->free_.emplace(p->size(), p);
      } else {
        delete p;
      }
    });
  }
  ~PlanePool() {
    for (auto& kv : free_) delete kv.second;
  }

 private:
  std::mutex mu_;
  std::multimap<size_t, DevBuf<float>*> free_;
};

class DenseSegmentationHip {
 public:
  DenseSegmentationHip(const vsg_options& o, int W, int H);
  ~DenseSegmentationHip();

  // DenseSegmentation::ProcessFrame (dense_segmentation.cpp:108-162).
  int ProcessFrame(bool flush, const uint8_t* bgr, size_t stride, const float* flow,
                   bool has_flow_stream, int mem);
  int ChunkSize() const { return options_.chunk_size; }

  int num_results() const { return (int)results_.size(); }
  const std::string& result_bytes(int i);
  const SegDesc& result(int i) const { return *results_[i]; }
  void last_merge_stats(int64_t* s3) const;
  const vsg_timings& last_timings() const { return last_timings_; }
  const GraphTimings& last_graph_timings() const { return last_graph_timings_; }
  void CopyLastSmoothed(float* out_interleaved_host);
  int W() const { return W_; }
  int H() const { return H_; }

  // Fresh stream in the middle of a video whose halo arrives later (see vsg.h).
  void ExpectHalo();
  // Back to the state right after construction, keeping every device allocation.
  void Restart();
  void ExportHalo(const int32_t** virt, const int32_t** cons, int64_t scalars[4]);
  void ImportHalo(const int32_t* virt, const int32_t* cons, int mem, const int64_t scalars[4]);

 private:
  typedef std::shared_ptr<DevBuf<float>> DevPlane;
  int MinRegionSize() const;
  void ChunkBoundaryOutput(bool flush);
  void SegmentAndOutputChunk(bool flush);
  void StartConstrainedGraph(const int32_t* virt_ids_dev, const int32_t* cons_ids_dev,
                             int max_label);
  void Retrieve(int frame, bool output_hierarchy, SegDesc* desc) const;

  QuiesceGuard quiesce_;   // first member: spans the release of every buffer below (device_cache.h)
  vsg_options options_;
  int W_, H_;
  size_t wh_;
  hipStream_t stream_ = nullptr;
  std::unique_ptr<DenseGraphHip> graph_;
  std::unique_ptr<Preprocessor> pre_;
  std::shared_ptr<PlanePool> planes_;
  bool graph_open_ = false;

  int input_frames_ = 0;
  int chunk_id_ = 0;
  int seg_chunk_id_ = 0;   // chunk id the current Segmentation object was created with
  int overlap_frames_ = 2;
  int constraint_frames_ = 1;
  int max_region_id_ = 0;
  int num_output_frames_ = 0;
  int curr_chunk_start_ = 0;
  bool assigned_constrained_ids_ = false;

  std::vector<DevPlane> feature_buffer_;
  std::vector<DevPlane> flow_dev_buffer_;     // W*H*2 f32 on the device, null = empty flow
  bool forget_pending_ = false;   // Restart: forget what the graph learned unless the video continues
  bool halo_deferred_ = false;    // ExpectHalo(): frames may precede ImportHalo()
  bool flow_stream_seen_ = false;
  int64_t frames_fed_ = 0;   // frames handed to this handle (has_flow_stream must not change)

  DevBuf<uint8_t> staging_bgr_;
  DevBuf<float> staging_flow_;
  DevBuf<int32_t> halo_ids_dev_[2];
  int32_t* halo_ids_host_ = nullptr;   // pinned staging of the two planes
  bool halo_valid_ = false;
  // pending import (multi-GPU chunk chain)
  bool pending_import_ = false;
  int pending_max_label_ = 0;

  std::vector<std::unique_ptr<SegDesc>> overlap_segmentations_;
  std::vector<std::unique_ptr<SegDesc>> results_;
  std::vector<std::string> encoded_;
  int64_t last_merge_stats_[3] = {0, 0, 0};
  vsg_timings last_timings_;
  GraphTimings last_graph_timings_;   // the graph's own record of the last segmented chunk (diagnostics)
  vsg_timings accum_;   // preprocess / edge time accumulated while the chunk is being built
};

}  // namespace vsg

#endif  // VSG_STREAM_H_
