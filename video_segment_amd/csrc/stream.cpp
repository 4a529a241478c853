// stream.cpp -- see stream.h.  Reference behaviour restated from
// segmentation/dense_segmentation.cpp:50-432 and segmentation/segmentation.cpp:392-773.
#include "stream.h"

#include <atomic>
#include <exception>
#include <thread>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace vsg {

namespace {
double NowMs() {
  using clk = std::chrono::steady_clock;
  return std::chrono::duration<double, std::milli>(clk::now().time_since_epoch()).count();
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// Preprocessor
// ---------------------------------------------------------------------------------------------
Preprocessor::Preprocessor(int W, int H, hipStream_t stream) : W_(W), H_(H), stream_(stream) {
  minmax_dev_.alloc(2);
}

// Bilateral tables exactly as imagefilter/image_filter.cpp:208-250 builds them on the host:
// space weights and the 12288-bin exp LUT use the double libm exp() rounded to float.
const Preprocessor::Lut& Preprocessor::GetLut(int umin, int umax) {
  auto key = std::make_pair(umin, umax);
  auto it = luts_.find(key);
  if (it != luts_.end()) return *it->second;
  const float c255 = (float)(1.0 / 255.0);
  const double min_val = (double)((float)umin * c255);
  const double max_val = (double)((float)umax * c255);
  const int cn = 3;
  const float sigma_color = 0.25f;
  const float diff_range = std::max<float>(
      1e-3f, (float)((max_val - min_val) * (max_val - min_val) * cn * (double)1.02f));
  const int num_bins = (1 << 12) * cn;
  const float scale = (float)num_bins / diff_range;
  const float color_coeff = (float)(-0.5 / (double)(sigma_color * sigma_color));
  std::vector<float> lut(num_bins);
  bool zero_reached = false;
  for (int i = 0; i < num_bins; ++i) {
    if (!zero_reached) {
      lut[i] = (float)std::exp((double)((float)i / scale * color_coeff));
      zero_reached = ((double)lut[i] < 1e-10);
    } else {
      lut[i] = 0;
    }
  }
  std::unique_ptr<Lut> l(new Lut);
  l->table.alloc((size_t)num_bins);
  l->scale = scale;
  VSG_HIP(hipMemcpyAsync(l->table.get(), lut.data(), lut.size() * sizeof(float),
                         hipMemcpyHostToDevice, stream_));
  VSG_HIP(hipStreamSynchronize(stream_));
  if (luts_.size() > 64) luts_.clear();   // bounded cache
  auto res = luts_.emplace(key, std::move(l));
  return *res.first->second;
}

void Preprocessor::Run(const uint8_t* bgr_dev, size_t stride, int presmoothing, float* out) {
  const double t0 = NowMs();
  if (presmoothing == 0) {
    LaunchConvertPlanar(bgr_dev, stride, W_, H_, out, stream_);
  } else if (presmoothing == 1) {
    // cv::getGaussianKernel(3, 1.5, CV_32F) (OpenCV 2.4 smooth.cpp; un-vendored, parity unpinned):
    // exp in double, stored as float, normalised by the double sum of the floats
    float cf[3];
    double sum = 0;
    const double scale2x = -0.5 / (1.5 * 1.5);
    for (int i = 0; i < 3; ++i) {
      const double x = i - 1.0;
      cf[i] = (float)std::exp(scale2x * x * x);
      sum += cf[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < 3; ++i) cf[i] = (float)(cf[i] * sum);
    LaunchGaussian3(bgr_dev, stride, W_, H_, cf[1], cf[0], out, stream_);
  } else {
    VSG_REQUIRE(presmoothing == 2, -1, "presmoothing has to be 0 (none), 1 (gaussian) or 2 (bilateral)");
    if (!space_uploaded_) {
      // space_weights, image_filter.cpp:216-225 (radius 4, sigma_space 3.0)
      float w[49];
      const float sigma_space = 3.0f;
      const int radius = (int)(sigma_space * 1.5f);
      const float coeff = -0.5f / (sigma_space * sigma_space);
      int k = 0;
      for (int i = -radius; i <= radius; ++i) {
        for (int j = -radius; j <= radius; ++j) {
          const int r2 = i * i + j * j;
          if (r2 > radius * radius) continue;
          w[k++] = (float)std::exp((double)(coeff * (float)r2));
        }
      }
      VSG_REQUIRE(k == 49, -4, "unexpected bilateral window");
      UploadSpaceWeights(w, stream_);
      space_uploaded_ = true;
    }
    LaunchMinMax(bgr_dev, stride, W_, H_, minmax_dev_.get(), stream_);
    int mm[2] = {0, 0};
    VSG_HIP(hipMemcpyAsync(mm, minmax_dev_.get(), sizeof(mm), hipMemcpyDeviceToHost, stream_));
    VSG_HIP(hipStreamSynchronize(stream_));
    const Lut& lut = GetLut(mm[0], mm[1]);
    LaunchBilateral(bgr_dev, stride, W_, H_, lut.table.get(), lut.scale, out, stream_);
  }
  last_ms_ = (float)(NowMs() - t0);
}

// ---------------------------------------------------------------------------------------------
// DenseSegmentationHip
// ---------------------------------------------------------------------------------------------
DenseSegmentationHip::DenseSegmentationHip(const vsg_options& o, int W, int H)
    : options_(o), W_(W), H_(H), wh_((size_t)W * H) {
  VSG_REQUIRE(options_.chunk_size >= 3, -1, "Chunk size needs to be at least 3 frames.");
  overlap_frames_ = (int)(options_.chunk_overlap_ratio * (float)options_.chunk_size + 0.5f);
  overlap_frames_ = std::min(overlap_frames_, 2);
  VSG_REQUIRE(overlap_frames_ < options_.chunk_size, -1, "Overlap needs to be smaller than chunk_size.");
  VSG_REQUIRE(overlap_frames_ == 2, -1, "chunk_overlap_ratio too small: the reference needs a 2 frame overlap");
  VSG_REQUIRE(options_.num_constraint_frames >= 1, -1, "num_constraint_frames >= 1");
  constraint_frames_ = std::min(options_.num_constraint_frames, overlap_frames_ - 1);
  // The caller (capi.cpp) has bound this thread to the handle's device.
  VSG_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  graph_.reset(new DenseGraphHip(W, H, options_.chunk_size + 1, options_.color_distance == 0, stream_));
  pre_.reset(new Preprocessor(W, H, stream_));
  planes_ = std::make_shared<PlanePool>();
  std::memset(&last_timings_, 0, sizeof(last_timings_));
  std::memset(&accum_, 0, sizeof(accum_));
}

DenseSegmentationHip::~DenseSegmentationHip() {
  quiesce_.Begin();   // one device synchronisation for all the buffers released below
  if (stream_) {
    (void)hipStreamSynchronize(stream_);
    graph_.reset();
    pre_.reset();
    feature_buffer_.clear();
    flow_dev_buffer_.clear();
    planes_.reset();
    (void)hipStreamDestroy(stream_);
  }
  if (halo_ids_host_) CacheFree(halo_ids_host_);
}

// dense_segmentation.cpp:268-279: float product truncated to int.
int DenseSegmentationHip::MinRegionSize() const {
  return (int)(options_.frac_min_region_size * (float)W_ * options_.frac_min_region_size *
               (float)H_ * (float)options_.chunk_size);
}

int DenseSegmentationHip::ProcessFrame(bool flush, const uint8_t* bgr, size_t stride,
                                       const float* flow, bool has_flow_stream, int mem) {
  results_.clear();
  encoded_.clear();
  if (forget_pending_) {   // first frame of another video after Restart
    graph_->ForgetLearned();
    forget_pending_ = false;
  }
  if (!graph_open_) {
    graph_->Reset(options_.chunk_size);
    graph_open_ = true;
    seg_chunk_id_ = chunk_id_;
  }
  if (bgr) {
    VSG_REQUIRE(stride >= (size_t)W_ * 3, -1, "stride smaller than a row");
    const uint8_t* bgr_dev = bgr;
    if (mem == VSG_MEM_HOST) {
      staging_bgr_.ensure(stride * (size_t)H_);
      VSG_HIP(hipMemcpyAsync(staging_bgr_.get(), bgr, stride * (size_t)(H_ - 1) + (size_t)W_ * 3,
                             hipMemcpyHostToDevice, stream_));
      bgr_dev = staging_bgr_.get();
    }
    DevPlane feat = planes_->Take(3 * wh_);
    pre_->Run(bgr_dev, stride, options_.presmoothing, feat->get());
    accum_.preprocess_ms += pre_->last_ms();
    accum_.preprocess_launches += 1;

    // "Flow always has to be passed or be absent" (dense_segmentation.cpp:140): the unit either
    // has a flow stream for the whole video or it has none.
    VSG_REQUIRE(frames_fed_ == 0 || has_flow_stream == flow_stream_seen_, -1,
                "has_flow_stream changed in the middle of a stream");
    ++frames_fed_;
    if (has_flow_stream) {
      flow_stream_seen_ = true;
      if (frames_fed_ == 1 && !pending_import_) {
        flow_dev_buffer_.push_back(nullptr);
      } else {
        VSG_REQUIRE(flow != nullptr, -1, "Flow always has to be passed or be absent.");
        // Deep copy (the caller's buffer is only valid during the call); the field stays on the
        // device: the edge kernels read it, the tube analysis samples it there.
        DevPlane fd = planes_->Take(2 * wh_);
        VSG_HIP(hipMemcpyAsync(fd->get(), flow, 2 * wh_ * sizeof(float),
                               mem == VSG_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                               stream_));
        flow_dev_buffer_.push_back(fd);
      }
    }

    const double te = NowMs();
    if (pending_import_) {
      // First frame after vsg_stream_import_halo: it is the constrained overlap frame.
      feature_buffer_.push_back(nullptr);   // virtual slot
      feature_buffer_.push_back(feat);
      if (has_flow_stream) {
        // keep [empty, flow] layout like after a chunk boundary
        DevPlane fd = flow_dev_buffer_.back();
        flow_dev_buffer_.clear();
        flow_dev_buffer_.push_back(nullptr);
        flow_dev_buffer_.push_back(fd);
      }
      curr_chunk_start_ = 1;
      if (halo_deferred_) {
        StartConstrainedGraph(nullptr, nullptr, 0);   // the labels follow (ImportHalo)
      } else {
        StartConstrainedGraph(halo_ids_dev_[0].get(), halo_ids_dev_[1].get(), pending_max_label_);
      }
      pending_import_ = false;
    } else {
      feature_buffer_.push_back(feat);
      graph_->AddFrame(feat->get(), nullptr);
      if (feature_buffer_.size() > 1) {
        const float* fl = flow_dev_buffer_.empty() ? nullptr : flow_dev_buffer_.back()->get();
        graph_->AddTemporal(feature_buffer_.end()[-1]->get(), feature_buffer_.end()[-2]->get(), fl,
                            false);
      }
    }
    VSG_HIP(hipStreamSynchronize(stream_));
    accum_.edges_ms += (float)(NowMs() - te);
    accum_.edge_launches += 1;
    ++input_frames_;
  }
  if (flush || (int)feature_buffer_.size() - curr_chunk_start_ >= options_.chunk_size) {
    ChunkBoundaryOutput(flush);
    return (int)results_.size();
  }
  return 0;
}

// AddVirtualImageConstrained + constrained AddFrame + virtual temporal edges
// (dense_segmentation.cpp:291-315).
void DenseSegmentationHip::StartConstrainedGraph(const int32_t* virt_ids_dev,
                                                 const int32_t* cons_ids_dev, int max_label) {
  graph_->Reset(curr_chunk_start_ + options_.chunk_size);
  graph_open_ = true;
  seg_chunk_id_ = chunk_id_;
  if (virt_ids_dev) {
    graph_->AddVirtualFrame(virt_ids_dev, std::max(max_label, 1));
  } else {
    graph_->AddVirtualFrameDeferred();
  }
  graph_->AddFrame(feature_buffer_[1]->get(), cons_ids_dev);
  const float* fl = flow_dev_buffer_.empty() ? nullptr : flow_dev_buffer_[1]->get();
  graph_->AddTemporal(nullptr, nullptr, fl, true);
}

void DenseSegmentationHip::ChunkBoundaryOutput(bool flush) {
  const double tb0 = NowMs();
  SegmentAndOutputChunk(flush);
  const double tb1 = NowMs();
  if (flush) {
    graph_open_ = false;
    return;
  }
  VSG_REQUIRE((int)overlap_segmentations_.size() == constraint_frames_ + 1, -4, "overlap size");
  // Render the two overlap segmentations to id images (SegmentationDescToIdImage): both at once,
  // into pinned memory (a pageable source made the two 8 MB copies 1.5 ms per chunk).
  if (!halo_ids_host_) {
    halo_ids_host_ = static_cast<int32_t*>(CacheAlloc(2 * wh_ * sizeof(int32_t), kCachePinned));
  }
  {
    auto render = [&](int k) {
      int32_t* ids = halo_ids_host_ + (size_t)k * wh_;
      std::fill(ids, ids + wh_, -1);
      RenderIdImage(*overlap_segmentations_[k], W_, ids);
    };
    // (a throwing render -- VSG_REQUIRE in RenderIdImage, bad_alloc -- must neither leave a joinable
    // thread behind nor escape from the second thread: both end in std::terminate)
    std::exception_ptr other_error;
    std::thread other([&] {
      try {
        render(1);
      } catch (...) {
        other_error = std::current_exception();
      }
    });
    try {
      render(0);
    } catch (...) {
      other.join();
      throw;
    }
    other.join();
    if (other_error) std::rethrow_exception(other_error);
  }
  for (int k = 0; k < 2; ++k) {
    halo_ids_dev_[k].ensure(wh_);
    VSG_HIP(hipMemcpyAsync(halo_ids_dev_[k].get(), halo_ids_host_ + (size_t)k * wh_, wh_ * sizeof(int32_t),
                           hipMemcpyHostToDevice, stream_));
  }
  // (the next write to the pinned planes is a whole chunk -- and several stream synchronisations -- away)
  halo_valid_ = true;
  const double tb2 = NowMs();
  StartConstrainedGraph(halo_ids_dev_[0].get(), halo_ids_dev_[1].get(), max_region_id_);
  overlap_segmentations_.clear();
  if (getenv("VSG_DEBUG_STATS")) {
    VSG_HIP(hipStreamSynchronize(stream_));
    std::fprintf(stderr, "[vsg] boundary: segment+output %.1f ms, halo planes %.1f ms, next graph start %.1f ms\n",
                 tb1 - tb0, tb2 - tb1, NowMs() - tb2);
  }
}

void DenseSegmentationHip::SegmentAndOutputChunk(bool flush) {
  std::vector<const float*> flows;   // device pointers, one per buffered slice
  const bool have_flows = !flow_dev_buffer_.empty();
  if (have_flows) {
    for (const auto& f : flow_dev_buffer_) flows.push_back(f ? f->get() : nullptr);
  }
  // RunOverSegmentation (segmentation.cpp:272-303)
  graph_->FinishBuilding();
  if (options_.two_stage_oversegment) graph_->SegmentSpatially();   // segmentation.cpp:280-283
  graph_->Segment(MinRegionSize(), true);
  graph_->ObtainResults(have_flows ? &flows : nullptr, options_.enforce_n4_connectivity != 0,
                        options_.enforce_spatial_connectedness != 0);
  const double t_host0 = NowMs();
  double t_h[6] = {0};
  const GraphTimings& gt = graph_->timings();
  last_merge_stats_[0] = gt.merges[0];
  last_merge_stats_[1] = gt.merges[1];
  last_merge_stats_[2] = gt.merges[2];

  const int buffered = (int)feature_buffer_.size();
  const int overlap_start = buffered - (flush ? 0 : overlap_frames_);
  const int last_output_frame = std::min(buffered - 1, overlap_start);
  VSG_REQUIRE(overlap_start > curr_chunk_start_, -3, "flush without buffered frames");
  const int max_result_frame = std::min(buffered - 1, last_output_frame + constraint_frames_);

  std::vector<RegionInfo>& regions = graph_->regions();
  const int lhs = 0, rhs = last_output_frame + 1;
  // ConstrainSegmentationToFrameInterval (segmentation.cpp:392-420)
  for (RegionInfo& r : regions) {
    if (!r.has_raster || r.raster.empty() || r.raster.front().frame >= rhs ||
        r.raster.back().frame < lhs) {
      r.removed = true;
    }
  }
  // AdjustRegionAreaToFrameInterval (segmentation.cpp:422-456)
  for (RegionInfo& r : regions) {
    if (!r.has_raster) continue;
    int inc = 0;
    for (const RasterSlice& sl : r.raster) {
      if (sl.frame < lhs || sl.frame >= rhs) inc -= RasterArea(sl.raster);
    }
    r.size += inc;
  }
  // AssignUniqueRegionIds (segmentation.cpp:537-582)
  const bool use_constraints = chunk_id_ > 0;
  assigned_constrained_ids_ = use_constraints;
  int max_id = -1;
  for (RegionInfo& r : regions) {
    r.region_id = (use_constraints && r.constrained_id >= 0) ? r.constrained_id
                                                             : r.index + max_region_id_;
    max_id = std::max(max_id, r.region_id);
  }
  max_region_id_ = std::max(max_region_id_, max_id + 1);

  t_h[0] = NowMs();
  const int chunk_size = last_output_frame - curr_chunk_start_ + 1;
  overlap_segmentations_.clear();
  const int hierarchy_frame_idx = num_output_frames_;
  // Per-frame results (RetrieveSegmentation3D, segmentation.cpp:458-535) and their serialized
  // form -- what the C ABI hands out -- are built here, inside the chunk boundary, one frame per
  // host thread (the frames only read the region table).
  const int num_result_frames = max_result_frame - curr_chunk_start_ + 1;
  std::vector<std::unique_ptr<SegDesc>> descs((size_t)num_result_frames);
  std::vector<std::string> wires((size_t)num_result_frames);
  {
    std::atomic<int> next(0);
    auto work = [&]() {
      for (int i = next.fetch_add(1); i < num_result_frames; i = next.fetch_add(1)) {
        const int frame_idx = curr_chunk_start_ + i;
        std::unique_ptr<SegDesc> desc(new SegDesc());
        Retrieve(frame_idx, frame_idx == curr_chunk_start_, desc.get());
        desc->chunk_size = chunk_size;
        desc->overlap_start = chunk_size;
        desc->hierarchy_frame_idx = hierarchy_frame_idx;
        if (frame_idx <= last_output_frame) wires[(size_t)i] = EncodeSegDesc(*desc);
        descs[(size_t)i] = std::move(desc);
      }
    };
    const int hw = (int)std::thread::hardware_concurrency();
    const int num_threads = std::max(1, std::min({num_result_frames, hw > 0 ? hw : 1, 32}));
    std::vector<std::thread> pool;
    for (int t = 1; t < num_threads; ++t) pool.emplace_back(work);
    work();
    for (std::thread& t : pool) t.join();
  }
  t_h[1] = NowMs();
  for (int i = 0; i < num_result_frames; ++i) {
    const int frame_idx = curr_chunk_start_ + i;
    std::unique_ptr<SegDesc>& desc = descs[(size_t)i];
    if (frame_idx < last_output_frame) {
      results_.push_back(std::move(desc));
      encoded_.push_back(std::move(wires[(size_t)i]));
      ++num_output_frames_;
      continue;
    }
    if (frame_idx == last_output_frame) {
      results_.push_back(std::unique_ptr<SegDesc>(new SegDesc(*desc)));
      encoded_.push_back(std::move(wires[(size_t)i]));
      ++num_output_frames_;
    }
    overlap_segmentations_.push_back(std::move(desc));
  }

  t_h[2] = NowMs();
  feature_buffer_.erase(feature_buffer_.begin(), feature_buffer_.begin() + last_output_frame);
  if (!flow_dev_buffer_.empty()) {
    flow_dev_buffer_.erase(flow_dev_buffer_.begin(), flow_dev_buffer_.begin() + last_output_frame);
  }
  curr_chunk_start_ = flush ? 0 : 1;
  if (!flush) {
    VSG_REQUIRE(overlap_frames_ == (int)feature_buffer_.size(), -4, "overlap buffer size");
    feature_buffer_[0].reset();
    if (!flow_dev_buffer_.empty()) {
      flow_dev_buffer_[0].reset();
    }
  }
  ++chunk_id_;
  t_h[3] = NowMs();
  if (getenv("VSG_DEBUG_STATS")) {
    std::fprintf(stderr, "[vsg] output: region table %.1f ms, retrieve+encode %.1f ms, collect %.1f ms, buffers %.1f ms\n",
                 t_h[0] - t_host0, t_h[1] - t_h[0], t_h[2] - t_h[1], t_h[3] - t_h[2]);
  }

  std::memset(&last_timings_, 0, sizeof(last_timings_));
  last_timings_.preprocess_ms = accum_.preprocess_ms;
  last_timings_.edges_ms = accum_.edges_ms;
  last_timings_.preprocess_launches = accum_.preprocess_launches;
  last_timings_.edge_launches = accum_.edge_launches;
  last_graph_timings_ = gt;
  last_timings_.merge_ms = gt.merge_ms;
  last_timings_.readout_ms = gt.readout_ms;
  last_timings_.host_post_ms = gt.host_post_ms + (float)(NowMs() - t_host0);
  last_timings_.edges_total = gt.edges_total;
  last_timings_.edges_active = gt.edges_active;
  last_timings_.merges = gt.merges[0] + gt.merges[1] + gt.merges[2];
  last_timings_.wave_kernel_ms = gt.wave_ms;
  last_timings_.wave_kernel_launches = gt.wave_launches;
  last_timings_.wave_kernel_edges = gt.wave_edges;
  last_timings_.filter_kernel_ms = gt.filter_ms;
  last_timings_.filter_kernel_launches = gt.filter_launches;
  last_timings_.spine_kernel_ms = gt.spine_ms;
  last_timings_.spine_kernel_launches = gt.spine_launches;
  last_timings_.spine_kernel_edges = gt.spine_edges;
  std::memset(&accum_, 0, sizeof(accum_));
}

// RetrieveSegmentation3D (segmentation.cpp:458-533, 671-773) without descriptors/vectorization.
void DenseSegmentationHip::Retrieve(int frame, bool output_hierarchy, SegDesc* desc) const {
  const std::vector<RegionInfo>& regions = graph_->regions();
  desc->frame_width = W_;
  desc->frame_height = H_;
  desc->chunk_id = seg_chunk_id_;
  desc->connectedness = options_.enforce_n4_connectivity ? 1 : 2;
  for (const RegionInfo& r : regions) {
    if (!r.has_raster) continue;
    auto it = std::lower_bound(r.raster.begin(), r.raster.end(), frame,
                               [](const RasterSlice& s, int f) { return s.frame < f; });
    if (it == r.raster.end() || it->frame != frame) continue;
    desc->regions.emplace_back();
    Region2DOut& o = desc->regions.back();
    o.id = r.region_id;
    o.raster = it->raster;
    MomentsFromRaster(o.raster, &o.moments);
  }
  if (assigned_constrained_ids_) {
    std::sort(desc->regions.begin(), desc->regions.end(),
              [](const Region2DOut& a, const Region2DOut& b) { return a.id < b.id; });
  }
  if (output_hierarchy) {
    desc->has_hierarchy = true;
    for (const RegionInfo& r : regions) {
      if (r.removed) continue;
      desc->hierarchy0.emplace_back();
      CompoundOut& c = desc->hierarchy0.back();
      c.id = r.region_id;
      c.size = r.size;
      for (int n : r.neighbors) {
        if (regions[n].removed) continue;
        c.neighbor_ids.push_back(regions[n].region_id);
      }
      if (assigned_constrained_ids_) std::sort(c.neighbor_ids.begin(), c.neighbor_ids.end());
      c.start_frame = r.raster.front().frame;
      c.end_frame = r.raster.back().frame;
    }
    if (assigned_constrained_ids_) {
      std::sort(desc->hierarchy0.begin(), desc->hierarchy0.end(),
                [](const CompoundOut& a, const CompoundOut& b) { return a.id < b.id; });
    }
  }
  // segmentation.cpp:527-532
  if (options_.compute_vectorization) ComputeFrameVectorization(desc);
}

const std::string& DenseSegmentationHip::result_bytes(int i) {
  if (encoded_.size() != results_.size()) {
    encoded_.clear();
    for (const auto& r : results_) encoded_.push_back(EncodeSegDesc(*r));
  }
  return encoded_[i];
}

void DenseSegmentationHip::last_merge_stats(int64_t* s3) const {
  s3[0] = last_merge_stats_[0];
  s3[1] = last_merge_stats_[1];
  s3[2] = last_merge_stats_[2];
}

void DenseSegmentationHip::CopyLastSmoothed(float* out) {
  VSG_REQUIRE(!feature_buffer_.empty() && feature_buffer_.back(), -3, "no buffered frame");
  DevBuf<float> tmp(3 * wh_);
  LaunchPlanarToInterleaved(feature_buffer_.back()->get(), wh_, tmp.get(), stream_);
  VSG_HIP(hipMemcpyAsync(out, tmp.get(), 3 * wh_ * sizeof(float), hipMemcpyDeviceToHost, stream_));
  VSG_HIP(hipStreamSynchronize(stream_));
}

void DenseSegmentationHip::ExportHalo(const int32_t** virt, const int32_t** cons,
                                      int64_t scalars[4]) {
  VSG_REQUIRE(halo_valid_, -3, "no chunk boundary has been processed yet");
  // (the planes are uploaded asynchronously at the boundary; whoever takes them reads them on a
  // stream of its own)
  VSG_HIP(hipStreamSynchronize(stream_));
  *virt = halo_ids_dev_[0].get();
  *cons = halo_ids_dev_[1].get();
  scalars[0] = max_region_id_;
  scalars[1] = chunk_id_;
  scalars[2] = num_output_frames_;
  scalars[3] = input_frames_;
}

void DenseSegmentationHip::ExpectHalo() {
  VSG_REQUIRE(frames_fed_ == 0 && !graph_open_ && !pending_import_, -3, "expect_halo needs a fresh stream");
  pending_import_ = true;
  halo_deferred_ = true;
  forget_pending_ = false;   // the same video goes on: what the graph learned about it stays
}

void DenseSegmentationHip::ImportHalo(const int32_t* virt, const int32_t* cons, int mem,
                                      const int64_t scalars[4]) {
  forget_pending_ = false;
  const bool late = halo_deferred_ && !pending_import_;   // frames were fed before the halo
  VSG_REQUIRE(late || (frames_fed_ == 0 && !graph_open_), -3, "import_halo needs a fresh stream");
  const hipMemcpyKind kind = mem == VSG_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  halo_ids_dev_[0].ensure(wh_);
  halo_ids_dev_[1].ensure(wh_);
  VSG_HIP(hipMemcpyAsync(halo_ids_dev_[0].get(), virt, wh_ * sizeof(int32_t), kind, stream_));
  VSG_HIP(hipMemcpyAsync(halo_ids_dev_[1].get(), cons, wh_ * sizeof(int32_t), kind, stream_));
  VSG_HIP(hipStreamSynchronize(stream_));
  max_region_id_ = (int)scalars[0];
  chunk_id_ = (int)scalars[1];
  num_output_frames_ = (int)scalars[2];
  pending_max_label_ = max_region_id_;
  if (late) {
    // input_frames_ counted the frames fed so far from zero; the constrained frame is fed again
    input_frames_ += (int)scalars[3] - 1;
    seg_chunk_id_ = chunk_id_;
    graph_->SetHaloLabels(halo_ids_dev_[0].get(), halo_ids_dev_[1].get(), std::max(max_region_id_, 1));
    halo_deferred_ = false;
  } else {
    input_frames_ = (int)scalars[3] - 1;   // the constrained frame is fed again
    pending_import_ = true;
    halo_deferred_ = false;
  }
}

void DenseSegmentationHip::Restart() {
  VSG_HIP(hipStreamSynchronize(stream_));
  results_.clear();
  encoded_.clear();
  overlap_segmentations_.clear();
  feature_buffer_.clear();
  flow_dev_buffer_.clear();
  graph_open_ = false;
  input_frames_ = 0;
  chunk_id_ = 0;
  seg_chunk_id_ = 0;
  max_region_id_ = 0;
  num_output_frames_ = 0;
  curr_chunk_start_ = 0;
  assigned_constrained_ids_ = false;
  pending_import_ = false;
  halo_deferred_ = false;
  halo_valid_ = false;
  flow_stream_seen_ = false;
  frames_fed_ = 0;
  forget_pending_ = true;   // unless the stream continues a video (ExpectHalo / ImportHalo)
  std::memset(&accum_, 0, sizeof(accum_));
}

}  // namespace vsg
