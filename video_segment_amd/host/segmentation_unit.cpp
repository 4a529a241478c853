// segmentation_unit.cpp -- see segmentation_unit.h.  Behaviour restated from the reference's
// DenseSegmentationUnit (segmentation/segmentation_unit.cpp:48-178): error conventions
// (LOG(ERROR) + return false from OpenStreams, CHECK-abort on contract violations), frame-set
// buffering until a chunk is segmented, the pts hand-over and the __STREAMING_SIZE__ marker.
#include "segmentation_unit.h"

#include <cstdio>
#include <cstring>

namespace segmentation {

// ---- minimal proto2 reader for SegmentationDesc -------------------------------------------
namespace {
struct Cursor {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  uint64_t Varint() {
    uint64_t v = 0;
    int shift = 0;
    while (p < end) {
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
      shift += 7;
      if (shift > 63) break;
    }
    ok = false;
    return 0;
  }
  // Returns the field number, sets wire type; length-delimited payload in [sub_p, sub_end).
  bool Next(int* field, int* wt, Cursor* sub, uint64_t* value) {
    if (p >= end || !ok) return false;
    const uint64_t tag = Varint();
    *field = (int)(tag >> 3);
    *wt = (int)(tag & 7);
    if (*wt == 0) {
      *value = Varint();
    } else if (*wt == 2) {
      const uint64_t n = Varint();
      if (!ok || n > (uint64_t)(end - p)) {
        ok = false;
        return false;
      }
      sub->p = p;
      sub->end = p + n;
      p += n;
    } else if (*wt == 5) {
      if (end - p < 4) { ok = false; return false; }
      p += 4;
    } else if (*wt == 1) {
      if (end - p < 8) { ok = false; return false; }
      p += 8;
    } else {
      ok = false;
      return false;
    }
    return ok;
  }
};
}  // namespace

bool SegmentationDesc::ToIdImage(int width, int height, std::vector<int32_t>* out) const {
  out->assign((size_t)width * height, -1);
  Cursor top{reinterpret_cast<const uint8_t*>(wire.data()),
             reinterpret_cast<const uint8_t*>(wire.data()) + wire.size()};
  int f, wt;
  uint64_t v;
  Cursor region{nullptr, nullptr};
  while (top.Next(&f, &wt, &region, &v)) {
    if (f != 2 || wt != 2) continue;            // SegmentationDesc.region = 2
    int id = -1;
    Cursor raster{nullptr, nullptr}, sub{nullptr, nullptr};
    bool has_raster = false;
    while (region.Next(&f, &wt, &sub, &v)) {
      if (f == 1 && wt == 0) id = (int)(int64_t)v;          // Region2D.id
      if (f == 3 && wt == 2) { raster = sub; has_raster = true; }   // Region2D.raster
    }
    if (!region.ok) return false;
    if (!has_raster) continue;
    Cursor scan{nullptr, nullptr};
    while (raster.Next(&f, &wt, &scan, &v)) {
      if (f != 1 || wt != 2) continue;          // Rasterization.scan_inter = 1
      int y = 0, lx = 0, rx = -1;
      Cursor none{nullptr, nullptr};
      while (scan.Next(&f, &wt, &none, &v)) {
        if (wt != 0) continue;
        if (f == 1) y = (int)(int64_t)v;
        if (f == 2) lx = (int)(int64_t)v;
        if (f == 3) rx = (int)(int64_t)v;
      }
      if (!scan.ok || y < 0 || y >= height || lx < 0 || rx >= width) return false;
      for (int x = lx; x <= rx; ++x) (*out)[(size_t)y * width + x] = id;
    }
    if (!raster.ok) return false;
  }
  return top.ok;
}

int SegmentationDesc::NumRegions() const {
  Cursor top{reinterpret_cast<const uint8_t*>(wire.data()),
             reinterpret_cast<const uint8_t*>(wire.data()) + wire.size()};
  int f, wt, n = 0;
  uint64_t v;
  Cursor sub{nullptr, nullptr};
  while (top.Next(&f, &wt, &sub, &v)) n += (f == 2 && wt == 2);
  return top.ok ? n : -1;
}

int SegmentationDesc::NumHierarchyLevels() const {
  Cursor top{reinterpret_cast<const uint8_t*>(wire.data()),
             reinterpret_cast<const uint8_t*>(wire.data()) + wire.size()};
  int f, wt, n = 0;
  uint64_t v;
  Cursor sub{nullptr, nullptr};
  while (top.Next(&f, &wt, &sub, &v)) n += (f == 3 && wt == 2);   // SegmentationDesc.hierarchy = 3
  return top.ok ? n : -1;
}

bool SegmentationDesc::FrameSize(int* width, int* height) const {
  Cursor top{reinterpret_cast<const uint8_t*>(wire.data()),
             reinterpret_cast<const uint8_t*>(wire.data()) + wire.size()};
  int f, wt, seen = 0;
  uint64_t v;
  Cursor sub{nullptr, nullptr};
  while (top.Next(&f, &wt, &sub, &v)) {
    if (wt != 0) continue;
    if (f == 4) { *width = (int)(int64_t)v; seen |= 1; }
    if (f == 5) { *height = (int)(int64_t)v; seen |= 2; }
  }
  return top.ok && seen == 3;
}

// ---- DenseSegmentation (host class over the C ABI) -----------------------------------------------
DenseSegmentation::DenseSegmentation(const DenseSegmentationOptions& options, int frame_width,
                                     int frame_height, int device)
    : options_(options), frame_width_(frame_width), frame_height_(frame_height) {
  // What the HIP path does not implement is rejected loudly (the reference would run it).
  VF_CHECK(!options_.thin_structure_suppression,
           "thin_structure_suppression is not supported by the HIP over-segmentation path");
  vsg_options o;
  vsg_default_options(&o);
  o.presmoothing = (int)options_.presmoothing;
  o.frac_min_region_size = options_.frac_min_region_size;
  o.chunk_size = options_.chunk_size;
  o.chunk_overlap_ratio = options_.chunk_overlap_ratio;
  o.num_constraint_frames = options_.num_constraint_frames;
  o.enforce_n4_connectivity = options_.enforce_n4_connectivity ? 1 : 0;
  o.enforce_spatial_connectedness = options_.enforce_spatial_connectedness ? 1 : 0;
  o.color_distance = (int)options_.color_distance;
  o.two_stage_oversegment = options_.two_stage_oversegment ? 1 : 0;
  o.compute_vectorization = options_.compute_vectorization ? 1 : 0;
  o.device = device;
  if (vsg_stream_create(&o, frame_width_, frame_height_, &stream_) != VSG_OK) stream_ = nullptr;
}

DenseSegmentation::~DenseSegmentation() {
  if (stream_) vsg_stream_destroy(stream_);
}

int DenseSegmentation::ProcessFrame(bool flush, const std::vector<MatView>* features,
                                    const MatView* flow,
                                    std::vector<std::unique_ptr<SegmentationDesc>>* results) {
  VF_CHECK(stream_ != nullptr, "DenseSegmentation was not created");
  VF_CHECK(results != nullptr, "results");
  const uint8_t* bgr = nullptr;
  size_t stride = 0;
  if (features) {
    VF_CHECK(features->size() == 1 && (*features)[0].type == MatView::TYPE_8UC3,
             "Expecting one BGR24 feature frame.");
    const MatView& f = (*features)[0];
    VF_CHECK(f.cols == frame_width_ && f.rows == frame_height_, "feature size differs from the stream's");
    bgr = static_cast<const uint8_t*>(f.data);
    stride = f.step;
  } else {
    VF_CHECK(flush, "features may only be omitted when flushing");
  }
  const float* flow_ptr = nullptr;
  if (flow && !flow->empty()) {
    VF_CHECK(flow->type == MatView::TYPE_32FC2 && flow->cols == frame_width_ &&
                 flow->rows == frame_height_ && flow->step == (size_t)frame_width_ * 2 * sizeof(float),
             "Expecting a dense W x H x 2 float flow field.");
    flow_ptr = static_cast<const float*>(flow->data);
  }
  int num_results = 0;
  const int rc = vsg_stream_process_frame(stream_, flush ? 1 : 0, bgr, stride, flow_ptr,
                                          flow != nullptr ? 1 : 0, VSG_MEM_HOST, &num_results);
  VF_CHECK(rc == VSG_OK, vsg_last_error());
  for (int k = 0; k < num_results; ++k) {
    const uint8_t* data = nullptr;
    size_t len = 0;
    VF_CHECK(vsg_stream_result_bytes(stream_, k, &data, &len) == VSG_OK, vsg_last_error());
    std::unique_ptr<SegmentationDesc> desc(new SegmentationDesc);
    desc->wire.assign(reinterpret_cast<const char*>(data), len);
    results->push_back(std::move(desc));
  }
  return num_results;
}

// ---- DenseSegmentationUnit ------------------------------------------------------------------
DenseSegmentationUnit::DenseSegmentationUnit(const DenseSegmentationUnitOptions& options,
                                             const DenseSegmentationOptions* dense_seg_options)
    : options_(options) {
  if (dense_seg_options) dense_seg_options_ = *dense_seg_options;
}

DenseSegmentationUnit::~DenseSegmentationUnit() {}

bool DenseSegmentationUnit::OpenStreams(StreamSet* set) {
  video_stream_idx_ = FindStreamIdx(options_.video_stream_name, set);
  if (video_stream_idx_ < 0) {
    std::fprintf(stderr, "ERROR: Could not find video stream!\n");
    return false;
  }
  const VideoStream& vid_stream = set->at(video_stream_idx_)->As<VideoStream>();
  frame_width_ = vid_stream.frame_width();
  frame_height_ = vid_stream.frame_height();
  if (vid_stream.pixel_format() != PIXEL_FORMAT_BGR24) {
    std::fprintf(stderr, "ERROR: Expecting video format to be BGR24.\n");
    return false;
  }
  if (!options_.flow_stream_name.empty()) {
    flow_stream_idx_ = FindStreamIdx(options_.flow_stream_name, set);
    if (flow_stream_idx_ < 0) {
      std::fprintf(stderr, "ERROR: Flow stream specified but not present\n");
      return false;
    }
  } else {
    flow_stream_idx_ = -1;
  }
  set->push_back(std::shared_ptr<DataStream>(
      new SegmentationStream(frame_width_, frame_height_, options_.segment_stream_name)));
  if (!OpenFeatureStreams(set)) {
    std::fprintf(stderr, "ERROR: Could not open feature streams.\n");
    return false;
  }
  dense_seg_ = CreateDenseSegmentation();
  if (!dense_seg_ || !dense_seg_->ok()) {
    std::fprintf(stderr, "ERROR: could not create HIP dense segmentation: %s\n", vsg_last_error());
    return false;
  }
  SetRateBufferSize(dense_seg_->ChunkSize() * 3);
  return true;
}

bool DenseSegmentationUnit::OpenFeatureStreams(StreamSet*) { return true; }

std::unique_ptr<DenseSegmentation> DenseSegmentationUnit::CreateDenseSegmentation() {
  return std::unique_ptr<DenseSegmentation>(
      new DenseSegmentation(dense_seg_options_, frame_width_, frame_height_, options_.device));
}

void DenseSegmentationUnit::ExtractFrameSetFeatures(FrameSetPtr input,
                                                    std::vector<MatView>* features) {
  VF_CHECK(features != nullptr, "features");
  const VideoFrame& video_frame = input->at(video_stream_idx_)->As<VideoFrame>();
  MatView view;   // appearance only
  view.data = video_frame.data();
  view.rows = video_frame.height();
  view.cols = video_frame.width();
  view.step = (size_t)video_frame.width_step();
  view.type = MatView::TYPE_8UC3;
  features->push_back(view);
}

void DenseSegmentationUnit::ProcessFrame(FrameSetPtr input, std::list<FrameSetPtr>* output) {
  std::vector<MatView> features;
  ExtractFrameSetFeatures(input, &features);
  MatView flow;   // stays empty for the first frame (segmentation_unit.cpp:124-130)
  flow.type = MatView::TYPE_32FC2;
  if (input_frames_ > 0 && flow_stream_idx_ >= 0) {
    const DenseFlowFrame& flow_frame = input->at(flow_stream_idx_)->As<DenseFlowFrame>();
    flow.data = flow_frame.flow();
    flow.rows = flow_frame.height();
    flow.cols = flow_frame.width();
    flow.step = (size_t)flow_frame.width() * 2 * sizeof(float);
  }
  frame_set_buffer_.push_back(input);
  ++input_frames_;
  std::vector<std::unique_ptr<SegmentationDesc>> results;
  if (dense_seg_->ProcessFrame(false, &features, flow_stream_idx_ >= 0 ? &flow : nullptr, &results) > 0) {
    OutputSegmentation(&results, output);
  }
}

bool DenseSegmentationUnit::PostProcess(std::list<FrameSetPtr>* append) {
  if (!dense_seg_ || input_frames_ == 0) return false;
  std::vector<std::unique_ptr<SegmentationDesc>> results;
  MatView no_flow;
  if (dense_seg_->ProcessFrame(true, nullptr, flow_stream_idx_ >= 0 ? &no_flow : nullptr, &results) > 0) {
    OutputSegmentation(&results, append);
  }
  return false;
}

void DenseSegmentationUnit::OutputSegmentation(
    std::vector<std::unique_ptr<SegmentationDesc>>* results, std::list<FrameSetPtr>* output) {
  for (size_t k = 0; k < results->size(); ++k) {
    VF_CHECK(!frame_set_buffer_.empty(), "more results than buffered frame sets");
    FrameSetPtr frame_set = frame_set_buffer_.front();
    frame_set_buffer_.pop_front();
    const int64_t pts = frame_set->at(video_stream_idx_)->pts();
    frame_set->push_back(std::shared_ptr<Frame>(
        new PointerFrame<SegmentationDesc>(std::move((*results)[k]), pts)));
    output->push_back(frame_set);
    ++output_frames_;
  }
  // Progress marker parsed by the reference's web front end (segmentation_unit.cpp:177).
  std::fprintf(stderr, "__STREAMING_SIZE__: %d\n", output_frames_);
}

// ---- RegionSegmentation / RegionSegmentationUnit ---------------------------------------------
RegionSegmentation::RegionSegmentation(const RegionSegmentationOptions& options, int frame_width, int frame_height)
    : options_(options) {
  vsg_regionseg_options o;
  vsg_regionseg_default_options(&o);
  o.min_region_num = options.min_region_num;
  o.max_region_num = options.max_region_num;
  o.level_cutoff_fraction = options.level_cutoff_fraction;
  o.small_region_penalizer = options.small_region_penalizer;
  o.luminance_bins = options.luminance_bins;
  o.color_bins = options.color_bins;
  o.flow_bins = options.flow_bins;
  o.chunk_set_size = options.chunk_set_size;
  o.chunk_set_overlap = options.chunk_set_overlap;
  o.constraint_chunks = options.constraint_chunks;
  o.use_appearance = options.use_appearance;
  o.use_flow = options.use_flow;
  o.use_size_penalizer = options.use_size_penalizer;
  o.compute_vectorization = options.compute_vectorization;
  o.save_descriptors = options.save_descriptors;
  if (vsg_regionseg_create(&o, frame_width, frame_height, &handle_) != VSG_OK) handle_ = nullptr;
}

RegionSegmentation::~RegionSegmentation() { vsg_regionseg_destroy(handle_); }

int RegionSegmentation::ProcessFrame(bool flush, const SegmentationDesc* segmentation,
                                     const std::vector<MatView>* features,
                                     std::vector<std::unique_ptr<SegmentationDesc>>* results) {
  VF_CHECK(handle_ != nullptr, "region segmentation was not created");
  VF_CHECK((segmentation == nullptr) == (features == nullptr),
           "Requring both segmentation and features to be either set or null.");
  int num_results = 0;
  if (segmentation) {
    VF_CHECK(!features->empty() && (*features)[0].type == MatView::TYPE_8UC3, "first feature has to be the BGR24 frame");
    const MatView& image = (*features)[0];
    const float* flow = nullptr;
    if (options_.use_flow && features->size() > 1 && !(*features)[1].empty()) {
      const MatView& fv = (*features)[1];
      VF_CHECK(fv.type == MatView::TYPE_32FC2 && fv.step == (size_t)fv.cols * 2 * sizeof(float), "flow has to be packed 32FC2");
      flow = static_cast<const float*>(fv.data);
    }
    VF_CHECK(vsg_regionseg_process_frame(handle_, flush ? 1 : 0,
                                         reinterpret_cast<const uint8_t*>(segmentation->wire.data()),
                                         segmentation->wire.size(), static_cast<const uint8_t*>(image.data),
                                         image.step, flow, &num_results) == VSG_OK,
             vsg_last_error());
  } else {
    VF_CHECK(vsg_regionseg_process_frame(handle_, flush ? 1 : 0, nullptr, 0, nullptr, 0, nullptr, &num_results) == VSG_OK,
             vsg_last_error());
  }
  for (int k = 0; k < num_results; ++k) {
    const uint8_t* data = nullptr;
    size_t len = 0;
    VF_CHECK(vsg_regionseg_result_bytes(handle_, k, &data, &len) == VSG_OK, vsg_last_error());
    std::unique_ptr<SegmentationDesc> desc(new SegmentationDesc);
    desc->wire.assign(reinterpret_cast<const char*>(data), len);
    results->push_back(std::move(desc));
  }
  return (int)results->size();
}

RegionSegmentationUnit::RegionSegmentationUnit(const RegionSegmentationUnitOptions& options,
                                               const RegionSegmentationOptions* region_options)
    : options_(options) {
  if (region_options) region_options_ = *region_options;
  SetRateBufferSize(300);
}

RegionSegmentationUnit::~RegionSegmentationUnit() {}

bool RegionSegmentationUnit::OpenStreams(StreamSet* set) {
  video_stream_idx_ = FindStreamIdx(options_.video_stream_name, set);
  if (video_stream_idx_ < 0) {
    std::fprintf(stderr, "ERROR: Could not find video stream!\n");
    return false;
  }
  const VideoStream& vid_stream = set->at(video_stream_idx_)->As<VideoStream>();
  frame_width_ = vid_stream.frame_width();
  frame_height_ = vid_stream.frame_height();
  if (vid_stream.pixel_format() != PIXEL_FORMAT_BGR24) {
    std::fprintf(stderr, "ERROR: Expecting video format to be BGR24.\n");
    return false;
  }
  if (!options_.flow_stream_name.empty()) {
    flow_stream_idx_ = FindStreamIdx(options_.flow_stream_name, set);
    if (flow_stream_idx_ < 0) {
      std::fprintf(stderr, "ERROR: Flow stream specified but not present\n");
      return false;
    }
  } else {
    flow_stream_idx_ = -1;
  }
  seg_stream_idx_ = FindStreamIdx(options_.segment_stream_name, set);
  if (seg_stream_idx_ < 0) {
    std::fprintf(stderr, "ERROR: Could not find Segmentation stream!\n");
    return false;
  }
  if (!OpenFeatureStreams(set)) {
    std::fprintf(stderr, "ERROR: Error opening feature streams!\n");
    return false;
  }
  region_seg_ = CreateRegionSegmentation();
  if (!region_seg_ || !region_seg_->ok()) {
    std::fprintf(stderr, "ERROR: could not create the region segmentation: %s\n", vsg_last_error());
    return false;
  }
  return true;
}

bool RegionSegmentationUnit::OpenFeatureStreams(StreamSet*) { return true; }

std::unique_ptr<RegionSegmentation> RegionSegmentationUnit::CreateRegionSegmentation() {
  region_options_.use_flow = flow_stream_idx_ >= 0;   // segmentation_unit.cpp:297
  return std::unique_ptr<RegionSegmentation>(new RegionSegmentation(region_options_, frame_width_, frame_height_));
}

void RegionSegmentationUnit::ExtractFrameSetFeatures(FrameSetPtr input, std::vector<MatView>* features) {
  VF_CHECK(features != nullptr, "features");
  const VideoFrame& frame = input->at(video_stream_idx_)->As<VideoFrame>();
  MatView image;
  image.data = frame.data();
  image.rows = frame.height();
  image.cols = frame.width();
  image.step = (size_t)frame.width_step();
  image.type = MatView::TYPE_8UC3;
  features->push_back(image);
  if (flow_stream_idx_ >= 0) {
    MatView flow;
    flow.type = MatView::TYPE_32FC2;
    if (num_input_frames_ > 0) {
      const DenseFlowFrame& flow_frame = input->at(flow_stream_idx_)->As<DenseFlowFrame>();
      flow.data = flow_frame.flow();
      flow.rows = flow_frame.height();
      flow.cols = flow_frame.width();
      flow.step = (size_t)flow_frame.width() * 2 * sizeof(float);
    }
    features->push_back(flow);   // an empty view for the first frame
  }
}

void RegionSegmentationUnit::ProcessFrame(FrameSetPtr input, std::list<FrameSetPtr>* output) {
  const PointerFrame<SegmentationDesc>& seg_frame = input->at(seg_stream_idx_)->As<PointerFrame<SegmentationDesc>>();
  const SegmentationDesc* desc = seg_frame.Ptr();
  std::vector<MatView> features;
  ExtractFrameSetFeatures(input, &features);
  frame_set_buffer_.push_back(input);
  std::vector<std::unique_ptr<SegmentationDesc>> results;
  region_seg_->ProcessFrame(false, desc, &features, &results);
  // The over-segmentation is replaced by the hierarchical result (OutputSegmentation); the frames
  // the next units do not need can go (segmentation_unit.cpp:255-262).
  if (options_.free_video_frames) input->at(video_stream_idx_).reset();
  if (flow_stream_idx_ >= 0 && options_.free_flow_frames) input->at(flow_stream_idx_).reset();
  if (!results.empty()) OutputSegmentation(&results, output);
  ++num_input_frames_;
}

bool RegionSegmentationUnit::PostProcess(std::list<FrameSetPtr>* append) {
  if (!region_seg_ || num_input_frames_ == 0) return false;
  std::vector<std::unique_ptr<SegmentationDesc>> results;
  if (region_seg_->ProcessFrame(true, nullptr, nullptr, &results) > 0) OutputSegmentation(&results, append);
  return false;
}

void RegionSegmentationUnit::OutputSegmentation(std::vector<std::unique_ptr<SegmentationDesc>>* results,
                                                std::list<FrameSetPtr>* output) {
  for (size_t k = 0; k < results->size(); ++k) {
    VF_CHECK(!frame_set_buffer_.empty(), "more results than buffered frame sets");
    FrameSetPtr frame_set = frame_set_buffer_.front();
    frame_set_buffer_.pop_front();
    const int64_t pts = frame_set->at(seg_stream_idx_)->pts();
    frame_set->at(seg_stream_idx_).reset(new PointerFrame<SegmentationDesc>(std::move((*results)[k]), pts));
    output->push_back(frame_set);
  }
}

}  // namespace segmentation
