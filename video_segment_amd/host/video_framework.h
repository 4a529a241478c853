// video_framework.h -- minimal restatement of the reference's video_framework operator API, kept so
// that DenseSegmentationUnit is a drop-in with the same names, argument meaning and error
// behaviour (reference: video_framework/video_unit.h:59-193, 198-290, 343-510;
// video_framework/video_unit.cpp:149-179, 317-483; video_framework/flow_reader.h:46-72).
//
// Only what the dense over-segmentation path touches is restated: typed frames and streams,
// FrameSet / StreamSet, and the VideoUnit tree with OpenStreams / ProcessFrame / PostProcess,
// its sender decorators, period statistics and the root's rate policy.  The threaded pipeline
// (VideoPipelineSink / Source / Invoker) is in video_pipeline.h.  Seeking and VideoPool stay out
// of scope.
#ifndef VSG_HOST_VIDEO_FRAMEWORK_H_
#define VSG_HOST_VIDEO_FRAMEWORK_H_

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <deque>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <list>
#include <memory>
#include <string>
#include <typeinfo>
#include <vector>

namespace video_framework {

#define VF_CHECK(cond, msg)                                                          \
  do {                                                                               \
    if (!(cond)) {                                                                   \
      std::fprintf(stderr, "CHECK failed: %s : %s (%s:%d)\n", #cond, msg, __FILE__, __LINE__); \
      std::abort();                                                                  \
    }                                                                                \
  } while (0)

enum VideoPixelFormat { PIXEL_FORMAT_RGB24, PIXEL_FORMAT_BGR24, PIXEL_FORMAT_ARGB32,
                        PIXEL_FORMAT_ABGR32, PIXEL_FORMAT_RGBA32, PIXEL_FORMAT_BGRA32,
                        PIXEL_FORMAT_YUV422, PIXEL_FORMAT_LUMINANCE };

// Checked down-casts (base/base.h:66-120).
class TypedType {
 public:
  explicit TypedType(const std::type_info* type) : type_(type) {}
  virtual ~TypedType() {}
  template <class T> const T& As() const {
    VF_CHECK(*type_ == typeid(T), "type mismatch in As<T>()");
    return *static_cast<const T*>(this);
  }
  template <class T> T& AsRef() {
    VF_CHECK(*type_ == typeid(T), "type mismatch in AsRef<T>()");
    return *static_cast<T*>(this);
  }

 private:
  const std::type_info* type_;
};

class Frame : public TypedType {
 protected:
  Frame(const std::type_info* type, int64_t pts) : TypedType(type), pts_(pts) {}

 public:
  int64_t pts() const { return pts_; }
  void set_pts(int64_t pts) { pts_ = pts; }

 private:
  int64_t pts_;
};

class DataFrame : public Frame {
 public:
  explicit DataFrame(int size = 0, int64_t pts = 0)
      : Frame(&typeid(DataFrame), pts), data_((size_t)size, 0) {}
  const uint8_t* data() const { return data_.data(); }
  uint8_t* mutable_data() { return data_.data(); }
  int size() const { return (int)data_.size(); }

 protected:
  DataFrame(const std::type_info* type, size_t size, int64_t pts) : Frame(type, pts), data_(size, 0) {}
  std::vector<uint8_t> data_;
};

// BGR24 frames have width_step = width*3 padded to a multiple of 4 in the reference's readers
// (video_reader_unit.cpp:200-206); any width_step >= width*channels is accepted here.
class VideoFrame : public DataFrame {
 public:
  VideoFrame(int width, int height, int channels, int width_step = 0, int64_t pts = 0)
      : DataFrame(&typeid(VideoFrame),
                  (size_t)(width_step ? width_step : width * channels) * (size_t)height, pts),
        width_(width), height_(height), channels_(channels),
        width_step_(width_step ? width_step : width * channels) {}
  int width() const { return width_; }
  int height() const { return height_; }
  int channels() const { return channels_; }
  int width_step() const { return width_step_; }

 private:
  int width_, height_, channels_, width_step_;
};

// Interleaved (x, y) f32 flow (flow_reader.h:46-72).
class DenseFlowFrame : public DataFrame {
 public:
  DenseFlowFrame(int width, int height, bool backward_flow, int64_t pts = 0)
      : DataFrame(&typeid(DenseFlowFrame), (size_t)width * height * 2 * sizeof(float), pts),
        width_(width), height_(height), backward_flow_(backward_flow) {}
  int width() const { return width_; }
  int height() const { return height_; }
  bool is_backward_flow() const { return backward_flow_; }
  const float* flow() const { return reinterpret_cast<const float*>(data()); }
  float* mutable_flow() { return reinterpret_cast<float*>(mutable_data()); }

 private:
  int width_, height_;
  bool backward_flow_;
};

template <class T>
class PointerFrame : public Frame {
 public:
  PointerFrame(std::unique_ptr<T> ptr, int64_t pts = 0)
      : Frame(&typeid(PointerFrame<T>), pts), ptr_(std::move(ptr)) {}
  const T* Ptr() const { return ptr_.get(); }
  T* MutablePtr() { return ptr_.get(); }
  const T& Ref() const { return *ptr_; }
  std::unique_ptr<T> release() { return std::move(ptr_); }

 private:
  std::unique_ptr<T> ptr_;
};

class DataStream : public TypedType {
 public:
  explicit DataStream(const std::string& stream_name)
      : TypedType(&typeid(DataStream)), stream_name_(stream_name) {}
  virtual std::string stream_name() { return stream_name_; }

 protected:
  DataStream(const std::type_info* type, const std::string& stream_name)
      : TypedType(type), stream_name_(stream_name) {}
  std::string stream_name_;
};

class VideoStream : public DataStream {
 public:
  VideoStream(int width, int height, int width_step, float fps = 0,
              VideoPixelFormat pixel_format = PIXEL_FORMAT_BGR24,
              const std::string& stream_name = "VideoStream")
      : DataStream(&typeid(VideoStream), stream_name), frame_width_(width), frame_height_(height),
        width_step_(width_step), fps_(fps), pixel_format_(pixel_format) {}
  int frame_width() const { return frame_width_; }
  int frame_height() const { return frame_height_; }
  int width_step() const { return width_step_; }
  float fps() const { return fps_; }
  VideoPixelFormat pixel_format() const { return pixel_format_; }

 private:
  int frame_width_, frame_height_, width_step_;
  float fps_;
  VideoPixelFormat pixel_format_;
};

class DenseFlowStream : public DataStream {
 public:
  DenseFlowStream(int width, int height, const std::string& stream_name = "BackwardFlowStream")
      : DataStream(&typeid(DenseFlowStream), stream_name), width_(width), height_(height) {}
  int width() const { return width_; }
  int height() const { return height_; }

 private:
  int width_, height_;
};

class SegmentationStream : public DataStream {
 public:
  SegmentationStream(int frame_width, int frame_height,
                     const std::string& stream_name = "SegmentationStream")
      : DataStream(&typeid(SegmentationStream), stream_name), frame_width_(frame_width),
        frame_height_(frame_height) {}
  int frame_width() const { return frame_width_; }
  int frame_height() const { return frame_height_; }

 private:
  int frame_width_, frame_height_;
};

typedef std::vector<std::shared_ptr<Frame>> FrameSet;
typedef std::vector<std::shared_ptr<DataStream>> StreamSet;
typedef std::shared_ptr<FrameSet> FrameSetPtr;

// Rate policy of a root unit (video_unit.h:296-340): a fixed ceiling (max_rate) and/or a ceiling
// that follows the slowest unit of the tree (dynamic_rate), throttled when pipeline queues fill up.
struct RatePolicy {
  float max_rate = 0;               // frames / s, 0 = as fast as possible
  bool dynamic_rate = false;
  float dynamic_rate_scale = 1.0f;
  int startup_frames = 0;
  float update_interval = 0;        // seconds
  int queue_throttle_threshold = 8;
  int num_throttle_frames = 4;
  float min_throttle_rate = 0.2f;
};

// VideoUnit tree (video_unit.h:343-510, video_unit.cpp:100-483): OpenStreams / ProcessFrame /
// PostProcess with their *FromSender decorators, PrepareProcessing + Run (or NextFrame) on the
// root, per-unit period statistics, rate limiting of the root.  A unit is driven by one thread
// at a time; units in different pipeline segments (video_pipeline.h) run on different threads and
// only share immutable FrameSets and the period statistics (guarded by a mutex).
class VideoUnit {
 public:
  VideoUnit() {}
  virtual ~VideoUnit() {}

  virtual bool OpenStreams(StreamSet* set) { return true; }
  virtual void ProcessFrame(FrameSetPtr input, std::list<FrameSetPtr>* output) {
    output->push_back(input);
  }
  virtual bool PostProcess(std::list<FrameSetPtr>* append) { return false; }

  // Decorators that name the calling unit (units with several inputs); by default every sender
  // is treated alike.
  virtual bool OpenStreamsFromSender(StreamSet* set, const VideoUnit* sender) {
    return OpenStreams(set);
  }
  virtual void ProcessFrameFromSender(FrameSetPtr input, std::list<FrameSetPtr>* output,
                                      const VideoUnit* sender) {
    ProcessFrame(input, output);
  }
  virtual bool PostProcessFromSender(std::list<FrameSetPtr>* append, const VideoUnit* sender) {
    return PostProcess(append);
  }

  // Rate management runs downstream from the root; only extensions such as pipeline sources
  // react (LimitRateImpl).
  void LimitRate(float fps) {
    LimitRateImpl(fps);
    for (VideoUnit* c : children_) c->LimitRate(fps);
  }

  void AddChild(VideoUnit* child) {
    RemoveChild(child);
    children_.push_back(child);
    child->parent_ = this;
  }
  void AttachTo(VideoUnit* parent) { parent->AddChild(this); }
  void RemoveChild(VideoUnit* child) {
    for (size_t i = 0; i < children_.size(); ++i) {
      if (children_[i] == child) {
        child->parent_ = nullptr;
        children_.erase(children_.begin() + (long)i);
        return;
      }
    }
  }
  void RemoveFrom(VideoUnit* parent) { parent->RemoveChild(this); }
  bool HasChild(VideoUnit* child) const {
    for (VideoUnit* c : children_) if (c == child) return true;
    return false;
  }
  VideoUnit* ParentUnit() const { return parent_; }
  VideoUnit* RootUnit() {
    VideoUnit* u = this;
    while (u->parent_) u = u->parent_;
    return u;
  }

  bool PrepareProcessing() {
    VF_CHECK(this == RootUnit(), "Only root unit can initiate setup.");
    StreamSet set;
    if (!OpenStreamsImpl(&set, nullptr)) return false;
    initialized_ = true;
    return true;
  }
  virtual bool Run() {
    if (!initialized_) {
      std::fprintf(stderr, "ERROR: Unit is not initialized, call PrepareProcessing first.\n");
      return false;
    }
    PostProcessImpl(nullptr);
    return true;
  }
  virtual bool RunRateLimited(const RatePolicy& rate_policy) {
    VF_CHECK(this == RootUnit(), "Only root unit can enforce rate policy.");
    rate_policy_ = rate_policy;
    rate_policy_updated_ = std::chrono::steady_clock::now();
    PostProcessImpl(nullptr);
    return true;
  }
  bool PrepareAndRun() { return PrepareProcessing() && Run(); }

  // Frame based processing: one PostProcess call of the root per NextFrame; false at the end.
  bool NextFrame() {
    if (!initialized_) {
      std::fprintf(stderr, "ERROR: Unit is not initialized, call PrepareProcessing first.\n");
      return false;
    }
    std::list<FrameSetPtr> append;
    const bool end_of_stream = NextFrameImpl(nullptr, &append);
    for (const FrameSetPtr& fs : append) {
      for (VideoUnit* c : children_) c->ProcessFrameImpl(fs, this);
    }
    if (end_of_stream) {
      for (VideoUnit* c : children_) c->PostProcessImpl(this);
      return false;
    }
    return true;
  }

  // Mean time spent in ProcessFrame (seconds) over the last SetRateBufferSize calls, its inverse,
  // the slowest unit of the subtree and the fullest pipeline queue of the subtree.
  float UnitPeriod() const {
    std::lock_guard<std::mutex> lock(buffer_mutex_);
    if (period_buffer_.empty()) return 0.f;
    double total = 0;
    for (float v : period_buffer_) total += v;
    return total > 0 ? (float)(total / (double)period_buffer_.size()) : 0.f;
  }
  float UnitRate() const {
    const float period = UnitPeriod();
    return period > 0 ? 1.0f / period : 1e3f;
  }
  float MinTreeRate() const {
    float r = UnitRate();
    for (const VideoUnit* c : children_) r = std::min(r, c->MinTreeRate());
    return r;
  }
  virtual int GetQueueSize() const { return 0; }
  int MaxTreeQueueSize() const {
    int q = GetQueueSize();
    for (const VideoUnit* c : children_) q = std::max(q, c->MaxTreeQueueSize());
    return q;
  }

 protected:
  int FindStreamIdx(const std::string& stream_name, const StreamSet* set) {
    for (size_t i = 0; i < set->size(); ++i) {
      if ((*set)[i]->stream_name() == stream_name) return (int)i;
    }
    return -1;
  }
  void SetRateBufferSize(int buffer_size) {
    std::lock_guard<std::mutex> lock(buffer_mutex_);
    period_capacity_ = (size_t)std::max(1, buffer_size);
    while (period_buffer_.size() > period_capacity_) period_buffer_.pop_front();
  }
  virtual void LimitRateImpl(float fps) {}
  // Units that hand their frames to another thread (pipeline sinks) do not pass the end of the
  // stream on themselves.
  virtual bool PostProcessingPassToChildren() { return true; }

  virtual bool OpenStreamsImpl(StreamSet* set, const VideoUnit* sender) {
    const int prev_stream_sz = (int)set->size();
    if (!OpenStreamsFromSender(set, sender)) return false;
    stream_sz_ = (int)set->size();
    for (int i = prev_stream_sz; i < stream_sz_; ++i) {   // duplicate names break FindStreamIdx
      const std::string name = (*set)[(size_t)i]->stream_name();
      if (FindStreamIdx(name, set) < i) {
        std::fprintf(stderr, "ERROR: Duplicate stream found: %s\n", name.c_str());
        return false;
      }
    }
    // The stream set is handed down the tree: a unit sees every stream its ancestors created.
    for (VideoUnit* c : children_) {
      if (!c->OpenStreamsImpl(set, this)) return false;
    }
    return true;
  }

  virtual void ProcessFrameImpl(const FrameSetPtr frame_set, const VideoUnit* sender) {
    std::list<FrameSetPtr> output;
    const auto t0 = std::chrono::steady_clock::now();
    ProcessFrameFromSender(frame_set, &output, sender);
    for (const FrameSetPtr& fs : output) {
      VF_CHECK((int)fs->size() == stream_sz_,
               "Number of streams set in OpenStreams not consistent with returned FrameSet.");
    }
    PushPeriod((float)std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), 1);
    for (const FrameSetPtr& fs : output) {
      for (VideoUnit* c : children_) c->ProcessFrameImpl(fs, this);
    }
  }

  // One PostProcess call of this unit; true at the end of the stream.
  bool NextFrameImpl(const VideoUnit* sender, std::list<FrameSetPtr>* output) {
    const auto t0 = std::chrono::steady_clock::now();
    const bool end_of_stream = !PostProcessFromSender(output, nullptr);
    if (!output->empty()) {
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      PushPeriod((float)(dt / (double)output->size()), (int)output->size());
      for (const FrameSetPtr& fs : *output) {
        VF_CHECK((int)fs->size() == stream_sz_,
                 "Number of streams set in PostProcessImpl not consistent with returned FrameSet.");
      }
    }
    return end_of_stream;
  }

  virtual void PostProcessImpl(const VideoUnit* sender) {
    for (;;) {
      std::list<FrameSetPtr> append;
      const bool end_of_stream = NextFrameImpl(sender, &append);
      if (rate_policy_.max_rate > 0) {   // only the root carries a policy
        VF_CHECK(this == RootUnit(), "Expected root unit.");
        const float target = 1.0f / rate_policy_.max_rate;
        float last = 0;
        {
          std::lock_guard<std::mutex> lock(buffer_mutex_);
          if (!period_buffer_.empty()) last = period_buffer_.back();
        }
        const int wait_us = (int)((target - last) * 1e6f);
        if (wait_us > 100) std::this_thread::sleep_for(std::chrono::microseconds(wait_us));
      }
      if (rate_policy_.dynamic_rate) UpdateDynamicRate();
      for (const FrameSetPtr& fs : append) {
        for (VideoUnit* c : children_) c->ProcessFrameImpl(fs, this);
      }
      if (end_of_stream) break;
      if (append.empty()) std::this_thread::sleep_for(std::chrono::microseconds(500));
    }
    if (PostProcessingPassToChildren()) {
      for (VideoUnit* c : children_) c->PostProcessImpl(this);
    }
  }

  const std::vector<VideoUnit*>& children() const { return children_; }

 private:
  void PushPeriod(float seconds, int times) {
    std::lock_guard<std::mutex> lock(buffer_mutex_);
    for (int i = 0; i < times; ++i) {
      period_buffer_.push_back(seconds);
      if (period_buffer_.size() > period_capacity_) period_buffer_.pop_front();
    }
  }
  // Dynamic ceiling of the root (video_unit.cpp:417-456): after the start-up frames, once per
  // update interval, max_rate follows the slowest unit of the tree; full queues halve it for
  // every num_throttle_frames above the threshold (never below min_throttle_rate).
  void UpdateDynamicRate() {
    VF_CHECK(this == RootUnit(), "Rate policy can only be adapted by root unit");
    size_t have;
    {
      std::lock_guard<std::mutex> lock(buffer_mutex_);
      have = period_buffer_.size();
      if ((size_t)rate_policy_.startup_frames >= period_capacity_) return;   // policy not enforceable
    }
    if ((size_t)rate_policy_.startup_frames >= have) return;
    const auto now = std::chrono::steady_clock::now();
    if (std::chrono::duration<float>(now - rate_policy_updated_).count() <= rate_policy_.update_interval) {
      return;
    }
    const float min_rate = MinTreeRate();
    float rate_scale = 1.0f;
    const int max_queue = MaxTreeQueueSize();
    if (max_queue > rate_policy_.queue_throttle_threshold) {
      rate_scale *= std::pow(0.5f, (float)(max_queue - rate_policy_.queue_throttle_threshold) /
                                       (float)rate_policy_.num_throttle_frames);
      rate_scale = std::max(rate_scale, rate_policy_.min_throttle_rate);
    }
    rate_policy_.max_rate = min_rate * rate_scale * rate_policy_.dynamic_rate_scale;
    LimitRate(min_rate);
    rate_policy_updated_ = now;
  }

  std::vector<VideoUnit*> children_;
  VideoUnit* parent_ = nullptr;
  int stream_sz_ = 0;
  bool initialized_ = false;
  std::deque<float> period_buffer_;
  size_t period_capacity_ = 64;
  mutable std::mutex buffer_mutex_;
  RatePolicy rate_policy_;
  std::chrono::steady_clock::time_point rate_policy_updated_;

  friend class VideoPipelineSource;
};

}  // namespace video_framework

#endif  // VSG_HOST_VIDEO_FRAMEWORK_H_
