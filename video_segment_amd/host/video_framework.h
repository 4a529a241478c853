// video_framework.h -- minimal restatement of the reference's video_framework operator API, kept so
// that DenseSegmentationUnit is a drop-in with the same names, argument meaning and error
// behaviour (reference: video_framework/video_unit.h:59-193, 198-290, 343-510;
// video_framework/video_unit.cpp:149-179, 317-483; video_framework/flow_reader.h:46-72).
//
// Only what the dense over-segmentation path touches is restated: typed frames and streams,
// FrameSet / StreamSet, and the VideoUnit tree with OpenStreams / ProcessFrame / PostProcess.
// Rate limiting, seeking, pools and the threaded pipeline stay out of scope.
#ifndef VSG_HOST_VIDEO_FRAMEWORK_H_
#define VSG_HOST_VIDEO_FRAMEWORK_H_

#include <chrono>
#include <cstdint>
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

// VideoUnit tree (video_unit.h:343-510).  Single threaded: PrepareProcessing() opens the streams
// down the tree, Run() pulls frames from the root's PostProcess until it returns false.
class VideoUnit {
 public:
  VideoUnit() {}
  virtual ~VideoUnit() {}

  virtual bool OpenStreams(StreamSet* set) { return true; }
  virtual void ProcessFrame(FrameSetPtr input, std::list<FrameSetPtr>* output) {
    output->push_back(input);
  }
  virtual bool PostProcess(std::list<FrameSetPtr>* append) { return false; }

  void AddChild(VideoUnit* child) {
    child->parent_ = this;
    children_.push_back(child);
  }
  void AttachTo(VideoUnit* parent) { parent->AddChild(this); }
  VideoUnit* ParentUnit() const { return parent_; }
  VideoUnit* RootUnit() {
    VideoUnit* u = this;
    while (u->parent_) u = u->parent_;
    return u;
  }

  // video_unit.cpp:168-179, 317-346
  bool PrepareProcessing() {
    StreamSet set;
    return OpenStreamsImpl(&set);
  }
  // video_unit.cpp:149-166, 389-483 (no rate policy)
  bool Run() {
    PostProcessImpl();
    return true;
  }
  bool PrepareAndRun() { return PrepareProcessing() && Run(); }

  float UnitPeriod() const { return frames_ ? (float)(seconds_ / frames_) : 0.f; }
  float UnitRate() const { return seconds_ > 0 ? (float)(frames_ / seconds_) : 0.f; }

 protected:
  int FindStreamIdx(const std::string& stream_name, const StreamSet* set) {
    for (size_t i = 0; i < set->size(); ++i) {
      if ((*set)[i]->stream_name() == stream_name) return (int)i;
    }
    return -1;
  }
  void SetRateBufferSize(int) {}

  bool OpenStreamsImpl(StreamSet* set) {
    if (!OpenStreams(set)) return false;
    stream_sz_ = (int)set->size();
    for (VideoUnit* c : children_) {
      StreamSet child_set(*set);
      if (!c->OpenStreamsImpl(&child_set)) return false;
    }
    return true;
  }

  // video_unit.cpp:348-387
  void ProcessFrameImpl(const FrameSetPtr& frame_set) {
    std::list<FrameSetPtr> output;
    const auto t0 = std::chrono::steady_clock::now();
    ProcessFrame(frame_set, &output);
    seconds_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ++frames_;
    Forward(output);
  }

  void PostProcessImpl() {
    for (;;) {
      std::list<FrameSetPtr> append;
      const bool more = PostProcess(&append);
      Forward(append);
      if (!more) break;
    }
    for (VideoUnit* c : children_) c->PostProcessImpl();
  }

  void Forward(const std::list<FrameSetPtr>& frames) {
    for (const FrameSetPtr& fs : frames) {
      VF_CHECK((int)fs->size() == stream_sz_, "FrameSet size differs from the unit's stream set");
      for (VideoUnit* c : children_) c->ProcessFrameImpl(fs);
    }
  }

 private:
  std::vector<VideoUnit*> children_;
  VideoUnit* parent_ = nullptr;
  int stream_sz_ = 0;
  double seconds_ = 0;
  long frames_ = 0;
};

}  // namespace video_framework

#endif  // VSG_HOST_VIDEO_FRAMEWORK_H_
