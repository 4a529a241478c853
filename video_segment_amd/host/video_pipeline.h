// video_pipeline.h -- the reference's threaded pipeline, restated on std::thread
// (video_framework/video_pipeline.h:46-182, video_pipeline.cpp:38-182, concurrent_queue.h).
//
// A unit tree is cut into segments that run on their own threads: the last unit of a segment is a
// VideoPipelineSink (it queues every FrameSet it is given), the first unit of the next segment is
// the VideoPipelineSource attached to that sink (its Run() drains the queue and feeds its
// children until the sink has seen the end of its stream and the queue is empty).  Usage, as in
// seg_tree_sample (seg_tree.cpp:155-163, 211-217, 339-364):
//   sink.AttachTo(reader);  VideoPipelineSource source(&sink);  dense_unit.AttachTo(&source); ...
//   root->PrepareProcessing();
//   VideoPipelineInvoker invoker;  invoker.RunRootRateLimited(policy, root);
//   invoker.RunPipelineSource(&source_0); ...;  last_source.Run();
//   invoker.WaitUntilPipelineFinished();
// With the MI355X DenseSegmentationUnit in the middle segment the reader, the GPU unit and the
// writer overlap: frames are decoded / generated and results written while a chunk is segmented.
#ifndef VSG_HOST_VIDEO_PIPELINE_H_
#define VSG_HOST_VIDEO_PIPELINE_H_

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "video_framework.h"

namespace video_framework {

// Producer / consumer queue (concurrent_queue.h): push never blocks, try_pop never waits.
template <class Data>
class concurrent_queue {
 public:
  void push(const Data& data) {
    {
      std::lock_guard<std::mutex> lock(mutex_);
      queue_.push_back(data);
      size_.store((int)queue_.size(), std::memory_order_relaxed);
    }
    data_available_.notify_one();
  }
  bool try_pop(Data* popped) {
    std::lock_guard<std::mutex> lock(mutex_);
    if (queue_.empty()) return false;
    *popped = queue_.front();
    queue_.pop_front();
    size_.store((int)queue_.size(), std::memory_order_relaxed);
    return true;
  }
  // Waits up to wait_duration milliseconds for an element.
  bool timed_wait_and_pop(Data* popped, int wait_duration = 1000) {
    std::unique_lock<std::mutex> lock(mutex_);
    if (!data_available_.wait_for(lock, std::chrono::milliseconds(wait_duration),
                                  [this] { return !queue_.empty(); })) {
      return false;
    }
    *popped = queue_.front();
    queue_.pop_front();
    size_.store((int)queue_.size(), std::memory_order_relaxed);
    return true;
  }
  bool empty() const { return size() == 0; }
  int size() const {
    std::lock_guard<std::mutex> lock(mutex_);
    return (int)queue_.size();
  }
  int unsafe_size() const { return size_.load(std::memory_order_relaxed); }

 private:
  std::deque<Data> queue_;
  mutable std::mutex mutex_;
  std::condition_variable data_available_;
  std::atomic<int> size_{0};
};

class VideoPipelineSource;

// Last unit of a pipeline segment: queues the FrameSets for the source attached to it.
class VideoPipelineSink : public VideoUnit {
 public:
  VideoPipelineSink() {}
  bool OpenStreams(StreamSet* set) override { return true; }
  void ProcessFrame(FrameSetPtr input, std::list<FrameSetPtr>* output) override {
    frameset_queue_.push(input);
    ++frame_number_;
  }
  bool PostProcess(std::list<FrameSetPtr>* append) override {
    source_exhausted_.store(true, std::memory_order_release);
    return GetQueueSize() != 0;   // stay alive while frames are queued
  }
  int GetQueueSize() const override { return frameset_queue_.unsafe_size(); }

 protected:
  // The attached source runs on another thread and signals the end of the stream itself.
  bool PostProcessingPassToChildren() override { return false; }

 private:
  bool TryFetchingFrameSet(FrameSetPtr* ptr) { return frameset_queue_.try_pop(ptr); }
  bool IsExhausted() const { return source_exhausted_.load(std::memory_order_acquire); }

  std::atomic<bool> source_exhausted_{false};
  concurrent_queue<FrameSetPtr> frameset_queue_;
  int frame_number_ = 0;
  friend class VideoPipelineSource;
};

// How a source reacts to LimitRate calls of the root.
struct SourceRatePolicy {
  SourceRatePolicy() {}
  SourceRatePolicy(bool respond, float scale) : respond_to_limit_rate(respond), rate_scale(scale) {}
  bool respond_to_limit_rate = false;
  float rate_scale = 1.0f;
  int sink_max_queue_size = 0;   // > 0: slow down when the monitored sink's queue is longer
};

// First unit of a pipeline segment: feeds the FrameSets of `sink` to its children from the
// thread that calls Run().
class VideoPipelineSource : public VideoUnit {
 public:
  VideoPipelineSource(VideoPipelineSink* sink, VideoUnit* idle_unit = nullptr,
                      const SourceRatePolicy& policy = SourceRatePolicy(), float max_fps = 0)
      : sink_(sink), idle_unit_(idle_unit), source_rate_policy_(policy), max_fps_(max_fps) {
    AttachTo(sink);
  }

  bool OpenStreams(StreamSet* set) override {
    return idle_unit_ ? idle_unit_->PrepareProcessing() : true;
  }
  void ProcessFrame(FrameSetPtr input, std::list<FrameSetPtr>* output) override {}
  bool PostProcess(std::list<FrameSetPtr>* append) override { return false; }

  // Polls the queue until the sink is exhausted and drained, then ends the stream for the
  // children of this segment.
  bool Run() override {
    using clock = std::chrono::steady_clock;
    while (!(sink_->IsExhausted() && sink_->GetQueueSize() == 0)) {
      float timeout_us = 200.f;
      if (max_fps_ > 0) timeout_us = 1.0f / max_fps_ * 1e6f;
      FrameSetPtr frame_set;
      if (sink_->TryFetchingFrameSet(&frame_set)) {
        if (frame_num_ > 0) {   // keep at least timeout_us between two frames
          for (;;) {
            const float passed =
                std::chrono::duration<float, std::micro>(clock::now() - prev_process_time_).count();
            const int wait_us = (int)(timeout_us - passed);
            if (wait_us <= 10) break;
            OnIdle();
            if (wait_us > 100) std::this_thread::sleep_for(std::chrono::microseconds(wait_us / 5));
          }
        }
        prev_process_time_ = clock::now();
        for (VideoUnit* child : children()) child->ProcessFrameImpl(frame_set, this);
        ++frame_num_;
      } else {
        OnIdle();
        std::this_thread::sleep_for(std::chrono::microseconds((int)(timeout_us / 5)));
      }
    }
    PostProcessImpl(this);
    return true;
  }

  void SetIdleUnit(VideoUnit* idle_unit) { idle_unit_ = idle_unit; }
  // The sink this segment writes to (watched when the root limits the rate).
  void SetMonitorSink(VideoPipelineSink* sink) { monitor_sink_ = sink; }

 protected:
  void LimitRateImpl(float fps) override {
    if (!source_rate_policy_.respond_to_limit_rate) return;
    max_fps_ = fps * source_rate_policy_.rate_scale;
    if (monitor_sink_ && source_rate_policy_.sink_max_queue_size > 0 &&
        monitor_sink_->GetQueueSize() > source_rate_policy_.sink_max_queue_size) {
      max_fps_ = fps * 0.1f;   // stall, but keep processing
    }
  }

 private:
  void OnIdle() {
    if (idle_unit_) idle_unit_->ProcessFrameImpl(FrameSetPtr(new FrameSet()), this);
  }

  VideoPipelineSink* sink_ = nullptr;
  VideoUnit* idle_unit_ = nullptr;
  VideoPipelineSink* monitor_sink_ = nullptr;
  SourceRatePolicy source_rate_policy_;
  std::atomic<float> max_fps_{0};
  int frame_num_ = 0;
  std::chrono::steady_clock::time_point prev_process_time_;
};

// Runs the segments of a pipeline on their own threads.
class VideoPipelineInvoker {
 public:
  VideoPipelineInvoker() {}
  ~VideoPipelineInvoker() { WaitUntilPipelineFinished(); }
  VideoPipelineInvoker(const VideoPipelineInvoker&) = delete;
  VideoPipelineInvoker& operator=(const VideoPipelineInvoker&) = delete;

  void RunRoot(VideoUnit* root) { threads_.emplace_back([root] { root->Run(); }); }
  void RunRootRateLimited(const RatePolicy& policy, VideoUnit* root) {
    threads_.emplace_back([policy, root] { root->RunRateLimited(policy); });
  }
  void RunPipelineSource(VideoPipelineSource* source) {
    threads_.emplace_back([source] { source->Run(); });
  }
  void WaitUntilPipelineFinished() {
    for (std::thread& t : threads_) {
      if (t.joinable()) t.join();
    }
    threads_.clear();
  }

 private:
  std::vector<std::thread> threads_;
};

}  // namespace video_framework

#endif  // VSG_HOST_VIDEO_PIPELINE_H_
