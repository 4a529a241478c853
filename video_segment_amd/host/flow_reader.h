// flow_reader.h -- reader of the reference's precomputed dense flow files (`<video>.flow`) and the
// unit that feeds them to DenseSegmentationUnit (video_framework/flow_reader.h:76-133,
// flow_reader.cpp:63-170).  The file is what the reference's DenseFlowUnit saves with
// --save_flow (flow_reader.cpp:240-248, 283-301):
//
//   int32 width, int32 height, int32 flow_type      (0 forward, 1 backward, 2 both)
//   per video frame k >= 1:  width*height*2 f32 interleaved (x, y), forward field first when both
//
// Frame 0 has no flow; the unit emits a 0x0 DenseFlowFrame for it (flow_reader.cpp:151-158).
#ifndef VSG_HOST_FLOW_READER_H_
#define VSG_HOST_FLOW_READER_H_

#include <cstdint>
#include <fstream>
#include <string>

#include "video_framework.h"

namespace video_framework {

enum DenseFlowType { FLOW_FORWARD = 0, FLOW_BACKWARD = 1, FLOW_BOTH = 2 };

class DenseFlowReader {
 public:
  explicit DenseFlowReader(const std::string& filename) : filename_(filename) {}
  bool OpenAndReadHeader();
  int RequiredBufferSize() const { return (int)sizeof(float) * width_ * height_ * 2; }
  // Returns false on a truncated field (the reference reads blindly; a short read is an error
  // here because it would silently segment against garbage flow).
  bool GetNextFlowFrame(uint8_t* buffer);
  bool MoreFramesAvailable();
  int width() const { return width_; }
  int height() const { return height_; }
  int FlowType() const { return flow_type_; }

 private:
  std::string filename_;
  int32_t width_ = 0, height_ = 0, flow_type_ = FLOW_FORWARD;
  std::ifstream ifs_;
  std::streamoff payload_left_ = 0;   // bytes of flow fields not read yet
};

// Writer side of the same format (what DenseFlowUnit does when flow_output_file is set).
class DenseFlowWriter {
 public:
  explicit DenseFlowWriter(const std::string& filename) : filename_(filename) {}
  bool OpenAndWriteHeader(int width, int height, int flow_type);
  void AddFlowFrame(const float* interleaved_xy);
  void Close() { ofs_.close(); }

 private:
  std::string filename_;
  int width_ = 0, height_ = 0;
  std::ofstream ofs_;
};

struct DenseFlowReaderOptions {
  std::string video_stream_name = "VideoStream";
  std::string backward_flow_stream_name = "BackwardFlowStream";
  std::string forward_flow_stream_name = "ForwardFlowStream";
};

class DenseFlowReaderUnit : public VideoUnit {
 public:
  DenseFlowReaderUnit(const DenseFlowReaderOptions& options, const std::string& file)
      : options_(options), reader_(file) {}
  bool OpenStreams(StreamSet* set) override;
  void ProcessFrame(FrameSetPtr input, std::list<FrameSetPtr>* output) override;
  bool PostProcess(std::list<FrameSetPtr>* append) override { return false; }

 private:
  DenseFlowReaderOptions options_;
  DenseFlowReader reader_;
  int vid_stream_idx_ = -1;
  int frame_width_ = 0, frame_height_ = 0;
  int frame_number_ = 0;
};

}  // namespace video_framework

#endif  // VSG_HOST_FLOW_READER_H_
