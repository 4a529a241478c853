// flow_reader.cpp -- see flow_reader.h.  Same file format and unit behaviour as
// video_framework/flow_reader.cpp:63-170.
#include "flow_reader.h"

#include <cstdio>
#include <memory>

namespace video_framework {

// The 12-byte header in one read; the number of fields the file holds follows from its length, so
// "is there another frame" is arithmetic instead of a peek at the stream.
bool DenseFlowReader::OpenAndReadHeader() {
  ifs_.open(filename_.c_str(), std::ios_base::in | std::ios_base::binary | std::ios_base::ate);
  const std::streamoff file_bytes = ifs_ ? (std::streamoff)ifs_.tellg() : -1;
  int32_t header[3] = {0, 0, -1};
  if (file_bytes >= (std::streamoff)sizeof(header)) {
    ifs_.seekg(0);
    ifs_.read(reinterpret_cast<char*>(header), sizeof(header));
  }
  if (file_bytes < 0) {
    std::fprintf(stderr, "ERROR: DenseFlowReader: can not open binary flow file %s\n", filename_.c_str());
    return false;
  }
  width_ = header[0];
  height_ = header[1];
  flow_type_ = header[2];
  const bool type_ok = flow_type_ == FLOW_FORWARD || flow_type_ == FLOW_BACKWARD || flow_type_ == FLOW_BOTH;
  if (!ifs_ || width_ <= 0 || height_ <= 0 || !type_ok) {
    std::fprintf(stderr, "ERROR: DenseFlowReader: malformed header in %s\n", filename_.c_str());
    return false;
  }
  payload_left_ = file_bytes - (std::streamoff)sizeof(header);
  return true;
}

bool DenseFlowReader::MoreFramesAvailable() { return payload_left_ > 0; }

bool DenseFlowReader::GetNextFlowFrame(uint8_t* buffer) {
  const std::streamsize want = RequiredBufferSize();
  ifs_.read(reinterpret_cast<char*>(buffer), want);
  payload_left_ -= ifs_.gcount();
  return ifs_.gcount() == want;
}

bool DenseFlowWriter::OpenAndWriteHeader(int width, int height, int flow_type) {
  ofs_.open(filename_.c_str(), std::ios_base::out | std::ios_base::binary | std::ios_base::trunc);
  if (!ofs_) return false;
  width_ = width;
  height_ = height;
  const int32_t header[3] = {width, height, flow_type};
  ofs_.write(reinterpret_cast<const char*>(header), sizeof(header));
  return true;
}

void DenseFlowWriter::AddFlowFrame(const float* interleaved_xy) {
  ofs_.write(reinterpret_cast<const char*>(interleaved_xy),
             sizeof(float) * 2 * (size_t)width_ * (size_t)height_);
}

bool DenseFlowReaderUnit::OpenStreams(StreamSet* set) {
  auto fail = [](const char* what) {
    std::fprintf(stderr, "ERROR: DenseFlowReaderUnit: %s\n", what);
    return false;
  };
  vid_stream_idx_ = FindStreamIdx(options_.video_stream_name, set);
  if (vid_stream_idx_ < 0) return fail("can not find video stream");
  if (!reader_.OpenAndReadHeader()) return false;
  const VideoStream& video = set->at(vid_stream_idx_)->As<VideoStream>();
  frame_width_ = video.frame_width();
  frame_height_ = video.frame_height();
  if (reader_.width() != frame_width_ || reader_.height() != frame_height_) {
    return fail("flow file has different dimension than input video");
  }
  // One plain DataStream per direction the file holds, forward before backward
  // (flow_reader.cpp:107-119).
  const int type = reader_.FlowType();
  const std::string* names[2] = {&options_.forward_flow_stream_name, &options_.backward_flow_stream_name};
  const bool present[2] = {type != FLOW_BACKWARD, type != FLOW_FORWARD};
  for (int d = 0; d < 2; ++d) {
    if (present[d]) set->push_back(std::make_shared<DataStream>(*names[d]));
  }
  frame_number_ = 0;
  return true;
}

void DenseFlowReaderUnit::ProcessFrame(FrameSetPtr input, std::list<FrameSetPtr>* output) {
  const int64_t pts = input->at(vid_stream_idx_)->pts();
  const int type = reader_.FlowType();
  const bool have = frame_number_ > 0 && reader_.MoreFramesAvailable();
  if (frame_number_ > 0 && !have) {
    std::fprintf(stderr, "WARNING: no more flow frames available, outputting empty flow frames\n");
  }
  // One frame per added stream, in stream order.  Frame 0 and frames past the end of the file
  // carry 0x0 fields (flow_reader.cpp:127-158).
  for (int backward = 0; backward < 2; ++backward) {
    const bool wanted = backward ? (type == FLOW_BACKWARD || type == FLOW_BOTH)
                                 : (type == FLOW_FORWARD || type == FLOW_BOTH);
    if (!wanted) continue;
    std::shared_ptr<DenseFlowFrame> frame;
    if (have) {
      frame.reset(new DenseFlowFrame(frame_width_, frame_height_, backward != 0, pts));
      VF_CHECK(reader_.GetNextFlowFrame(frame->mutable_data()), "truncated flow field in .flow file");
    } else {
      frame.reset(new DenseFlowFrame(0, 0, backward != 0, pts));
    }
    input->push_back(frame);
  }
  output->push_back(input);
  ++frame_number_;
}

}  // namespace video_framework
