// flow_reader.cpp -- see flow_reader.h.  Follows video_framework/flow_reader.cpp:63-170.
#include "flow_reader.h"

#include <cstdio>

namespace video_framework {

bool DenseFlowReader::OpenAndReadHeader() {
  ifs_.open(filename_.c_str(), std::ios_base::in | std::ios_base::binary);
  if (!ifs_) {
    std::fprintf(stderr, "ERROR: DenseFlowReader: can not open binary flow file %s\n",
                 filename_.c_str());
    return false;
  }
  ifs_.read(reinterpret_cast<char*>(&width_), sizeof(width_));
  ifs_.read(reinterpret_cast<char*>(&height_), sizeof(height_));
  ifs_.read(reinterpret_cast<char*>(&flow_type_), sizeof(flow_type_));
  if (!ifs_ || width_ <= 0 || height_ <= 0 || flow_type_ < FLOW_FORWARD || flow_type_ > FLOW_BOTH) {
    std::fprintf(stderr, "ERROR: DenseFlowReader: malformed header in %s\n", filename_.c_str());
    return false;
  }
  return true;
}

bool DenseFlowReader::MoreFramesAvailable() {
  return ifs_.peek() != std::char_traits<char>::eof();
}

bool DenseFlowReader::GetNextFlowFrame(uint8_t* buffer) {
  ifs_.read(reinterpret_cast<char*>(buffer), RequiredBufferSize());
  return ifs_.gcount() == RequiredBufferSize();
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
  vid_stream_idx_ = FindStreamIdx(options_.video_stream_name, set);
  if (vid_stream_idx_ < 0) {
    std::fprintf(stderr, "ERROR: DenseFlowReaderUnit: can not find video stream\n");
    return false;
  }
  const VideoStream& vid_stream = set->at(vid_stream_idx_)->As<VideoStream>();
  frame_width_ = vid_stream.frame_width();
  frame_height_ = vid_stream.frame_height();
  if (!reader_.OpenAndReadHeader()) return false;
  if (reader_.width() != frame_width_ || reader_.height() != frame_height_) {
    std::fprintf(stderr, "ERROR: flow file has different dimension than input video\n");
    return false;
  }
  // Forward stream first, then backward (flow_reader.cpp:107-119); plain DataStreams.
  if (reader_.FlowType() == FLOW_FORWARD || reader_.FlowType() == FLOW_BOTH) {
    set->push_back(std::shared_ptr<DataStream>(new DataStream(options_.forward_flow_stream_name)));
  }
  if (reader_.FlowType() == FLOW_BACKWARD || reader_.FlowType() == FLOW_BOTH) {
    set->push_back(std::shared_ptr<DataStream>(new DataStream(options_.backward_flow_stream_name)));
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
