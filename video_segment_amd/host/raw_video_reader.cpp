// raw_video_reader.cpp -- see raw_video_reader.h.
#include "raw_video_reader.h"

#include <cstdio>
#include <cstring>

namespace video_framework {

bool RawVideoReaderUnit::OpenStreams(StreamSet* set) {
  ifs_.open(video_file_.c_str(), std::ios_base::in | std::ios_base::binary);
  if (!ifs_) {
    std::fprintf(stderr, "ERROR: could not open video file %s\n", video_file_.c_str());
    return false;
  }
  char tag[4];
  int32_t pixel_format = -1;
  ifs_.read(tag, 4);
  ifs_.read(reinterpret_cast<char*>(&width_), 4);
  ifs_.read(reinterpret_cast<char*>(&height_), 4);
  ifs_.read(reinterpret_cast<char*>(&pixel_format), 4);
  ifs_.read(reinterpret_cast<char*>(&frames_), 4);
  ifs_.read(reinterpret_cast<char*>(&fps_), 4);
  if (!ifs_ || std::memcmp(tag, "RAWV", 4) != 0 || width_ <= 0 || height_ <= 0 || frames_ < 0 ||
      pixel_format != 0) {
    std::fprintf(stderr, "ERROR: %s is not a BGR24 raw video file\n", video_file_.c_str());
    return false;
  }
  if (options_.trim_frames > 0 && options_.trim_frames < frames_) frames_ = options_.trim_frames;
  width_step_ = (width_ * 3 + 3) / 4 * 4;
  set->push_back(std::shared_ptr<DataStream>(
      new VideoStream(width_, height_, width_step_, fps_, PIXEL_FORMAT_BGR24, options_.stream_name)));
  next_frame_ = 0;
  return true;
}

bool RawVideoReaderUnit::PostProcess(std::list<FrameSetPtr>* append) {
  if (next_frame_ >= frames_) return false;
  const int64_t pts = fps_ > 0 ? (int64_t)((double)next_frame_ / (double)fps_ * 1e6) : next_frame_;
  std::shared_ptr<VideoFrame> frame(new VideoFrame(width_, height_, 3, width_step_, pts));
  uint8_t* dst = frame->mutable_data();
  for (int y = 0; y < height_; ++y) {
    ifs_.read(reinterpret_cast<char*>(dst + (size_t)y * width_step_), (std::streamsize)width_ * 3);
  }
  VF_CHECK((bool)ifs_, "raw video file is shorter than its header says");
  FrameSetPtr fs(new FrameSet);
  fs->push_back(frame);
  append->push_back(fs);
  ++next_frame_;
  return true;
}

bool WriteRawVideoHeader(std::ofstream* ofs, int width, int height, int frames, float fps) {
  const int32_t h[4] = {width, height, 0, frames};
  ofs->write("RAWV", 4);
  ofs->write(reinterpret_cast<const char*>(h), sizeof(h));
  ofs->write(reinterpret_cast<const char*>(&fps), sizeof(fps));
  return (bool)*ofs;
}

}  // namespace video_framework
