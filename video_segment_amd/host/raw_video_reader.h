// raw_video_reader.h -- root unit that reads uncompressed BGR24 frames from a file.  It stands in
// for the reference's ffmpeg-based VideoReaderUnit (video_framework/video_reader_unit.{h,cpp}; no
// codec is available in this image) so that real footage, decoded elsewhere, can be pushed through
// DenseSegmentationUnit together with a precomputed `.flow` file (flow_reader.h).
//
// File layout (little endian, this project's own container -- the reference has no raw format):
//   "RAWV"  int32 width  int32 height  int32 pixel_format (0 = BGR24)  int32 frames  float32 fps
//   frames x height x width x 3 bytes, rows tightly packed
// Like the reference's reader the unit emits VideoFrames whose width_step is padded to a multiple
// of 4 bytes (video_reader_unit.cpp:200-206) and pts = frame_index / fps in microseconds.
#ifndef VSG_HOST_RAW_VIDEO_READER_H_
#define VSG_HOST_RAW_VIDEO_READER_H_

#include <cstdint>
#include <fstream>
#include <string>

#include "video_framework.h"

namespace video_framework {

struct RawVideoReaderOptions {
  std::string stream_name = "VideoStream";
  int trim_frames = 0;   // like VideoReaderOptions::trim_frames: 0 = all
};

class RawVideoReaderUnit : public VideoUnit {
 public:
  RawVideoReaderUnit(const RawVideoReaderOptions& options, const std::string& video_file)
      : options_(options), video_file_(video_file) {}
  bool OpenStreams(StreamSet* set) override;
  bool PostProcess(std::list<FrameSetPtr>* append) override;
  int frame_width() const { return width_; }
  int frame_height() const { return height_; }
  int num_frames() const { return frames_; }

 private:
  RawVideoReaderOptions options_;
  std::string video_file_;
  std::ifstream ifs_;
  int32_t width_ = 0, height_ = 0, frames_ = 0;
  float fps_ = 0;
  int width_step_ = 0;
  int next_frame_ = 0;
};

// Writer of the same layout (used by tests and by anybody who decodes footage elsewhere).
bool WriteRawVideoHeader(std::ofstream* ofs, int width, int height, int frames, float fps);

}  // namespace video_framework

#endif  // VSG_HOST_RAW_VIDEO_READER_H_
