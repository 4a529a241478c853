// segmentation_io.h -- writer for the reference's chunked segmentation container
// (segment_util/segmentation_io.h:31-66, segmentation_io.cpp:46-154) and the unit that feeds it
// (SegmentationWriterUnit, segmentation/segmentation_unit.cpp:333-415), so that the HIP path's
// output is a file-level drop-in for segment_converter / segment_renderer / segment_viewer.
//
//   HEAD  int32 M, int32 flags[M]            flags = {use_vectorization = 1, shape_moments = 0}
//   CHNK  int32 id, int32 N, int64 offs[N], int64 pts[N], int64 next_header_offset
//   N x   SEGD  int32 size, bytes[size]      serialized SegmentationDesc
//   TERM  int32 number_of_chunks
#ifndef VSG_HOST_SEGMENTATION_IO_H_
#define VSG_HOST_SEGMENTATION_IO_H_

#include <cstdint>
#include <fstream>
#include <string>
#include <vector>

#include "segmentation_unit.h"

namespace segmentation {

class SegmentationWriter {
 public:
  explicit SegmentationWriter(const std::string& filename) : filename_(filename) {}
  bool OpenFile(const std::vector<int>& header_entries = std::vector<int>());
  void AddSegmentationDataToChunk(const std::string& data, int64_t pts = 0);
  void WriteChunk();
  void WriteTermHeaderAndClose();

 private:
  // A frame waiting for its chunk: the serialized message and its time stamp.
  struct Pending {
    std::string wire;
    int64_t pts;
  };
  // Bytes a chunk header with n frames occupies: tag, id, n, n offsets, n time stamps, link.
  static int64_t ChunkHeaderBytes(int64_t n) { return 4 + 4 + 4 + 16 * n + 8; }

  std::string filename_;
  std::ofstream out_;
  std::vector<Pending> pending_;
  int64_t file_pos_ = 0;        // bytes written so far (the stream is append-only)
  int32_t chunks_written_ = 0;
  int frames_written_ = 0;
};

// Reader of the same container (segment_util/segmentation_io.h:117-170, segmentation_io.cpp:168-300):
// walks the CHNK headers up to TERM, then serves frames by file offset.
class SegmentationReader {
 public:
  explicit SegmentationReader(const std::string& filename) : filename_(filename) {}
  bool OpenFileAndReadHeaders();
  // Width / height of the first frame (SegmentationDesc.frame_width / frame_height).
  bool SegmentationResolution(int* width, int* height);
  bool ReadNextFrameBinary(std::string* data);
  bool ReadNextFrame(SegmentationDesc* desc);
  const std::vector<int32_t>& GetHeaderFlags() const { return header_flags_; }
  const std::vector<int64_t>& TimeStamps() const { return time_stamps_; }
  bool SeekToFrame(int frame);
  int NumFrames() const { return (int)file_offsets_.size(); }
  int RemainingFrames() const { return NumFrames() - curr_frame_; }
  void CloseFile() { ifs_.close(); }

 private:
  std::string filename_;
  std::ifstream ifs_;
  std::vector<int64_t> file_offsets_, time_stamps_;
  std::vector<int32_t> header_flags_;
  int curr_frame_ = 0;
};

struct SegmentationWriterUnitOptions {
  std::string video_stream_name = "VideoStream";
  std::string segment_stream_name = "SegmentationStream";
  std::string filename;
};

// Like the reference's unit, frames are only buffered while streaming and the whole video is
// written as ONE chunk when the stream ends.
class SegmentationWriterUnit : public VideoUnit {
 public:
  explicit SegmentationWriterUnit(const SegmentationWriterUnitOptions& options)
      : options_(options), writer_(options.filename) {}
  bool OpenStreams(StreamSet* set) override;
  void ProcessFrame(FrameSetPtr input, std::list<FrameSetPtr>* output) override;
  bool PostProcess(std::list<FrameSetPtr>* append) override;

 private:
  SegmentationWriterUnitOptions options_;
  SegmentationWriter writer_;
  int seg_stream_idx_ = -1;
  int frame_number_ = 0;
};

struct SegmentationReaderUnitOptions {
  std::string filename;
  std::string segment_stream_name = "SegmentationStream";
};

// Adds the frames of a segmentation container to the FrameSets passing through, or -- used as the
// root of a tree -- produces one FrameSet per stored frame (segmentation_unit.cpp:417-476).
class SegmentationReaderUnit : public VideoUnit {
 public:
  explicit SegmentationReaderUnit(const SegmentationReaderUnitOptions& options)
      : options_(options), reader_(options.filename) {}
  bool OpenStreams(StreamSet* set) override;
  void ProcessFrame(FrameSetPtr input, std::list<FrameSetPtr>* output) override;
  bool PostProcess(std::list<FrameSetPtr>* append) override;

 protected:
  void ReadNextFrame(FrameSetPtr input);

 private:
  SegmentationReaderUnitOptions options_;
  SegmentationReader reader_;
  int seg_stream_index_ = -1;
  int frame_width_ = 0, frame_height_ = 0;
};

}  // namespace segmentation

#endif  // VSG_HOST_SEGMENTATION_IO_H_
