// segmentation_io.cpp -- see segmentation_io.h.
#include "segmentation_io.h"

#include <cstdio>

namespace segmentation {

namespace {
template <class T>
void Put(std::ofstream& o, const T& v) {
  o.write(reinterpret_cast<const char*>(&v), sizeof(T));
}
}  // namespace

bool SegmentationWriter::OpenFile(const std::vector<int>& header_entries) {
  header_entries_ = header_entries;
  ofs_.open(filename_.c_str(), std::ios_base::out | std::ios_base::binary | std::ios_base::trunc);
  if (!ofs_) {
    std::fprintf(stderr, "ERROR: Could not open %s to write!\n", filename_.c_str());
    return false;
  }
  num_chunks_ = 0;
  ofs_.write("HEAD", 4);
  const int32_t num_entries = (int32_t)header_entries_.size();
  Put(ofs_, num_entries);
  for (int e : header_entries_) Put(ofs_, (int32_t)e);
  curr_offset_ = 4 + 4 + (int64_t)num_entries * 4;
  return true;
}

void SegmentationWriter::AddSegmentationDataToChunk(const std::string& data, int64_t pts) {
  file_offsets_.push_back(curr_offset_);
  chunk_buffer_.push_back(data);
  curr_offset_ += (int64_t)data.size() + 4 + (int64_t)sizeof(int32_t);
  time_stamps_.push_back(pts);
}

void SegmentationWriter::WriteChunk() {
  const int32_t num_frames = (int32_t)file_offsets_.size();
  const int32_t chunk_id = num_chunks_++;
  ofs_.write("CHNK", 4);
  Put(ofs_, chunk_id);
  Put(ofs_, num_frames);
  const int64_t size_of_header = 4 + 2 * (int64_t)sizeof(int32_t) +
                                 (int64_t)num_frames * 2 * (int64_t)sizeof(int64_t) +
                                 (int64_t)sizeof(int64_t);
  curr_offset_ += size_of_header;
  for (int64_t& o : file_offsets_) o += size_of_header;
  for (int64_t o : file_offsets_) Put(ofs_, o);
  for (int64_t t : time_stamps_) Put(ofs_, t);
  Put(ofs_, curr_offset_);
  for (const std::string& frame : chunk_buffer_) {
    ofs_.write("SEGD", 4);
    Put(ofs_, (int32_t)frame.size());
    ofs_.write(frame.data(), (std::streamsize)frame.size());
  }
  total_frames_ += (int)chunk_buffer_.size();
  chunk_buffer_.clear();
  file_offsets_.clear();
  time_stamps_.clear();
}

void SegmentationWriter::WriteTermHeaderAndClose() {
  if (!chunk_buffer_.empty()) WriteChunk();
  ofs_.write("TERM", 4);
  Put(ofs_, num_chunks_);
  ofs_.close();
  std::fprintf(stderr, "Wrote a total of %d frames.\n", total_frames_);
}

bool SegmentationWriterUnit::OpenStreams(StreamSet* set) {
  if (!options_.video_stream_name.empty() && FindStreamIdx(options_.video_stream_name, set) < 0) {
    std::fprintf(stderr, "ERROR: Could not find Video stream!\n");
    return false;
  }
  seg_stream_idx_ = FindStreamIdx(options_.segment_stream_name, set);
  if (seg_stream_idx_ < 0) {
    std::fprintf(stderr, "ERROR: Could not find Segmentation stream!\n");
    return false;
  }
  frame_number_ = 0;
  return writer_.OpenFile(std::vector<int>{1, 0});   // use vectorization, no shape moments
}

void SegmentationWriterUnit::ProcessFrame(FrameSetPtr input, std::list<FrameSetPtr>* output) {
  const PointerFrame<SegmentationDesc>& seg_frame =
      input->at(seg_stream_idx_)->As<PointerFrame<SegmentationDesc>>();
  writer_.AddSegmentationDataToChunk(seg_frame.Ref().wire, seg_frame.pts());
  output->push_back(input);
  ++frame_number_;
}

bool SegmentationWriterUnit::PostProcess(std::list<FrameSetPtr>* append) {
  writer_.WriteTermHeaderAndClose();
  return false;
}

}  // namespace segmentation
