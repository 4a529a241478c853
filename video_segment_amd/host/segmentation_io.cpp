// segmentation_io.cpp -- see segmentation_io.h.
#include "segmentation_io.h"

#include <cstdio>
#include <cstring>

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

// ---- SegmentationReader ------------------------------------------------------------------------
bool SegmentationReader::OpenFileAndReadHeaders() {
  ifs_.open(filename_.c_str(), std::ios_base::in | std::ios_base::binary);
  if (!ifs_) {
    std::fprintf(stderr, "ERROR: could not open segmentation file %s\n", filename_.c_str());
    return false;
  }
  file_offsets_.clear();
  time_stamps_.clear();
  header_flags_.clear();
  curr_frame_ = 0;
  int32_t prev_header_id = -1;
  for (;;) {
    char tag[5] = {0, 0, 0, 0, 0};
    ifs_.read(tag, 4);
    if (!ifs_) {
      std::fprintf(stderr, "ERROR: segmentation file ends without a TERM header\n");
      return false;
    }
    if (std::strcmp(tag, "TERM") == 0) break;
    if (std::strcmp(tag, "HEAD") == 0) {
      int32_t num_entries = 0;
      ifs_.read(reinterpret_cast<char*>(&num_entries), sizeof(num_entries));
      if (!ifs_ || num_entries < 0 || num_entries > (1 << 20)) return false;
      header_flags_.resize((size_t)num_entries);
      ifs_.read(reinterpret_cast<char*>(header_flags_.data()), sizeof(int32_t) * (size_t)num_entries);
      continue;
    }
    if (std::strcmp(tag, "CHNK") != 0) {
      std::fprintf(stderr, "ERROR: parsing error, expected chunk header, found %s\n", tag);
      return false;
    }
    int32_t header_id = 0, num_frames = 0;
    ifs_.read(reinterpret_cast<char*>(&header_id), sizeof(header_id));
    ifs_.read(reinterpret_cast<char*>(&num_frames), sizeof(num_frames));
    if (!ifs_ || header_id != prev_header_id + 1 || num_frames < 0) return false;
    prev_header_id = header_id;
    const size_t old = file_offsets_.size();
    file_offsets_.resize(old + (size_t)num_frames);
    time_stamps_.resize(old + (size_t)num_frames);
    ifs_.read(reinterpret_cast<char*>(file_offsets_.data() + old), sizeof(int64_t) * (size_t)num_frames);
    ifs_.read(reinterpret_cast<char*>(time_stamps_.data() + old), sizeof(int64_t) * (size_t)num_frames);
    int64_t next_header_pos = 0;
    ifs_.read(reinterpret_cast<char*>(&next_header_pos), sizeof(next_header_pos));
    if (!ifs_) return false;
    ifs_.seekg(next_header_pos);
  }
  return true;
}

bool SegmentationReader::SeekToFrame(int frame) {
  if (frame < 0 || frame >= NumFrames()) return false;
  curr_frame_ = frame;
  return true;
}

bool SegmentationReader::ReadNextFrameBinary(std::string* data) {
  if (curr_frame_ >= NumFrames()) return false;
  ifs_.clear();
  ifs_.seekg(file_offsets_[(size_t)curr_frame_]);
  char tag[5] = {0, 0, 0, 0, 0};
  ifs_.read(tag, 4);
  int32_t frame_sz = 0;
  ifs_.read(reinterpret_cast<char*>(&frame_sz), sizeof(frame_sz));
  if (!ifs_ || std::strcmp(tag, "SEGD") != 0 || frame_sz < 0) {
    std::fprintf(stderr, "ERROR: expecting segmentation header, error parsing file\n");
    return false;
  }
  data->resize((size_t)frame_sz);
  if (frame_sz) ifs_.read(&(*data)[0], frame_sz);
  if (!ifs_) return false;
  ++curr_frame_;
  return true;
}

bool SegmentationReader::ReadNextFrame(SegmentationDesc* desc) {
  return ReadNextFrameBinary(&desc->wire) && desc->NumRegions() >= 0;
}

bool SegmentationReader::SegmentationResolution(int* width, int* height) {
  const int playhead = curr_frame_;
  SegmentationDesc first;
  if (!SeekToFrame(0) || !ReadNextFrame(&first)) return false;
  const bool ok = first.FrameSize(width, height);
  if (playhead < NumFrames()) SeekToFrame(playhead);
  return ok;
}

// ---- SegmentationReaderUnit -----------------------------------------------------------------
bool SegmentationReaderUnit::OpenStreams(StreamSet* set) {
  const bool res = reader_.OpenFileAndReadHeaders();
  if (res) reader_.SegmentationResolution(&frame_width_, &frame_height_);
  set->push_back(std::shared_ptr<DataStream>(
      new SegmentationStream(frame_width_, frame_height_, options_.segment_stream_name)));
  seg_stream_index_ = (int)set->size() - 1;
  return res;
}

void SegmentationReaderUnit::ProcessFrame(FrameSetPtr input, std::list<FrameSetPtr>* output) {
  ReadNextFrame(input);
  output->push_back(input);
}

bool SegmentationReaderUnit::PostProcess(std::list<FrameSetPtr>* append) {
  if (reader_.RemainingFrames() > 0) {   // the reader is the source of the tree
    VF_CHECK(seg_stream_index_ == 0, "Reader encountered remaining frames but not used as source.");
    FrameSetPtr input(new FrameSet);
    ReadNextFrame(input);
    append->push_back(input);
    return true;
  }
  return false;
}

void SegmentationReaderUnit::ReadNextFrame(FrameSetPtr input) {
  const int frame = reader_.NumFrames() - reader_.RemainingFrames();
  std::unique_ptr<SegmentationDesc> segmentation(new SegmentationDesc());
  if (!reader_.ReadNextFrame(segmentation.get())) {
    std::fprintf(stderr, "ERROR: Could not read from segmentation.\n");
    return;
  }
  const int64_t pts = frame < (int)reader_.TimeStamps().size() ? reader_.TimeStamps()[(size_t)frame] : 0;
  input->push_back(std::shared_ptr<Frame>(
      new PointerFrame<SegmentationDesc>(std::move(segmentation), pts)));
}

}  // namespace segmentation
