// segmentation_io.cpp -- see segmentation_io.h.
#include "segmentation_io.h"

#include <cstdio>
#include <cstring>

namespace segmentation {

namespace {
// Little-endian POD records, appended to a byte string that goes out in one write per section.
template <class T>
void Append(std::string* bytes, const T& v) {
  bytes->append(reinterpret_cast<const char*>(&v), sizeof(T));
}
}  // namespace

// Layout: see the header.  Every section is assembled in memory and written in one go; the file
// position is tracked here (file_pos_) because the chunk header stores absolute offsets of the
// frames that follow it and of the next header.
bool SegmentationWriter::OpenFile(const std::vector<int>& header_entries) {
  out_.open(filename_.c_str(), std::ios_base::out | std::ios_base::binary | std::ios_base::trunc);
  if (!out_) {
    std::fprintf(stderr, "ERROR: Could not open %s to write!\n", filename_.c_str());
    return false;
  }
  std::string head("HEAD");
  Append(&head, (int32_t)header_entries.size());
  for (int flag : header_entries) Append(&head, (int32_t)flag);
  out_.write(head.data(), (std::streamsize)head.size());
  file_pos_ = (int64_t)head.size();
  chunks_written_ = 0;
  frames_written_ = 0;
  pending_.clear();
  return true;
}

void SegmentationWriter::AddSegmentationDataToChunk(const std::string& data, int64_t pts) {
  pending_.push_back(Pending{data, pts});
}

void SegmentationWriter::WriteChunk() {
  const int64_t n = (int64_t)pending_.size();
  // absolute offset of every frame record (tag + size + payload) behind the header, and of what
  // follows the chunk
  int64_t at = file_pos_ + ChunkHeaderBytes(n);
  std::string header("CHNK");
  Append(&header, (int32_t)chunks_written_);
  Append(&header, (int32_t)n);
  for (const Pending& f : pending_) {
    Append(&header, at);
    at += 4 + (int64_t)sizeof(int32_t) + (int64_t)f.wire.size();
  }
  for (const Pending& f : pending_) Append(&header, f.pts);
  Append(&header, at);   // where the next chunk header (or TERM) starts
  out_.write(header.data(), (std::streamsize)header.size());
  for (const Pending& f : pending_) {
    std::string rec("SEGD");
    Append(&rec, (int32_t)f.wire.size());
    out_.write(rec.data(), (std::streamsize)rec.size());
    out_.write(f.wire.data(), (std::streamsize)f.wire.size());
  }
  file_pos_ = at;
  frames_written_ += (int)n;
  ++chunks_written_;
  pending_.clear();
}

void SegmentationWriter::WriteTermHeaderAndClose() {
  if (!pending_.empty()) WriteChunk();
  std::string term("TERM");
  Append(&term, (int32_t)chunks_written_);
  out_.write(term.data(), (std::streamsize)term.size());
  out_.close();
  std::fprintf(stderr, "Wrote a total of %d frames.\n", frames_written_);
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
