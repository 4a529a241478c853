// seg_tree_synth.cpp -- the caller of the hot path, mirroring seg_tree_sample's over-segmentation
// wiring (seg_tree_sample/seg_tree.cpp:85-367): root unit -> [flow] -> DenseSegmentationUnit ->
// sink, run single threaded.  The H.264 reader and the .flow reader of the reference are replaced
// by an in-process synthetic source (no codec is available in this image).
//
//   seg_tree_synth --width 64 --height 48 --frames 45 --flow 1 --input probe
// prints the number of over-segmented frames, Region2D counts and the FNV-1a-32 hash of all region
// id images (the quantity pinned in SURVEY.md App. B), then __SEGMENTATION_FINISHED__.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

#include "flow_reader.h"
#include "raw_video_reader.h"
#include "segmentation_io.h"
#include "segmentation_unit.h"

using namespace video_framework;
using namespace segmentation;

namespace {

uint32_t PcgHash(uint32_t v) {
  const uint32_t state = v * 747796405u + 2891336453u;
  const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}

// Same generators as tests/synth.py (probe_frame / bench_frame / const_flow).
class SyntheticVideoUnit : public VideoUnit {
 public:
  SyntheticVideoUnit(int width, int height, int frames, bool flow, bool bench,
                     const std::string& save_flow = std::string())
      : width_(width), height_(height), frames_(frames), flow_(flow), bench_(bench),
        save_flow_(save_flow) {}

  bool OpenStreams(StreamSet* set) override {
    width_step_ = (width_ * 3 + 3) / 4 * 4;   // padded like video_reader_unit.cpp:200-206
    set->push_back(std::shared_ptr<DataStream>(
        new VideoStream(width_, height_, width_step_, 25.0f, PIXEL_FORMAT_BGR24, "VideoStream")));
    if (flow_) {
      set->push_back(std::shared_ptr<DataStream>(
          new DenseFlowStream(width_, height_, "BackwardFlowStream")));
    }
    if (!save_flow_.empty()) {   // what DenseFlowUnit does with --save_flow (flow_reader.cpp:240-248)
      flow_writer_.reset(new DenseFlowWriter(save_flow_));
      if (!flow_writer_->OpenAndWriteHeader(width_, height_, FLOW_BACKWARD)) return false;
    }
    return true;
  }

  bool PostProcess(std::list<FrameSetPtr>* append) override {
    if (k_ >= frames_) {
      if (flow_writer_) flow_writer_->Close();
      return false;
    }
    FrameSetPtr fs(new FrameSet);
    std::shared_ptr<VideoFrame> vf(new VideoFrame(width_, height_, 3, width_step_, (int64_t)k_ * 40000));
    uint8_t* d = vf->mutable_data();
    const int cw = bench_ ? std::max(1, 16 * width_ / 64) : 16;
    const int ch = bench_ ? std::max(1, 12 * width_ / 64) : 12;
    const uint32_t base = PcgHash((uint32_t)(1234 + k_));
    for (int y = 0; y < height_; ++y) {
      uint8_t* row = d + (size_t)y * width_step_;
      for (int x = 0; x < width_; ++x) {
        int b = x * 255 / width_, g = y * 255 / height_;
        const int chk = (((x + 2 * k_) / cw) % 2) ^ ((y / ch) % 2);
        int r = bench_ ? chk * 160 + 40 : (chk ? 200 : 40);
        if (bench_) {
          const uint32_t i = (uint32_t)((y * width_ + x) * 3);
          b += (int)(PcgHash(i + base) % 7u) - 3;
          g += (int)(PcgHash(i + 1 + base) % 7u) - 3;
          r += (int)(PcgHash(i + 2 + base) % 7u) - 3;
        }
        row[3 * x] = (uint8_t)std::min(255, std::max(0, b));
        row[3 * x + 1] = (uint8_t)std::min(255, std::max(0, g));
        row[3 * x + 2] = (uint8_t)std::min(255, std::max(0, r));
      }
    }
    fs->push_back(vf);
    if (flow_ || flow_writer_) {
      std::shared_ptr<DenseFlowFrame> ff(new DenseFlowFrame(width_, height_, true, vf->pts()));
      float* f = ff->mutable_flow();
      for (size_t i = 0; i < (size_t)width_ * height_; ++i) {
        f[2 * i] = -2.0f;
        f[2 * i + 1] = 0.0f;
      }
      if (flow_writer_ && k_ > 0) flow_writer_->AddFlowFrame(f);   // no field for frame 0
      if (flow_) fs->push_back(ff);
    }
    append->push_back(fs);
    ++k_;
    return true;
  }

 private:
  int width_, height_, frames_;
  bool flow_, bench_;
  std::string save_flow_;
  std::unique_ptr<DenseFlowWriter> flow_writer_;
  int width_step_ = 0;
  int k_ = 0;
};

// Consumes "SegmentationStream" like the reference's writer / renderer units do
// (segmentation_unit.cpp:376-379, 562-565).
class HashSinkUnit : public VideoUnit {
 public:
  bool OpenStreams(StreamSet* set) override {
    seg_idx_ = FindStreamIdx("SegmentationStream", set);
    if (seg_idx_ < 0) return false;
    const SegmentationStream& s = set->at(seg_idx_)->As<SegmentationStream>();
    width_ = s.frame_width();
    height_ = s.frame_height();
    return true;
  }
  void ProcessFrame(FrameSetPtr input, std::list<FrameSetPtr>* output) override {
    const SegmentationDesc& desc = input->at(seg_idx_)->As<PointerFrame<SegmentationDesc>>().Ref();
    std::vector<int32_t> ids;
    VF_CHECK(desc.ToIdImage(width_, height_, &ids), "malformed SegmentationDesc");
    for (int32_t v : ids) {
      for (int b = 0; b < 4; ++b) {
        hash_ ^= (uint32_t)((uint32_t)v >> (8 * b)) & 0xffu;
        hash_ *= 16777619u;
      }
    }
    if (frames_ == 0) first_regions_ = desc.NumRegions();
    total_regions_ += desc.NumRegions();
    bytes_ += desc.wire.size();
    ++frames_;
    output->push_back(input);
  }
  uint32_t hash() const { return hash_; }
  int frames() const { return frames_; }
  int first_regions() const { return first_regions_; }
  long total_regions() const { return total_regions_; }
  size_t bytes() const { return bytes_; }

 private:
  int seg_idx_ = -1, width_ = 0, height_ = 0, frames_ = 0, first_regions_ = 0;
  long total_regions_ = 0;
  size_t bytes_ = 0;
  uint32_t hash_ = 2166136261u;
};

}  // namespace

// --read_pb FILE: reads a segmentation container back with SegmentationReader and prints what the
// sink prints for a live run (frames, regions, label hash), without touching the GPU.
int ReadBack(const std::string& file) {
  SegmentationReader reader(file);
  if (!reader.OpenFileAndReadHeaders()) return 1;
  int width = 0, height = 0;
  if (reader.NumFrames() > 0 && !reader.SegmentationResolution(&width, &height)) return 1;
  uint32_t hash = 2166136261u;
  long total_regions = 0;
  int first_regions = 0;
  size_t bytes = 0;
  std::vector<int32_t> ids;
  for (int k = 0; reader.RemainingFrames() > 0; ++k) {
    SegmentationDesc desc;
    if (!reader.ReadNextFrame(&desc) || !desc.ToIdImage(width, height, &ids)) return 1;
    for (int32_t v : ids) {
      for (int b = 0; b < 4; ++b) {
        hash ^= (uint32_t)((uint32_t)v >> (8 * b)) & 0xffu;
        hash *= 16777619u;
      }
    }
    if (k == 0) first_regions = desc.NumRegions();
    total_regions += desc.NumRegions();
    bytes += desc.wire.size();
  }
  std::printf("frames=%d first_frame_regions=%d total_regions=%ld label_fnv1a32=%08x bytes=%zu "
              "width=%d height=%d header_flags=%zu last_pts=%lld\n",
              reader.NumFrames(), first_regions, total_regions, hash, bytes, width, height,
              reader.GetHeaderFlags().size(),
              reader.NumFrames() ? (long long)reader.TimeStamps().back() : 0ll);
  return 0;
}

int main(int argc, char** argv) {
  int width = 64, height = 48, frames = 45, chunk = 20, flow = 1, device = -1;
  std::string input = "probe", write_to_file, flow_file, save_flow, input_file, read_pb;
  for (int i = 1; i + 1 < argc; i += 2) {
    const std::string k = argv[i];
    const char* v = argv[i + 1];
    if (k == "--width") width = atoi(v);
    else if (k == "--height") height = atoi(v);
    else if (k == "--frames") frames = atoi(v);
    else if (k == "--chunk_size") chunk = atoi(v);
    else if (k == "--flow") flow = atoi(v);
    else if (k == "--input") input = v;
    else if (k == "--device") device = atoi(v);
    else if (k == "--write_to_file") write_to_file = v;   // seg_tree.cpp:65
    else if (k == "--flow_file") flow_file = v;           // <input>.flow, seg_tree.cpp:121-125
    else if (k == "--save_flow") save_flow = v;           // seg_tree.cpp:67, 177-179
    else if (k == "--input_file") input_file = v;         // raw BGR24 video, seg_tree.cpp:45
    else if (k == "--read_pb") read_pb = v;
    else {
      std::fprintf(stderr, "unknown flag %s\n", k.c_str());
      return 2;
    }
  }
  if (!read_pb.empty()) return ReadBack(read_pb);
  // With --input_file the video comes from a raw BGR24 file and, as in seg_tree.cpp:120-126, the
  // flow from "<input base>.flow" when that file exists.
  std::unique_ptr<RawVideoReaderUnit> raw_reader;
  if (!input_file.empty()) {
    RawVideoReaderOptions ro;
    ro.trim_frames = 0;
    raw_reader.reset(new RawVideoReaderUnit(ro, input_file));
    if (flow && flow_file.empty()) {
      const std::string candidate = input_file.substr(0, input_file.find_last_of(".")) + ".flow";
      if (std::ifstream(candidate.c_str()).good()) flow_file = candidate;
      else flow = 0;   // no flow unit in this build: segment without temporal displacement
    }
  }
  // With --flow_file the flow comes from DenseFlowReaderUnit instead of the source
  // (seg_tree.cpp:164-169).
  const bool flow_from_file = flow != 0 && !flow_file.empty();
  SyntheticVideoUnit source(width, height, frames, flow != 0 && !flow_from_file, input == "bench",
                            save_flow);
  std::unique_ptr<DenseFlowReaderUnit> flow_reader;
  VideoUnit* root_unit = raw_reader ? static_cast<VideoUnit*>(raw_reader.get()) : &source;
  VideoUnit* input_unit = root_unit;
  if (flow_from_file) {
    flow_reader.reset(new DenseFlowReaderUnit(DenseFlowReaderOptions(), flow_file));
    flow_reader->AttachTo(input_unit);
    input_unit = flow_reader.get();
  }
  DenseSegmentationUnitOptions unit_options;
  if (!flow) unit_options.flow_stream_name.clear();   // seg_tree.cpp:195-198
  unit_options.device = device;
  DenseSegmentationOptions seg_options;
  seg_options.chunk_size = chunk;
  DenseSegmentationUnit dense_unit(unit_options, &seg_options);
  HashSinkUnit sink;
  dense_unit.AttachTo(input_unit);
  sink.AttachTo(&dense_unit);
  std::unique_ptr<SegmentationWriterUnit> writer;
  if (!write_to_file.empty()) {   // seg_tree.cpp:296-312
    SegmentationWriterUnitOptions wo;
    wo.filename = write_to_file;
    writer.reset(new SegmentationWriterUnit(wo));
    writer->AttachTo(&sink);
  }

  if (!root_unit->PrepareProcessing()) {
    std::fprintf(stderr, "ERROR: setup failed\n");
    return 1;
  }
  const auto t0 = std::chrono::steady_clock::now();
  root_unit->Run();
  if (raw_reader) frames = raw_reader->num_frames();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("frames=%d first_frame_regions=%d total_regions=%ld label_fnv1a32=%08x bytes=%zu "
              "seconds=%.3f fps=%.2f\n",
              sink.frames(), sink.first_regions(), sink.total_regions(), sink.hash(), sink.bytes(),
              dt, sink.frames() / dt);
  std::fprintf(stderr, "__SEGMENTATION_FINISHED__\n");
  return sink.frames() == frames ? 0 : 3;
}
