// seg_tree_synth.cpp -- the caller of the hot path, mirroring seg_tree_sample's over-segmentation
// wiring (seg_tree_sample/seg_tree.cpp:85-367): root unit -> [flow] -> DenseSegmentationUnit ->
// sink, as a threaded pipeline (--use_pipeline, the default as in the reference) or single threaded.  The H.264 reader and the .flow reader of the reference are replaced
// by an in-process synthetic source (no codec is available in this image).
//
//   seg_tree_synth --width 64 --height 48 --frames 45 --flow --input probe [--nouse_pipeline]
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
#include "video_pipeline.h"

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
  // kind: 0 probe, 1 bench, 2 soft (tests/synth.py: soft_frame, the input of the hierarchical stage)
  SyntheticVideoUnit(int width, int height, int frames, bool flow, int kind,
                     const std::string& save_flow = std::string())
      : width_(width), height_(height), frames_(frames), flow_(flow), bench_(kind != 0), soft_(kind == 2),
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
        int b = soft_ ? 96 + x * 64 / width_ : x * 255 / width_;
        int g = soft_ ? 96 + y * 64 / height_ : y * 255 / height_;
        const int chk = (((x + 2 * k_) / cw) % 2) ^ ((y / ch) % 2);
        int r = soft_ ? chk * 30 + 110 : (bench_ ? chk * 160 + 40 : (chk ? 200 : 40));
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
  bool flow_, bench_, soft_;
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
    if (frames_ == 0) {
      first_regions_ = desc.NumRegions();
      first_levels_ = desc.NumHierarchyLevels();
    }
    total_regions_ += desc.NumRegions();
    bytes_ += desc.wire.size();
    ++frames_;
    output->push_back(input);
  }
  uint32_t hash() const { return hash_; }
  int frames() const { return frames_; }
  int first_regions() const { return first_regions_; }
  int first_levels() const { return first_levels_; }
  long total_regions() const { return total_regions_; }
  size_t bytes() const { return bytes_; }

 private:
  int seg_idx_ = -1, width_ = 0, height_ = 0, frames_ = 0, first_regions_ = 0, first_levels_ = 0;
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

// Flags: the reference's names where it has them (seg_tree.cpp:52-72, dense_segmentation.cpp:39-46),
// gflags syntax (--flag=value, --flag value, --flag / --noflag for booleans).
struct Flags {
  // seg_tree_sample
  bool flow = true;
  std::string input_file;
  bool use_pipeline = true;
  bool over_segment = false;       // as in the reference: also turns the vectorization on (this
                                   // driver always stops after the dense over-segmentation)
  bool write_to_file = false;      // writes <input_file>.pb (or --output_file)
  bool save_flow = false;          // writes <input base>.flow from the synthetic source
  // dense_segmentation.cpp
  std::string dense_smoothing = "bilateral";   // none | bilateral
  std::string dense_color_dist = "l2";         // l1 | l2
  double dense_min_region_size = 0.01;         // frac_min_region_size
  // this driver only
  int width = 64, height = 48, frames = 45, chunk_size = 20, device = -1;
  std::string input = "probe";     // synthetic generator: probe | bench | soft
  // seg_tree.cpp:219-241 runs the RegionSegmentationUnit unless --over_segment is given; here it is
  // opt-in, so that the over-segmentation hashes of the default run stay what the pins say
  bool region_segmentation = false;
  int chunk_set_size = 6, chunk_set_overlap = 2, min_region_num = 10;
  std::string output_file, flow_file, read_pb;
  std::string flow_output_file;    // DenseFlowOptions::flow_output_file: explicit path for --save_flow
  double pipeline_max_rate = 0;    // seg_tree.cpp:349 uses 20 frames/s for its root
  bool two_stage_oversegment = false;
};

bool ParseFlags(int argc, char** argv, Flags* f) {
  auto as_bool = [](const std::string& v) { return !(v == "0" || v == "false" || v == "no"); };
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a.rfind("--", 0) != 0) {
      std::fprintf(stderr, "unexpected argument %s\n", a.c_str());
      return false;
    }
    a = a.substr(2);
    std::string v;
    bool has_v = false;
    const size_t eq = a.find('=');
    if (eq != std::string::npos) {
      v = a.substr(eq + 1);
      a = a.substr(0, eq);
      has_v = true;
    }
    static const char* kBools[] = {"flow", "use_pipeline", "over_segment", "write_to_file", "save_flow",
                                   "two_stage_oversegment", "region_segmentation"};
    bool is_bool = false, negated = false;
    for (const char* b : kBools) {
      if (a == b) is_bool = true;
      if (a == std::string("no") + b) {
        is_bool = negated = true;
        a = b;
      }
    }
    if (!has_v && !(is_bool && (i + 1 >= argc || std::string(argv[i + 1]).rfind("--", 0) == 0)) &&
        !negated) {
      if (i + 1 >= argc) {
        std::fprintf(stderr, "flag --%s needs a value\n", a.c_str());
        return false;
      }
      v = argv[++i];
      has_v = true;
    }
    const bool bv = negated ? false : (has_v ? as_bool(v) : true);
    if (a == "flow") f->flow = bv;
    else if (a == "use_pipeline") f->use_pipeline = bv;
    else if (a == "over_segment") f->over_segment = bv;
    else if (a == "write_to_file") f->write_to_file = bv;
    else if (a == "save_flow") f->save_flow = bv;
    else if (a == "two_stage_oversegment") f->two_stage_oversegment = bv;
    else if (a == "region_segmentation") f->region_segmentation = bv;
    else if (a == "chunk_set_size") f->chunk_set_size = atoi(v.c_str());
    else if (a == "chunk_set_overlap") f->chunk_set_overlap = atoi(v.c_str());
    else if (a == "min_region_num") f->min_region_num = atoi(v.c_str());
    else if (a == "input_file") f->input_file = v;
    else if (a == "dense_smoothing") f->dense_smoothing = v;
    else if (a == "dense_color_dist") f->dense_color_dist = v;
    else if (a == "dense_min_region_size") f->dense_min_region_size = atof(v.c_str());
    else if (a == "width") f->width = atoi(v.c_str());
    else if (a == "height") f->height = atoi(v.c_str());
    else if (a == "frames") f->frames = atoi(v.c_str());
    else if (a == "chunk_size") f->chunk_size = atoi(v.c_str());
    else if (a == "device") f->device = atoi(v.c_str());
    else if (a == "input") f->input = v;
    else if (a == "output_file") f->output_file = v;
    else if (a == "flow_file") f->flow_file = v;
    else if (a == "flow_output_file") f->flow_output_file = v;
    else if (a == "read_pb") f->read_pb = v;
    else if (a == "pipeline_max_rate") f->pipeline_max_rate = atof(v.c_str());
    else {
      std::fprintf(stderr, "unknown flag --%s\n", a.c_str());
      return false;
    }
  }
  return true;
}

int main(int argc, char** argv) {
  Flags FLAGS;
  if (!ParseFlags(argc, argv, &FLAGS)) return 2;
  if (!FLAGS.read_pb.empty()) return ReadBack(FLAGS.read_pb);
  bool use_flow = FLAGS.flow;
  int frames = FLAGS.frames;
  std::string flow_file = FLAGS.flow_file;
  const std::string input_base = FLAGS.input_file.substr(0, FLAGS.input_file.find_last_of("."));

  // Root: a raw BGR24 file (--input_file) or the synthetic source; with --input_file the flow
  // comes from "<input base>.flow" when that file exists (seg_tree.cpp:120-126).
  std::unique_ptr<RawVideoReaderUnit> raw_reader;
  if (!FLAGS.input_file.empty()) {
    RawVideoReaderOptions ro;
    ro.trim_frames = 0;
    raw_reader.reset(new RawVideoReaderUnit(ro, FLAGS.input_file));
    if (use_flow && flow_file.empty()) {
      const std::string candidate = input_base + ".flow";
      if (std::ifstream(candidate.c_str()).good()) flow_file = candidate;
      else use_flow = false;   // no flow unit in this build: segment without temporal displacement
    }
  }
  const bool flow_from_file = use_flow && !flow_file.empty();
  const std::string save_flow =
      !FLAGS.flow_output_file.empty() ? FLAGS.flow_output_file
      : FLAGS.save_flow ? (FLAGS.input_file.empty() ? std::string("synth.flow") : input_base + ".flow")
                        : std::string();
  SyntheticVideoUnit source(FLAGS.width, FLAGS.height, frames, use_flow && !flow_from_file,
                            FLAGS.input == "bench" ? 1 : (FLAGS.input == "soft" ? 2 : 0), save_flow);
  VideoUnit* root = raw_reader ? static_cast<VideoUnit*>(raw_reader.get()) : &source;
  VideoUnit* input = root;

  // Pipeline segments as in seg_tree.cpp:155-163, 211-217: reader | [flow reader] dense
  // segmentation | sink + writer, each on its own thread.
  std::vector<std::unique_ptr<VideoPipelineSource>> sources;
  std::vector<std::unique_ptr<VideoPipelineSink>> sinks;
  auto cut = [&]() {
    sinks.emplace_back(new VideoPipelineSink());
    sinks.back()->AttachTo(input);
    sources.emplace_back(new VideoPipelineSource(sinks.back().get()));
    input = sources.back().get();
  };
  if (FLAGS.use_pipeline) cut();

  std::unique_ptr<DenseFlowReaderUnit> flow_reader;
  if (flow_from_file) {   // seg_tree.cpp:164-169
    flow_reader.reset(new DenseFlowReaderUnit(DenseFlowReaderOptions(), flow_file));
    flow_reader->AttachTo(input);
    input = flow_reader.get();
  }

  DenseSegmentationUnitOptions unit_options;
  if (!use_flow) unit_options.flow_stream_name.clear();   // seg_tree.cpp:195-198
  unit_options.device = FLAGS.device;
  DenseSegmentationOptions seg_options;
  seg_options.chunk_size = FLAGS.chunk_size;
  seg_options.two_stage_oversegment = FLAGS.two_stage_oversegment;
  if (FLAGS.over_segment) seg_options.compute_vectorization = true;   // seg_tree.cpp:202-204
  // dense_segmentation.cpp:79-101: the flags override the options.
  seg_options.frac_min_region_size = (float)FLAGS.dense_min_region_size;
  if (FLAGS.dense_smoothing == "none") seg_options.presmoothing = DenseSegmentationOptions::PRESMOOTH_NONE;
  else if (FLAGS.dense_smoothing == "bilateral") seg_options.presmoothing = DenseSegmentationOptions::PRESMOOTH_BILATERAL;
  else if (FLAGS.dense_smoothing == "gaussian") seg_options.presmoothing = DenseSegmentationOptions::PRESMOOTH_GAUSSIAN;
  else {
    std::fprintf(stderr, "ERROR: --dense_smoothing %s is not supported (none | gaussian | bilateral)\n",
                 FLAGS.dense_smoothing.c_str());
    return 2;
  }
  if (FLAGS.dense_color_dist == "l1") seg_options.color_distance = DenseSegmentationOptions::COLOR_DISTANCE_L1;
  else if (FLAGS.dense_color_dist == "l2") seg_options.color_distance = DenseSegmentationOptions::COLOR_DISTANCE_L2;
  else {
    std::fprintf(stderr, "ERROR: unknown --dense_color_dist %s (l1 | l2)\n", FLAGS.dense_color_dist.c_str());
    return 2;
  }
  DenseSegmentationUnit dense_unit(unit_options, &seg_options);
  dense_unit.AttachTo(input);
  input = &dense_unit;
  if (FLAGS.use_pipeline) cut();

  std::unique_ptr<RegionSegmentationUnit> region_unit;   // seg_tree.cpp:219-241
  if (FLAGS.region_segmentation && !FLAGS.over_segment) {
    RegionSegmentationUnitOptions ro;
    if (!use_flow) ro.flow_stream_name.clear();
    RegionSegmentationOptions rso;
    rso.chunk_set_size = FLAGS.chunk_set_size;
    rso.chunk_set_overlap = FLAGS.chunk_set_overlap;
    rso.min_region_num = FLAGS.min_region_num;
    region_unit.reset(new RegionSegmentationUnit(ro, &rso));
    region_unit->AttachTo(input);
    input = region_unit.get();
    if (FLAGS.use_pipeline) cut();
  }

  HashSinkUnit sink;
  sink.AttachTo(input);
  input = &sink;
  std::unique_ptr<SegmentationWriterUnit> writer;
  if (FLAGS.write_to_file || !FLAGS.output_file.empty()) {   // seg_tree.cpp:296-312
    SegmentationWriterUnitOptions wo;
    wo.filename = !FLAGS.output_file.empty() ? FLAGS.output_file
                  : (FLAGS.input_file.empty() ? std::string("synth.pb") : FLAGS.input_file + ".pb");
    writer.reset(new SegmentationWriterUnit(wo));
    writer->AttachTo(input);
    input = writer.get();
  }

  if (!root->PrepareProcessing()) {
    std::fprintf(stderr, "ERROR: Setup failed.\n");
    return 1;
  }
  const auto t0 = std::chrono::steady_clock::now();
  if (!FLAGS.use_pipeline) {
    root->Run();
  } else {   // seg_tree.cpp:339-364
    VideoPipelineInvoker invoker;
    RatePolicy pipeline_policy;
    pipeline_policy.max_rate = (float)FLAGS.pipeline_max_rate;
    invoker.RunRootRateLimited(pipeline_policy, root);
    for (size_t k = 0; k + 1 < sources.size(); ++k) invoker.RunPipelineSource(sources[k].get());
    sources.back()->Run();   // the last segment runs on the main thread
    invoker.WaitUntilPipelineFinished();
  }
  if (raw_reader) frames = raw_reader->num_frames();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("frames=%d first_frame_regions=%d total_regions=%ld label_fnv1a32=%08x bytes=%zu "
              "seconds=%.3f fps=%.2f pipeline=%d hierarchy_levels=%d\n",
              sink.frames(), sink.first_regions(), sink.total_regions(), sink.hash(), sink.bytes(),
              dt, sink.frames() / dt, FLAGS.use_pipeline ? 1 : 0, sink.first_levels());
  std::fprintf(stderr, "__SEGMENTATION_FINISHED__\n");
  return sink.frames() == frames ? 0 : 3;
}
