// segmentation_unit.h -- DenseSegmentationUnit: drop-in for the reference's unit of the same name
// (segmentation/segmentation_unit.h:52-124, segmentation_unit.cpp:48-178), running the dense
// over-segmentation on an MI355X through the C ABI in include/vsg.h.
//
// Same stream contract: needs a BGR24 "VideoStream", optionally a "BackwardFlowStream", appends one
// "SegmentationStream" whose frames are PointerFrame<SegmentationDesc>.  Without the protobuf C++
// runtime (not present in this image) SegmentationDesc here is the serialized message plus a small
// read-only decoder; a build against the reference parses the same bytes with
// segmentation::SegmentationDesc::ParseFromArray.
#ifndef VSG_HOST_SEGMENTATION_UNIT_H_
#define VSG_HOST_SEGMENTATION_UNIT_H_

#include <list>
#include <memory>
#include <string>
#include <vector>

#include "../../include/vsg.h"
#include "video_framework.h"

namespace segmentation {

using namespace video_framework;

// Serialized segmentation.proto SegmentationDesc (segment_util/segmentation.proto:55-172).
struct SegmentationDesc {
  std::string wire;
  // Renders Region2D ids into a W*H image (SegmentationDescToIdImage level 0,
  // segment_util/segmentation_util.cpp:741-770).  Returns false on malformed input.
  bool ToIdImage(int width, int height, std::vector<int32_t>* out) const;
  int NumRegions() const;
  int NumHierarchyLevels() const;
  // frame_width (field 4) / frame_height (field 5); false if absent or malformed.
  bool FrameSize(int* width, int* height) const;
};

// Same fields and defaults as the reference (dense_segmentation.h:42-95).
struct DenseSegmentationOptions {
  enum Presmoothing { PRESMOOTH_NONE = 0, PRESMOOTH_GAUSSIAN = 1, PRESMOOTH_BILATERAL = 2 };
  Presmoothing presmoothing = PRESMOOTH_BILATERAL;
  float frac_min_region_size = 0.01f;
  int chunk_size = 20;
  float chunk_overlap_ratio = 0.2f;
  bool two_stage_oversegment = false;
  int num_constraint_frames = 1;
  bool thin_structure_suppression = false;
  bool enforce_n4_connectivity = true;
  bool enforce_spatial_connectedness = true;
  enum ColorDistance { COLOR_DISTANCE_L1 = 0, COLOR_DISTANCE_L2 = 1 };
  ColorDistance color_distance = COLOR_DISTANCE_L2;
  bool compute_vectorization = false;
};

// Non-owning view of image memory, standing in for the cv::Mat views the reference hands around
// (VideoFrame::MatView, DenseFlowFrame::MatViewInterleaved; video_unit.cpp:75-78,
// flow_reader.cpp:55-61).  type: 0 = 8UC3 (BGR24), 1 = 32FC2 (interleaved flow).
struct MatView {
  enum Type { TYPE_8UC3 = 0, TYPE_32FC2 = 1 };
  const void* data = nullptr;
  int rows = 0, cols = 0;
  size_t step = 0;   // bytes between rows
  Type type = TYPE_8UC3;
  bool empty() const { return data == nullptr; }
};

// Host-side drop-in for segmentation::DenseSegmentation (dense_segmentation.h:112-186): same
// ProcessFrame / ChunkSize contract, executed by the MI355X library behind the C ABI
// (vsg_stream_*, include/vsg.h).  Results are owned by the caller like the reference's
// std::unique_ptr<SegmentationDesc>.
class DenseSegmentation {
 public:
  // device: HIP device ordinal (-1: the caller's current device).
  DenseSegmentation(const DenseSegmentationOptions& options, int frame_width, int frame_height,
                    int device = -1);
  virtual ~DenseSegmentation();
  DenseSegmentation(const DenseSegmentation&) = delete;
  DenseSegmentation& operator=(const DenseSegmentation&) = delete;

  // features: {BGR24 frame} (null only together with flush); flow: null when the unit has no
  // flow stream at all, an empty view for the first frame.  Returns the number of results.
  int ProcessFrame(bool flush, const std::vector<MatView>* features, const MatView* flow,
                   std::vector<std::unique_ptr<SegmentationDesc>>* results);
  int ChunkSize() const { return options_.chunk_size; }
  // False when the library could not create the stream (no HIP device, bad options); the
  // message is in vsg_last_error().
  bool ok() const { return stream_ != nullptr; }

 protected:
  const DenseSegmentationOptions& options() const { return options_; }
  int frame_width() const { return frame_width_; }
  int frame_height() const { return frame_height_; }

 private:
  DenseSegmentationOptions options_;
  int frame_width_, frame_height_;
  vsg_stream* stream_ = nullptr;
};

struct DenseSegmentationUnitOptions {
  std::string video_stream_name = "VideoStream";
  std::string flow_stream_name = "BackwardFlowStream";
  std::string segment_stream_name = "SegmentationStream";
  int device = -1;   // HIP device ordinal (-1: current)
};

// Derive to redefine the features handed to the segmentation (segmentation_unit.h:63-124).
class DenseSegmentationUnit : public VideoUnit {
 public:
  DenseSegmentationUnit(const DenseSegmentationUnitOptions& options,
                        const DenseSegmentationOptions* dense_seg_options);
  virtual ~DenseSegmentationUnit();

  virtual bool OpenStreams(StreamSet* set);
  virtual void ProcessFrame(FrameSetPtr input, std::list<FrameSetPtr>* output);
  virtual bool PostProcess(std::list<FrameSetPtr>* append);

  int output_frames() const { return output_frames_; }

 protected:
  // The reference's extension points (segmentation_unit.h:79-88).
  // Override to look up feature specific streams; called during OpenStreams.
  virtual bool OpenFeatureStreams(StreamSet* set);
  // Extracts the features of a FrameSet (default: the BGR24 video frame); they are passed to
  // DenseSegmentation::ProcessFrame.
  virtual void ExtractFrameSetFeatures(FrameSetPtr input, std::vector<MatView>* features);
  // Returns the DenseSegmentation to use; called during OpenStreams after OpenFeatureStreams.
  virtual std::unique_ptr<DenseSegmentation> CreateDenseSegmentation();

  int video_stream_idx() const { return video_stream_idx_; }
  int flow_stream_idx() const { return flow_stream_idx_; }   // -1: no flow
  int frame_width() const { return frame_width_; }
  int frame_height() const { return frame_height_; }
  const DenseSegmentationUnitOptions& options() const { return options_; }
  const DenseSegmentationOptions& dense_seg_options() const { return dense_seg_options_; }

 private:
  void OutputSegmentation(std::vector<std::unique_ptr<SegmentationDesc>>* results,
                          std::list<FrameSetPtr>* output);

  int video_stream_idx_ = -1;
  int flow_stream_idx_ = -1;
  DenseSegmentationUnitOptions options_;
  DenseSegmentationOptions dense_seg_options_;
  int frame_width_ = 0, frame_height_ = 0;
  int input_frames_ = 0, output_frames_ = 0;
  std::unique_ptr<DenseSegmentation> dense_seg_;
  std::list<FrameSetPtr> frame_set_buffer_;
};

// ---- hierarchical stage -----------------------------------------------------------------------
// Same fields and defaults as the reference (region_segmentation.h:41-83).
struct RegionSegmentationOptions {
  int min_region_num = 10;
  int max_region_num = 10000;
  float level_cutoff_fraction = 0.8f;
  float small_region_penalizer = 0.25f;
  int luminance_bins = 10;
  int color_bins = 20;
  int flow_bins = 16;
  int chunk_set_size = 6;
  int chunk_set_overlap = 2;
  int constraint_chunks = 1;
  bool use_appearance = true;
  bool use_flow = true;
  bool use_size_penalizer = true;
  bool compute_vectorization = true;
  bool save_descriptors = false;
};

// Host-side drop-in for segmentation::RegionSegmentation (region_segmentation.h:131-216) behind
// the C ABI (vsg_regionseg_*, include/vsg.h).
class RegionSegmentation {
 public:
  RegionSegmentation(const RegionSegmentationOptions& options, int frame_width, int frame_height);
  virtual ~RegionSegmentation();
  RegionSegmentation(const RegionSegmentation&) = delete;
  RegionSegmentation& operator=(const RegionSegmentation&) = delete;

  // segmentation + features ({BGR24 frame[, flow: empty view for the first frame]}) are either
  // both set or both null (flush only).  Returns the number of results.
  int ProcessFrame(bool flush, const SegmentationDesc* segmentation, const std::vector<MatView>* features,
                   std::vector<std::unique_ptr<SegmentationDesc>>* results);
  bool ok() const { return handle_ != nullptr; }

 private:
  RegionSegmentationOptions options_;
  vsg_regionseg* handle_ = nullptr;
};

struct RegionSegmentationUnitOptions {   // segmentation_unit.h:126-137
  std::string video_stream_name = "VideoStream";
  std::string flow_stream_name = "BackwardFlowStream";
  std::string segment_stream_name = "SegmentationStream";
  bool free_video_frames = false;
  bool free_flow_frames = true;
};

// Drop-in for the reference's RegionSegmentationUnit (segmentation_unit.h:139-199,
// segmentation_unit.cpp:180-331): replaces the over-segmentation in "SegmentationStream" by the
// hierarchical segmentation, chunk set by chunk set.
class RegionSegmentationUnit : public VideoUnit {
 public:
  RegionSegmentationUnit(const RegionSegmentationUnitOptions& options,
                         const RegionSegmentationOptions* region_options);
  virtual ~RegionSegmentationUnit();

  virtual bool OpenStreams(StreamSet* set);
  virtual void ProcessFrame(FrameSetPtr input, std::list<FrameSetPtr>* output);
  virtual bool PostProcess(std::list<FrameSetPtr>* append);

 protected:
  virtual bool OpenFeatureStreams(StreamSet* set);
  virtual std::unique_ptr<RegionSegmentation> CreateRegionSegmentation();
  virtual void ExtractFrameSetFeatures(FrameSetPtr input, std::vector<MatView>* features);

  int video_stream_idx() const { return video_stream_idx_; }
  int flow_stream_idx() const { return flow_stream_idx_; }
  int frame_width() const { return frame_width_; }
  int frame_height() const { return frame_height_; }
  const RegionSegmentationUnitOptions& options() const { return options_; }

 private:
  void OutputSegmentation(std::vector<std::unique_ptr<SegmentationDesc>>* results,
                          std::list<FrameSetPtr>* output);

  int video_stream_idx_ = -1, flow_stream_idx_ = -1, seg_stream_idx_ = -1;
  RegionSegmentationUnitOptions options_;
  RegionSegmentationOptions region_options_;
  std::unique_ptr<RegionSegmentation> region_seg_;
  int frame_width_ = 0, frame_height_ = 0, num_input_frames_ = 0;
  std::list<FrameSetPtr> frame_set_buffer_;
};

}  // namespace segmentation

#endif  // VSG_HOST_SEGMENTATION_UNIT_H_
