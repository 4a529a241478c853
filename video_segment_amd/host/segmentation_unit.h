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

struct DenseSegmentationUnitOptions {
  std::string video_stream_name = "VideoStream";
  std::string flow_stream_name = "BackwardFlowStream";
  std::string segment_stream_name = "SegmentationStream";
  int device = -1;   // HIP device ordinal (-1: current)
};

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
  int video_stream_idx() const { return video_stream_idx_; }
  int flow_stream_idx() const { return flow_stream_idx_; }
  int frame_width() const { return frame_width_; }
  int frame_height() const { return frame_height_; }

 private:
  void OutputSegmentation(int num_results, std::list<FrameSetPtr>* output);

  int video_stream_idx_ = -1;
  int flow_stream_idx_ = -1;
  DenseSegmentationUnitOptions options_;
  DenseSegmentationOptions dense_seg_options_;
  int frame_width_ = 0, frame_height_ = 0;
  int input_frames_ = 0, output_frames_ = 0;
  vsg_stream* dense_seg_ = nullptr;
  std::list<FrameSetPtr> frame_set_buffer_;
};

}  // namespace segmentation

#endif  // VSG_HOST_SEGMENTATION_UNIT_H_
