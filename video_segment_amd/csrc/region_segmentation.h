// region_segmentation.h -- hierarchical region segmentation on top of the over-segmentation
// (BASELINE configs[4]): the drop-in for
// segmentation::RegionSegmentation (segmentation/region_segmentation.h:131-216), host C++.
//
// The pixel-heavy hot path of the system is the dense over-segmentation (GPU); this stage works
// on a few thousand regions per chunk set -- an agglomerative clustering whose every merge
// re-evaluates float histogram distances in a fixed order -- and stays on the host, as SURVEY.md
// 8(f) plans it.  It consumes the dense unit's serialized SegmentationDesc messages unchanged.
#ifndef VSG_REGION_SEGMENTATION_H_
#define VSG_REGION_SEGMENTATION_H_

#include <memory>
#include <string>
#include <vector>

#include "host_model.h"

namespace vsg {

// RegionSegmentationOptions (region_segmentation.h:41-83).
struct RegionSegOptions {
  int min_region_num = 10;
  int max_region_num = 10000;
  float level_cutoff_fraction = 0.8f;
  float small_region_penalizer = 0.25f;
  int luminance_bins = 10, color_bins = 20, flow_bins = 16;
  int chunk_set_size = 6, chunk_set_overlap = 2, constraint_chunks = 1;
  bool use_appearance = true, use_flow = true, use_size_penalizer = true;
  bool compute_vectorization = true;
  bool save_descriptors = false;
};

// cv::cvtColor(BGR -> Lab) for 8-bit frames (OpenCV's fixed-point algorithm restated; un-vendored
// third-party arithmetic, parity unpinned).  dst: W*H*3 bytes, rows packed.
void BgrToLab8(const uint8_t* src, size_t stride, int W, int H, uint8_t* dst);

class RegionSegmentationHost {
 public:
  RegionSegmentationHost(const RegionSegOptions& options, int frame_width, int frame_height);
  ~RegionSegmentationHost();

  // RegionSegmentation::ProcessFrame (region_segmentation.cpp:97-205).  overseg == nullptr: no
  // new frame (flush only).  flow: W*H*2 f32 of this frame or nullptr (first frame / no flow).
  // Returns the number of results now available.  Throws Error(-1) where the reference aborts
  // (glog CHECK).
  int ProcessFrame(bool flush, const SegDesc* overseg, const uint8_t* bgr, size_t stride, const float* flow);
  int num_results() const { return (int)results_.size(); }
  const std::string& result_bytes(int i);
  const SegDesc& result(int i) const { return *results_[(size_t)i]; }

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
  std::vector<std::unique_ptr<SegDesc>> results_;
  std::vector<std::string> encoded_;
};

}  // namespace vsg

#endif  // VSG_REGION_SEGMENTATION_H_
