// host_model.h -- host-side data model of the product: scan-interval rasters, region table,
// SegmentationDesc and its proto2 wire encoding.
//
// Mirrors (without copying) the reference's data model:
//   segment_util/segmentation.proto:55-172        SegmentationDesc and nested messages
//   segmentation/segmentation_common.h:39-116     RegionInformation
//   segment_util/segmentation_util.h:230          Rasterization3D
#ifndef VSG_HOST_MODEL_H_
#define VSG_HOST_MODEL_H_

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace vsg {

struct Interval {
  int y, lx, rx;
};
typedef std::vector<Interval> Raster;

struct RasterSlice {
  int frame;
  Raster raster;
};
typedef std::vector<RasterSlice> Raster3D;   // ordered by frame

struct Moments {
  float size = 0, mean_x = 0, mean_y = 0, xx = 0, xy = 0, yy = 0;
};

// One over-segmentation region (RegionInformation restricted to what the dense path uses).
struct RegionInfo {
  int index = -1;
  int size = 0;
  bool removed = false;        // FLAGGED_FOR_REMOVAL
  bool has_raster = false;     // raster != nullptr in the reference
  std::vector<int> neighbors;  // sorted region indices
  Raster3D raster;
  int constrained_id = -1;
  int region_id = -1;
};

struct PolygonOut {
  std::vector<int> coord_idx;   // index of the point's x coordinate in SegDesc::vector_mesh
  bool hole = false;
};

struct Region2DOut {
  int id = 0;
  Raster raster;
  Moments moments;
  std::vector<PolygonOut> polygons;   // Region2D.vectorization (compute_vectorization)
};

struct CompoundOut {
  int id = 0, size = 0;
  std::vector<int> neighbor_ids;
  bool has_parent = false;        // CompoundRegion.parent_id is set (every level but the top one)
  int parent_id = -1;
  std::vector<int> child_ids;     // CompoundRegion.child_id (levels above the over-segmentation)
  int start_frame = 0, end_frame = 0;
};

struct SegDesc {
  std::vector<Region2DOut> regions;
  bool has_hierarchy = false;
  std::vector<CompoundOut> hierarchy0;
  // SegmentationDesc.hierarchy beyond level 0 (hierarchical RegionSegmentation; empty for the
  // dense unit): upper_levels[l - 1] = HierarchyLevel l.
  std::vector<std::vector<CompoundOut>> upper_levels;
  int frame_width = 0, frame_height = 0;
  int chunk_size = 0, overlap_start = 0, chunk_id = -1, hierarchy_frame_idx = 0;
  int connectedness = 1;   // N4_CONNECT = 1, N8_CONNECT = 2
  bool has_vector_mesh = false;
  std::vector<float> vector_mesh;   // x, y pairs (SegmentationDesc.vector_mesh)
  std::vector<uint32_t> feature_ids;   // SegmentationDesc.features: RegionFeatures.id (save_descriptors)
};

// Boundaries of all regions of a frame and their vectorization (boundary.cpp): fills
// Region2DOut::polygons and SegDesc::vector_mesh.  Regions have to be sorted by id.
void ComputeFrameVectorization(SegDesc* desc);

// Raster utilities (postprocess.cpp).
int RasterArea(const Raster& r);                                   // segmentation_util.cpp:644-650
void MomentsFromRaster(const Raster& r, Moments* m);               // segmentation_util.cpp:652-693
void MergeRasters(const Raster& a, const Raster& b, Raster* out);  // segmentation_util.cpp:484-570
// N4 connected components of a raster, ordered by first interval (segmentation_util.cpp:1025-1101).
void SplitComponentsN4(const Raster& r, std::vector<Raster>* comps);

// Tube analysis of one region (EnforceSpatialConnectedness, dense_segmentation_graph.h:666-861):
// returns the final tubes (each a Raster3D) and their areas; tube_to_keep = index of the tube
// that keeps the region's identity.  flows[frame] = W*H*2 f32 or nullptr; flows may be empty.
struct TubeResult {
  std::vector<Raster3D> tubes;
  std::vector<float> areas;
  int tube_to_keep = -1;
  int tubes_matched = 0;   // tubes after the temporal matching, before the joins (statistics)
};
// One point where the matching reads the backward flow: flow[frame](y, x).
struct FlowRequest {
  int frame, y, x;
};
// The analysis in two steps (see postprocess.cpp): Prepare lists the flow samples it needs,
// Finish takes them (2 floats per request, in request order; null: no flow).
class TubeSplitter {
 public:
  TubeSplitter();
  ~TubeSplitter();
  TubeSplitter(TubeSplitter&&) noexcept;
  void Prepare(const Raster3D& raster, std::vector<FlowRequest>* requests);
  bool MaySplit() const;
  void Finish(int W, int H, const float* flow_xy, TubeResult* out);
#ifdef VSG_TEST_MODELS
  void FinishPlain(int W, int H, const float* flow_xy, TubeResult* out);   // tests/host/tube_plain_model.inc
#endif

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};
// Both steps with host-resident flow fields (flows[frame] = W*H*2 f32).
void SplitRegionIntoTubes(const Raster3D& raster, int W, int H,
                          const std::vector<const float*>& flows, bool have_flows,
                          TubeResult* out);

// proto2 wire encoding of SegmentationDesc (field order = field number order).
std::string EncodeSegDesc(const SegDesc& d);
// ... and decoding of what a consumer of the dense unit's output needs: Region2D ids and
// rasterizations, the hierarchy levels, the frame / chunk fields (shape moments and vector data are
// recomputed by whoever needs them).  Returns false on a malformed message.
bool DecodeSegDesc(const uint8_t* data, size_t len, SegDesc* d);
// Renders the Region2D ids into a W*H image (SegmentationDescToIdImage, level 0).
void RenderIdImage(const SegDesc& d, int W, int32_t* out);

}  // namespace vsg

#endif  // VSG_HOST_MODEL_H_
