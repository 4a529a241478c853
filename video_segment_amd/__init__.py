"""MI355X-native dense video over-segmentation (drop-in for the reference's
DenseSegmentationUnit hot path).  See DESIGN.md and include/vsg.h."""
from .dense_segmentation import ChunkChain, DenseSegGraph, DenseSegmentation, default_options  # noqa: F401
from .pipelined import PipelinedDenseSegmentation  # noqa: F401
from .region_segmentation import RegionSegmentation, bgr_to_lab, default_region_options  # noqa: F401
from ._lib import (VsgDiagnostics, VsgError, VsgMemoryStats, VsgOptions, VsgRegionOptions, VsgTimings,  # noqa: F401
                   memory_limit, memory_stats, memory_trim)
