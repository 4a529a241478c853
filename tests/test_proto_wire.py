"""The serialized results must be valid `segmentation.SegmentationDesc` messages of the reference's
schema (segment_util/segmentation.proto:55-172): a consumer built against the reference (converter,
renderer, RegionSegmentationUnit) parses them with protobuf.

No protoc is available, so the schema is rebuilt here programmatically (field numbers and types
as in the reference's .proto) and the protobuf runtime parses the bytes.
"""
import numpy as np
import pytest
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

import oracle_lib as ol
import synth

F = descriptor_pb2.FieldDescriptorProto


def _field(msg, name, number, ftype, label=F.LABEL_OPTIONAL, type_name=None, packed=None,
           default=None):
    f = msg.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
        f.type_name = type_name
    if packed is not None:
        f.options.packed = packed
    if default is not None:
        f.default_value = default
    return f


def build_schema():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "segmentation_test.proto"
    fd.package = "segmentation"
    fd.syntax = "proto2"
    rf = fd.message_type.add()
    rf.name = "RegionFeatures"
    _field(rf, "id", 1, F.TYPE_FIXED32, F.LABEL_REQUIRED)
    sd = fd.message_type.add()
    sd.name = "SegmentationDesc"
    ras = sd.nested_type.add()
    ras.name = "Rasterization"
    si = ras.nested_type.add()
    si.name = "ScanInterval"
    for i, n in enumerate(["y", "left_x", "right_x"]):
        _field(si, n, i + 1, F.TYPE_INT32, F.LABEL_REQUIRED)
    _field(ras, "scan_inter", 1, F.TYPE_MESSAGE, F.LABEL_REPEATED,
           ".segmentation.SegmentationDesc.Rasterization.ScanInterval")
    sm = sd.nested_type.add()
    sm.name = "ShapeMoments"
    for i, n in enumerate(["size", "mean_x", "mean_y", "moment_xx", "moment_xy", "moment_yy"]):
        _field(sm, n, i + 1, F.TYPE_FLOAT)
    vm = sd.nested_type.add()
    vm.name = "VectorMesh"
    _field(vm, "coord", 1, F.TYPE_FLOAT, F.LABEL_REPEATED, packed=True)
    pg = sd.nested_type.add()
    pg.name = "Polygon"
    _field(pg, "coord_idx", 1, F.TYPE_INT32, F.LABEL_REPEATED, packed=True)
    _field(pg, "hole", 2, F.TYPE_BOOL, default="false")
    vz = sd.nested_type.add()
    vz.name = "Vectorization"
    _field(vz, "polygon", 1, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".segmentation.SegmentationDesc.Polygon")
    r2 = sd.nested_type.add()
    r2.name = "Region2D"
    _field(r2, "id", 1, F.TYPE_INT32, F.LABEL_REQUIRED)
    _field(r2, "raster", 3, F.TYPE_MESSAGE, type_name=".segmentation.SegmentationDesc.Rasterization")
    _field(r2, "shape_moments", 5, F.TYPE_MESSAGE, type_name=".segmentation.SegmentationDesc.ShapeMoments")
    _field(r2, "vectorization", 6, F.TYPE_MESSAGE, type_name=".segmentation.SegmentationDesc.Vectorization")
    cr = sd.nested_type.add()
    cr.name = "CompoundRegion"
    _field(cr, "id", 1, F.TYPE_INT32, F.LABEL_REQUIRED)
    _field(cr, "size", 2, F.TYPE_INT32, F.LABEL_REQUIRED)
    _field(cr, "neighbor_id", 3, F.TYPE_INT32, F.LABEL_REPEATED)
    _field(cr, "parent_id", 4, F.TYPE_INT32, default="-1")
    _field(cr, "child_id", 5, F.TYPE_INT32, F.LABEL_REPEATED)
    _field(cr, "start_frame", 6, F.TYPE_INT32)
    _field(cr, "end_frame", 7, F.TYPE_INT32)
    hl = sd.nested_type.add()
    hl.name = "HierarchyLevel"
    _field(hl, "region", 2, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".segmentation.SegmentationDesc.CompoundRegion")
    en = sd.enum_type.add()
    en.name = "Connectedness"
    for n, v in (("N4_CONNECT", 1), ("N8_CONNECT", 2)):
        e = en.value.add()
        e.name, e.number = n, v
    _field(sd, "region", 2, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".segmentation.SegmentationDesc.Region2D")
    _field(sd, "hierarchy", 3, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".segmentation.SegmentationDesc.HierarchyLevel")
    _field(sd, "frame_width", 4, F.TYPE_INT32, default="0")
    _field(sd, "frame_height", 5, F.TYPE_INT32, default="0")
    _field(sd, "chunk_size", 6, F.TYPE_INT32)
    _field(sd, "overlap_start", 7, F.TYPE_INT32)
    _field(sd, "chunk_id", 8, F.TYPE_INT32, default="-1")
    _field(sd, "hierarchy_frame_idx", 9, F.TYPE_INT32, default="0")
    _field(sd, "features", 10, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".segmentation.RegionFeatures")
    _field(sd, "vector_mesh", 11, F.TYPE_MESSAGE, type_name=".segmentation.SegmentationDesc.VectorMesh")
    _field(sd, "connectedness", 12, F.TYPE_ENUM, type_name=".segmentation.SegmentationDesc.Connectedness",
           default="N4_CONNECT")
    _field(sd, "rasterization_removed", 13, F.TYPE_BOOL, default="false")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    desc = pool.FindMessageTypeByName("segmentation.SegmentationDesc")
    try:
        return message_factory.GetMessageClass(desc)
    except AttributeError:  # older protobuf
        return message_factory.MessageFactory(pool).GetPrototype(desc)


def test_results_parse_as_segmentation_desc():
    Msg = build_schema()
    W, H, N, chunk = 64, 48, 20, 8
    s = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=True)
    fl = synth.const_flow(W, H)
    frame_idx = 0
    chunk_ids = []
    for k in range(N):
        n = s.process_frame(synth.bench_frame(W, H, k), fl if k > 0 else None, flush=(k == N - 1))
        for i in range(n):
            raw = s.result_bytes(i)
            m = Msg()
            m.ParseFromString(raw)
            assert m.IsInitialized()
            assert m.SerializeToString() == raw          # canonical field order
            assert (m.frame_width, m.frame_height) == (W, H)
            assert m.connectedness == 1                   # N4_CONNECT
            ids = [r.id for r in m.region]
            if m.chunk_id > 0:
                assert ids == sorted(ids)
            covered = np.zeros((H, W), np.int32)
            for r in m.region:
                area = 0
                for iv in r.raster.scan_inter:
                    covered[iv.y, iv.left_x:iv.right_x + 1] += 1
                    area += iv.right_x - iv.left_x + 1
                assert r.shape_moments.size == float(area)
            assert (covered == 1).all()                   # a partition of the frame
            assert (len(m.hierarchy) == 1) == (i == 0)    # hierarchy on the chunk's first frame
            if i == 0:
                assert m.hierarchy_frame_idx == frame_idx
                hier_ids = {c.id for c in m.hierarchy[0].region}
                assert set(ids) <= hier_ids
                for c in m.hierarchy[0].region:
                    assert not c.HasField("parent_id") and len(c.child_id) == 0
                    assert c.start_frame <= c.end_frame
            chunk_ids.append(m.chunk_id)
            frame_idx += 1
    assert frame_idx == N
    assert chunk_ids == sorted(chunk_ids) and chunk_ids[0] == 0 and chunk_ids[-1] >= 1
