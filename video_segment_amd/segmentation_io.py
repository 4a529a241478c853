"""Reader / writer of the reference's chunked segmentation container (`.pb` files).

Format (segment_util/segmentation_io.h:31-66, little endian):

    HEAD  int32 M, int32 flags[M]
    CHNK  int32 chunk_id, int32 N, int64 frame_offsets[N], int64 pts[N], int64 next_header_offset
    N x   SEGD  int32 size, bytes[size]         (serialized segmentation.SegmentationDesc)
    ...
    TERM  int32 num_chunks

The frames are the byte strings the DenseSegmentation drop-in returns, so a file written here is
consumable by the reference's segment_converter / segment_renderer / segment_viewer.
"""
import struct


class SegmentationWriter:
    """Mirror of segmentation::SegmentationWriter (segmentation_io.cpp:46-154)."""

    def __init__(self, filename):
        self.filename = filename
        self.f = None

    def open_file(self, header_entries=()):
        self.f = open(self.filename, "wb")
        self.num_chunks = 0
        self.total_frames = 0
        self.f.write(b"HEAD" + struct.pack("<i", len(header_entries)))
        for e in header_entries:
            self.f.write(struct.pack("<i", int(e)))
        self.curr_offset = 8 + 4 * len(header_entries)
        self.offsets, self.pts, self.buf = [], [], []
        return True

    def add_segmentation_data_to_chunk(self, data, pts=0):
        self.offsets.append(self.curr_offset)
        self.buf.append(bytes(data))
        self.curr_offset += len(data) + 8
        self.pts.append(int(pts))

    def write_chunk(self):
        n = len(self.offsets)
        chunk_id = self.num_chunks
        self.num_chunks += 1
        header = 4 + 8 + n * 16 + 8
        self.curr_offset += header
        offs = [o + header for o in self.offsets]
        self.f.write(b"CHNK" + struct.pack("<ii", chunk_id, n))
        self.f.write(struct.pack("<%dq" % n, *offs))
        self.f.write(struct.pack("<%dq" % n, *self.pts))
        self.f.write(struct.pack("<q", self.curr_offset))
        for frame in self.buf:
            self.f.write(b"SEGD" + struct.pack("<i", len(frame)) + frame)
        self.total_frames += n
        self.offsets, self.pts, self.buf = [], [], []

    def write_term_header_and_close(self):
        if self.buf:
            self.write_chunk()
        self.f.write(b"TERM" + struct.pack("<i", self.num_chunks))
        self.f.close()
        self.f = None


def read_segmentation_file(filename):
    """Returns (header_flags, [(pts, frame_bytes)], num_chunks); validates every offset."""
    data = open(filename, "rb").read()
    assert data[:4] == b"HEAD", "not a segmentation file"
    (m,) = struct.unpack_from("<i", data, 4)
    flags = list(struct.unpack_from("<%di" % m, data, 8))
    pos = 8 + 4 * m
    frames = []
    chunks = 0
    while True:
        tag = data[pos:pos + 4]
        if tag == b"TERM":
            (num_chunks,) = struct.unpack_from("<i", data, pos + 4)
            assert num_chunks == chunks, (num_chunks, chunks)
            assert pos + 8 == len(data)
            return flags, frames, num_chunks
        assert tag == b"CHNK", tag
        chunk_id, n = struct.unpack_from("<ii", data, pos + 4)
        assert chunk_id == chunks
        offs = struct.unpack_from("<%dq" % n, data, pos + 12)
        pts = struct.unpack_from("<%dq" % n, data, pos + 12 + 8 * n)
        (next_header,) = struct.unpack_from("<q", data, pos + 12 + 16 * n)
        p = pos + 12 + 16 * n + 8
        for i in range(n):
            assert offs[i] == p, (offs[i], p)
            assert data[p:p + 4] == b"SEGD"
            (sz,) = struct.unpack_from("<i", data, p + 4)
            frames.append((pts[i], data[p + 8:p + 8 + sz]))
            p += 8 + sz
        assert next_header == p, (next_header, p)
        pos = p
        chunks += 1
