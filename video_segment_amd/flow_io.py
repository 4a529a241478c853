"""Reader / writer of the reference's precomputed dense-flow files (`<video>.flow`).

Format (video_framework/flow_reader.cpp:63-88 reader, :240-248 / :283-301 writer; little endian):

    int32 width, int32 height, int32 flow_type        0 forward, 1 backward, 2 both
    per video frame k >= 1: width*height*2 f32 interleaved (x, y); forward first when both

Frame 0 of the video has no field.  `DenseFlowReader.fields()` yields what DenseFlowReaderUnit
pushes into the frame set for each video frame (None for frame 0), ready for
`DenseSegmentation.process_frame(frame, flow)`.
"""
import struct

import numpy as np

FLOW_FORWARD, FLOW_BACKWARD, FLOW_BOTH = 0, 1, 2


class DenseFlowReader:
    """Mirror of video_framework::DenseFlowReader (flow_reader.h:76-101)."""

    def __init__(self, filename):
        self.filename = filename
        self.f = None
        self.width = self.height = 0
        self.flow_type = FLOW_FORWARD

    def open_and_read_header(self):
        self.f = open(self.filename, "rb")
        head = self.f.read(12)
        if len(head) != 12:
            raise ValueError("malformed .flow header in %s" % self.filename)
        self.width, self.height, self.flow_type = struct.unpack("<iii", head)
        if self.width <= 0 or self.height <= 0 or self.flow_type not in (0, 1, 2):
            raise ValueError("malformed .flow header in %s" % self.filename)
        return True

    def required_buffer_size(self):
        return 4 * self.width * self.height * 2

    def more_frames_available(self):
        pos = self.f.tell()
        more = len(self.f.read(1)) == 1
        self.f.seek(pos)
        return more

    def get_next_flow_frame(self):
        buf = self.f.read(self.required_buffer_size())
        if len(buf) != self.required_buffer_size():
            raise ValueError("truncated flow field in %s" % self.filename)
        return np.frombuffer(buf, dtype="<f4").reshape(self.height, self.width, 2).copy()

    def fields(self, backward=True):
        """Per video frame: None for frame 0, then the backward (default) or forward field."""
        if self.f is None:
            self.open_and_read_header()
        want = FLOW_BACKWARD if backward else FLOW_FORWARD
        if self.flow_type not in (want, FLOW_BOTH):
            raise ValueError("file holds no %s flow" % ("backward" if backward else "forward"))
        yield None
        while self.more_frames_available():
            if self.flow_type == FLOW_BOTH:
                fwd = self.get_next_flow_frame()
                bwd = self.get_next_flow_frame()
                yield bwd if backward else fwd
            else:
                yield self.get_next_flow_frame()

    def close(self):
        if self.f:
            self.f.close()
            self.f = None


class DenseFlowWriter:
    """What DenseFlowUnit does when `flow_output_file` is set (flow_reader.cpp:240-301)."""

    def __init__(self, filename):
        self.filename = filename
        self.f = None

    def open_and_write_header(self, width, height, flow_type=FLOW_BACKWARD):
        self.f = open(self.filename, "wb")
        self.width, self.height = width, height
        self.f.write(struct.pack("<iii", width, height, flow_type))
        return True

    def add_flow_frame(self, field):
        a = np.ascontiguousarray(field, dtype="<f4")
        assert a.shape == (self.height, self.width, 2), a.shape
        self.f.write(a.tobytes())

    def close(self):
        self.f.close()
        self.f = None
