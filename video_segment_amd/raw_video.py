"""Raw BGR24 video container read by the host mirror's RawVideoReaderUnit
(video_segment_amd/host/raw_video_reader.h): the stand-in for the reference's ffmpeg reader.

    "RAWV"  int32 width  int32 height  int32 pixel_format (0 = BGR24)  int32 frames  float32 fps
    frames x height x width x 3 bytes, rows tightly packed
"""
import struct

import numpy as np


def write_raw_video(path, frames, fps=25.0):
    frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
    h, w, c = frames[0].shape
    assert c == 3
    with open(path, "wb") as f:
        f.write(b"RAWV" + struct.pack("<iiiif", w, h, 0, len(frames), float(fps)))
        for fr in frames:
            assert fr.shape == (h, w, 3)
            f.write(fr.tobytes())


def read_raw_video(path):
    data = open(path, "rb").read()
    assert data[:4] == b"RAWV", "not a raw video file"
    w, h, fmt, n, fps = struct.unpack_from("<iiiif", data, 4)
    assert fmt == 0
    frames = np.frombuffer(data, np.uint8, count=n * h * w * 3, offset=24).reshape(n, h, w, 3)
    return frames, fps
