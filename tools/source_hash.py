"""sha256 (first 16 hex digits) over the kernel sources of the library (video_segment_amd/csrc/*.hip,
*.h, *.cpp, sorted by name) and over what else decides which kernels run: the build flags
(csrc/Makefile) and the public header (include/vsg.h).  tools/measure_round.sh stores it in the summaries it writes under
profiles/, and bench.py only quotes counters from a summary whose hash equals the one of the
sources it runs from: a kernel change without a new measurement yields `traffic: null`, not stale
numbers."""
import glob
import hashlib
import os
import sys


def source_hash(root=None):
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    files = []
    for pat in ("*.hip", "*.h", "*.cpp"):
        files += glob.glob(os.path.join(root, "video_segment_amd", "csrc", pat))
    files = sorted(files)
    files.append(os.path.join(root, "video_segment_amd", "csrc", "Makefile"))
    files.append(os.path.join(root, "include", "vsg.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash(sys.argv[1] if len(sys.argv) > 1 else None))
