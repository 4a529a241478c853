"""Finds the device buffer whose first reader expects zeros: runs a parity case in a child process
with VSG_DEVICE_CACHE_POISON=1 restricted to a range of allocation serial numbers
(csrc/device_cache.cpp), halves the range while the case still fails, and prints the call stack of
the allocation that is left.  usage: python tools/poison_bisect.py [case-code]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE = sys.argv[1] if len(sys.argv) > 1 else "run_streams(vsg, 96, 64, 28, 'bench', True, 8)"
CODE = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import video_segment_amd as vsg\n"
        "from test_gpu_parity import run_streams\n"
        "%s\nprint('case ok')\n" % (ROOT, os.path.join(ROOT, "tests"), CASE))


def run(extra):
    env = dict(os.environ, **extra)
    p = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, env=env, timeout=900)
    return p.returncode == 0 and "case ok" in p.stdout, p


base = {}
for k in ("VSG_SPINE_MIN", "VSG_SPINE_CHECK"):
    if k in os.environ:
        base[k] = os.environ[k]
ok, p = run(dict(base))
print("without poison:", "ok" if ok else "FAILS", flush=True)
ok, p = run(dict(base, VSG_DEVICE_CACHE_POISON="1"))
print("all poisoned:", "ok" if ok else "FAILS")
if ok:
    sys.exit(0)
print(p.stderr[-600:])
lo, hi = 0, 4096
while hi - lo > 1:
    mid = (lo + hi) // 2
    ok, _ = run(dict(base, VSG_DEVICE_CACHE_POISON="1", VSG_DEVICE_CACHE_POISON_FROM=str(lo),
                     VSG_DEVICE_CACHE_POISON_TO=str(mid)))
    print("poison [%d, %d): %s" % (lo, mid, "ok" if ok else "FAILS"), flush=True)
    if ok:
        lo = mid
    else:
        hi = mid
ok, p = run(dict(base, VSG_DEVICE_CACHE_POISON="1", VSG_DEVICE_CACHE_POISON_FROM=str(lo),
                 VSG_DEVICE_CACHE_POISON_TO=str(hi), VSG_DEVICE_CACHE_TRACE=str(lo)))
print("allocation %d alone: %s" % (lo, "ok" if ok else "FAILS"))
print("\n".join(l for l in p.stderr.splitlines() if "vsg" in l or "libvsg" in l)[:6000])
