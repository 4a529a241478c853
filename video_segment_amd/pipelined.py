"""One video, two chunk engines on one GPU: the chunk chain of multi_gpu.py folded into a single
process.

A DenseSegmentation stream is synchronous: the call that completes a chunk returns after the
merge, the read-out and the host post-processing, and only then the caller can feed the next
chunk's frames -- whose pre-filter, edge weights and bucket sort do not depend on the chunk before
(SURVEY.md 8(e)).  Here even chunks go to one engine and odd chunks to the other, each on its own
host thread: an engine builds the graph of its chunk while the other one still merges, imports the
two label planes of the hand-off right before the frame that completes its chunk
(vsg_stream_expect_halo / vsg_stream_import_halo) and passes its own on afterwards -- the same
protocol, byte for byte the same output, as the multi-GPU chain (run_chain) and as a single stream.
The merge of chunk c+1 still needs the final labels of chunk c: what overlaps is everything else.

Interface of DenseSegmentation (process_frame / result_bytes / close); results arrive in frame
order, possibly a chunk later than from a single stream (`process_frame` returns what is ready,
`flush=True` waits for the rest)."""
import queue
import threading
import time

from .dense_segmentation import DenseSegmentation
from .multi_gpu import product_halo


class _Shared:
    """What the engine threads and the caller share.  The threads hold this object, not the
    PipelinedDenseSegmentation itself, so that an instance nobody closed can still be collected
    (its __del__ then closes the engines)."""

    def __init__(self, chunk):
        self.inq = [queue.Queue(maxsize=2 * chunk) for _ in range(2)]
        self.halo = [queue.Queue() for _ in range(2)]
        self.cond = threading.Condition()
        self.done = {}          # chunk -> ([bytes], VsgTimings, completion time)
        self.error = None


def _drain(q):
    """After an error: keeps taking (and dropping) what the caller feeds, so that a caller blocked in
    a full queue gets to the point where it sees the error."""
    while q.get() is not None:
        pass


def _engine_loop(sh, eng, p, device, hand_over):
    used = False
    try:
        if device is not None:
            import torch
            torch.cuda.set_device(device)
        while True:
            msg = sh.inq[p].get()
            if msg is None:
                return
            c, frame, flow, first, last, flush = msg
            if first:
                if used:
                    eng.restart()
                used = True
                if c > 0:
                    eng.expect_halo()
            if last and c > 0:
                halo = sh.halo[p].get()     # the labels of chunk c - 1 (the other engine)
                if halo is None:            # the other engine failed (or the unit is being closed)
                    _drain(sh.inq[p])
                    return
                virt, cons, scal = halo
                eng.import_halo(virt, cons, scal)
            n = eng.process_frame(frame, flow, flush=flush)
            if last:
                res = [eng.result_bytes(i) for i in range(n)]
                t = eng.last_timings() if hasattr(eng, "last_timings") else None
                if not flush:
                    sh.halo[1 - p].put(hand_over(eng))
                with sh.cond:
                    sh.done[c] = (res, t, time.perf_counter())
                    sh.cond.notify_all()
            elif n:
                raise RuntimeError("results before the end of a chunk")
    except BaseException as e:   # noqa: BLE001 -- handed to the caller's thread
        with sh.cond:
            sh.error = e
            sh.cond.notify_all()
        sh.halo[1 - p].put(None)
        _drain(sh.inq[p])


class PipelinedDenseSegmentation:
    """engine_factory / halo_of: any two engines with the stream interface (process_frame,
    result_bytes, restart, expect_halo, import_halo, close) and a function that returns what an
    engine hands over after a chunk, (labels_virtual, labels_constrained, scalars[4]) -- the CPU
    tests run the routing and the hand-off with two oracle streams.  Default: two product engines
    on `device`, label planes copied on the device (product_halo)."""

    def __init__(self, W, H, options, has_flow=True, device=None, engine_factory=None, halo_of=None):
        self.W, self.H = W, H
        self.chunk = int(options.chunk_size)
        if self.chunk < 3:
            raise ValueError("chunk_size >= 3")
        self.stride = self.chunk - 1
        self.device = None
        if engine_factory is None:
            import torch
            # The library resolves device -1 to the caller's current HIP device (capi.cpp,
            # ResolveDevice); the torch side has to mean the same device, and both engines are
            # pinned to it explicitly -- under torchrun the current device is the local rank's.
            if device is not None:
                dev_index = torch.device(device).index
                if dev_index is None:
                    dev_index = torch.cuda.current_device()
            else:
                dev_index = int(options.device) if int(options.device) >= 0 else torch.cuda.current_device()
            self.device = torch.device("cuda", dev_index)
            pinned = type(options).from_buffer_copy(options)
            pinned.device = dev_index
            engine_factory = lambda: DenseSegmentation(W, H, pinned, has_flow=has_flow)   # noqa: E731
        self._halo_of = halo_of
        self.engines = [engine_factory() for _ in range(2)]
        self._sh = _Shared(self.chunk)
        self._next_chunk = 0     # next chunk whose results are handed out
        self._frames = 0
        self._chunks_fed = 0
        self._ready = []
        self.timings = []        # VsgTimings of the boundaries handed out by the last call
        self.stamps = []         # time.perf_counter() at which every chunk was complete
        hand_over = self._make_hand_over()
        self._threads = [threading.Thread(target=_engine_loop,
                                          args=(self._sh, self.engines[p], p, self.device, hand_over), daemon=True)
                         for p in range(2)]
        for t in self._threads:
            t.start()

    def _make_hand_over(self):
        """What an engine passes on after a chunk: (labels_virtual, labels_constrained, scalars).
        (A function that does not refer to the unit: the engine threads must not keep it alive.)"""
        if self._halo_of is not None:
            return self._halo_of
        W, H, device = self.W, self.H, self.device

        def hand_over(eng):
            import torch
            virt, cons, scal = product_halo(eng, W, H, device)
            torch.cuda.current_stream().synchronize()   # the copies, before the other thread reads them
            return virt, cons, scal.numpy()
        return hand_over

    # ---- caller side ------------------------------------------------------------------------------
    def _put(self, p, item):
        """Queue.put that notices a failed engine instead of blocking for ever in a full queue."""
        sh = self._sh
        while True:
            try:
                sh.inq[p].put(item, timeout=0.2)
                return
            except queue.Full:
                with sh.cond:
                    if sh.error is not None:
                        raise sh.error

    def _collect(self, wait_for_chunks):
        """Moves the finished chunks, in order, to the ready list; waits until `wait_for_chunks`
        chunks have been handed out in total."""
        ready, timings = [], []
        sh = self._sh
        with sh.cond:
            while True:
                if sh.error is not None:
                    raise sh.error
                while self._next_chunk in sh.done:
                    res, t, stamp = sh.done.pop(self._next_chunk)
                    ready += res
                    timings.append(t)
                    self.stamps.append(stamp)
                    self._next_chunk += 1
                if self._next_chunk >= wait_for_chunks:
                    break
                sh.cond.wait(0.5)
        self._ready, self.timings = ready, timings
        return len(ready)

    def process_frame(self, bgr, flow=None, flush=False, wait=False):
        """Feeds frame k of the video.  Returns the number of results that are ready
        (result_bytes(i)); wait=True: blocks until every chunk that is complete with this frame
        has been handed out (what a single stream does)."""
        if bgr is None:
            raise ValueError("the pipelined unit flushes with the last frame (flush=True)")
        k, s = self._frames, self.stride
        self._frames += 1
        ends_chunk = k > 0 and k % s == 0
        if ends_chunk:
            c = k // s - 1
            self._put(c % 2, (c, bgr, flow, False, True, flush))
            self._chunks_fed = c + 1
            if not flush:
                self._put((c + 1) % 2, (c + 1, bgr, flow, True, False, False))
        else:
            c = k // s
            self._put(c % 2, (c, bgr, flow, k == 0, flush, flush))
            if flush:
                self._chunks_fed = c + 1
        if flush:
            n = self._collect(self._chunks_fed)
            self._shutdown()
            return n
        return self._collect(self._chunks_fed if wait else 0)

    def result_bytes(self, i):
        return self._ready[i]

    def _shutdown(self):
        # (an engine that failed keeps draining its queue: these puts cannot block for long)
        for p in range(2):
            self._sh.inq[p].put(None)
        for t in self._threads:
            t.join()
        self._threads = []

    def close(self):
        if self._threads:
            for p in range(2):
                self._sh.halo[p].put(None)
            self._shutdown()
        for e in self.engines:
            e.close()
        self.engines = []

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass
