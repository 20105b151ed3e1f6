"""Frame source with the reference FrameReader's surface (io_video.py:13-82) over frames that are already
decoded (a sequence / array of BGR uint8 images): get_frame / get_n_frames with the same bookkeeping --
out-of-range requests return an all-zero "null" frame numbered -1 (:40-44), a failed read re-delivers the last
good frame and bumps read_errors (:51-53), and, like VideoReader/HDF5Reader, end_frame is the frame COUNT while
the range test is inclusive, so the frame "one past the end" is requested once and served by that fallback
(SURVEY appendix).  Video decoding itself (cv2.VideoCapture / HDF5, io_video.py:85-165) is out of scope."""
import datetime

import numpy as np


class ArrayReader:
    def __init__(self, frames, fps=30.0, start=0, end=0, filepath=None):
        self.frames = frames
        self.filepath = filepath
        self.fps = fps
        self.start_frame = start
        self.end_frame = end if end > 0 else len(frames)
        self.next_frame_number = self.start_frame
        self.total_frames = self.end_frame - self.start_frame
        self.frame_shape = tuple(frames[0].shape) if len(frames) else (0, 0, 0)
        self.last_read_frame = None
        self.frames_read = 0
        self.read_errors = 0
        self._midnight = datetime.datetime.combine(datetime.date.today(), datetime.time())

    def read_frame(self, frame_number, increment=True):
        frame = self.frames[frame_number] if 0 <= frame_number < len(self.frames) else None
        if increment:
            self.next_frame_number += 1
        return frame

    def frame_number_to_timestamp(self, frame_number):
        """:74-82: midnight today + frame_number / fps, rounded to microseconds with pandas' own integer-nanosecond
        arithmetic (io_data.frame_timestamp_ns), so the stamps join the exporter's per-frame table exactly."""
        from .io_data import frame_timestamp_ns
        return self._midnight + datetime.timedelta(microseconds=frame_timestamp_ns(frame_number, self.fps) // 1000)

    def get_frame(self, frame_number=None):
        if frame_number is None:
            frame_number = self.next_frame_number
        if not self.start_frame <= frame_number <= self.end_frame:
            self.next_frame_number += 0
            return np.zeros(self.frame_shape).astype(np.uint8), -1, "00:00:00.000"
        frame = self.read_frame(frame_number)
        timestamp = self.frame_number_to_timestamp(frame_number)
        if frame is None:
            frame = self.last_read_frame
            self.read_errors += 1
        else:
            self.frame_shape = frame.shape
            self.last_read_frame = frame
            self.frames_read += 1
        return frame, frame_number, timestamp

    def get_n_frames(self, n):
        frames, numbers, stamps = [], [], []
        for _ in range(n):
            f, k, t = self.get_frame()
            frames.append(f); numbers.append(k); stamps.append(t)
        return frames, numbers, stamps


class PresegmentingReader:
    """A frame reader that segments ahead of the counting loop.  The reference's loop (__main__.py:71-92) is strictly serial -- read a
    queue-ful, segment it, classify and track its frames, read the next -- so its GPU work per window is a chain of latencies (§6 of
    DESIGN.md: 2.4 ms for 21 frames with the classifier on, whatever the host does).  A reader knows which frames come next: this one
    wraps any reader with the FrameReader surface (ArrayReader, RawFileReader, io_roi_stream.RoiStreamReader), reads `windows` queue-fuls
    ahead in a background thread and segments them in ONE library call (data_structures.segment_windows: the same call the loop's
    windows_per_call mode makes), and starts the classifier on the batch once a classifier has asked for scores.  get_n_frames hands
    out the same frames, numbers and timestamps as the wrapped reader; FrameQueue.segment_queue recognises the window by its frames
    and attaches the prepared segments instead of calling the GPU.  The loop itself stays the reference's, call by call; results are
    identical (windows are independent).  Needs to know what segment_queue will be asked: crop_region, min_seg_size, queue_size (a ROI
    stream carries the first two in its header).  The stage images of such a window are produced only if somebody reads them."""

    def __init__(self, reader, crop_region=None, queue_size=21, windows=8, min_seg_size=None, device=0, params=None, classifier=None):
        import collections
        import weakref
        self.reader = reader
        self.crop_region = crop_region if crop_region is not None else reader.crop_region
        self.min_seg_size = tuple(min_seg_size if min_seg_size is not None else getattr(reader, "min_seg_size", (24, 24)))
        self.queue_size, self.windows, self.device, self.params = int(queue_size), max(int(windows), 1), device, params
        if hasattr(reader, "ahead"):
            reader.ahead = max(reader.ahead, self.windows)
        self._ready = collections.deque()            # windows segmented and not yet handed out: (frames, numbers, stamps, counters)
        self._job = None                             # the batch being read and segmented: (thread event with .result / .error)
        # the classifier whose scoring is started with every batch's segmentation: handed in (classifier=..., or by the counting loop:
        # pipeline.swift_counting_algorithm sets it), else learned from the first batch whose scores a classifier asks for
        # (data_structures.WindowBatch) -- which never happens in time when a video starts with frames without segments (the
        # classifier returns early on an empty list): such a video then scored frame by frame until a batch's first window had some
        self._classifier_hint = weakref.ref(classifier) if classifier is not None else None
        self._delivered = 0
        self._handed = collections.deque()           # keys of the windows handed out last (their entries go when the caller has moved on)
        self._exhausted = False
        self._sync_counters()

    def set_classifier(self, classifier):
        """Score every batch for this classifier as soon as it is segmented (the reference builds its classifier after the reader)."""
        import weakref
        self._classifier_hint = weakref.ref(classifier) if classifier is not None else None

    # the wrapped reader runs ahead: the counters the caller sees are those of the windows handed out
    def _sync_counters(self, snap=None):
        r = self.reader
        snap = snap or (r.next_frame_number, r.frames_read, r.read_errors, r.last_read_frame, getattr(r, "frame_shape", None))
        self.next_frame_number, self.frames_read, self.read_errors, self.last_read_frame, self.frame_shape = snap

    def __getattr__(self, name):                     # fps, total_frames, start_frame, end_frame, filepath, read_frame, get_frame, ...
        return getattr(self.reader, name)

    def _work(self, done):
        import threading  # noqa: F401
        from . import data_structures as ds
        try:
            r = self.reader
            triples, snaps = [], []
            for _ in range(self.windows):
                if r.next_frame_number > r.end_frame:            # only null frames from here on: the loop will have stopped
                    self._exhausted = True
                    break
                triples.append(r.get_n_frames(self.queue_size))
                snaps.append((r.next_frame_number, r.frames_read, r.read_errors, r.last_read_frame, getattr(r, "frame_shape", None)))
            out = []
            if triples:
                hint = self._classifier_hint() if self._classifier_hint is not None else None
                info = {}
                popped = ds.segment_windows(triples, self.crop_region, self.min_seg_size, device=self.device, params=self.params,
                                            classifier=hint, owner=self, info=info)
                for (frames, numbers, stamps), snap, frs, iters in zip(triples, snaps, popped, info["iters"]):
                    ds.PRESEGMENTED[id(frames[0])] = ds.Presegmented(list(frames), [f.segments for f in frs], self.crop_region,
                                                                       self.min_seg_size, self.params, iters, self.device)
                    out.append((frames, numbers, stamps, snap))
            done.result = out
        except BaseException as exc:                  # surfaces in get_n_frames
            done.error = exc
        done.set()

    def _start(self):
        import threading
        done = threading.Event()
        done.result, done.error = None, None
        threading.Thread(target=self._work, args=(done,), daemon=True).start()
        self._job = done

    def _collect(self):
        done, self._job = self._job, None
        done.wait()
        if done.error is not None:
            raise done.error
        self._ready.extend(done.result)

    def get_n_frames(self, n):
        if n != self.queue_size or (self._exhausted and not self._ready and self._job is None):
            if self._job is not None:
                self._collect()
            if not self._ready:                       # another window size, or past the end: the wrapped reader as it is
                out = self.reader.get_n_frames(n)
                self._sync_counters()
                return out
        if not self._ready:
            if self._job is None:
                self._start()
            self._collect()
            if not self._ready:
                out = self.reader.get_n_frames(n)
                self._sync_counters()
                return out
        frames, numbers, stamps, snap = self._ready.popleft()
        self._sync_counters(snap)
        self._delivered += 1
        from . import data_structures as ds
        self._handed.append(id(frames[0]))
        while len(self._handed) > 2:                  # a window nobody segmented: its prepared segments are dropped
            ds.PRESEGMENTED.pop(self._handed.popleft(), None)
        # the next batch is started from the second window on (by then a classifier, if there is one, has asked for the first batch's
        # scores: they are computed before the context moves on, and the next batch's are started with its segmentation)
        if self._job is None and not self._exhausted and self._delivered >= 2 and len(self._ready) <= self.windows - 2:
            self._start()
        return frames, numbers, stamps

    def close(self):
        from . import data_structures as ds
        if self._job is not None:
            try:
                self._collect()
            except BaseException:
                pass
        for frames, _, _, _ in self._ready:
            ds.PRESEGMENTED.pop(id(frames[0]), None)
        self._ready.clear()
        while self._handed:
            ds.PRESEGMENTED.pop(self._handed.popleft(), None)
        if hasattr(self.reader, "close"):
            self.reader.close()


class RawFileReader(ArrayReader):
    """ArrayReader over a file of decoded frames that is memory-mapped instead of loaded (SURVEY section 8f rank 2:
    "pre-extracted streams"; the image has no video codec): either a .npy array of shape (frames, H, W, 3) uint8 or a
    headerless raw file of consecutive H x W x 3 BGR frames (frame_shape required).  Frames are handed out as views
    of the mapping, so only the pages the ROI crop touches are ever read from disk."""

    def __init__(self, path, frame_shape=None, fps=30.0, start=0, end=0):
        if str(path).endswith(".npy"):
            frames = np.load(path, mmap_mode="r")
        else:
            if frame_shape is None:
                raise ValueError("a raw frame file needs frame_shape=(H, W, 3)")
            frames = np.memmap(path, dtype=np.uint8, mode="r")
            per = int(np.prod(frame_shape))
            frames = frames[:frames.size // per * per].reshape((-1,) + tuple(frame_shape))
        if frames.dtype != np.uint8 or frames.ndim != 4:
            raise ValueError("expected uint8 frames of shape (frames, H, W, channels)")
        ArrayReader.__init__(self, frames, fps=fps, start=start, end=end, filepath=path)
