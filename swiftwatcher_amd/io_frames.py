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
