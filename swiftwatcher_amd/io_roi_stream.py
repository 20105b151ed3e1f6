"""Pre-extracted ROI streams: the frame source that makes 10 k frames/s a 3 GB/s problem instead of a 62 GB/s one
(SURVEY.md section 7 "Feeding 10k fps", section 8f rank 2).

The counting loop only ever looks at the chimney's crop region of a frame -- plus, for segment images, at most half the minimum
segment size beyond it (extract_segment_images grows small boxes, image_filtering.py:338-369).  A ROI stream file holds exactly
that rectangle of every frame of a video (424 x 212 + 12 px margin = 317 KB instead of 6.2 MB per 1080p frame), written once by
RoiStreamWriter from any source of decoded frames (the image has no video codec: decoding itself, io_video.py:85-165, stays out of
scope).

RoiStreamReader has the reference FrameReader's surface and bookkeeping (io_video.py:13-82): get_frame / get_n_frames / read_frame,
frames past the end are all-zero "null" frames numbered -1 (:40-44), a failed read re-delivers the last good frame (:51-53) -- and
because end_frame is the frame COUNT while the range test is inclusive, the frame one past the end is requested once and served by
that rule, exactly like VideoReader / HDF5Reader.  What it hands out are RoiFrame objects: they index like the full frame
(frame[y0:y1, x0:x1] in full-frame coordinates, which is all crop_frame and extract_segment_images do) but hold only the stored
rectangle.

get_n_frames(n) delivers a window as views of ONE page-locked block in the order of the file (one read per window).  FrameQueue
pushes with appendleft -- the last frame read is queue position 0 -- and hands the block to the library as it is, with a negative
frame stride: no staging copy, no reversal.  A background thread fills the next window's block while the current window is classified
and tracked.  A block is reused a few windows later; frames still alive then (the tracker's cached frame) get a private copy of their
pixels first, and a window's segment images are cut into one buffer of their own when the window is segmented, so segments kept in
long tracks or in events do not hold frames.
"""
import datetime
import json
import os
import struct
import threading
import weakref

import numpy as np

from . import _lib

MAGIC = b"SWKROI1\n"
_PIXELS_LOCK = threading.Lock()          # RoiFrame.__getitem__ against RoiFrame.detach (two threads of one reader's frames)
# Page-locked blocks of closed readers, by (device, shape): pinning memory costs tens of milliseconds per 100 MB (a 4K ROI's ring of three
# 8-window blocks is 600 MB: 40 ms of a 100-ms video), so a reader that is closed hands the blocks nobody looks at any more to the
# next reader of the same geometry (one video after another on a GPU; a video played again).
_BLOCK_POOL, _BLOCK_POOL_LOCK, _BLOCK_POOL_BYTES = {}, threading.Lock(), 1 << 30          # at most 1 GiB kept page-locked for later readers


_FILL_POOL = None


def _fill_pool():
    global _FILL_POOL
    if _FILL_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _FILL_POOL = ThreadPoolExecutor(4, thread_name_prefix="swk-roi-read")
    return _FILL_POOL


def _take_block(shape, device):
    with _BLOCK_POOL_LOCK:
        free = _BLOCK_POOL.get((device, tuple(shape)))
        if free:
            return free.pop()
    return _lib.pinned_empty(tuple(shape), np.uint8, device=device)


def _give_blocks(blocks, device):
    with _BLOCK_POOL_LOCK:
        held = sum(a.nbytes for free in _BLOCK_POOL.values() for a in free)
        for b in blocks:
            if held + b.nbytes > _BLOCK_POOL_BYTES:
                break          # (the rest is unpinned and freed with its last reference)
            _BLOCK_POOL.setdefault((device, tuple(b.shape)), []).append(b)
            held += b.nbytes


def margin_rect(frame_hw, crop_region, min_seg_size=(24, 24)):
    """(ya, yb, xa, xb): the crop region grown by half the minimum segment size, clipped to the frame -- every pixel
    crop_frame and extract_segment_images can touch."""
    Hf, Wf = frame_hw[:2]
    (x0, y0), (x1, y1) = crop_region
    x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, Wf), min(y1, Hf)
    my, mx = int(min_seg_size[0]) // 2, int(min_seg_size[1]) // 2
    return max(y0 - my, 0), min(y1 + my, Hf), max(x0 - mx, 0), min(x1 + mx, Wf)


class RoiFrame:
    """The stored rectangle of one frame, indexed in FULL-frame coordinates.  frame[ya:yb, xa:xb] (the two slices crop_frame and
    extract_segment_images make) is translated and clipped like numpy clips a slice at the full frame's edges; anything else of
    ndarray's interface is not offered -- the counting loop does not use it."""
    __slots__ = ("roi", "origin", "shape", "dtype", "ndim", "block", "slot", "recycled", "__weakref__")

    def __init__(self, roi, origin, full_shape, block=None, slot=-1):
        self.roi, self.origin, self.shape = roi, origin, tuple(full_shape)
        self.dtype, self.ndim = roi.dtype, len(self.shape)
        self.block, self.slot = block, slot
        self.recycled = False          # True once the pixels sit in a buffer that returns to a reader's spare list with this frame

    def __getitem__(self, key):
        if not (isinstance(key, tuple) and len(key) >= 2 and isinstance(key[0], slice) and isinstance(key[1], slice)):
            raise TypeError("a RoiFrame is indexed with two slices in full-frame coordinates: frame[y0:y1, x0:x1]")
        oy, ox = self.origin
        ys, ye, _ = key[0].indices(self.shape[0])
        xs, xe, _ = key[1].indices(self.shape[1])
        if key[0].step not in (None, 1) or key[1].step not in (None, 1):
            raise TypeError("strided access to a RoiFrame")
        h, w = self.roi.shape[:2]          # (the stored rectangle's size never changes)
        if ys < oy or xs < ox or ye > oy + h or xe > ox + w:
            if ye > ys and xe > xs:
                raise IndexError("rows %d:%d, columns %d:%d leave the stored rectangle (rows %d:%d, columns %d:%d)"
                                 % (ys, ye, xs, xe, oy, oy + h, ox, ox + w))
        # Pixels that live in a reader's block are copied out UNDER THE LOCK detach() takes: the reader's read-ahead thread gives a
        # frame private pixels (and then overwrites the block) while the counting loop's thread may be slicing it -- without the
        # lock a slice could pair the old pixels with the cleared `block` (an uncopied view of a block about to be overwritten) or
        # copy a block that is being refilled.
        with _PIXELS_LOCK:
            roi, block = self.roi, self.block
            out = roi[max(ys - oy, 0):max(ye - oy, 0), max(xs - ox, 0):max(xe - ox, 0)]
            if len(key) > 2:
                out = out[(slice(None), slice(None)) + tuple(key[2:])]
            if block is not None:
                return out.copy()
        # pixels that live in a reader's block are handed out as a copy: the block is reused a few windows later.  Private pixels that
        # go back to the reader's spare list when the frame dies (detach(spare)) are copied too: a view kept beyond the frame's life
        # would change content when the buffer is reused
        return out.copy() if self.recycled else out

    def detach(self, spare=None):
        """Private copy of the pixels: the block they live in is about to be reused.  spare: a list of arrays of the right shape that
        earlier detached frames have given back (fresh pages are expensive; a long video recycles a handful of buffers)."""
        buf = spare.pop() if spare else np.empty_like(self.roi)
        with _PIXELS_LOCK:
            np.copyto(buf, self.roi)
            if spare is not None:
                self.recycled = True
                weakref.finalize(self, spare.append, buf)
            self.roi = buf
            self.block, self.slot = None, -1

    def as_full_frame(self, fill=128):
        """A full-size ndarray with the stored rectangle pasted in (ROI-mask generation reads the first frame's crop region)."""
        full = np.full(self.shape, fill, np.uint8)
        oy, ox = self.origin
        full[oy:oy + self.roi.shape[0], ox:ox + self.roi.shape[1]] = self.roi
        return full


class RoiStreamWriter:
    """with RoiStreamWriter(path, frame_hw, crop_region, fps) as w: w.append(frame) ...   (frames: full decoded BGR frames)"""

    def __init__(self, path, frame_hw, crop_region, fps=30.0, min_seg_size=(24, 24), channels=3):
        self.rect = margin_rect(frame_hw, crop_region, min_seg_size)
        ya, yb, xa, xb = self.rect
        self.meta = dict(version=1, fps=float(fps), frame_hw=[int(frame_hw[0]), int(frame_hw[1])], channels=int(channels),
                         crop_region=[[int(crop_region[0][0]), int(crop_region[0][1])], [int(crop_region[1][0]), int(crop_region[1][1])]],
                         min_seg_size=[int(min_seg_size[0]), int(min_seg_size[1])], rect=[ya, yb, xa, xb], frames=0)
        self._fh = open(path, "wb")
        self._fh.write(MAGIC + struct.pack("<I", 4096))
        self._fh.write(b"\0" * (4096 - len(MAGIC) - 4))              # the JSON header is written on close (frame count)
        self.path = path

    def append(self, frame):
        ya, yb, xa, xb = self.rect
        if tuple(frame.shape[:2]) != tuple(self.meta["frame_hw"]) or frame.dtype != np.uint8:
            raise ValueError("frame shape / dtype differs from the stream's")
        self._fh.write(np.ascontiguousarray(frame[ya:yb, xa:xb]).tobytes())
        self.meta["frames"] += 1

    def close(self):
        if self._fh is None:
            return
        blob = json.dumps(self.meta).encode()
        if len(blob) > 4096 - len(MAGIC) - 4 - 1:
            raise ValueError("header too large")
        self._fh.seek(len(MAGIC) + 4)
        self._fh.write(blob)
        self._fh.close()
        self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def write_roi_stream(path, frames, crop_region, fps=30.0, min_seg_size=(24, 24)):
    first = frames[0]
    with RoiStreamWriter(path, first.shape[:2], crop_region, fps, min_seg_size, channels=1 if first.ndim == 2 else first.shape[2]) as w:
        for f in frames:
            w.append(f)
    return path


class RoiStreamReader:
    """FrameReader over a ROI stream file (io_video.py:13-82 bookkeeping).  prefetch=True reads the next window ahead in a thread."""

    BLOCKS = 6          # one window per block (ahead = 1): a block is reused five windows later, by then few of its frames are still
                        # referenced (tracks last 1-2 windows); blocks of `ahead` windows: a ring of three or more
    MAX_BLOCKS = 32     # the ring grows when most frames of the block in turn are still in use (a caller that reads many windows ahead)

    def __init__(self, path, start=0, end=0, prefetch=True, device=0, ahead=1):
        self.filepath = path
        with open(path, "rb") as fh:
            head = fh.read(len(MAGIC) + 4)
            if head[:len(MAGIC)] != MAGIC:
                raise ValueError("%s is not a ROI stream" % path)
            hdr = struct.unpack("<I", head[len(MAGIC):])[0]
            self.meta = json.loads(fh.read(hdr - len(MAGIC) - 4).split(b"\0", 1)[0].decode())
        m = self.meta
        ya, yb, xa, xb = m["rect"]
        self.rect, self.origin = (ya, yb, xa, xb), (ya, xa)
        self.crop_region = [tuple(m["crop_region"][0]), tuple(m["crop_region"][1])]
        self.min_seg_size = tuple(m["min_seg_size"])
        ch = m["channels"]
        self.roi_shape = (yb - ya, xb - xa) + ((ch,) if ch > 1 else ())
        self.full_shape = (m["frame_hw"][0], m["frame_hw"][1]) + ((ch,) if ch > 1 else ())
        self.count = int(m["frames"])
        # frames are read with preadv straight into their slot of a page-locked block (one system call per frame, no file position
        # shared between threads).  A memory map would do too, but every first touch of a mapped page is a page fault, and 1,600 of
        # them per 21-frame window cost more than the window's GPU work here (measured: 4.3 ms per window against 3.2 from RAM).
        self._fd = os.open(path, os.O_RDONLY)
        self._hdr = hdr
        self._frame_bytes = int(np.prod(self.roi_shape))
        self.fps = m["fps"]
        self.start_frame = start
        self.end_frame = end if end > 0 else self.count
        self.next_frame_number = self.start_frame
        self.total_frames = self.end_frame - self.start_frame
        self.frame_shape = self.full_shape
        self.last_read_frame = None
        self.frames_read = 0
        self.read_errors = 0
        self.device = device
        self._midnight = datetime.datetime.combine(datetime.date.today(), datetime.time())
        self._blocks, self._alive, self._turn, self._used = [], [], 0, 0
        self._spare = []                     # pixel buffers of detached frames that have died since
        self._prefetch = prefetch
        self.ahead = max(int(ahead), 1)      # windows read ahead of the caller (a caller that takes several windows per GPU call sets it)
        self._pending = []                   # windows being read ahead, oldest first: (first frame number, n, (block, alive), "done" event)
        self._jobs = None                    # queue of the read-ahead thread (started with the first window)

    # ---- the reference reader's surface ----
    def frame_number_to_timestamp(self, frame_number):
        from .io_data import frame_timestamp_ns
        return self._midnight + datetime.timedelta(microseconds=frame_timestamp_ns(frame_number, self.fps) // 1000)

    def read_frame(self, frame_number, increment=True):
        """One frame as a RoiFrame with private pixels (None past the end of the file, like a failed cv2 read)."""
        frame = None
        if 0 <= frame_number < self.count:
            pixels = np.empty(self.roi_shape, np.uint8)
            self._read_into(frame_number, pixels)
            frame = RoiFrame(pixels, self.origin, self.full_shape)
        if increment:
            self.next_frame_number += 1
        return frame

    def get_frame(self, frame_number=None):
        if frame_number is None:
            frame_number = self.next_frame_number
        if not self.start_frame <= frame_number <= self.end_frame:
            return RoiFrame(np.zeros(self.roi_shape, np.uint8), self.origin, self.full_shape), -1, "00:00:00.000"
        frame = self.read_frame(frame_number)
        timestamp = self.frame_number_to_timestamp(frame_number)
        if frame is None:
            frame = self.last_read_frame
            self.read_errors += 1
        else:
            self.last_read_frame = frame
            self.frames_read += 1
        return frame, frame_number, timestamp

    def _read_into(self, frame_number, dst):
        got = os.preadv(self._fd, [memoryview(dst).cast("B")], self._hdr + frame_number * self._frame_bytes)
        if got != self._frame_bytes:
            raise IOError("short read of frame %d of %s" % (frame_number, self.filepath))

    def close(self):
        if getattr(self, "_fd", None) is not None:
            for pend in self._pending:
                pend[3].wait()
            self._pending = []
            if self._jobs is not None:
                self._jobs.put(None)
                self._jobs = None
            os.close(self._fd)
            self._fd = None
            if self.last_read_frame is not None and self.last_read_frame.block is not None:
                self.last_read_frame.detach()          # the reader's own reference must not keep a 100-MB block out of the pool
            self._release_blocks()

    def _release_blocks(self):
        """Blocks no live frame looks into go to the pool of page-locked blocks (a frame still alive keeps its block to itself)."""
        free = []
        for blk, alive in zip(self._blocks, self._alive):
            if not any(fr is not None and fr.block is blk for fr in (ref() for ref in alive)):
                free.append(blk)
        self._blocks, self._alive = [], []
        if free:
            _give_blocks(free, self.device)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- a window at a time: one page-locked block in queue order, next window read ahead ----
    def _block(self, n):
        """Where the next window goes: (page-locked block, first slot, its list of live frames).  A block holds `ahead` consecutive
        windows side by side (a caller that takes several windows per GPU call -- the reader that segments ahead, windows_per_call --
        then hands the library ONE contiguous piece of it: no staging copy for a batch either); windows are placed in the order they
        are asked for.  The ring's next block is taken when the current one is full; frames that still live in a block about to be
        reused get private pixels first."""
        group = max(int(self.ahead), 1)
        if not self._blocks or self._blocks[0].shape[0] != group * n:
            ring = self.BLOCKS if group == 1 else max(3, -(-self.BLOCKS // group) + 2)
            self._release_blocks()
            self._blocks = [_take_block((group * n,) + self.roi_shape, self.device) for _ in range(ring)]
            self._alive = [[] for _ in range(ring)]
            self._turn, self._used = 0, group          # (the first call below moves on to block 0)
            self._turn = len(self._blocks) - 1
        if self._used < group:
            base = self._used * n
            self._used += 1
            return self._blocks[self._turn], base, self._alive[self._turn]
        b = (self._turn + 1) % len(self._blocks)
        live = [fr for fr in (ref() for ref in self._alive[b]) if fr is not None and fr.block is self._blocks[b]]
        if len(live) > (group * n) // 2 and len(self._blocks) < self.MAX_BLOCKS:
            # most of that block is still in use (its frames wait to be segmented, or sit in long tracks): a new block instead
            self._blocks.insert(b, _take_block((group * n,) + self.roi_shape, self.device))
            self._alive.insert(b, [])
        else:
            for fr in live:
                fr.detach(self._spare)
            self._alive[b] = []
        self._turn, self._used = b, 1
        return self._blocks[b], 0, self._alive[b]

    def _fill(self, block, first, n):
        """Pixels of frames first .. first + n - 1 into the block, frame k at slot k (the order of the file: ONE read for the window's
        real frames; FrameQueue hands the block over with a negative frame stride, queue position 0 = slot n - 1); nulls are zeros, the
        frame one past the end repeats the last one (get_frame's fallback)."""
        real = [k for k in range(n) if self.start_frame <= first + k <= self.end_frame and first + k < self.count]
        if real:
            k0, k1 = real[0], real[-1]
            want = (k1 - k0 + 1) * self._frame_bytes
            if want >= (8 << 20) and k1 > k0:
                # a large window (a 4K ROI: 23 MB) in four pieces at once: one thread copies out of the page cache at 12-14 GB/s, which
                # at 1.7 ms per window was the slowest stage of a video's loop; preadv releases the GIL
                parts = min(4, k1 - k0 + 1)
                cuts = [k0 + (k1 - k0 + 1) * i // parts for i in range(parts + 1)]

                def piece(a, b):
                    return os.preadv(self._fd, [memoryview(block[a:b]).cast("B")], self._hdr + (first + a) * self._frame_bytes)
                got = sum(_fill_pool().map(piece, cuts[:-1], cuts[1:]))
            else:
                got = os.preadv(self._fd, [memoryview(block[k0:k1 + 1]).cast("B")], self._hdr + (first + k0) * self._frame_bytes)
            if got != want:
                raise IOError("short read of frames %d..%d of %s" % (first + k0, first + k1, self.filepath))
        for k in range(n):
            number = first + k
            if not self.start_frame <= number <= self.end_frame:
                block[k] = 0
            elif number >= self.count:
                if k > 0 and number - 1 < self.count:
                    block[k] = block[k - 1]
                elif self.count > 0 and number - 1 >= self.start_frame:
                    self._read_into(self.count - 1, block[k])          # the window starts with the re-delivered frame: the file's last one
                else:
                    block[k] = 0

    def _reader_loop(self, jobs):
        while True:
            job = jobs.get()
            if job is None:
                return
            block, first, n, done = job
            try:
                self._fill(block, first, n)
            except BaseException as exc:          # surfaces in get_n_frames
                done.error = exc
            done.set()

    def _start_prefetch(self, first, n):
        """Keep `ahead` windows in flight beyond the one just delivered (first = the frame number the next window starts at)."""
        if not self._prefetch:
            return
        if self._pending:
            first = self._pending[-1][0] + n
        while len(self._pending) < self.ahead and first <= self.end_frame:
            if self._jobs is None:
                import queue
                self._jobs = queue.Queue()
                threading.Thread(target=self._reader_loop, args=(self._jobs,), daemon=True).start()
            block, base, alive = self._block(n)
            done = threading.Event()
            done.error = None
            self._jobs.put((block[base:base + n], first, n, done))
            self._pending.append((first, n, (block, base, alive), done))
            first += n

    def _wait(self, pend):
        pend[3].wait()
        if pend[3].error is not None:
            raise pend[3].error

    def get_n_frames(self, n):
        first = self.next_frame_number
        if self._pending and self._pending[0][0] == first and self._pending[0][1] == n:
            pend = self._pending.pop(0)
            self._wait(pend)
            block, base, alive = pend[2]
        else:
            for pend in self._pending:           # read ahead for another position or window size: let it finish, then read here
                self._wait(pend)
            self._pending = []
            block, base, alive = self._block(n)
            self._fill(block[base:base + n], first, n)
        frames, numbers, stamps = [], [], []
        for k in range(n):
            number = self.next_frame_number
            if not self.start_frame <= number <= self.end_frame:
                fr, num, ts = RoiFrame(block[base + k], self.origin, self.full_shape, block, base + k), -1, "00:00:00.000"
            else:
                fr = RoiFrame(block[base + k], self.origin, self.full_shape, block, base + k)
                num, ts = number, self.frame_number_to_timestamp(number)
                self.next_frame_number += 1
                if number < self.count:
                    self.last_read_frame = fr
                    self.frames_read += 1
                else:
                    self.read_errors += 1
            alive.append(weakref.ref(fr))
            frames.append(fr); numbers.append(num); stamps.append(ts)
        self._start_prefetch(self.next_frame_number, n)
        return frames, numbers, stamps
