"""Drop-in for the reference's swiftwatcher/data_structures.py (FrameQueue :116-217, Frame
:33-63, Segment :16-30): same attributes and call order, but preprocess_queue + segment_queue
run the whole window as ONE call into the HIP library (swk_batch_run) instead of six Python
list comprehensions over OpenCV / SciPy / scikit-image.

Not mirrored: Frame.export_segments (:65-113, PNG debug output, cv2.imwrite)."""
from collections import OrderedDict, deque

import numpy as np

from . import _lib
from . import image_filtering as img

STAGE_KEYS = OrderedDict([("gray", "grayscale"), ("rpca", "RPCA"), ("bilateral", "bilateral"),
                          ("thresh", "thresh_15"), ("opened", "opened"), ("labels", "cc_labeling")])


class Segment:
    """data_structures.py:16-30.  The reference copies every public regionprops attribute
    (about 9 ms per segment); only label/bbox/centroid/area are ever read downstream
    (segment_tracking.py:139-222, segment_classification.py:30,42), so only those exist here."""

    def __init__(self, regionprops, frame_number, timestamp, segment_image):
        self.parent_frame_number = frame_number
        self.parent_timestamp = timestamp
        self.segment_image = segment_image
        self.segment_history = []
        self.status = None
        self.label = regionprops.label
        self.bbox = regionprops.bbox
        self.centroid = regionprops.centroid
        self.area = regionprops.area


class _LazyStages(OrderedDict):
    """processed_frames: values may be zero-argument callables that are resolved on first read
    (a stage image that still lives on the GPU side of the boundary)."""

    def __getitem__(self, key):
        v = OrderedDict.__getitem__(self, key)
        if callable(v):
            v = v()
            OrderedDict.__setitem__(self, key, v)
        return v

    def values(self):
        return [self[k] for k in self.keys()]

    def items(self):
        return [(k, self[k]) for k in self.keys()]


class Frame:
    """data_structures.py:33-63."""

    src_video = None

    def __init__(self, frame=None, frame_number=-1, timestamp="00:00:00.000"):
        self.frame_number = frame_number
        self.timestamp = timestamp
        self.frame = frame
        self.processed_frames = _LazyStages()
        self.segments = []
        self.null = frame_number < 0

    def get_frame(self):
        return self.frame

    def get_processed_frame(self, process_name):
        return self.processed_frames[process_name]

    def get_num_segments(self):
        return len(self.segments)

    def set_segments(self, regionprops_list, segment_images):
        self.segments = [Segment(rp, self.frame_number, self.timestamp, seg)
                         for rp, seg in zip(regionprops_list, segment_images)]


class FrameQueue(deque):
    """data_structures.py:116-217.  Index 0 is the newest frame (appendleft, :134); RPCA column j
    is queue index j (image_filtering.py:234-237)."""

    def __init__(self, queue_size=21, device=0, params=None, keep_stages=True):
        deque.__init__(self, maxlen=queue_size)
        self.frames_read = 0
        self.frames_processed = 0
        self.device = device
        self.params = params
        self.keep_stages = keep_stages
        self.last_iters = None
        self._staging = None

    # ---- container behaviour (reference :126-169): newest frame at index 0, oldest popped first ----
    def is_empty(self):
        return not len(self)

    def push_frame(self, input_frame, frame_number, timestamp):
        self.appendleft(Frame(input_frame, frame_number, timestamp))
        self.frames_read += 1

    def push_list_of_frames(self, frame_list, frame_number_list, timestamp_list):
        for triple in zip(frame_list, frame_number_list, timestamp_list):
            self.push_frame(*triple)

    def pop_frame(self):
        oldest = self.pop()
        self.frames_processed += 0 if oldest.null else 1          # padding frames are not counted (:146-147)
        return oldest

    def store_processed_queue(self, processed_frame_list, process_name):
        for slot, image in zip(self, processed_frame_list):
            slot.processed_frames[process_name] = image

    def store_segmented_queue(self, regionprops_lists, segment_image_list):
        for slot, props, crops in zip(self, regionprops_lists, segment_image_list):
            slot.set_segments(props, crops)

    def get_queue(self):
        return [slot.frame for slot in self]

    def get_processed_queue(self, process_name):
        return [slot.processed_frames[process_name] for slot in self]

    def get_last_processed_queue(self):
        latest = lambda stages: stages[next(reversed(stages))]      # noqa: E731  last stage stored
        return [latest(slot.processed_frames) for slot in self]

    # ---- the hot path, :171-217 ----
    def preprocess_queue(self, crop_region, resize_dim=None):
        """:171-185.  "crop" is a view like in the reference; "grayscale" is produced on the GPU by
        segment_queue's single library call and is resolved lazily if somebody reads it earlier."""
        crops = [img.crop_frame(f, crop_region) for f in self.get_queue()]
        self.store_processed_queue(crops, "crop")
        for pos in range(len(self)):
            crop = crops[pos]
            self[pos].processed_frames["grayscale"] = (lambda c=crop: img.convert_grayscale(c))

    def _stack_crops(self):
        crops = self.get_processed_queue("crop")
        shape = (len(crops),) + crops[0].shape
        if self._staging is None or self._staging.shape != shape:
            self._staging = _lib.pinned_empty(shape, np.uint8)      # page-locked: the upload is one DMA
        for i, c in enumerate(crops):
            self._staging[i] = c
        return self._staging

    def segment_queue(self, min_seg_size, crop_region):
        """:187-217: RPCA -> bilateral -> threshold -> opening -> CCL -> region properties ->
        segment crops, one swk_batch_run for the whole window."""
        if "crop" not in self[0].processed_frames:
            raise RuntimeError("preprocess_queue must run before segment_queue")
        roi = self._stack_crops()
        n = roi.shape[0]
        ctx = _lib.default_context(self.device)
        stages = tuple(STAGE_KEYS) if self.keep_stages else ()
        res = ctx.batch_run(roi, 1, n, params=self.params, stages=stages)
        self.last_iters = int(res["iters"][0])
        for key, name in STAGE_KEYS.items():
            if key in res:
                self.store_processed_queue([res[key][i] for i in range(n)], name)
        if np.any(res["nseg"] > res["segs"].shape[1]):
            raise _lib.SwkError("more regions in a frame than seg_cap")      # cannot happen: labels are u8
        regionprops_lists = [img.regionprops_from_records(res["segs"][i, :res["nseg"][i]]) for i in range(n)]
        segment_images = [img.extract_segment_images(rps, frame, min_seg_size, crop_region)
                          for frame, rps in zip(self.get_queue(), regionprops_lists)]
        self.store_segmented_queue(regionprops_lists, segment_images)


def segment_windows(windows, crop_region, min_seg_size=(24, 24), device=0, params=None):
    """Several FrameQueue-fuls in ONE library call.  windows: list of (frames, frame_numbers, timestamps) triples as
    FrameReader.get_n_frames returns them (oldest frame first), all of the same length n.  Returns one list of Frame
    objects per window in POP order (oldest first, the order __main__.py:81-92 consumes them), segments attached
    exactly as preprocess_queue + segment_queue would have (data_structures.py:171-217).  Windows are independent in
    the reference too (the queue is emptied between them), so batching changes nothing but the launch count."""
    if not windows:
        return []
    n = len(windows[0][0])
    (x0, y0), (x1, y1) = crop_region
    stack = np.empty((len(windows) * n, y1 - y0, x1 - x0) + windows[0][0][0].shape[2:], np.uint8)
    for w, (frames, _, _) in enumerate(windows):
        if len(frames) != n:
            raise ValueError("every window needs the same number of frames")
        for k, f in enumerate(frames):                       # queue index 0 = newest = last frame read (:134)
            stack[w * n + (n - 1 - k)] = f[y0:y1, x0:x1]
    ctx = _lib.default_context(device)
    res = ctx.batch_run(stack, len(windows), n, params=params, stages=())
    if np.any(res["nseg"] > res["segs"].shape[1]):
        raise _lib.SwkError("more regions in a frame than seg_cap")
    out = []
    for w, (frames, numbers, stamps) in enumerate(windows):
        popped = []
        for k in range(n):                                   # oldest first
            slot = w * n + (n - 1 - k)
            fr = Frame(frames[k], numbers[k], stamps[k])
            props = img.regionprops_from_records(res["segs"][slot, :res["nseg"][slot]])
            fr.set_segments(props, img.extract_segment_images(props, frames[k], min_seg_size, crop_region))
            popped.append(fr)
        out.append(popped)
    return out
