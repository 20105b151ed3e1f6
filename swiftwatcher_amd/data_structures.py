"""Drop-in for the reference's swiftwatcher/data_structures.py (FrameQueue :116-217, Frame
:33-63, Segment :16-30): same attributes and call order, but preprocess_queue + segment_queue
run the whole window as ONE call into the HIP library (swk_batch_run) instead of six Python
list comprehensions over OpenCV / SciPy / scikit-image.

What a window leaves on the GPU stays there until somebody asks for it:
  * the six stage images of every frame (:183-208 stores them, the counting loop reads none) are values of
    Frame.processed_frames that copy ONE image to the host on first read (_LazyStages over _lib.DevicePlanes);
  * the ROI frames (with a margin of half the minimum segment size, so that segment boxes grow into it exactly like
    they grow into the full frame, image_filtering.py:338-369) and the region records: SegmentClassifier scores ALL
    segments of the window in one batch at the window's first classifier(frame.segments) call (__main__.py:84-85) and
    answers the following calls from that table (WindowBatch);
  * Segment.segment_image is cut (a view of the full frame, like the reference's) when it is first read.

Not mirrored: Frame.export_segments (:65-113, PNG debug output, cv2.imwrite)."""
import weakref
from collections import OrderedDict, deque

import numpy as np

from . import _lib
from . import image_filtering as img

STAGE_KEYS = OrderedDict([("gray", "grayscale"), ("rpca", "RPCA"), ("bilateral", "bilateral"),
                          ("thresh", "thresh_15"), ("opened", "opened"), ("labels", "cc_labeling")])


class _Cut(tuple):
    """(frame, bbox, min_seg_size, crop_region) of a segment image that has not been cut yet."""
    __slots__ = ()


class _Crop(tuple):
    """(buffer, offset, rows, columns, trailing shape) of a segment image that sits in its window's crop buffer (frames of a ROI-stream
    reader: their memory is reused, so the window's crops are copied out in one call and the segments do not hold the frames)."""
    __slots__ = ()


def _window_crops(frames, nseg, live, min_seg_size, crop_region):
    """One buffer with every segment image of the window (extract_segment_images' boxes, image_filtering.py:345-366), for RoiFrames."""
    first = frames[0]
    Hf, Wf = first.shape[:2]
    oy0, ox0 = first.origin
    r0, c0, r1, c1 = (live[k].astype(np.int64) for k in ("r0", "c0", "r1", "c1"))
    d = np.maximum(int(min_seg_size[0]) - (r1 - r0), 0)
    r0, r1 = r0 - d // 2, r1 + (d - d // 2)
    d = np.maximum(int(min_seg_size[1]) - (c1 - c0), 0)
    c0, c1 = c0 - d // 2, c1 + (d - d // 2)
    oy, ox = crop_region[0][1], crop_region[0][0]
    boxes = np.stack([np.clip(r0 + oy, 0, Hf) - oy0, np.clip(r1 + oy, 0, Hf) - oy0,
                      np.clip(c0 + ox, 0, Wf) - ox0, np.clip(c1 + ox, 0, Wf) - ox0], 1)
    h, w = first.roi.shape[:2]
    if len(boxes) and (boxes[:, 0].min() < 0 or boxes[:, 2].min() < 0 or boxes[:, 1].max() > h or boxes[:, 3].max() > w):
        raise ValueError("a segment box leaves the rectangle the ROI stream holds")
    frame_of = np.repeat(np.arange(len(frames)), nseg)
    buf, offsets = _lib.cut_boxes([f.roi for f in frames], frame_of, boxes)
    return buf, offsets.tolist(), np.maximum(boxes[:, 1] - boxes[:, 0], 0).tolist(), np.maximum(boxes[:, 3] - boxes[:, 2], 0).tolist()


def window_segments(segs, nseg, slots, min_seg_size, crop_region, batch=None):
    """Segment objects of a whole batch_run at once: slots = the Frame objects in the batch's frame order; segs / nseg = its
    region records.  The same attributes Segment.__init__ sets (label, bbox, centroid = sum / area in float64, area), made
    from whole-batch arrays instead of one record at a time; segment images are cut when first read."""
    F, cap = segs.shape
    live = segs[np.arange(cap)[None, :] < nseg[:, None]]              # frame order, ascending label
    area = live["area"]
    fa = area.astype(np.float64)
    rows = zip(live["label"].tolist(), live["r0"].tolist(), live["c0"].tolist(), live["r1"].tolist(), live["c1"].tolist(),
               (live["sum_r"].astype(np.float64) / fa).tolist(), (live["sum_c"].astype(np.float64) / fa).tolist(), area.tolist())
    new = Segment.__new__
    k = 0
    crops = None
    if len(slots) and hasattr(slots[0].frame, "roi"):
        buf, offs, hs, ws = _window_crops([s.frame for s in slots], nseg, live, min_seg_size, crop_region)
        tail = tuple(slots[0].frame.roi.shape[2:])
        crops = (_Crop((buf, o, h, w, tail)) for o, h, w in zip(offs, hs, ws))
    for slot, count in zip(slots, nseg.tolist()):
        number, stamp, frame = slot.frame_number, slot.timestamp, slot.frame
        out = []
        for _ in range(count):
            lab, r0, c0, r1, c1, cy, cx, ar = next(rows)
            s = new(Segment)
            bbox = (r0, c0, r1, c1)
            s.__dict__ = {"parent_frame_number": number, "parent_timestamp": stamp,
                          "_image": next(crops) if crops is not None else _Cut((frame, bbox, min_seg_size, crop_region)),
                          "segment_history": [], "status": None, "label": lab, "bbox": bbox, "centroid": (cy, cx), "area": ar,
                          "_batch": batch, "_index": k}
            if batch is None:
                del s.__dict__["_batch"], s.__dict__["_index"]
            out.append(s)
            k += 1
        slot.segments = out


class Segment:
    """data_structures.py:16-30.  The reference copies every public regionprops attribute
    (about 9 ms per segment); only label/bbox/centroid/area are ever read downstream
    (segment_tracking.py:139-222, segment_classification.py:30,42), so only those exist here.
    segment_image may be handed in as a zero-argument callable: it is then cut when first read."""

    def __init__(self, regionprops, frame_number, timestamp, segment_image):
        self.parent_frame_number = frame_number
        self.parent_timestamp = timestamp
        self._image = segment_image
        self.segment_history = []
        self.status = None
        self.label = regionprops.label
        self.bbox = regionprops.bbox
        self.centroid = regionprops.centroid
        self.area = regionprops.area

    @property
    def segment_image(self):
        im = self._image
        if type(im) is _Cut:                 # (frame, bbox, min_seg_size, crop_region): the view is made on first read
            r0, r1, c0, c1 = img.segment_crop_box(im[1], im[0].shape, im[2], im[3])
            im = self._image = im[0][r0:r1, c0:c1]
        elif type(im) is _Crop:              # (buffer, offset, rows, columns, trailing shape): a view of the window's crop buffer
            buf, off, h, w, tail = im
            count = h * w * int(np.prod(tail, dtype=np.int64))
            im = self._image = buf[off:off + count].reshape((h, w) + tail)
        elif callable(im):
            im = self._image = im()
        return im

    @segment_image.setter
    def segment_image(self, value):
        self._image = value

    def __deepcopy__(self, memo):
        """copy.deepcopy(tracker.detected_events) (__main__.py:100): plain data only -- the image is resolved, the link to the
        window's device-side state is dropped."""
        import copy
        new = Segment.__new__(Segment)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("_batch", "_index"):
                continue
            if k == "_image":
                v = self.segment_image
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new


class _LazyStages(OrderedDict):
    """processed_frames: values may be zero-argument callables that are resolved on first read
    (a stage image that still lives on the GPU side of the boundary)."""

    def __getitem__(self, key):
        v = OrderedDict.__getitem__(self, key)
        if callable(v):
            v = v()
            OrderedDict.__setitem__(self, key, v)
        return v

    def get(self, key, default=None):
        return self[key] if key in self else default

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

    def export_segments(self, min_seg_size, crop_region, export_dir):
        """data_structures.py:65-113 (`--export`, debug output; host code, no GPU): per segment a PNG of the crop with the segment's box
        tinted red (cv2.rectangle filled, both corners inclusive, blended 0.6 / 0.4 by cv2.addWeighted) under <export_dir>/overlay,
        and the segment's >= min_seg_size cut from the full frame under <export_dir>.  Files are named like the reference's
        ('"<video stem>"_<frame>_<label>_<segments in the frame>.png').  PNGs are written with Pillow (cv2 is not a dependency): the
        pixels are cv2.imwrite's (BGR arrays stored as RGB images), the compressed bytes are not.  PARITY UNPINNED (cv2 arithmetic:
        float32 blend, round half to even, restated)."""
        import math
        from pathlib import Path
        from PIL import Image
        export_dir = Path(export_dir)
        (export_dir / "overlay").mkdir(parents=True, exist_ok=True)
        color_img = np.asarray(self.processed_frames["crop"])
        oy, ox = crop_region[0][1], crop_region[0][0]
        for segment in self.segments:
            name = '"{}"_{}_{}_{}.png'.format(self.src_video, self.frame_number, segment.label, len(self.segments))
            bbox = list(segment.bbox)
            overlay = color_img.copy()
            r0, c0 = max(bbox[0], 0), max(bbox[1], 0)
            overlay[r0:bbox[2] + 1, c0:bbox[3] + 1] = (0, 0, 255)                      # cv2.rectangle(..., -1): corners inclusive, BGR red
            blend = overlay.astype(np.float32) * np.float32(0.6) + color_img.astype(np.float32) * np.float32(0.4)
            output = np.clip(np.rint(blend), 0, 255).astype(np.uint8)
            Image.fromarray(np.ascontiguousarray(output[..., ::-1])).save(str(export_dir / "overlay" / name))
            h, w = bbox[2] - bbox[0], bbox[3] - bbox[1]
            if h < min_seg_size[0]:
                d = min_seg_size[0] - h
                bbox[0] -= math.floor(d / 2)
                bbox[2] += math.ceil(d / 2)
            if w < min_seg_size[1]:
                d = min_seg_size[1] - w
                bbox[1] -= math.floor(d / 2)
                bbox[3] += math.ceil(d / 2)
            # (a box that leaves the frame's top / left is clamped like extract_segment_images here clamps it: INTEGRATION.md)
            seg = self.frame[max(bbox[0] + oy, 0):max(bbox[2] + oy, 0), max(bbox[1] + ox, 0):max(bbox[3] + ox, 0)]
            seg = np.asarray(seg)
            if seg.size:
                Image.fromarray(np.ascontiguousarray(seg[..., ::-1] if seg.ndim == 3 else seg)).save(str(export_dir / name))


class Presegmented:
    """One window that a reader segmented ahead of the counting loop (io_frames.PresegmentingReader): the frames as get_n_frames handed
    them out (oldest first) and, per frame, the Segment objects segment_queue would have made.  Registered under the oldest frame's
    identity; FrameQueue.segment_queue takes it when its queue holds exactly these frames and was asked for the same regions."""
    __slots__ = ("frames", "segments", "crop_region", "min_seg_size", "params", "iters", "device")

    def __init__(self, frames, segments, crop_region, min_seg_size, params, iters, device):
        self.frames, self.segments, self.crop_region = frames, segments, [tuple(crop_region[0]), tuple(crop_region[1])]
        self.min_seg_size, self.params, self.iters, self.device = tuple(min_seg_size), params, iters, device


PRESEGMENTED = {}          # id(oldest frame of the window) -> Presegmented


class WindowBatch:
    """What one segment_queue call left on the device for the classifier: the window's frames and region records, held by
    the library context until its next batch.  Segment k of the batch (frames in queue order, ascending label) carries
    (_batch, _index = k); SegmentClassifier asks predictions(classifier) for the whole table once per window.  Returns None
    when the context has moved on (the caller then classifies from the segments' images, as before)."""

    def __init__(self, ctx, generation, total, min_seg_size, queue=None):
        self.ctx, self.generation, self.total, self.min_seg_size = ctx, generation, total, tuple(min_seg_size)
        self.queue = weakref.ref(queue) if queue is not None else None
        self._tables = {}          # id(classifier) -> numpy int array (total,) of predicted classes, or a pending device tensor
        self.used = False

    def alive(self):
        return self.generation == self.ctx.generation

    def launch(self, classifier):
        """Start scoring the window's segments on the GPU without waiting for the result."""
        key = id(classifier)
        if key in self._tables or not self.alive() or self.total == 0:
            return
        dev = getattr(classifier, "device", None)
        if dev is None or dev.type != "cuda" or (dev.index or 0) != self.ctx.device:
            return                               # a classifier on another device scores the segments' images
        try:
            self._tables[key] = classifier.predict_last_batch(self.ctx, self.generation, self.total, self.min_seg_size)
        except _lib.StaleBatch:
            pass

    def predictions(self, classifier):
        key = id(classifier)
        if key not in self._tables:
            self.launch(classifier)
        table = self._tables.get(key)
        if table is None:
            return None
        if not isinstance(table, np.ndarray):
            table = self._tables[key] = table.cpu().numpy()          # waits for the forward that launch() started
        self.used = True
        q = self.queue() if self.queue is not None else None
        if q is not None:
            q._classifier_hint = weakref.ref(classifier)             # the next window is scored as soon as it is segmented
        return table


class FrameQueue(deque):
    """data_structures.py:116-217.  Index 0 is the newest frame (appendleft, :134); RPCA column j
    is queue index j (image_filtering.py:234-237).

    keep_stages=True (the default, like the reference) makes the six stage images available under
    Frame.processed_frames; they stay on the GPU until read.  keep_stages=False does not produce them at all."""

    def __init__(self, queue_size=21, device=0, params=None, keep_stages=True):
        deque.__init__(self, maxlen=queue_size)
        self.frames_read = 0
        self.frames_processed = 0
        self.device = device
        self.params = params
        self.keep_stages = keep_stages
        self.last_iters = None
        self._staging = None
        self._classifier_hint = None      # weakref to the classifier that asked for the last window's scores
        self._last_batch = None

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
        """:171-185.  "crop" is the reference's view of the frame, made when it is first read; "grayscale" is produced on the
        GPU by segment_queue's single library call and is resolved lazily if somebody reads it earlier."""
        region = [tuple(crop_region[0]), tuple(crop_region[1])]          # the caller's list may change before a value is read
        for slot in self:
            frame = slot.frame
            slot.processed_frames["crop"] = (lambda f=frame: img.crop_frame(f, region))
            slot.processed_frames["grayscale"] = (lambda f=frame: img.convert_grayscale(img.crop_frame(f, region)))

    def _take_presegmented(self, min_seg_size, crop_region):
        """The window a PresegmentingReader segmented ahead, if the queue holds exactly its frames and the same regions are asked for."""
        n = len(self)
        pre = PRESEGMENTED.get(id(self[-1].frame))
        if pre is None or len(pre.frames) != n or any(pre.frames[k] is not self[n - 1 - k].frame for k in range(n)):
            return False
        if (pre.crop_region != [tuple(crop_region[0]), tuple(crop_region[1])] or pre.min_seg_size != tuple(min_seg_size)
                or pre.params is not self.params or pre.device != self.device):
            return False
        del PRESEGMENTED[id(self[-1].frame)]
        for k in range(n):
            self[n - 1 - k].segments = pre.segments[k]
        self.last_iters = pre.iters
        self._last_batch = None
        if self.keep_stages:
            read = _stages_on_request(self.get_queue(), crop_region, tuple(min_seg_size), self.params, self.device)
            for key, name in STAGE_KEYS.items():
                for i, slot in enumerate(self):
                    slot.processed_frames[name] = (lambda k=key, i=i: read(k, i))
        return True

    def _stage_window(self, min_seg_size, crop_region):
        """The window's ROI crops plus a margin of half the minimum segment size (clipped to the frame), stacked in
        page-locked memory: (staging array (n, Hm, Wm, 3), (x, y) of the ROI inside it, ROI (Hc, Wc))."""
        frames = self.get_queue()

        def buffer(shape):
            if self._staging is None or self._staging.shape != shape:
                self._staging = _lib.pinned_empty(shape, np.uint8, device=self.device)  # page-locked: the upload is one DMA
            return self._staging
        return stack_frames(frames, crop_region, min_seg_size, buffer)

    def segment_queue(self, min_seg_size, crop_region):
        """:187-217: RPCA -> bilateral -> threshold -> opening -> CCL -> region properties ->
        segment crops, one swk_batch_run for the whole window."""
        if "crop" not in self[0].processed_frames:
            raise RuntimeError("preprocess_queue must run before segment_queue")
        if PRESEGMENTED and self._take_presegmented(min_seg_size, crop_region):
            return
        stack, (rx, ry), (Hc, Wc), backwards = self._stage_window(min_seg_size, crop_region)
        n = stack.shape[0]
        ctx = _lib.default_context(self.device)
        stages = tuple(STAGE_KEYS) if self.keep_stages else ()
        res = ctx.batch_run(stack, 1, n, crop=(rx, ry, Wc, Hc), params=self.params, stages=stages, device_stages=True,
                            reverse_frames=backwards)
        generation = res["generation"]
        self.last_iters = int(res["iters"][0])
        nseg = res["nseg"]
        if np.any(nseg > res["segs"].shape[1]):
            raise _lib.SwkError("more regions in a frame than seg_cap")      # cannot happen: labels are u8
        # the classifier that scored the last window gets this one's segments right away: its forward runs while the
        # Python objects below are made.  A window whose scores nobody asked for turns that off again.
        prev, self._last_batch = self._last_batch, None
        if prev is not None and not prev.used:
            self._classifier_hint = None
        batch = WindowBatch(ctx, generation, int(nseg.sum()), min_seg_size, queue=self) if stack.ndim == 4 else None
        self._last_batch = batch
        hint = self._classifier_hint() if self._classifier_hint is not None else None
        if hint is not None and batch is not None:
            batch.launch(hint)
        planes = res.get("planes")
        if planes is not None:
            for key, name in STAGE_KEYS.items():
                for i, slot in enumerate(self):
                    slot.processed_frames[name] = (lambda k=key, i=i, p=planes: p.read(k, i))
        window_segments(res["segs"], nseg, list(self), tuple(min_seg_size), crop_region, batch)


def _stages_on_request(frames, crop_region, min_seg_size, params, device):
    """processed_frames values of a window that was segmented ahead without stage images: the first read runs the window once more
    with the images switched on (a debugging read, not part of the counting loop)."""
    state = {}

    def read(key, pos):
        if "planes" not in state:
            ctx = _lib.default_context(device)
            stack, (rx, ry), (Hc, Wc), backwards = stack_frames(frames, crop_region, min_seg_size, ctx.staging)
            res = ctx.batch_run(stack, 1, len(frames), crop=(rx, ry, Wc, Hc), params=params, stages=tuple(STAGE_KEYS), device_stages=True,
                                reverse_frames=backwards)
            state["planes"] = res["planes"]
        return state["planes"].read(key, pos)
    return read


def stack_frames(frames, crop_region, min_seg_size, buffer):
    """frames (in the batch's frame order) -> (stack (F, Hm, Wm[, C]) uint8 in page-locked memory, (x, y) of the ROI inside a stacked
    frame, ROI (Hc, Wc), reversed: the stack holds the frames in the opposite order).  Full decoded frames are cropped to the ROI plus its margin into buffer(shape) (swk_stage_frames).
    RoiFrames of a ROI-stream reader (io_roi_stream.py) already ARE that rectangle: when they sit, in this order, in one of the
    reader's page-locked blocks, the block itself is the stack -- no copy at all."""
    first = frames[0]
    (ya, yb, xa, xb), (x0, y0, x1, y1) = _margin_rect(first.shape, crop_region, min_seg_size)
    if hasattr(first, "roi"):
        oy, ox = first.origin
        h, w = first.roi.shape[:2]
        if ya < oy or xa < ox or yb > oy + h or xb > ox + w:
            raise ValueError("the ROI stream does not hold the crop region plus its margin")
        block, s0, last = first.block, first.slot, len(frames) - 1
        # (one snapshot per frame: the reader's thread may give a frame private pixels meanwhile -- then it no longer matches and the
        #  copy below is taken)
        where = [(f.block, f.slot) for f in frames]
        if block is not None and s0 - last >= 0 and all(b is block and sl == s0 - i for i, (b, sl) in enumerate(where)):
            return block[s0 - last:s0 + 1], (x0 - ox, y0 - oy), (y1 - y0, x1 - x0), True          # the piece lies in file order: read it backwards
        if block is not None and all(b is block and sl == s0 + i for i, (b, sl) in enumerate(where)):
            return block[s0:s0 + last + 1], (x0 - ox, y0 - oy), (y1 - y0, x1 - x0), False
        stack = buffer((len(frames), yb - ya, xb - xa) + first.shape[2:])
        _lib.stage_frames([f.roi for f in frames], ya - oy, yb - oy, xa - ox, xb - ox, stack)
        return stack, (x0 - xa, y0 - ya), (y1 - y0, x1 - x0), False
    stack = buffer((len(frames), yb - ya, xb - xa) + first.shape[2:])
    _lib.stage_frames(frames, ya, yb, xa, xb, stack)
    return stack, (x0 - xa, y0 - ya), (y1 - y0, x1 - x0), False


def _margin_rect(frame_shape, crop_region, min_seg_size):
    """(ya, yb, xa, xb) of the ROI plus half the minimum segment size, clipped to the frame, and the ROI (x0, y0, x1, y1) clipped."""
    Hf, Wf = frame_shape[:2]
    (x0, y0), (x1, y1) = crop_region
    x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, Wf), min(y1, Hf)
    my, mx = int(min_seg_size[0]) // 2, int(min_seg_size[1]) // 2
    return (max(y0 - my, 0), min(y1 + my, Hf), max(x0 - mx, 0), min(x1 + mx, Wf)), (x0, y0, x1, y1)


def segment_windows(windows, crop_region, min_seg_size=(24, 24), device=0, params=None, classifier=None, owner=None, info=None):
    """Several FrameQueue-fuls in ONE library call.  windows: list of (frames, frame_numbers, timestamps) triples as
    FrameReader.get_n_frames returns them (oldest frame first), all of the same length n.  Returns one list of Frame
    objects per window in POP order (oldest first, the order __main__.py:81-92 consumes them), segments attached
    exactly as preprocess_queue + segment_queue would have (data_structures.py:171-217).  Windows are independent in
    the reference too (the queue is emptied between them), so batching changes nothing but the launch count.
    classifier: its scoring of the batch's segments is started on the GPU before the Python objects are made.  owner: an object whose
    _classifier_hint the batch sets when a classifier asks for its scores (a reader that segments ahead).  info: a dict that receives
    'iters' (IALM iterations per window)."""
    if not windows:
        return []
    n = len(windows[0][0])
    ctx = _lib.default_context(device)
    for frames, _, _ in windows:
        if len(frames) != n:
            raise ValueError("every window needs the same number of frames")
    # windows of a ROI-stream reader that lie side by side in one of its page-locked blocks (it places the windows it reads ahead
    # that way): taken in REVERSE order the batch is that piece of the block read backwards -- every window newest frame first --
    # and goes to the library as it lies; windows are independent, only the bookkeeping below has to know the order
    W = len(windows)
    order = list(range(W))
    if W > 1 and all(getattr(w[0][0], "block", None) is not None for w in windows):
        blk, at = windows[0][0][0].block, windows[0][0][0].slot
        if all(fr.block is blk and fr.slot == at + i for i, fr in enumerate(f for w in windows for f in w[0])):
            order.reverse()
    pos = {w: i for i, w in enumerate(order)}                # window w is the pos[w]-th of the batch
    ordered = []
    for w in order:
        ordered.extend(windows[w][0][::-1])                  # queue index 0 = newest = last frame read (:134)
    stack, (rx, ry), (Hc, Wc), backwards = stack_frames(ordered, crop_region, min_seg_size, ctx.staging)
    res = ctx.batch_run(stack, len(windows), n, crop=(rx, ry, Wc, Hc), params=params, stages=(), reverse_frames=backwards)
    nseg = res["nseg"]
    if np.any(nseg > res["segs"].shape[1]):
        raise _lib.SwkError("more regions in a frame than seg_cap")
    batch = WindowBatch(ctx, res["generation"], int(nseg.sum()), min_seg_size, queue=owner) if stack.ndim == 4 else None
    if batch is not None and classifier is not None:
        batch.launch(classifier)
    if info is not None:
        info["iters"] = [int(res["iters"][pos[w]]) for w in range(W)]
    slots = [None] * (W * n)
    out = []
    for w, (frames, numbers, stamps) in enumerate(windows):
        popped = [Frame(frames[k], numbers[k], stamps[k]) for k in range(n)]          # oldest first
        for k, fr in enumerate(popped):
            slots[pos[w] * n + (n - 1 - k)] = fr
        out.append(popped)
    window_segments(res["segs"], nseg, slots, tuple(min_seg_size), crop_region, batch)
    return out
