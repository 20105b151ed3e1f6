"""The reference's per-video loop (swift_counting_algorithm, __main__.py:56-100) over the MI355X segment path:
read queue_size frames -> preprocess_queue + segment_queue (one GPU call) -> per popped frame: optional
classifier, tracker step -> events -> swift count.  Frame I/O, ROI-mask generation and CSV export stay with the
caller (out of scope): frames come from any iterable, crop_region and roi_mask are arguments."""
import numpy as np

from .data_structures import FrameQueue
from .segment_tracking import SegmentTracker
from . import event_classification as ec


def null_frame_like(frame):
    return np.zeros_like(frame)


def swift_counting_algorithm(frames, crop_region, roi_mask, queue_size=21, classifier=None, min_seg_size=(24, 24),
                             timestamps=None, device=0, keep_stages=False):
    """frames: sequence of full BGR uint8 frames, oldest first.  Returns the tracker's detected events
    (lists of Segment objects, the same structure the reference hands to event classification)."""
    total = len(frames)
    queue = FrameQueue(queue_size, device=device, keep_stages=keep_stages)
    tracker = SegmentTracker(roi_mask)
    read = 0
    while queue.frames_processed < total:
        batch, numbers, stamps = [], [], []
        for _ in range(queue.maxlen):                              # FrameReader.get_n_frames pads with null frames
            if read < total:
                batch.append(frames[read]); numbers.append(read)
                stamps.append(timestamps[read] if timestamps is not None else "%010.3f" % (read / 30.0))
            else:
                batch.append(null_frame_like(frames[0])); numbers.append(-1); stamps.append("00:00:00.000")
            read += 1
        queue.push_list_of_frames(batch, numbers, stamps)
        queue.preprocess_queue(crop_region, None)
        queue.segment_queue(min_seg_size, crop_region)
        while not queue.is_empty():
            frame = queue.pop_frame()
            if classifier is not None:
                frame.segments = classifier(frame.segments)
            tracker.step(frame)
    return tracker.detected_events


def count_swifts(frames, crop_region, roi_mask, **kw):
    events = swift_counting_algorithm(frames, crop_region, roi_mask, **kw)
    return ec.count_swifts(events), events
