"""The reference's per-video loop (swift_counting_algorithm, __main__.py:56-100) over the MI355X segment path:
read queue_size frames -> preprocess_queue + segment_queue (one GPU call) -> per popped frame: optional
classifier, tracker step -> events -> swift count.  ROI-mask generation and CSV export stay with the caller
(out of scope): crop_region and roi_mask are arguments; the reader is anything with the
reference FrameReader's get_n_frames / total_frames (swiftwatcher_amd.io_frames.ArrayReader for decoded frames)."""
from .data_structures import FrameQueue
from .io_frames import ArrayReader
from .segment_tracking import SegmentTracker
from . import event_classification as ec


def swift_counting_algorithm(reader, crop_region, roi_mask, queue_size=21, classifier=None, min_seg_size=(24, 24),
                             device=0, keep_stages=False):
    """Same call order as __main__.py:67-100.  Returns the tracker's detected events (lists of Segment objects,
    the structure the reference hands to event classification)."""
    queue = FrameQueue(queue_size, device=device, keep_stages=keep_stages)
    tracker = SegmentTracker(roi_mask)
    while queue.frames_processed < reader.total_frames:
        frames, numbers, stamps = reader.get_n_frames(n=queue.maxlen)          # :73 (pads with null frames)
        queue.push_list_of_frames(frames, numbers, stamps)                     # :74
        queue.preprocess_queue(crop_region, None)                              # :77
        queue.segment_queue(min_seg_size, crop_region)                         # :78
        while not queue.is_empty():                                            # :81
            frame = queue.pop_frame()
            if classifier is not None:                                         # :84-85 (--classify)
                frame.segments = classifier(frame.segments)
            tracker.step(frame)                                                # :87-92
    return tracker.detected_events


def count_swifts(frames, crop_region, roi_mask, fps=30.0, **kw):
    """Decoded frames (oldest first) -> (swift count, events)."""
    events = swift_counting_algorithm(ArrayReader(frames, fps=fps), crop_region, roi_mask, **kw)
    return ec.count_swifts(events), events
