"""The reference's per-video loop (swift_counting_algorithm, __main__.py:56-100) over the MI355X segment path:
read queue_size frames -> preprocess_queue + segment_queue (one GPU call) -> per popped frame: optional
classifier, tracker step -> events -> swift count.  ROI-mask generation and CSV export stay with the caller
(out of scope): crop_region and roi_mask are arguments; the reader is anything with the
reference FrameReader's get_n_frames / total_frames (swiftwatcher_amd.io_frames.ArrayReader for decoded frames)."""
from .data_structures import FrameQueue, segment_windows
from .io_frames import ArrayReader
from .segment_tracking import SegmentTracker
from . import event_classification as ec


def swift_counting_algorithm(reader, crop_region, roi_mask, queue_size=21, classifier=None, min_seg_size=(24, 24),
                             device=0, keep_stages=False, windows_per_call=1):
    """Same call order as __main__.py:67-100.  Returns the tracker's detected events (lists of Segment objects,
    the structure the reference hands to event classification).  windows_per_call > 1 reads that many queue-fuls
    ahead and segments (and classifies) them in one GPU call each; the tracker still sees the frames one by one in
    the reference's order, so the events are the same."""
    tracker = SegmentTracker(roi_mask)
    if windows_per_call > 1:
        # producer thread: reads ahead and segments (GPU call and array copies release the GIL);
        # this thread: classifier and the strictly sequential tracker
        import queue as _queue
        import threading
        ready = _queue.Queue(maxsize=2)

        def produce():
            try:
                ahead = 0
                while ahead < reader.total_frames:
                    windows = []
                    while len(windows) < windows_per_call and ahead < reader.total_frames:
                        triple = reader.get_n_frames(n=queue_size)                  # :73 (pads with null frames)
                        windows.append(triple)
                        ahead += sum(1 for k in triple[1] if k >= 0)                # null frames are not counted (:146-147)
                    ready.put(segment_windows(windows, crop_region, min_seg_size, device=device))
                ready.put(None)
            except BaseException as exc:                                            # surfaces in the consumer
                ready.put(exc)

        worker = threading.Thread(target=produce, daemon=True)
        worker.start()
        while True:
            batches = ready.get()
            if batches is None:
                break
            if isinstance(batches, BaseException):
                raise batches
            if classifier is not None:
                classifier.classify_frames([fr for popped in batches for fr in popped])
            for popped in batches:
                for frame in popped:
                    tracker.step(frame)
        worker.join()
        return tracker.detected_events
    queue = FrameQueue(queue_size, device=device, keep_stages=keep_stages)
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
