"""The reference's per-video loop (swift_counting_algorithm, __main__.py:56-100) over the MI355X segment path:
read queue_size frames -> preprocess_queue + segment_queue (one GPU call) -> per popped frame: optional
classifier, tracker step -> events -> swift count.  The regions come from the chimney corners like in the reference
(generate_regions on the first frame, __main__.py:62-63) or are handed in; the reader is anything with the
reference FrameReader's read_frame / get_n_frames / total_frames (swiftwatcher_amd.io_frames.ArrayReader for decoded
frames)."""
from .data_structures import FrameQueue, segment_windows
from .io_frames import ArrayReader
from .segment_tracking import SegmentTracker, apply_hungarian_algorithm
from . import event_classification as ec
from . import image_filtering as img


def swift_counting_algorithm(reader, crop_region=None, roi_mask=None, queue_size=21, classifier=None, min_seg_size=(24, 24),
                             device=0, keep_stages=False, windows_per_call=1, corners=None, export_dir=None):
    """Same call order as __main__.py:62-100.  corners = ((x1, y1), (x2, y2)) of the chimney's top edge: crop region
    and ROI mask are then generated from the video's first frame (:62-63) instead of being passed in.  Returns the tracker's detected events (lists of Segment objects,
    the structure the reference hands to event classification).  windows_per_call > 1 reads that many queue-fuls
    ahead and segments (and classifies) them in one GPU call each; the tracker still sees the frames one by one in
    the reference's order, so the events are the same."""
    if corners is not None:
        first_frame = reader.read_frame(0, increment=False)                        # :62
        crop_region, roi_mask, _ = img.generate_regions(first_frame, corners)      # :63
    if crop_region is None or roi_mask is None:
        raise ValueError("either corners or crop_region + roi_mask are needed")
    tracker = SegmentTracker(roi_mask)
    if classifier is not None and getattr(reader, "_classifier_hint", False) is None:
        import weakref
        reader._classifier_hint = weakref.ref(classifier)          # a reader that segments ahead scores every batch for this classifier
    if windows_per_call > 1:
        if hasattr(reader, "ahead"):                 # a reader that reads ahead (io_roi_stream): as far as one GPU call takes
            reader.ahead = max(reader.ahead, windows_per_call)
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
                    ready.put(segment_windows(windows, crop_region, min_seg_size, device=device, classifier=classifier))
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
            tracker.set_current_frame(frame)                                   # :87-92, call by call
            cost_matrix = tracker.formulate_cost_matrix()
            tracker.store_assignments(apply_hungarian_algorithm(cost_matrix))
            tracker.link_matching_segments()
            tracker.check_for_events()
            tracker.cache_current_frame()
            if export_dir is not None:                                         # :94-96 (--export): needs keep_stages=True (the "crop" image)
                frame.export_segments(min_seg_size, crop_region, export_dir)
    return tracker.detected_events


def count_swifts(frames, crop_region=None, roi_mask=None, fps=30.0, **kw):
    """Decoded frames (oldest first) -> (swift count, events).  Regions either explicit or from corners=..."""
    events = swift_counting_algorithm(ArrayReader(frames, fps=fps), crop_region, roi_mask, **kw)
    return ec.count_swifts(events), events
