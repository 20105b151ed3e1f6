"""CPU side of the end-to-end parity checks -- TEST INFRASTRUCTURE ONLY (tests/, and bench.py's cpu_baseline / parity leg, where it is
the checker and never the thing measured): the reference's per-video loop (__main__.py:56-100) driven by the CPU oracle's segments
instead of the HIP path's.  Reader bookkeeping, tracker and event classification are the product's host code (pinned to the reference's
own traces in tests/test_tracking_counts.py); everything per-pixel and the classifier come from oracle/."""
import numpy as np


def oracle_frames(clip, crop_region, queue_size=21, min_seg_size=(24, 24)):
    """clip: decoded BGR frames, oldest first.  Runs the oracle window by window the way the counting loop reads them
    (__main__.py:71-82) and returns, in pop order (oldest first), one dict per popped frame:
    number, timestamp, segments (oracle region dicts), crops (extract_segment_images' views of the full frame)."""
    from oracle import reference_path as orc
    from swiftwatcher_amd.io_frames import ArrayReader
    (x0, y0), (x1, y1) = crop_region
    reader = ArrayReader(list(clip))
    processed, n, out = 0, queue_size, []
    while processed < reader.total_frames:
        frames, numbers, stamps = reader.get_n_frames(n)                    # :73 (null frames past the end)
        roi = np.stack([f[y0:y1, x0:x1] for f in frames][::-1])             # queue order: newest first (:134)
        ref = orc.window(np.ascontiguousarray(roi))
        for pos in range(n - 1, -1, -1):                                    # pop order: oldest first (:81-82)
            k = n - 1 - pos
            crops = []
            for s in ref["segments"][pos]:
                r0, c0, r1, c1 = orc.segment_crop_box(s["bbox"], min_seg_size, crop_region)
                crops.append(frames[k][max(r0, 0):max(r1, 0), max(c0, 0):max(c1, 0)])
            out.append(dict(number=numbers[k], timestamp=stamps[k], segments=ref["segments"][pos], crops=crops))
            processed += 0 if numbers[k] < 0 else 1
    return out


def track(frames_info, roi_mask, keep=None):
    """The tracker over oracle_frames() output.  keep: per-frame boolean lists from the classifier (None = --classify
    off, __main__.py:84); kept segments are relabelled 1..k like segment_classification.py:41-42."""
    from swiftwatcher_amd.segment_tracking import SegmentTracker
    from swiftwatcher_amd.data_structures import Frame, Segment
    from swiftwatcher_amd.image_filtering import RegionProps
    tracker = SegmentTracker(roi_mask)
    for i, info in enumerate(frames_info):
        fr = Frame(None, info["number"], info["timestamp"])
        segs = info["segments"]
        if keep is not None:
            segs = [dict(s, label=j + 1) for j, s in enumerate(s for s, kp in zip(segs, keep[i]) if kp)]
        fr.segments = [Segment(RegionProps(s["label"], s["bbox"], s["centroid"], s["area"]), fr.frame_number, fr.timestamp, None)
                       for s in segs]
        tracker.step(fr)
    return tracker.detected_events


def oracle_events(clip, crop_region, roi_mask, queue_size=21):
    return track(oracle_frames(clip, crop_region, queue_size), roi_mask)


def event_signature(events):
    return [[(s.parent_frame_number, s.label, s.bbox, s.centroid) for s in e] for e in events]


def classify_keep(sd, frames_info, margin=1e-3):
    """The batch-1 CPU classifier (oracle/classifier_ref.py, segment_classification.py:26-44 in eval mode) on every crop of
    oracle_frames() output.  Returns (per-frame keep lists, per-frame lists of |score_1 - score_0| -- a decision whose margin is below
    `margin` can legitimately differ between float32 summation orders -- and the number of such unsure decisions)."""
    from oracle import classifier_ref
    keep, margins, unsure = [], [], 0
    for info in frames_info:
        scores, kp = classifier_ref.classify(sd, info["crops"])
        m = [float(abs(s[1] - s[0])) for s in scores]
        unsure += sum(1 for v in m if v < margin)
        keep.append([bool(k) for k in kp])
        margins.append(m)
    return keep, margins, unsure
