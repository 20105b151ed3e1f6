"""cProfile of the reference-pattern counting loop (pipeline.swift_counting_algorithm, one FrameQueue window per GPU call, classifier
per popped frame, six tracker calls) on a synthetic 1080p clip in host memory: where the host time of a window goes.
    python tools/prof_pipeline.py [windows] [classify 0/1]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swiftwatcher_amd import pipeline, synthetic                      # noqa: E402
from swiftwatcher_amd.io_frames import ArrayReader                    # noqa: E402

crop_region = [(748, 452), (1172, 664)]
queue = 21
n_windows = int(sys.argv[1]) if len(sys.argv) > 1 else 16
classify = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
total = n_windows * queue
clip = synthetic.full_frames(5, total, crop_region, birds=12)[::-1]
frames = [clip[i] for i in range(total)]
roi_mask = np.zeros((212, 424), np.uint8)
roi_mask[100:, 42:382] = 255
clf = None
if classify:
    import torch
    from swiftwatcher_amd.segment_classification import SegmentClassifier
    from oracle import classifier_ref as cref                         # test infrastructure: only to calibrate a head that keeps about half
    from swiftwatcher_amd.data_structures import FrameQueue
    q = FrameQueue()
    q.push_list_of_frames(frames[:queue], list(range(queue)), ["t"] * queue)
    q.preprocess_queue(crop_region, None)
    q.segment_queue((24, 24), crop_region)
    crops = [s.segment_image for f in q for s in f.segments]
    sd = cref.calibrate_head(cref.random_state_dict(4), crops[:60])
    clf = SegmentClassifier.from_state_dict(sd, batch_size=2048)
_AR = ArrayReader
make_reader = lambda fr: _AR(fr)                                     # noqa: E731
if len(sys.argv) > 3 and sys.argv[3] == "roi":
    import tempfile
    from swiftwatcher_amd.io_roi_stream import RoiStreamReader, write_roi_stream
    tmpdir = tempfile.mkdtemp()
    paths = {}

    def make_reader(fr):                                              # noqa: F811
        if len(fr) not in paths:
            paths[len(fr)] = write_roi_stream(os.path.join(tmpdir, "clip%d.swkroi" % len(fr)), fr, crop_region)
        return RoiStreamReader(paths[len(fr)])
ArrayReader = make_reader
for _ in range(2):
    pipeline.swift_counting_algorithm(ArrayReader(frames[:4 * queue]), crop_region, roi_mask, classifier=clf, keep_stages=True)
t0 = time.perf_counter()
ev = pipeline.swift_counting_algorithm(ArrayReader(frames), crop_region, roi_mask, classifier=clf, keep_stages=True)
dt = time.perf_counter() - t0
print("unprofiled: %.3f ms per window, %.0f frames/s, %d events" % (dt / n_windows * 1e3, total / dt, len(ev)))
pr = cProfile.Profile()
pr.enable()
pipeline.swift_counting_algorithm(ArrayReader(frames), crop_region, roi_mask, classifier=clf, keep_stages=True)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(24)
