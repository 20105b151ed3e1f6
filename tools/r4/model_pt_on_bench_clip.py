"""Round 4: what model.pt's weights (tests/golden/classifier_model_pt.npz) do on bench.py's count_loop clip: kept share, decision margins,
count -- to decide whether the bench's classifier can BE model.pt instead of random weights with a calibrated head.
    python tools/r4/model_pt_on_bench_clip.py [windows]          (GPU box)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from swiftwatcher_amd import pipeline, synthetic                       # noqa: E402
from swiftwatcher_amd import event_classification as ec               # noqa: E402
from swiftwatcher_amd import image_filtering as img                   # noqa: E402
from swiftwatcher_amd.io_frames import ArrayReader                    # noqa: E402
from swiftwatcher_amd.segment_classification import SegmentClassifier  # noqa: E402

windows = int(sys.argv[1]) if len(sys.argv) > 1 else 24
g = np.load(os.path.join(ROOT, "tests", "golden", "classifier_model_pt.npz"))
sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:")}
clf = SegmentClassifier.from_state_dict(sd, device=torch.device("cuda", 0))
corners = [(790, 620), (1130, 622)]
crop_region = img.generate_crop_region(corners)
roi_mask = np.zeros((212, 424), np.uint8)
roi_mask[100:, 42:382] = 255
for label, kw in (("bench clip (12 birds 30-50 x 12-20 px, contrast 40-90)", dict(birds=12)),
                  ("config-3 test birds (14 of 5-8 x 4-6 px, contrast 25-40)", dict(birds=14, bird_len=(5, 8), bird_wid=(4, 6), contrast=(25, 40))),
                  ("mid birds (12 of 12-20 x 6-10 px, contrast 30-60)", dict(birds=12, bird_len=(12, 20), bird_wid=(6, 10), contrast=(30, 60)))):
    clip = synthetic.full_frames(5, 21 * windows, crop_region, **kw)[::-1]
    flist = [clip[i] for i in range(len(clip))]
    # all segments of the clip and their scores
    from swiftwatcher_amd.data_structures import FrameQueue
    q = FrameQueue()
    reader = ArrayReader(flist)
    crops = []
    while q.frames_processed < reader.total_frames:
        fr, nu, st = reader.get_n_frames(n=21)
        q.push_list_of_frames(fr, nu, st)
        q.preprocess_queue(crop_region, None)
        q.segment_queue((24, 24), crop_region)
        while not q.is_empty():
            f = q.pop_frame()
            crops += [s.segment_image for s in f.segments]
    s = clf.scores(crops).cpu().numpy()
    m = np.abs(s[:, 1] - s[:, 0])
    keep = np.argmax(s, 1) == 1
    t0 = time.perf_counter()
    events = pipeline.swift_counting_algorithm(ArrayReader(flist), crop_region, roi_mask, queue_size=21, classifier=clf)
    dt = time.perf_counter() - t0
    print("%s: %d segments, kept %d (%.1f %%), margins min %.2e, below 1e-3: %d, below 1e-2: %d; events %d count %d; %.0f frames/s" % (
        label, len(crops), int(keep.sum()), 100.0 * keep.mean(), m.min(), int((m < 1e-3).sum()), int((m < 1e-2).sum()), len(events),
        int(ec.count_swifts(events)), len(flist) / dt), flush=True)
