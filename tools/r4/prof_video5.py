"""Where a config-5 video (4K, 850 x 425 ROI, ROI stream file) spends its time in the video_sharded leg's loop: wall-clock per stage."""
import os, sys, time, tempfile
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from swiftwatcher_amd import pipeline, synthetic, data_structures as ds, _lib
from swiftwatcher_amd import image_filtering as img
from swiftwatcher_amd.io_frames import PresegmentingReader
from swiftwatcher_amd.io_roi_stream import RoiFrame, RoiStreamReader, RoiStreamWriter, margin_rect
from swiftwatcher_amd.segment_classification import SegmentClassifier
from swiftwatcher_amd import segment_tracking as stt

frame_hw, corners = (2160, 3840), [(1580, 1240), (2260, 1244)]
crop_region = img.generate_crop_region(corners)
(x0, y0), (x1, y1) = crop_region
Hc, Wc = y1 - y0, x1 - x0
nf = 21 * 24
dev = torch.device("cuda", 0)
roi = synthetic.roi_stream_torch(dev, nf, Hc, Wc, seed=77000, **synthetic.SWIFT_LIKE).cpu().numpy()
ya, yb, xa, xb = margin_rect(frame_hw, crop_region)
tmp = tempfile.TemporaryDirectory()
path = os.path.join(tmp.name, "v.swkroi")
with RoiStreamWriter(path, frame_hw, crop_region) as w:
    rect = np.full((yb - ya, xb - xa, 3), 128, np.uint8)
    for k in range(nf):
        rect[y0 - ya:y1 - ya, x0 - xa:x1 - xa] = roi[k]
        w.append(RoiFrame(rect, (ya, xa), frame_hw + (3,)))
g = np.load(os.path.join(ROOT, "tests", "golden", "classifier_model_pt.npz"))
sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:")}
clf = SegmentClassifier.from_state_dict(sd, batch_size=8192)
roi_mask = np.zeros((Hc, Wc), np.uint8); roi_mask[int(0.47 * Hc):, int(0.1 * Wc):int(0.9 * Wc)] = 255
log = []
def wrap(obj, name, tag):
    fn = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); log.append((tag, t0, time.perf_counter())); return r
    setattr(obj, name, w)
wrap(RoiStreamReader, "_fill", "fill")
wrap(RoiStreamReader, "_block", "block")
wrap(RoiStreamReader, "get_n_frames", "get_n")
wrap(PresegmentingReader, "get_n_frames", "pre_get_n")
wrap(ds, "stack_frames", "stack")
wrap(ds, "window_segments", "objects")
wrap(_lib.Context, "batch_run", "batch_run")
wrap(clf, "predict_last_batch", "clf_launch")
wrap(stt.SegmentTracker, "step", "track") if hasattr(stt.SegmentTracker, "step") else None
wrap(_lib, "cut_boxes", "cut_boxes")
for rep in range(3):
    del log[:]
    pre = PresegmentingReader(RoiStreamReader(path), crop_region, queue_size=21, windows=8)
    t0 = time.perf_counter()
    ev = pipeline.swift_counting_algorithm(pre, crop_region, roi_mask, queue_size=21, classifier=clf)
    dt = time.perf_counter() - t0
    pre.close()
print("total %.1f ms, %.0f frames/s, events %d" % (dt * 1e3, nf / dt, len(ev)))
agg = {}
for tag, a, b in log:
    agg.setdefault(tag, [0, 0.0]); agg[tag][0] += 1; agg[tag][1] += b - a
for tag, (c, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-12s calls %4d  total %7.2f ms" % (tag, c, s * 1e3))
for tag, a, b in sorted(log, key=lambda r: r[1])[:40]:
    if True:
        print("%-12s %8.2f -> %8.2f ms" % (tag, (a - t0) * 1e3, (b - t0) * 1e3))
