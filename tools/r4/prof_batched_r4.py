"""Where a batched counting loop (windows_per_call = 8) spends its time: wall-clock of the producer's stages and the consumer's."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from swiftwatcher_amd import pipeline, synthetic, data_structures as ds, _lib
from swiftwatcher_amd.io_frames import ArrayReader
from swiftwatcher_amd.segment_classification import SegmentClassifier
from oracle import classifier_ref as cref

crop_region = [(748, 452), (1172, 664)]
n, nw = 21, 24
total = n * nw
clip = synthetic.full_frames(5, total, crop_region, **(synthetic.SWIFT_LIKE if "swift" in sys.argv else dict(birds=12)))[::-1]
frames = [clip[i] for i in range(total)]
roi_mask = np.zeros((212, 424), np.uint8); roi_mask[100:, 42:382] = 255
q = ds.FrameQueue(); q.push_list_of_frames(frames[:n], list(range(n)), ["t"] * n); q.preprocess_queue(crop_region, None); q.segment_queue((24, 24), crop_region)
crops = [s.segment_image for f in q for s in f.segments]
sd = cref.calibrate_head(cref.random_state_dict(4), crops[:60])
if "pt" in sys.argv:
    g = np.load(os.path.join(ROOT, "tests", "golden", "classifier_model_pt.npz"))
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:")}
clf = SegmentClassifier.from_state_dict(sd, batch_size=8192)
log = []
def wrap(obj, name, tag):
    fn = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); log.append((tag, t0, time.perf_counter())); return r
    setattr(obj, name, w)
wrap(ds, "stack_frames", "stage")
_ws = ds.window_segments
def _ws_prof(*a, **k):
    import cProfile, pstats, io
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable(); r = _ws(*a, **k); pr.disable()
    t1 = time.perf_counter()
    log.append(("objects", t0, t1))
    if t1 - t0 > 0.02:
        buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(8); print(buf.getvalue())
    return r
ds.window_segments = _ws_prof
wrap(_lib.Context, "batch_run", "batch_run"); wrap(clf, "predict_last_batch", "clf_launch"); wrap(clf, "classify_frames", "classify_frames")
wrap(ArrayReader, "get_n_frames", "read")
from swiftwatcher_amd import segment_tracking as stt
wrap(stt.SegmentTracker, "step", "track")
if "serial-first" in sys.argv:
    t0 = time.perf_counter()
    for rep in range(2):
        pipeline.swift_counting_algorithm(ArrayReader(frames), crop_region, roi_mask, classifier=clf, keep_stages=True)
    print("serial x2: %.1f ms; graphs %d, graph error %r" % ((time.perf_counter() - t0) * 1e3, len(clf._graphs), clf._graph_error))
import gc
if "gc-off" in sys.argv:
    gc.collect(); gc.disable()
print("gc counts", gc.get_count(), "tracked objects", len(gc.get_objects()), "switch interval", sys.getswitchinterval())
for rep in range(2):
    del log[:]
    t0 = time.perf_counter()
    ev = pipeline.swift_counting_algorithm(ArrayReader(frames), crop_region, roi_mask, classifier=clf, windows_per_call=8)
    dt = time.perf_counter() - t0
print("total %.1f ms, %.0f frames/s, events %d" % (dt * 1e3, total / dt, len(ev)))
agg = {}
for tag, a, b in log:
    agg.setdefault(tag, [0, 0.0]); agg[tag][0] += 1; agg[tag][1] += b - a
for tag, (c, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-16s calls %4d  total %7.2f ms" % (tag, c, s * 1e3))
for tag, a, b in log:
    if tag in ("batch_run", "clf_launch", "classify_frames", "objects", "stage"):
        print("%-16s %8.2f -> %8.2f ms" % (tag, (a - t0) * 1e3, (b - t0) * 1e3))
