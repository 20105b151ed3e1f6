"""End-to-end counting loop (reader -> segment -> [classify] -> track -> events -> count) on a synthetic 1080p clip
held in host memory: frames/s for one queue-ful per GPU call (the reference's loop shape) and for batched calls."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swiftwatcher_amd import pipeline, synthetic
from swiftwatcher_amd.segment_classification import SegmentClassifier, SqueezeNet10

crop_region = [(748, 452), (1172, 664)]                       # 424 x 212
n_windows = int(sys.argv[1]) if len(sys.argv) > 1 else 24
queue = 21
total = n_windows * queue
clip = synthetic.full_frames(5, total, crop_region, birds=12)[::-1]     # oldest first
frames = [clip[i] for i in range(total)]
roi_mask = np.zeros((212, 424), np.uint8); roi_mask[100:, :] = 255
torch.manual_seed(0)
clf = SegmentClassifier.from_state_dict(SqueezeNet10(2).state_dict(), batch_size=2048)
out = {}
for name, kw in (("per_window", dict()), ("batch_8", dict(windows_per_call=8)), ("batch_24", dict(windows_per_call=24)),
                 ("batch_8_classify", dict(windows_per_call=8, classifier=clf))):
    pipeline.count_swifts(frames if "classifier" in kw else frames[:queue * 2], crop_region, roi_mask, **kw)   # warm-up (MIOpen searches per shape)
    t0 = time.perf_counter()
    count, events = pipeline.count_swifts(frames, crop_region, roi_mask, **kw)
    dt = time.perf_counter() - t0
    out[name] = {"frames_per_s": round(total / dt, 1), "count": int(count), "events": len(events)}
print(json.dumps(out))
