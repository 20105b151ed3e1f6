"""Throughput of the segment classifier on the MI355X: batched SqueezeNet-1.0 forward
through PyTorch-ROCm (MIOpen), fp32, eval mode.  1.465 GFLOP per segment (0.7326 GMAC).  Random weights of the
right shapes (tests use the same generator); preprocessing (the HIP resize/pad/normalise kernel, fed from host
crops) is timed separately."""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import classifier_ref as ref                                  # noqa: E402  (weights generator only)
from swiftwatcher_amd.segment_classification import SegmentClassifier    # noqa: E402

GFLOP_PER_SEGMENT = 1.465


def main():
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "w.pt")
        torch.save(ref.random_state_dict(0), path)
        clf = SegmentClassifier(path, batch_size=2048)
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, size=(24, 24, 3), dtype=np.uint8) if i % 2 else
            rng.integers(0, 256, size=(int(rng.integers(24, 60)), int(rng.integers(24, 80)), 3), dtype=np.uint8)
            for i in range(4096)]
    t0 = time.perf_counter()
    x = clf.preprocess(imgs[:2048])
    torch.cuda.synchronize()
    t_pre = time.perf_counter() - t0
    t0 = time.perf_counter()
    xw = clf.preprocess(imgs[:2048], window=True)
    torch.cuda.synchronize()
    t_pre_w = time.perf_counter() - t0
    out = {}
    which = (("full", clf.model, x), ("cropped", clf.cropped, xw))
    if "--cropped-only" in sys.argv:
        which = which[1:]
    for name, fn, xin in which:
        for bs in (256, 1024, 2048):
            xb = xin[:bs]
            with torch.no_grad():
                for _ in range(3):
                    fn(xb)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reps = 10
                for _ in range(reps):
                    fn(xb)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            # tflops: the full network's 1.465 GFLOP per segment for both (the cropped path does 0.31 of them)
            out["%s_batch_%d" % (name, bs)] = {"segments_per_s": round(bs / dt, 1),
                                               "equivalent_tflops": round(bs * GFLOP_PER_SEGMENT / dt / 1e3, 2)}
    t0 = time.perf_counter()
    clf.scores(imgs)
    torch.cuda.synchronize()
    out["end_to_end_scores_segments_per_s"] = round(len(imgs) / (time.perf_counter() - t0), 1)   # packing + H2D + kernel + net
    out["preprocess_segments_per_s"] = round(2048 / t_pre, 1)      # swk_classifier_input incl. host packing + H2D
    out["preprocess_window_segments_per_s"] = round(2048 / t_pre_w, 1)
    out["dtype"] = "f32"
    out["peak_f32_matrix_tflops"] = 157.3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
