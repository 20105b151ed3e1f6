"""Per-layer times of the cropped classifier's own kernels inside a real forward (batch 4096 by default): every
swk_nhwc_* call of CroppedSqueezeNet10._forward_hip_glue is bracketed by events on PyTorch's stream.  Prints one row per
call with its multiply-accumulate rate and the activation bytes it has to move, and a JSON summary as the last line.

    python3 tools/bench_convs.py [batch] [reps] [1x1 ring knob] [Winograd one-block-waves knob]
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import classifier_ref as ref                                  # noqa: E402  (weights generator only)
from swiftwatcher_amd import _lib                                         # noqa: E402
from swiftwatcher_amd.segment_classification import SegmentClassifier    # noqa: E402


class TimedLib:
    def __init__(self, lib):
        self._lib = lib
        self.records = []
        self.on = False

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("swk_nhwc_"):
            return fn

        def call(*a):
            if not self.on:
                return fn(*a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            self.records.append((name, a, e0, e1))
            return rc
        return call


def describe(name, a):
    """(label, MACs, activation bytes) of one call from its arguments (include/swk.h)."""
    if name == "swk_nhwc_conv1x1_bias_relu_place":
        n, cin, h, w, cout = a[2], a[5], a[8], a[9], a[12]
        return "1x1 %3d->%3d %2dx%2d" % (cin, cout, h, w), n * h * w * cin * cout, 4 * n * h * w * (cin + cout)
    if name in ("swk_nhwc_conv3x3_bias_relu_place", "swk_nhwc_conv3x3_winograd_bias_relu_place"):
        n, t, cin, cout = a[2], a[3], a[4], a[7]
        o = t - 2
        # direct-convolution multiply-accumulates for both (the Winograd kernel executes 16/36 of them, padded to even sizes)
        return ("w3x3" if "winograd" in name else "3x3 ") + "%3d->%3d %2dx%2d" % (cin, cout, o, o), n * o * o * 9 * cin * cout, 4 * n * (t * t * cin + o * o * cout)
    if name == "swk_nhwc_maxpool3s2_conv1x1_bias_relu_place":
        n, t, cin, cout = a[2], a[3], a[4], a[7]
        p = (t - 3) // 2 + 1
        live = a[16] if len(a) > 16 and a[14] else t          # with a shared ring only the live square is fetched per segment
        return "p+1x1 %3d->%3d %2dx%2d" % (cin, cout, p, p), n * p * p * cin * cout, 4 * n * (live * live * cin + p * p * cout)
    if name == "swk_nhwc_maxpool3s2":
        n, h, w, c = a[2], a[3], a[4], a[5]
        oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
        return "pool %3d %2dx%2d" % (c, h, w), 0, 4 * n * c * (h * w + oh * ow)
    if name == "swk_nhwc_conv7x7s2_bias_relu":
        n, side, m = a[2], a[3], a[5]
        return "7x7s2  3-> 96 %2dx%2d" % (m, m), n * m * m * 147 * 96, 4 * n * (side * side * 3 + m * m * 96)
    if name == "swk_nhwc_head2_relu_mean":
        n, px, c = a[2], a[3], a[4]
        return "head %3d->  2 %3d px" % (c, px), n * px * c * 2, 4 * n * px * c
    if name == "swk_nhwc_bias_relu_place":
        n, c, h, w = a[2], a[5], a[8], a[9]
        return "place %3d %2dx%2d" % (c, h, w), 0, 8 * n * c * h * w
    return name, 0, 0


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    ring = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if _lib.load().swk_set_cnn_tuning(0, ring):
        raise SystemExit("swk_set_cnn_tuning refused %d" % ring)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "w.pt")
        torch.save(ref.random_state_dict(0), path)
        clf = SegmentClassifier(path, batch_size=batch)
    timed = TimedLib(_lib.load())
    _lib.load = lambda: timed
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn((batch, 3, 40, 40), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    net = clf.cropped
    with torch.no_grad():
        for _ in range(2):
            net(x)
        torch.cuda.synchronize()
        whole = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            net(x)
            e1.record()
            whole.append((e0, e1))
        torch.cuda.synchronize()
        whole_ms = float(np.mean([a.elapsed_time(b) for a, b in whole]))
        # the classifier's own entry: two chains on two streams from 1,024 rows (SegmentClassifier._forward_two_streams)
        for _ in range(2):
            clf._forward(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            clf._forward(x)
        e1.record()
        torch.cuda.synchronize()
        chains_ms = e0.elapsed_time(e1) / reps
        timed.on = True
        for _ in range(reps):
            net(x)
        torch.cuda.synchronize()
    per = len(timed.records) // reps
    rows = []
    for i in range(per):
        name, a = timed.records[i][0], timed.records[i][1]
        ms = float(np.mean([timed.records[r * per + i][2].elapsed_time(timed.records[r * per + i][3]) for r in range(reps)]))
        label, macs, nbytes = describe(name, a)
        rows.append({"call": label, "us": round(ms * 1e3, 1), "tflops": round(2 * macs / ms / 1e9, 1), "gbs": round(nbytes / ms / 1e6, 0)})
        print("%-24s %9.1f us %7.1f TFLOP/s %7.0f GB/s" % (label, ms * 1e3, 2 * macs / ms / 1e9, nbytes / ms / 1e6))
    own = sum(r["us"] for r in rows) / 1e3
    groups = {}
    for r in rows:
        groups[r["call"][:4].strip()] = round(groups.get(r["call"][:4].strip(), 0.0) + r["us"] / 1e3, 3)
    print(json.dumps({"batch": batch, "forward_ms": round(whole_ms, 3), "forward_two_chains_ms": round(chains_ms, 3), "own_kernels_ms": round(own, 3), "by_kind_ms": groups, "rows": rows}))


if __name__ == "__main__":
    main()
