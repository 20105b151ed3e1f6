"""Latency of the drop-in path: FrameQueue.preprocess_queue + segment_queue per window, from host frames,
including staging, H2D/D2H and Python object creation (what the reference's loop would see)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swiftwatcher_amd import synthetic                      # noqa: E402
from swiftwatcher_amd.data_structures import FrameQueue     # noqa: E402
from swiftwatcher_amd import _lib                            # noqa: E402

crop_region = [(748, 452), (1172, 664)]                     # the 424x212 ROI inside 1080p frames
out = {}
for n, keep in [(21, True), (21, False), (64, True), (64, False)]:
    frames = synthetic.full_frames(3, n, crop_region)        # (n, 1080, 1920, 3)
    q = FrameQueue(queue_size=n, keep_stages=keep)
    times = []
    for rep in range(6):
        q.push_list_of_frames([frames[i] for i in range(n - 1, -1, -1)], list(range(n)), ["t"] * n)
        t0 = time.perf_counter()
        q.preprocess_queue(crop_region, None)
        q.segment_queue((24, 24), crop_region)
        times.append(time.perf_counter() - t0)
        nseg = sum(len(f.segments) for f in q)
        while not q.is_empty():
            q.pop_frame()
    t = float(np.median(times[1:]))
    # where the time goes on the device side: one more window with the library's per-family HIP events on
    ctx = _lib.default_context(0)
    ctx.prof_enable(True)
    ctx.prof_reset()
    q.push_list_of_frames([frames[i] for i in range(n - 1, -1, -1)], list(range(n)), ["t"] * n)
    q.preprocess_queue(crop_region, None)
    q.segment_queue((24, 24), crop_region)
    fam = {k: round(v[0], 3) for k, v in ctx.prof().items() if isinstance(v, tuple) and v[0] > 0}
    ctx.prof_enable(False)
    while not q.is_empty():
        q.pop_frame()
    out["n%d_keep%d" % (n, keep)] = {"ms_per_window": round(t * 1e3, 2), "frames_per_s": round(n / t, 1), "segments": nseg,
                                    "iters": q.last_iters, "device_ms_by_family": fam, "eig_sweeps_last": ctx.last_eig_sweeps}
print(json.dumps(out))
