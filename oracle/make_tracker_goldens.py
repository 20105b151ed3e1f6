"""Golden traces of the REFERENCE tracker / event classifier.  Test infrastructure only; run in the build
container:  /opt/conda/bin/python3.9 oracle/make_tracker_goldens.py

The reference's segment_tracking / event_classification / io_data / data_structures modules are imported
unchanged (empty placeholder for the `import cv2` line of data_structures; nothing of cv2 is called) and driven
with synthetic per-frame segment lists: birds on straight lines, some of which fly into a rectangular chimney ROI
and vanish there.  The fixture stores the inputs (centroids per frame, ROI mask) and what the reference produced
(assignments per frame, events, angles, labels, total)."""
import os
import sys
import types

import numpy as np

sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, "/root/reference")
import swiftwatcher.data_structures as ds           # noqa: E402
import swiftwatcher.segment_tracking as st          # noqa: E402
import swiftwatcher.event_classification as ec      # noqa: E402
import swiftwatcher.io_data as dio                  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def make_stream(seed, n_frames, H=212, W=424, spawn=0.35, grid=0, burst=3):
    """grid > 0: centroids snapped to multiples of `grid` pixels (equal distances: the assignment's tie-breaking decides);
    burst: most birds that appear in one frame."""
    rng = np.random.default_rng(seed)
    roi = np.zeros((H, W), np.uint8)
    roi[int(H * 0.55):int(H * 0.8), int(W * 0.12):int(W * 0.88)] = 255
    birds = []                                        # [r, c, vr, vc, frames_left]
    frames = []
    for t in range(n_frames):
        if rng.random() < spawn or t == 0:
            for _ in range(int(rng.integers(1, burst + 1))):
                r, c = rng.uniform(5, H * 0.5), rng.uniform(5, W - 5)
                if rng.random() < 0.5:                # aims at the chimney mouth
                    tr, tc = rng.uniform(H * 0.6, H * 0.75), rng.uniform(W * 0.2, W * 0.8)
                    steps = int(rng.integers(4, 12))
                    birds.append([r, c, (tr - r) / steps, (tc - c) / steps, steps + 1])
                else:
                    ang = rng.uniform(0, 2 * np.pi)
                    sp = rng.uniform(5, 22)
                    birds.append([r, c, sp * np.sin(ang), sp * np.cos(ang), int(rng.integers(3, 15))])
        cents = []
        for b in birds:
            jr, jc = rng.normal(0, 0.7, 2)
            cr, cc = float(b[0] + jr), float(b[1] + jc)
            if grid:
                cr, cc = float(round(cr / grid) * grid), float(round(cc / grid) * grid)
            cents.append((cr, cc))
            b[0] += b[2]; b[1] += b[3]; b[4] -= 1
        birds = [b for b in birds if b[4] > 0 and 0 <= b[0] < H and 0 <= b[1] < W]
        cents = [c for c in cents if 0 <= c[0] < H and 0 <= c[1] < W]
        order = rng.permutation(len(cents))            # label order is unrelated to identity
        frames.append([cents[i] for i in order])
    return roi, frames


def run_reference(roi, frames, fps=30.0):
    tracker = st.SegmentTracker(roi)
    assigns = []
    for t, cents in enumerate(frames):
        fr = ds.Frame(None, t, "ts%05d" % t)
        segs = []
        for i, c in enumerate(cents):
            s = ds.Segment.__new__(ds.Segment)
            s.parent_frame_number = t; s.parent_timestamp = "ts%05d" % t
            s.segment_image = None; s.segment_history = []; s.status = None
            s.label = i + 1; s.centroid = c
            segs.append(s)
        fr.segments = segs
        tracker.set_current_frame(fr)
        cm = tracker.formulate_cost_matrix()
        a = st.apply_hungarian_algorithm(cm)
        assigns.append(np.asarray(a, np.int64))
        tracker.store_assignments(a)
        tracker.link_matching_segments()
        tracker.check_for_events()
        tracker.cache_current_frame()
    events = tracker.detected_events
    ev_last_frame = [e[-1].parent_frame_number for e in events]
    ev_len = [len(e) for e in events]
    ev_first = [e[0].centroid for e in events]
    ev_last = [e[-1].centroid for e in events]
    out = dict(ev_last_frame=np.array(ev_last_frame, np.int64), ev_len=np.array(ev_len, np.int64),
               ev_first=np.array(ev_first, np.float64).reshape(-1, 2), ev_last=np.array(ev_last, np.float64).reshape(-1, 2))
    if events:
        df_events = ec.convert_events_to_dataframe(events, ["parent_frame_number", "parent_timestamp", "centroid"])
        feats = ec.generate_angle_features(df_events)
        out["angles_all"] = feats["angle"].to_numpy(np.float64)
        df_labels = ec.classify_events(df_events)
        out["angles_kept"] = df_labels["angle"].to_numpy(np.float64)
        out["labels"] = df_labels["label"].to_numpy(np.int64)
        out["mode"] = np.float64(ec.compute_mode(ec.filter_false_angles(feats)))
        out["total"] = np.int64(int((df_labels["label"] > 0).sum()))
    else:
        out.update(angles_all=np.zeros(0), angles_kept=np.zeros(0), labels=np.zeros(0, np.int64), mode=np.float64(-90), total=np.int64(0))
    return assigns, out


if __name__ == "__main__":
    # round 3: a crowd (up to eight new birds per frame), centroids on a 4-pixel grid (ties), a long clip
    cases = [("tracker_a", 1, 260, {}), ("tracker_b", 2, 400, {}), ("tracker_sparse", 3, 120, dict(spawn=0.08)),
             ("tracker_crowd", 11, 200, dict(spawn=0.8, burst=8)), ("tracker_grid", 12, 300, dict(grid=4, spawn=0.5)),
             ("tracker_long", 13, 900, dict(spawn=0.3))]
    only = sys.argv[1:]
    for name, seed, nf, kw in cases:
        if only and name not in only:
            continue
        roi, frames = make_stream(seed, nf, **kw)
        assigns, out = run_reference(roi, frames)
        counts = np.array([len(f) for f in frames], np.int64)
        flat = np.array([c for f in frames for c in f], np.float64).reshape(-1, 2)
        a_flat = np.concatenate(assigns) if assigns else np.zeros(0, np.int64)
        a_len = np.array([len(a) for a in assigns], np.int64)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), roi=roi, counts=counts, centroids=flat,
                            assign_flat=a_flat, assign_len=a_len, **out)
        print(name, "frames", nf, "segments", int(counts.sum()), "events", len(out["ev_len"]), "total", int(out["total"]), "mode", float(out["mode"]))
