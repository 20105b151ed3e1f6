"""Drop-in for the counting half of the reference's swiftwatcher/event_classification.py (:47-141) and of
io_data.export_results' total (io_data.py:19-30, :113) -- SURVEY.md section 8f rank 4 -- on plain numpy (the
reference goes through pandas DataFrames; the CSV export is not mirrored).

    events  = tracker.detected_events            (lists of Segment objects, oldest first)
    result  = classify_events(events)            angle / label per event, the estimated mode
    total   = count_swifts(events)               == export_results(...)'s return value
"""
import math
import sys

import numpy as np

EPSILON = sys.float_info.epsilon


def compute_angle(centroid_list):
    """:75-83: direction from the first to the last centroid of the motion path, image coordinates."""
    del_y = centroid_list[0][0] - centroid_list[-1][0]
    del_x = -1 * (centroid_list[0][1] - centroid_list[-1][1])
    return math.degrees(math.atan2(del_y, del_x))


def compute_mode(angles):
    """:120-141: mode of the angle histogram (36 bins over [-180, 180]), interpolated; -90 unless the fullest
    bin starts inside (-135, -45)."""
    hist, edges = np.histogram(np.asarray(angles, np.float64), bins=36, range=[-180 - EPSILON, 180 + EPSILON])
    i_max = int(np.argmax(hist))
    xl = edges[i_max]
    if -135 < xl < -45:
        f0, f_1, f1 = hist[i_max], hist[i_max - 1], hist[i_max + 1]
        w = abs(edges[1] - edges[0])
        return xl + ((f0 - f_1) / (2 * f0 - f1 - f_1)) * w
    return -90


def classify_events(events):
    """:47-117.  Returns a dict of parallel lists over the events that survive filter_false_angles:
    framenumber, timestamp, angle, label (1 = swift entered: mode-30 < angle <= mode+30, the (a, b] bins of
    pandas.cut), plus 'mode'.

    filter_false_angles (:86-100) drops by INDEX LABEL: `df.drop(df[angle % 15 == 0].index)` removes every row whose
    (timestamp, framenumber) equals that of an event with an exact multiple of 15 degrees -- so an event that ends on the
    same frame as such an event goes with it.  Reproduced (the reference's own CSV fixtures need it)."""
    rows = []
    for event in events:
        centroids = [s.centroid for s in event]
        rows.append((event[-1].parent_frame_number, event[-1].parent_timestamp, compute_angle(centroids)))
    dropped = {(r[1], r[0]) for r in rows if r[2] % 15 == 0}
    rows = [r for r in rows if (r[1], r[0]) not in dropped]
    angles = [r[2] for r in rows]
    mode = compute_mode(angles)
    lo, hi = mode - 30, mode + 30
    labels = [1 if (lo < a <= hi) else 0 for a in angles]
    return dict(framenumber=[r[0] for r in rows], timestamp=[r[1] for r in rows], angle=angles, label=labels, mode=mode)


def count_swifts(events):
    """The number export_results returns (io_data.py:113): events labelled 1."""
    if not events:
        return 0
    return int(sum(classify_events(events)["label"]))
