"""Golden CSV tables of the REFERENCE exporter (io_data.py:19-135) and its event classifier.  Test infrastructure only;
run in the build container:  /opt/conda/bin/python3.9 oracle/make_export_goldens.py

The reference's event_classification and io_data modules are imported unchanged and driven with synthetic events
(lists of objects carrying parent_frame_number / parent_timestamp / centroid, the attributes
convert_events_to_dataframe keeps, __main__.py:41-45); timestamps are made with the reference reader's own formula
(io_video.py:74-82).  The fixture stores the inputs and the six CSV files export_results wrote, with the run date (the
reference's timestamps are "today at midnight + t") replaced by the token <DATE>.  pandas here is 2.3.3, the reference
pins 0.25: the tables' number formatting is pinned against THIS pandas."""
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import pandas as pd

sys.path.insert(0, "/root/reference")
import swiftwatcher.event_classification as ec      # noqa: E402
import swiftwatcher.io_data as dio                  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "export_tables.json")


class Seg:
    def __init__(self, frame_number, timestamp, centroid):
        self.parent_frame_number = frame_number
        self.parent_timestamp = timestamp
        self.centroid = centroid
        self.status = "D"                       # an attribute convert_events_to_dataframe must drop


def timestamp(frame_number, fps):               # io_video.py:74-82
    return (pd.Timestamp("00:00:00.000") + pd.Timedelta(frame_number / fps, 's')).round(freq='us')


def scenario(seed, fps, start, end, n_events, clump):
    rng = np.random.default_rng(seed)
    events = []
    frames = sorted(int(f) for f in rng.integers(start + 3, end, size=n_events))
    for i, f in enumerate(frames):
        if clump and i % 4 == 1:
            f = frames[i - 1]                   # several events on one frame: rows merge, counts add up
        length = int(rng.integers(2, 6))
        if rng.random() < 0.7:                  # heading down into the chimney: angle near -90
            ang = np.radians(rng.normal(-90, 12))
        else:
            ang = rng.uniform(-np.pi, np.pi)
        step = rng.uniform(6, 20)
        r0, c0 = rng.uniform(20, 80), rng.uniform(50, 350)
        if i % 7 == 3:                          # an "unnatural" angle, exact multiple of 15 degrees: filtered out
            cents = [(50.0, 100.0 + 10.0 * k) for k in range(length)]
        else:
            cents = [(float(r0 - k * step * np.sin(ang)), float(c0 + k * step * np.cos(ang))) for k in range(length)]
        events.append([Seg(f - length + 1 + k, timestamp(f - length + 1 + k, fps), cents[k]) for k in range(length)])
    return events


def run(name, seed, fps, start, end, n_events, clump):
    events = scenario(seed, fps, start, end, n_events, clump)
    df_events = ec.convert_events_to_dataframe(events, ["parent_frame_number", "parent_timestamp", "centroid"])
    df_labels = ec.classify_events(df_events)
    with tempfile.TemporaryDirectory() as d:
        total = dio.export_results(Path(d), df_labels, fps, start, end)
        files = {}
        today = str(pd.Timestamp("00:00:00").date())
        for p in sorted(Path(d).iterdir()):
            files[p.name] = p.read_text().replace(today, "<DATE>")
    return dict(name=name, fps=fps, start=start, end=end, total=int(total),
                events=[[dict(frame=s.parent_frame_number, centroid=list(s.centroid)) for s in e] for e in events],
                angles=[float(a) for a in df_labels["angle"]], labels=[int(v) for v in df_labels["label"]],
                label_frames=[int(i[1]) for i in df_labels.index], files=files)


if __name__ == "__main__":
    cases = [run("short_30fps", 1, 30.0, 0, 300, 14, True),          # < 1 minute: the per-minute table prints dates only
             run("long_2997", 2, 29.97, 0, 4000, 40, True),          # > 2 minutes, fractional microseconds
             run("offset_start_60fps", 3, 60.0, 120, 2000, 9, False),
             # start * 1e9 / fps with a fractional nanosecond: pandas casts it to int64 (truncation), which moves one or two
             # rows of the per-microsecond table by 1 us against a rounding restatement
             run("offset_start_2997", 6, 29.97, 17, 1234, 12, True),      # (seed 4 has no rejected event: the reference's combine_first raises under pandas 2.x)
             run("offset_start_23976", 5, 23.976, 17, 900, 8, False)]
    json.dump(dict(pandas=pd.__version__, numpy=np.__version__, cases=cases), open(OUT, "w"), indent=0)
    for c in cases:
        print(c["name"], "total", c["total"], "files", list(c["files"]), "labels", sum(c["labels"]), "/", len(c["labels"]))
    print(cases[0]["files"][[k for k in cases[0]["files"] if "events-only_usec" in k][0]][:400])
    print(cases[0]["files"][[k for k in cases[0]["files"] if "full_min" in k][0]][:200])
    print(cases[1]["files"][[k for k in cases[1]["files"] if "events-only_sec" in k][0]][:300])
