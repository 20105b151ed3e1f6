"""CSV result tables (io_data.py:19-135) against files the reference's own export_results wrote
(tests/golden/export_tables.json, oracle/make_export_goldens.py): angles, labels, the total and all six tables byte for
byte (the run date, which the reference bakes into every timestamp, is a token in the fixture)."""
import datetime
import json
import os

import numpy as np
import pytest


class _Seg:
    def __init__(self, frame, stamp, centroid):
        self.parent_frame_number, self.parent_timestamp, self.centroid = frame, stamp, tuple(centroid)


@pytest.fixture(scope="module")
def cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "export_tables.json")))["cases"]


@pytest.mark.parametrize("idx", [0, 1, 2, 3, 4])
def test_tables_match_the_reference_files(cases, idx, tmp_path):
    from swiftwatcher_amd import event_classification as ec, io_data
    from swiftwatcher_amd.io_frames import ArrayReader
    c = cases[idx]
    reader = ArrayReader([np.zeros((2, 2, 3), np.uint8)] * 2, fps=c["fps"], start=c["start"], end=c["end"])
    events = [[_Seg(s["frame"], reader.frame_number_to_timestamp(s["frame"]), s["centroid"]) for s in e] for e in c["events"]]
    labels = ec.classify_events(events)
    assert labels["framenumber"] == c["label_frames"]
    np.testing.assert_allclose(labels["angle"], c["angles"], rtol=0, atol=0)
    assert labels["label"] == c["labels"]
    today = datetime.date(2031, 5, 17)
    total = io_data.export_results(tmp_path, labels, c["fps"], c["start"], c["end"], today=today)
    assert total == c["total"] == ec.count_swifts(events)
    assert sorted(os.listdir(tmp_path)) == sorted(c["files"])
    for name, text in c["files"].items():
        got = open(os.path.join(tmp_path, name)).read()
        assert got == text.replace("<DATE>", today.isoformat()), name


def test_reader_timestamps_are_the_table_keys():
    """Every frame's reader timestamp (io_video.py:74-82) is a key of the exporter's empty table (io_data.py:33-62) for
    the usual frame rates, so events join existing rows instead of adding new ones."""
    from swiftwatcher_amd import io_data
    for fps in (30.0, 29.97, 25.0, 60.0, 23.976, 18.0):
        table = io_data.create_empty_table(fps, 0, 5000)
        for f in (0, 1, 2, 77, 1234, 4999, 5000):
            assert (io_data.frame_timestamp_ns(f, fps), f) in table, (fps, f)


def test_empty_table_equals_pandas_for_offset_starts():
    """create_empty_table against the reference's statements (io_data.py:33-48) run on the installed pandas, for starts
    whose start * 1e9 / fps has a fractional nanosecond (pandas truncates it; 3 of 63 combinations differed by 1 us when the
    restatement rounded)."""
    pd = pytest.importorskip("pandas")
    from swiftwatcher_amd import io_data
    midnight = pd.Timestamp("00:00:00.000000")
    for fps in (29.97, 59.94, 23.976, 30.0, 29.97002997):
        for start, end in ((17, 1234), (100, 4421), (17, 900), (1234, 1300)):
            nano = (1 / fps) * 1e9
            count = end - start + 1
            first = midnight + pd.Timedelta(start * nano, "ns")
            last = first + pd.Timedelta((count - 1) * nano, "ns")
            stamps = pd.date_range(start=first, end=last, periods=count).round(freq="us")
            expect = [(int(t.value - midnight.value), f) for t, f in zip(stamps, range(start, end + 1))]
            assert list(io_data.create_empty_table(fps, start, end)) == expect, (fps, start, end)


def test_formatting_rules():
    from swiftwatcher_amd import io_data
    today = datetime.date(2030, 1, 2)
    ms = io_data._formatter([0, 20_000_000, 40_000_000], today)          # 50 fps: millisecond resolution -> 3 digits
    assert ms(20_000_000) == "2030-01-02 00:00:00.020"
    assert io_data._formatter([0], today)(0) == "2030-01-02"             # all at midnight: dates only
    assert io_data._formatter([0, 60 * io_data.NS], today)(60 * io_data.NS) == "2030-01-02 00:01:00"
    assert io_data._formatter([0, 33_333_000], today)(0) == "2030-01-02 00:00:00.000000"
