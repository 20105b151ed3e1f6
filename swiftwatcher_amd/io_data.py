"""Drop-in for the results-exporting half of the reference's swiftwatcher/io_data.py (export_results :19-30,
create_empty_dataframe :33-62, split_labeled_events :65-85, fill_and_group :88-116, save_to_csv :119-136) -- SURVEY.md
section 8f rank 4 -- without pandas: the six CSV tables (per-microsecond, per-second, per-minute counts, each full and
events-only) are produced byte for byte as the reference writes them through pandas.DataFrame.to_csv, including its
timestamp arithmetic (integer nanoseconds: Timedelta-from-float rounding, date_range's linspace, round-half-even to
microseconds) and to_csv's formatting rules (the resolution a datetime column is printed at depends on ALL its values;
a column that is at midnight throughout prints dates only; counts print as floats).

    labels = event_classification.classify_events(events)
    total  = export_results(directory, labels, reader.fps, reader.start_frame, reader.end_frame)

Pinned by tests/golden/export_tables.json, written by the reference's own export_results (generator script committed next to the other golden generators)
under pandas 2.3.3.  The research helpers of the reference file (:143-213) are not mirrored.
"""
import datetime
import os

import numpy as np

NS = 1_000_000_000


def _timedelta_ns(value, unit_ns):
    """pandas.Timedelta(float value, unit) in nanoseconds (tslibs/conversion cast_from_unit): integer part exact, fraction
    rounded to the unit's nanosecond digits, then truncated.  For unit 'ns' there are no digits to round to: the float is
    cast to int64, i.e. truncated (Timedelta(33366666.6667, 'ns').value == 33366666 under pandas 2.3.3)."""
    if unit_ns == 1:
        return int(value)
    base = int(value)
    frac = round(value - base, 9)
    return base * unit_ns + int(frac * unit_ns)


def _round_us(ns):
    """Timestamp.round('us') / DatetimeIndex.round('us'): nearest multiple of 1000 ns, ties to even."""
    ns = np.asarray(ns, np.int64)
    q, r = np.divmod(ns, 1000)
    up = (r > 500) | ((r == 500) & (q % 2 == 1))
    return (q + up.astype(np.int64)) * 1000


def frame_timestamp_ns(frame_number, fps):
    """io_video.py:74-82: nanoseconds after midnight of the timestamp the reference's reader gives a frame."""
    q, r = divmod(_timedelta_ns(frame_number / fps, NS), 1000)          # _round_us on one value, in plain integers
    return (q + (1 if r > 500 or (r == 500 and q & 1) else 0)) * 1000


def create_empty_table(fps, start, end):
    """:33-62: every frame of the video, (timestamp ns after midnight, frame number) -> [predicted, rejected]."""
    nano = (1 / fps) * 1e9
    count = end - start + 1
    start_ns = _timedelta_ns(start * nano, 1)
    duration = _timedelta_ns((count - 1) * nano, 1)
    stamps = _round_us(np.linspace(0, duration, count, dtype=np.int64) + start_ns)
    return {(int(t), int(f)): [0.0, 0.0] for t, f in zip(stamps, range(start, end + 1))}


def _as_ns(timestamp):
    if isinstance(timestamp, (int, np.integer)):
        return int(timestamp)
    midnight = datetime.datetime.combine(timestamp.date(), datetime.time())
    d = timestamp - midnight
    return (d.days * 86400 + d.seconds) * NS + d.microseconds * 1000


def fill_and_group(table, labels):
    """:65-116: events merged per (timestamp, frame) -- label > 0 counts as predicted, label 0 as rejected; a key the
    empty table does not hold is added (the outer join combine_first makes).  Returns total and the exact / per-second
    / per-minute tables as sorted lists of (key, predicted, rejected)."""
    for t, f, lab in zip(labels["timestamp"], labels["framenumber"], labels["label"]):
        row = table.setdefault((_as_ns(t), int(f)), [0.0, 0.0])
        row[0 if lab > 0 else 1] += 1.0
    exact = [(k, v[0], v[1]) for k, v in sorted(table.items())]

    def grouped(unit):
        out = {}
        for (t, _), p, r in exact:
            g = out.setdefault(t - t % unit, [0.0, 0.0])
            g[0] += p
            g[1] += r
        return [(k, v[0], v[1]) for k, v in sorted(out.items())]
    total = int(sum(p for _, p, _ in exact))
    return total, grouped(60 * NS), grouped(NS), exact


def _formatter(stamps_ns, today):
    """to_csv's rendering of a datetime column: the coarsest of date-only / seconds / milli / micro / nano that
    represents every value of the column."""
    stamps_ns = list(stamps_ns)
    date = today.isoformat()
    if all(t % (86400 * NS) == 0 for t in stamps_ns):
        return lambda t: (today + datetime.timedelta(days=t // (86400 * NS))).isoformat()
    digits = 0
    for d, unit in ((3, 1_000_000), (6, 1000), (9, 1)):
        if any(t % (unit * 1000) for t in stamps_ns):
            digits = d

    def fmt(t):
        day, rem = divmod(t, 86400 * NS)
        s, frac = divmod(rem, NS)
        text = "%s %02d:%02d:%02d" % ((today + datetime.timedelta(days=day)).isoformat() if day else date, s // 3600, s // 60 % 60, s % 60)
        if digits:
            text += "." + ("%09d" % frac)[:digits]
        return text
    return fmt


def _num(v):
    return repr(float(v))


def save_to_csv(save_directory, count, minutes, seconds, exact, today=None):
    """:119-136: six files named "<count>-swifts_<table>.csv"."""
    today = today or datetime.date.today()
    os.makedirs(str(save_directory), exist_ok=True)
    nonzero = lambda rows: [r for r in rows if not (r[1] == 0 and r[2] == 0)]      # noqa: E731
    exact_fmt = _formatter([k[0] for k, _, _ in exact], today)       # a MultiIndex level is rendered from ALL its values

    def write(name, rows, multi):
        fmt = exact_fmt if multi else _formatter([k for k, _, _ in rows], today)
        path = os.path.join(str(save_directory), "%d-swifts_%s.csv" % (count, name))
        with open(path, "w", newline="") as fh:
            fh.write("timestamp,framenumber,predicted,rejected\n" if multi else "timestamp,predicted,rejected\n")
            for k, p, r in rows:
                if multi:
                    fh.write("%s,%d,%s,%s\n" % (fmt(k[0]), k[1], _num(p), _num(r)))
                else:
                    fh.write("%s,%s,%s\n" % (fmt(k), _num(p), _num(r)))
        return path
    return [write("full_usec", exact, True), write("events-only_usec", nonzero(exact), True),
            write("full_sec", seconds, False), write("events-only_sec", nonzero(seconds), False),
            write("full_min", minutes, False), write("events-only_min", nonzero(minutes), False)]


def export_results(save_directory, labels, fps, start, end, today=None):
    """:19-30.  labels: event_classification.classify_events' result (framenumber, timestamp, label per surviving
    event).  Returns the predicted total, the swift count (:113)."""
    print("[-]     Saving results to csv files...")
    total, minutes, seconds, exact = fill_and_group(create_empty_table(fps, start, end), labels)
    save_to_csv(save_directory, total, minutes, seconds, exact, today)
    return total
