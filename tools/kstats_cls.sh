#!/bin/bash
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 tools/bench_classifier.py "$@" > "$out/bench.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 30:
            print("%-110s calls %5s total_ms %9.3f avg_us %9.1f" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
tail -n 1 "$out/bench.log"
