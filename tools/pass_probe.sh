#!/bin/bash
set -e
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
./tools/pass_probe > "$out/plain.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- ./tools/pass_probe > "$out/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -- ./tools/pass_probe > "$out/write.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
elems = 128 * 64 * 89888
for (k, c), v in sorted(acc.items()):
    if "k_probe" in k: print("%-40s %-10s counted B/elt %.3f" % (k[:40], c, v[-1] * 1024 / elems))
PY
