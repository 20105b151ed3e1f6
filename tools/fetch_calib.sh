#!/bin/bash
# bash tools/fetch_calib.sh <outdir>   (on the GPU box)
set -e
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
./tools/fetch_calib > "$out/plain.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- ./tools/fetch_calib > "$out/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -- ./tools/fetch_calib > "$out/write.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
known = {"h": 1, "f": 4, "d": 8}
acc = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
elems = 128 * 64 * 89888
for (k, c), v in sorted(acc.items()):
    if "k_shape" not in k: continue
    size = 1 if "unsigned char" in k or "<h" in k else (4 if "float" in k else 8)
    last = v[-1]
    print("%-60s %-10s counter_KB %14.1f  known_B/elt %d  counted_B/elt %.3f  factor %.3f" %
          (k[:60], c, last, size, last * 1024 / elems, size / (last * 1024 / elems) if last else float("nan")))
PY
