#!/bin/bash
# the driver's command several times in a row (GPU box): the threaded side measurements must neither fail nor change their counts
out=${1:-gpurun_out/repeat}; runs=${2:-4}
mkdir -p $out
for i in $(seq 1 $runs); do
  timeout -k 10 240 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $out/b$i.json 2> $out/b$i.err || { echo "run $i failed"; tail -n 5 $out/b$i.err; exit 1; }
done
python3 - "$out" "$runs" <<'PY'
import json, sys
out, runs = sys.argv[1], int(sys.argv[2])
bad = 0
for i in range(1, runs + 1):
    d = json.load(open("%s/b%d.json" % (out, i)))
    cl = d["count_loop"]
    modes = {k: v for k, v in cl.items() if isinstance(v, dict)}
    errs = [k for k, v in modes.items() if "error" in v] + (["count_loop"] if "error" in cl else [])
    counts = {(v.get("count"), v.get("events")) for k, v in modes.items() if k != "no_classify" and "error" not in v} | {(cl.get("count"), cl.get("events"))}
    bad += bool(errs) or len(counts) != 1
    print(i, d["value"], d["drop_in"]["value"], cl.get("value"), sorted(counts), {k: v.get("value") for k, v in modes.items()}, "ERRORS %s" % errs if errs else "")
sys.exit(1 if bad else 0)
PY
