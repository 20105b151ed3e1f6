#!/bin/bash
# round 2, first GPU call: new parity tests, the default bench line, kernel stats of the same command
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r2a; mkdir -p $out
step() { name=$1; shift; echo "== $name"; date; timeout -k 10 "$TMO" "$@" > $out/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -n 3 $out/$name.log | cut -c1-600; if [ $rc -ge 124 ]; then exit $rc; fi; }
TMO=600 step tests python3 -m pytest tests/test_baseline_configs.py tests/test_classifier.py tests/test_distributed.py -m gpu -q -x --durations=15
TMO=400 step bench python3 bench.py
TMO=500 step stats rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv 2>/dev/null; rm -rf $out/stats
head -n 40 $out/kernel_stats.csv | cut -c1-200
