#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r2b; mkdir -p $out
step() { name=$1; shift; echo "== $name"; date; timeout -k 10 "$TMO" "$@" > $out/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -n 3 $out/$name.log | cut -c1-1500; if [ $rc -ge 124 ]; then exit $rc; fi; }
TMO=900 step tests python3 -m pytest tests/test_baseline_configs.py tests/test_classifier.py tests/test_distributed.py -m gpu -q --durations=15
TMO=400 step bench python3 bench.py
TMO=300 step bench_b512 python3 bench.py --no-cpu-baseline --cls-batch 512
TMO=300 step bench_b1024 python3 bench.py --no-cpu-baseline --cls-batch 1024
TMO=300 step framequeue python3 tools/bench_framequeue.py
