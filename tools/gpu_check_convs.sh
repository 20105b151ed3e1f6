#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/check_convs; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_classifier.py -m gpu -x -q -k "winograd or conv1x1 or conv3x3" > $out/tests_k.log 2>&1; rc=$?; tail -n 15 $out/tests_k.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/bench_convs.py 4096 5 0 > $out/convs.log 2>&1 || { tail $out/convs.log; exit 1; }
grep -v "^{" $out/convs.log | grep -E "w3x3|3x3 " | cut -c1-70; tail -n 1 $out/convs.log | cut -c1-200
if [ "$1" = "pmc" ]; then timeout -k 10 900 bash tools/pmc_convs.sh gpurun_out/r2o || exit 1; fi
if [ "$1" = "full" ]; then
timeout -k 10 400 python3 -m pytest tests/test_classifier.py tests/test_baseline_configs.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?; tail -n 5 $out/tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $out/bench.log 2>&1; grep "^{" $out/bench.log | cut -c1-200
fi
