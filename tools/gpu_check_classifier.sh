#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/check_classifier; mkdir -p $out
timeout -k 10 400 python3 -m pytest tests/test_classifier.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?; tail -n 3 $out/tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/bench_convs.py 4096 5 > $out/convs.log 2>&1 || { tail $out/convs.log; exit 1; }
grep -v "^{" $out/convs.log | cut -c1-70; tail -n 1 $out/convs.log | cut -c1-200
