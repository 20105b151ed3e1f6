#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r2m; mkdir -p $out
timeout -k 10 400 python3 -m pytest tests/test_classifier.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?; tail -n 3 $out/tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/bench_convs.py 4096 5 1 > $out/convs_ring1.log 2>&1 || { tail $out/convs_ring1.log; exit 1; }
timeout -k 10 300 python3 tools/bench_convs.py 4096 5 0 > $out/convs_ring0.log 2>&1 || { tail $out/convs_ring0.log; exit 1; }
paste <(grep -v "^{" $out/convs_ring1.log | cut -c1-70) <(grep -v "^{" $out/convs_ring0.log | cut -c25-70)
tail -n 1 $out/convs_ring1.log | cut -c1-200; tail -n 1 $out/convs_ring0.log | cut -c1-200
