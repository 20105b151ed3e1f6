#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r2t; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_classifier.py -m gpu -x -q -k "winograd or conv1x1 or conv3x3" > $out/tests_k.log 2>&1; rc=$?; tail -n 5 $out/tests_k.log; [ $rc -ne 0 ] && exit $rc
for k in 1 0; do
timeout -k 10 300 python3 tools/bench_convs.py 4096 5 0 $k > $out/convs_k$k.log 2>&1 || { tail $out/convs_k$k.log; exit 1; }
echo "knob $k"; grep "w3x3" $out/convs_k$k.log | cut -c1-70; tail -n 1 $out/convs_k$k.log | cut -c1-160
done
