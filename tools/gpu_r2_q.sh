#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r2q; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_classifier.py -m gpu -x -q -k "winograd or conv1x1 or conv3x3" > $out/tests_k.log 2>&1; rc=$?; tail -n 5 $out/tests_k.log; [ $rc -ne 0 ] && exit $rc
for b in 4096 2048 1024 512 256; do
timeout -k 10 300 python3 tools/bench_convs.py $b 5 0 > $out/convs_$b.log 2>&1 || { tail $out/convs_$b.log; exit 1; }
echo "batch $b: $(tail -n 1 $out/convs_$b.log | cut -c1-170)"
done
grep -v "^{" $out/convs_4096.log | cut -c1-70
timeout -k 10 900 bash tools/pmc_convs.sh gpurun_out/r2o || exit 1
