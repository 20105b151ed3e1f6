#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r2j; mkdir -p $out
step() { name=$1; shift; echo "== $name"; date; timeout -k 10 "$TMO" "$@" > $out/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -n 8 $out/$name.log | cut -c1-500; if [ $rc -ge 124 ]; then exit $rc; fi; }
TMO=200 step t_conv python3 -m pytest tests/test_classifier.py -m gpu -q -x -k "fused_conv" --durations=5
TMO=300 step bench_fused python3 bench.py --no-cpu-baseline
SWK_FUSED_3X3=0 TMO=300 step bench_no3x3 python3 bench.py --no-cpu-baseline
TMO=300 step bench_fused2 python3 bench.py --no-cpu-baseline
TMO=600 step t_cls python3 -m pytest tests/test_classifier.py tests/test_baseline_configs.py -m gpu -q -x --durations=5
