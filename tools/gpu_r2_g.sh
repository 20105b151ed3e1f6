#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r2g; mkdir -p $out
step() { name=$1; shift; echo "== $name"; date; timeout -k 10 "$TMO" "$@" > $out/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -n 12 $out/$name.log | cut -c1-600; if [ $rc -ge 124 ]; then exit $rc; fi; }
TMO=300 step ab21 python3 tools/ab_pass.py --n 21 --windows 384 v3:0 v4:0
TMO=300 step framequeue python3 tools/bench_framequeue.py
TMO=900 step t_parity python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x --durations=8
TMO=400 step bench python3 bench.py --cpu-windows 1
