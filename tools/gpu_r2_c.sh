#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r2c; mkdir -p $out
step() { name=$1; shift; echo "== $name"; date; timeout -k 10 "$TMO" "$@" > $out/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -n 4 $out/$name.log | cut -c1-1200; if [ $rc -ge 124 ]; then exit $rc; fi; }
TMO=400 step t_new python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "kstep or duplicated" --durations=5
TMO=500 step ab64 python3 tools/ab_pass.py v3:0 v5:0 v5:1 v4:0 v4:1 v4:3
TMO=400 step ab21 python3 tools/ab_pass.py --n 21 --windows 384 v3:0 v5:0 v5:1 v4:0 v4:1
TMO=600 step t_cfg python3 -m pytest tests/test_baseline_configs.py -m gpu -q --durations=5
TMO=400 step bench python3 bench.py --cpu-windows 1
TMO=300 step bench_b4096 python3 bench.py --no-cpu-baseline --cls-batch 4096
