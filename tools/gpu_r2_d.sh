#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r2d; mkdir -p $out
step() { name=$1; shift; echo "== $name"; date; timeout -k 10 "$TMO" "$@" > $out/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -n 8 $out/$name.log | cut -c1-1200; if [ $rc -ge 124 ]; then exit $rc; fi; }
TMO=120 step probe ./tools/f64_pipe_probe
TMO=600 step t_cfg3 python3 -m pytest tests/test_baseline_configs.py tests/test_abi_and_host.py -m gpu -q -k "config3 or runtime" --durations=5
