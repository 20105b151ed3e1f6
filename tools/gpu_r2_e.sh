#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r2e; mkdir -p $out
step() { name=$1; shift; echo "== $name"; date; timeout -k 10 "$TMO" "$@" > $out/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -n 12 $out/$name.log | cut -c1-400; if [ $rc -ge 124 ]; then exit $rc; fi; }
TMO=120 step chain ./tools/f64_mfma_chain_probe
TMO=400 step t_new python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "kstep or duplicated or sparse_store or golden_without" --durations=5
TMO=400 step ab64 python3 tools/ab_pass.py v3:0 v4:0 v4:1
TMO=400 step ab21 python3 tools/ab_pass.py --n 21 --windows 384 v3:0 v4:0 v4:1
TMO=300 step ab49 python3 tools/ab_pass.py --n 49 --windows 128 --rounds 1 v3:0 v4:0
