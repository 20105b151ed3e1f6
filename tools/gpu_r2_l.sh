#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r2l; mkdir -p $out
step() { name=$1; shift; echo "== $name"; date; timeout -k 10 "$TMO" "$@" > $out/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -n 4 $out/$name.log | cut -c1-400; if [ $rc -ge 124 ]; then exit $rc; fi; }
TMO=200 step t_conv python3 -m pytest tests/test_classifier.py -m gpu -q -x -k "fused_conv or model_pt" --durations=5
TMO=300 step bench python3 bench.py --no-cpu-baseline
TMO=300 step bench2 python3 bench.py --no-cpu-baseline
timeout -k 10 300 python3 tools/bench_classifier.py --cropped-only > $out/prewarm.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/st -- python3 tools/bench_classifier.py --cropped-only > $out/cls.log 2>&1; echo "rc=$?"
cp $(ls $out/st/*/*kernel_stats.csv | head -1) $out/cls_kernel_stats.csv; rm -rf $out/st
grep "swk::" $out/cls_kernel_stats.csv | cut -c1-60,180-260 | head -14
