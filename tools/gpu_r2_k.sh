#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r2k; mkdir -p $out
timeout -k 10 300 python3 tools/bench_classifier.py --cropped-only > $out/prewarm.log 2>&1; echo "prewarm rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/st -- python3 tools/bench_classifier.py --cropped-only > $out/cls.log 2>&1; echo "rc=$?"
cp $(ls $out/st/*/*kernel_stats.csv | head -1) $out/cls_kernel_stats.csv; rm -rf $out/st
grep -v naive $out/cls_kernel_stats.csv | head -n 24 | cut -c1-180
tail -n 1 $out/cls.log | cut -c1-500
