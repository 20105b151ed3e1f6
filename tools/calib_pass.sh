#!/bin/bash
# re-run only the pass-shape calibration of tools/measure_r1.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/meas_r1; mkdir -p $out
bash tools/pass_probe.sh $out/pprobe > $out/pass_probe.txt 2>&1
cat $out/pprobe/plain.log >> $out/pass_probe.txt
rm -rf $out/pprobe
cat $out/pass_probe.txt
