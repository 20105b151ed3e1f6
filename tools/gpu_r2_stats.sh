#!/bin/bash
# kernel stats of the default bench with MIOpen's find results already in the user db (a first, untraced run of the same
# command makes them), so the trace holds the steady-state kernels and not MIOpen's search
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/meas_r2; mkdir -p $out
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > $out/bench_prewarm.log 2>&1; echo "prewarm rc=$?"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > $out/stats.log 2>&1; echo "stats rc=$?"
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv; rm -rf $out/stats
head -n 30 $out/kernel_stats.csv | cut -c1-150
grep "^{" $out/stats.log | tail -n 1 | cut -c1-400
