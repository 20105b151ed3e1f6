#!/bin/bash
# Round-1 measurement set (run on the GPU box): rocprofv3 kernel stats + PMC passes of the default bench,
# plus the side numbers quoted in DESIGN.md.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/meas_r1
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > $out/stats.log 2>&1
bash tools/pmc_pass.sh $out/pmc --steps 1 --warmup 0 > $out/pmc.log 2>&1
python3 tools/pmc_summary.py $out/pmc $out/pmc_summary.json > $out/pmc_summary.txt 2>&1
python3 bench.py --no-cpu-baseline --n 21 --windows 384 > $out/bench_n21.log 2>&1
python3 bench.py --no-cpu-baseline --host-input --windows 32 > $out/bench_host.log 2>&1
python3 bench.py --no-cpu-baseline --windows 1 --steps 10 --warmup 2 > $out/bench_w1.log 2>&1
python3 bench.py --no-cpu-baseline --size P1 --windows 512 > $out/bench_p1.log 2>&1
tail -n 1 $out/bench_n21.log $out/bench_host.log $out/bench_w1.log $out/bench_p1.log
