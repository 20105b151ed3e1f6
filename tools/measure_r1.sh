#!/bin/bash
# Round-1 measurement set (run on the GPU box): rocprofv3 kernel stats + PMC passes of the default bench, the
# FETCH_SIZE calibration for the pass's access mix, and the side numbers quoted in DESIGN.md.
#   bash tools/measure_r1.sh            -> everything under gpurun_out/meas_r1/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/meas_r1
rm -rf $out; mkdir -p $out
echo "[1] kernel stats"; date
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > $out/stats.log 2>&1
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
rm -rf $out/stats
echo "[2] pmc"; date
bash tools/pmc_pass.sh $out/pmc --steps 1 --warmup 0 > $out/pmc.log 2>&1
python3 tools/pmc_summary.py $out/pmc $out/pmc_summary.json > $out/pmc_summary.txt 2>&1
rm -rf $out/pmc/*/
echo "[3] calibration"; date
bash tools/pass_probe.sh $out/pprobe > $out/pass_probe.txt 2>&1
cat $out/pprobe/plain.log >> $out/pass_probe.txt
bash tools/fetch_calib.sh $out/calib > $out/fetch_calib.txt 2>&1
cat $out/calib/plain.log >> $out/fetch_calib.txt
rm -rf $out/pprobe $out/calib
echo "[4] side benches"; date
python3 bench.py --no-cpu-baseline > $out/bench_default.log 2>&1
python3 bench.py --no-cpu-baseline --variant 2 > $out/bench_v2.log 2>&1
python3 bench.py --no-cpu-baseline --n 21 --windows 384 > $out/bench_n21.log 2>&1
python3 bench.py --no-cpu-baseline --host-input --windows 32 > $out/bench_host.log 2>&1
python3 bench.py --no-cpu-baseline --windows 1 --steps 10 --warmup 2 > $out/bench_w1.log 2>&1
python3 bench.py --no-cpu-baseline --size P1 --windows 512 > $out/bench_p1.log 2>&1
python3 bench.py --no-cpu-baseline --size P3 --windows 32 > $out/bench_p3.log 2>&1
python3 bench.py --no-cpu-baseline --classify > $out/bench_classify.log 2>&1
python3 tools/bench_classifier.py > $out/bench_classifier.log 2>&1
python3 tools/bench_framequeue.py > $out/bench_framequeue.log 2>&1 || true
echo "[5] done"; date
for f in default v2 n21 host w1 p1 p3 classify; do echo "== $f"; tail -n 1 $out/bench_$f.log | cut -c1-400; done
tail -n 1 $out/bench_classifier.log
tail -n 3 $out/bench_framequeue.log
