#!/bin/bash
# Round-2 measurement set (run on the GPU box): rocprofv3 kernel stats of the DEFAULT bench (segment + classify), PMC
# passes of the segment part, the FETCH_SIZE calibration for the pass's access mix, and the side numbers quoted in
# DESIGN.md.   bash tools/measure_r2.sh   -> everything under gpurun_out/meas_r2/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/meas_r2
part=${1:-a}
mkdir -p $out
step() { name=$1; shift; echo "[$name]"; date; timeout -k 10 "$TMO" "$@" > $out/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi; }
if [ "$part" = "a" ]; then
TMO=400 step bench_prewarm python3 bench.py --no-cpu-baseline --steps 2 --warmup 1     # MIOpen find results into the user db: the trace below holds steady-state kernels
TMO=500 step stats rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv; rm -rf $out/stats
echo "[pmc]"; date
timeout -k 10 700 bash tools/pmc_pass.sh $out/pmc --no-classify --steps 1 --warmup 0 > $out/pmc.log 2>&1 || exit 1
python3 tools/pmc_summary.py $out/pmc $out/pmc_summary.json > $out/pmc_summary.txt 2>&1
rm -rf $out/pmc/*/
echo "[calibration]"; date
timeout -k 10 300 bash tools/pass_probe.sh $out/pprobe > $out/pass_probe.txt 2>&1; cat $out/pprobe/plain.log >> $out/pass_probe.txt
timeout -k 10 300 bash tools/fetch_calib.sh $out/calib > $out/fetch_calib.txt 2>&1; cat $out/calib/plain.log >> $out/fetch_calib.txt
rm -rf $out/pprobe $out/calib
TMO=400 step bench_default python3 bench.py
TMO=300 step bench_segment python3 bench.py --no-cpu-baseline --no-classify
python3 tools/pmc_to_json.py $out $out/pmc_ialm_pass.json > $out/pmc_to_json.log 2>&1
cat $out/pmc_to_json.log | tail -n 12
grep "^{" $out/bench_default.log | tail -n 1 | cut -c1-3000
exit 0
fi
TMO=300 step bench_spec_off python3 bench.py --no-cpu-baseline --no-classify --sparse-spec 0 --norm-spec 0
TMO=300 step bench_v2 python3 bench.py --no-cpu-baseline --no-classify --variant 2
# (round 2 also ran --variant 3 here: round 1's M-state kernel, removed in round 3)
TMO=300 step bench_n21 python3 bench.py --no-cpu-baseline --no-classify --n 21 --windows 384
TMO=300 step bench_n49 python3 bench.py --no-cpu-baseline --no-classify --n 49 --windows 128
TMO=300 step bench_host python3 bench.py --no-cpu-baseline --no-classify --host-input --windows 32
TMO=300 step bench_w1 python3 bench.py --no-cpu-baseline --no-classify --windows 1 --steps 10 --warmup 2
TMO=300 step bench_w1_n21 python3 bench.py --no-cpu-baseline --no-classify --windows 1 --n 21 --steps 10 --warmup 2
TMO=300 step bench_p1 python3 bench.py --no-cpu-baseline --no-classify --size P1 --windows 512
TMO=300 step bench_p3 python3 bench.py --no-cpu-baseline --no-classify --size P3 --windows 32
TMO=300 step bench_p3_n21 python3 bench.py --no-cpu-baseline --size P3 --n 21 --windows 96
TMO=300 step bench_nofind python3 bench.py --no-cpu-baseline --no-cudnn-benchmark
TMO=300 step bench_overlap python3 bench.py --no-cpu-baseline --overlap --steps 4
# (round 2 also ran --groups 2 here: window groups on CU-masked side streams, removed in round 3)
TMO=300 step bench_classifier python3 tools/bench_classifier.py
TMO=300 step bench_framequeue python3 tools/bench_framequeue.py
TMO=300 step bench_pipeline python3 tools/bench_pipeline.py
TMO=600 step soak python3 tools/spec_soak.py 30
echo "[done]"; date
for f in default segment spec_off v2 v3 n21 n49 host w1 w1_n21 p1 p3 p3_n21 nofind overlap groups2; do echo "== $f"; grep "^{" $out/bench_$f.log | tail -n 1 | cut -c1-330; done
tail -n 1 $out/bench_classifier.log | cut -c1-600
tail -n 1 $out/bench_framequeue.log | cut -c1-900
tail -n 3 $out/bench_pipeline.log | cut -c1-600
tail -n 2 $out/soak.log
cat $out/pmc_to_json.log | tail -n 12
