#!/bin/bash
# final round-3 pass (run on the GPU box; every step under its own timeout, the set stops at the first failure):
#   the driver's default line (with the CPU baseline), rocprofv3 kernel stats of the same command, the FrameQueue drop-in, the profile of
#   the serial counting loop, per-layer classifier times at 4,096 and 256 rows, config 5's per-GPU line, the n = 21 line, and -- when
#   tools/libswk_stamp.so was built (tools/small_stamp.sh, where hipcc is) -- the stamps of the small-matrix step
set -o pipefail
out=gpurun_out/r3m
mkdir -p $out
export TMPDIR=/tmp
step() { local name=$1 tmo=$2; shift 2; timeout -k 10 $tmo "$@" > $out/$name 2> $out/$name.err || { echo "FAILED: $name"; tail -n 5 $out/$name.err; exit 1; }; }
step default.json 500 python bench.py
rm -rf $out/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline --no-drop-in --steps 2 --warmup 1 > $out/prof_bench.json 2> $out/prof_bench.err || { echo "FAILED: rocprofv3"; exit 1; }
cp $(ls $out/prof/*/*kernel_stats.csv | head -1) $out/r3_kernel_stats_default_bench.csv
rm -rf $out/prof
step framequeue.json 200 python tools/bench_framequeue.py
step r3_count_loop_profile.txt 200 python tools/prof_pipeline.py 16 1
step r3_cnn_layers.txt 200 python tools/bench_convs.py 4096 5
step r3_cnn_layers_batch256.txt 200 python tools/bench_convs.py 256 20
step p3_n21.json 300 python bench.py --no-cpu-baseline --no-drop-in --size P3 --n 21 --windows 96 --steps 5
step n21.json 300 python bench.py --no-cpu-baseline --no-drop-in --n 21 --windows 384 --steps 5
if [ -f tools/libswk_stamp.so ]; then
  step r3_small_stamp_n21.txt 100 python3 tools/small_stamp.py 21
  step r3_small_stamp_n64.txt 100 python3 tools/small_stamp.py 64
fi
head -4 $out/r3_kernel_stats_default_bench.csv | cut -c1-200
cut -c1-220 $out/default.json
