#!/bin/bash
# Round-4 measurement set (run on the GPU box; every step under its own timeout, the set stops at the first failure):
#   part a: the driver's default line, rocprofv3 kernel stats of the same command (classifier on, sub-results off), FETCH_SIZE / WRITE_SIZE
#           of the IALM pass in two separate --pmc runs (kernel trace only), per-layer classifier times
#   part b: the n = 21 and config-5 per-GPU lines, the two-rank launcher rehearsal with the video-sharded leg
set -o pipefail
part=${1:-a}
out=gpurun_out/r4m
mkdir -p $out
export TMPDIR=/tmp
step() { local name=$1 tmo=$2; shift 2; timeout -k 10 $tmo "$@" > $out/$name 2> $out/$name.err || { echo "FAILED: $name"; tail -n 5 $out/$name.err; exit 1; }; }
if [ "$part" = a ]; then
  step default.json 600 python bench.py
  rm -rf $out/prof
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline --no-drop-in --video-windows 0 --steps 2 --warmup 1 > $out/prof_bench.json 2> $out/prof_bench.err || { echo "FAILED: rocprofv3 stats"; exit 1; }
  cp $(ls $out/prof/*/*kernel_stats.csv | head -1) $out/r4_kernel_stats_default_bench.csv
  rm -rf $out/prof
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $out/pmc_$c
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --no-cpu-baseline --no-drop-in --video-windows 0 --no-classify --steps 1 --warmup 0 > $out/pmc_$c.json 2> $out/pmc_$c.err || { echo "FAILED: rocprofv3 pmc $c"; exit 1; }
  done
  python3 tools/pmc_summary.py $out $out/pmc_summary.json > $out/pmc_summary.txt
  rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
  step r4_cnn_layers.txt 200 python tools/bench_convs.py 4096 5
  step r4_cnn_layers_8192.txt 200 python tools/bench_convs.py 8192 5
  head -4 $out/r4_kernel_stats_default_bench.csv | cut -c1-200
  cut -c1-300 $out/default.json
else
  step n21.json 300 python bench.py --no-cpu-baseline --no-drop-in --video-windows 0 --n 21 --windows 384 --steps 5
  step p3_n21.json 300 python bench.py --no-cpu-baseline --no-drop-in --video-windows 0 --size P3 --n 21 --windows 96 --steps 5
  SWK_DIST_BACKEND=gloo step launcher_2ranks_gloo_one_gpu.json 600 python bench.py --gpus 2 --steps 3 --warmup 1 --windows 32 --no-cpu-baseline --no-drop-in --video-windows 12 --verify-all-videos
  step video_config5_one_gpu.json 300 python bench.py --steps 1 --warmup 0 --windows 8 --no-cpu-baseline --no-drop-in --video-config 5 --video-windows 24
  cut -c1-200 $out/n21.json; cut -c1-200 $out/p3_n21.json
fi
