#!/bin/bash
# Round-3 measurement pass on the GPU box (one gpurun call each part; outputs under gpurun_out/r3m, copied to profiles/ by hand).
#   part a: the driver's default command + rocprofv3 kernel stats of the same command at fewer steps
#   part b: the other lines quoted in DESIGN.md section 6
set -e
mkdir -p gpurun_out/r3m
export TMPDIR=/tmp
case "$1" in
a)
  python bench.py > gpurun_out/r3m/default.json 2> gpurun_out/r3m/default.err
  rm -rf gpurun_out/r3m/prof
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3m/prof -- python3 bench.py --no-cpu-baseline --no-drop-in --steps 2 --warmup 1 > gpurun_out/r3m/prof_bench.json 2> gpurun_out/r3m/prof_bench.err
  cp $(ls gpurun_out/r3m/prof/*/*kernel_stats.csv | head -1) gpurun_out/r3m/r3_kernel_stats_default_bench.csv
  rm -rf gpurun_out/r3m/prof
  head -12 gpurun_out/r3m/r3_kernel_stats_default_bench.csv
  ;;
c)
  python tools/prof_pipeline.py 16 1 > gpurun_out/r3m/r3_count_loop_profile.txt 2>&1
  python tools/prof_batched.py > gpurun_out/r3m/r3_batched_loop_profile.txt 2>&1
  SWK_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 3 --warmup 1 --windows 32 --no-cpu-baseline > gpurun_out/r3m/launcher_2ranks_gloo_one_gpu.json 2> gpurun_out/r3m/launcher.err
  tail -c 900 gpurun_out/r3m/launcher_2ranks_gloo_one_gpu.json
  python bench.py > gpurun_out/r3m/default.json 2> gpurun_out/r3m/default.err
  ;;
b)
  python bench.py --no-cpu-baseline --no-drop-in --size P3 --n 21 --windows 96 --steps 5 > gpurun_out/r3m/p3_n21.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-drop-in --n 21 --windows 384 --steps 5 > gpurun_out/r3m/n21.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-drop-in --no-classify --steps 5 > gpurun_out/r3m/segment.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-drop-in --no-classify --steps 5 --norm-spec 0 --sparse-spec 0 > gpurun_out/r3m/spec_off.json 2>/dev/null
  python tools/bench_convs.py 4096 5 > gpurun_out/r3m/r3_cnn_layers.txt 2>&1
  python tools/bench_convs.py 256 20 > gpurun_out/r3m/r3_cnn_layers_batch256.txt 2>&1
  python tools/bench_framequeue.py > gpurun_out/r3m/framequeue.json 2>/dev/null
  python tools/prof_pipeline.py 16 1 > gpurun_out/r3m/r3_count_loop_profile.txt 2>&1
  python tools/prof_batched.py > gpurun_out/r3m/r3_batched_loop_profile.txt 2>&1
  SWK_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 3 --warmup 1 --windows 32 --no-cpu-baseline > gpurun_out/r3m/launcher_2ranks_gloo_one_gpu.json 2> gpurun_out/r3m/launcher.err
  tail -c 600 gpurun_out/r3m/launcher_2ranks_gloo_one_gpu.json
  ;;
esac
