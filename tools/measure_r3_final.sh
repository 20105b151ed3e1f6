#!/bin/bash
# final round-3 pass: the driver's default line (with the CPU baseline), rocprofv3 kernel stats of the same command, per-layer classifier times
mkdir -p gpurun_out/r3m
export TMPDIR=/tmp
python bench.py > gpurun_out/r3m/default.json 2> gpurun_out/r3m/default.err
rm -rf gpurun_out/r3m/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3m/prof -- python3 bench.py --no-cpu-baseline --no-drop-in --steps 2 --warmup 1 > gpurun_out/r3m/prof_bench.json 2> gpurun_out/r3m/prof_bench.err
cp $(ls gpurun_out/r3m/prof/*/*kernel_stats.csv | head -1) gpurun_out/r3m/r3_kernel_stats_default_bench.csv
rm -rf gpurun_out/r3m/prof
python tools/bench_convs.py 4096 5 > gpurun_out/r3m/r3_cnn_layers.txt 2>&1
python tools/bench_convs.py 256 20 > gpurun_out/r3m/r3_cnn_layers_batch256.txt 2>&1
python bench.py --no-cpu-baseline --no-drop-in --size P3 --n 21 --windows 96 --steps 5 > gpurun_out/r3m/p3_n21.json 2>/dev/null
python bench.py --no-cpu-baseline --no-drop-in --n 21 --windows 384 --steps 5 > gpurun_out/r3m/n21.json 2>/dev/null
head -4 gpurun_out/r3m/r3_kernel_stats_default_bench.csv | cut -c1-200
