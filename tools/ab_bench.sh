#!/bin/bash
# A/B of two builds of libswk.so on the same GPU box (box-to-box variance is +-15 %):
#   bash tools/ab_bench.sh tools/ab/libswk_a.so [bench args...]     -> alternates A, B, A, B
a=$1; shift
for i in 1 2; do
  SWK_LIB=$a python3 bench.py --no-cpu-baseline "$@" | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('A', d['value'], d['roofline']['avg_launch_ms'], d['kernel_ms_per_step'])"
  python3 bench.py --no-cpu-baseline "$@" | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('B', d['value'], d['roofline']['avg_launch_ms'], d['kernel_ms_per_step'])"
done
