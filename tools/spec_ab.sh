#!/bin/bash
for i in 1 2; do
for args in "--norm-spec 0 --sparse-spec 0" "--norm-spec 0" "" ; do
  python3 bench.py --no-cpu-baseline $args | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ARGS [$args]', d['value'], d['roofline']['avg_launch_ms'], d['kernel_ms_per_step']['ialm_pass'])"
done; done
