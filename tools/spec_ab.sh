#!/bin/bash
for i in 1 2; do
for args in "--integer-start 0" "" ; do
  python3 bench.py --no-cpu-baseline $args | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ARGS [$args]', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done; done
