#!/bin/bash
for args in "" "--groups 2" "--groups 2 --eig-cus 16" "--groups 4 --eig-cus 16" "--windows 256" ; do
  python3 bench.py --no-cpu-baseline $args | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ARGS [$args]', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done
