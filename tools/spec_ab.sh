#!/bin/bash
python3 bench.py --no-cpu-baseline --size P1 --windows 512 | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('P1', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
