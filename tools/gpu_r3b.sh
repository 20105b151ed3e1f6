#!/bin/bash
# round-3 checkpoint on the GPU box: new parity tests, the default bench line with its sub-results, config 5's per-GPU line
mkdir -p gpurun_out/r3b
python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py -q -m gpu -k "guard_band or model_pt or from_corners or start_switch" > gpurun_out/r3b/t2.log 2>&1
tail -8 gpurun_out/r3b/t2.log
python bench.py --no-cpu-baseline > gpurun_out/r3b/bench_default.json 2> gpurun_out/r3b/bench_default.err
python - <<PY
import json
d = json.load(open("gpurun_out/r3b/bench_default.json"))
print(d["value"], d["ms_per_step"], d["guard_windows"], d["redo_batches"])
print(json.dumps(d["drop_in"])); print(json.dumps(d["count_loop"])); print(json.dumps(d["pcie_inclusive"]))
print(d["segment_only"]["value"], d["roofline"]["frac"], d["roofline"]["hbm"])
PY
python bench.py --no-cpu-baseline --no-drop-in --size P3 --n 21 --windows 96 --steps 5 > gpurun_out/r3b/bench_p3_n21.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r3b/bench_p3_n21.json')); print(d['value'], d['config']['segments_per_frame'], d['config']['ialm_iters_mean'], d['segment_only']['value'])"
