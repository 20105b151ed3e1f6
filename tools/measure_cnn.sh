#!/bin/bash
# evidence behind DESIGN.md's classifier section: per-layer times inside a forward, PMC counters of the convolution kernels,
# s_memtime shares of the Winograd kernel's phases, the f32 pipe probe.   bash tools/measure_cnn.sh   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/meas_cnn; mkdir -p $out
timeout -k 10 300 python3 tools/bench_convs.py 4096 5 > $out/cnn_layers.txt 2>&1 || { tail $out/cnn_layers.txt; exit 1; }
timeout -k 10 900 bash tools/pmc_convs.sh $out/pmc > $out/pmc.log 2>&1 || { tail $out/pmc.log; exit 1; }
cp $out/pmc/summary.txt $out/pmc_summary_cnn_kernels.txt
timeout -k 10 300 bash tools/wino_stamp.sh > $out/wino_stamp.txt 2>&1 || { tail $out/wino_stamp.txt; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/f32_pipe_probe tools/f32_pipe_probe.hip > /dev/null 2>&1
timeout -k 10 120 ./tools/f32_pipe_probe > $out/f32_pipe_probe.txt 2>&1
grep -v "^{" $out/cnn_layers.txt | cut -c1-70 | tail -30; tail -n 28 $out/wino_stamp.txt
