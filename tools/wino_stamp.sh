#!/bin/bash
# builds the diagnostic (stamped) Winograd kernel and prints the shares of a phase's parts; run on the GPU box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DSWK_WINO_STAMP -I include -I swiftwatcher_amd/csrc -shared -o tools/libwino_stamp.so swiftwatcher_amd/csrc/cnn_wino3x3.hip 2> gpurun_out/r2s/build.log || { tail gpurun_out/r2s/build.log; exit 1; }
timeout -k 10 300 python3 tools/wino_stamp.py > gpurun_out/r2s/stamp.txt 2>&1; cat gpurun_out/r2s/stamp.txt
