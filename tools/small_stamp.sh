#!/bin/bash
# Diagnostic build of the library with wall-clock stamps in the small-matrix kernel (-DSWK_SMALL_STAMP -> tools/libswk_stamp.so); build it
# where hipcc is (it cross-compiles), run tools/small_stamp.py on the GPU box.
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DSWK_SMALL_STAMP -I include -I swiftwatcher_amd/csrc \
    -shared -pthread -o tools/libswk_stamp.so swiftwatcher_amd/csrc/*.hip swiftwatcher_amd/csrc/*.cpp
