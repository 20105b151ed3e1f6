#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the default bench per kernel (GPU box): bash tools/pmc_hbm.sh <outdir>
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/$c" -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > "$out/$c.log" 2>&1
done
python3 tools/pmc_summary.py "$out" "$out/summary.json" | grep -A3 "k_ialm_pass_m<16, 2\|k_ialm_pass_m<16, 1"
rm -rf "$out"/FETCH_SIZE "$out"/WRITE_SIZE
