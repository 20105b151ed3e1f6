#!/bin/bash
# Cache-side PMC passes over the classifier's own kernels inside a real forward (tools/bench_convs.py): what the memory-bound ones
# (pool + squeeze, 1x1) wait for.  bash tools/r4/pmc_cache.sh <outdir>
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
run() { name=$1; shift; timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- python3 tools/bench_convs.py 4096 2 > "$out/$name.log" 2>&1 || { echo "FAILED $name"; tail -5 "$out/$name.log"; exit 1; }; }
run c1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run c2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
run c3 TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum GRBM_GUI_ACTIVE
run c4 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TD_TD_BUSY_sum
python3 tools/pmc_summary.py "$out" "$out/summary.json" > "$out/summary.txt" 2>&1
rm -rf "$out"/c1 "$out"/c2 "$out"/c3 "$out"/c4
grep -A 18 "pool_squeeze<3" "$out/summary.txt" | head -24
