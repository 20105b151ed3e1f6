#!/bin/bash
# PMC passes over the classifier's own kernels inside a real forward (tools/bench_convs.py).  bash tools/pmc_convs.sh <outdir>
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
run() { name=$1; shift; timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- python3 tools/bench_convs.py 4096 2 > "$out/$name.log" 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM
run sq3 SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAVES
run fetch FETCH_SIZE GRBM_COUNT
run write WRITE_SIZE GRBM_COUNT
python3 tools/pmc_summary.py "$out" "$out/summary.json" > "$out/summary.txt" 2>&1
rm -rf "$out"/sq1 "$out"/sq2 "$out"/sq3 "$out"/fetch "$out"/write
grep -A 30 "wino3x3_relu_place<8" "$out/summary.txt" | head -40
