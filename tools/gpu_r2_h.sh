#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r2h; mkdir -p $out
echo "== pmc"; date
timeout -k 10 600 bash tools/pmc_pass.sh $out/pmc --no-classify --steps 1 --warmup 0 > $out/pmc.log 2>&1; echo "pmc rc=$?"
python3 tools/pmc_summary.py $out/pmc $out/pmc_summary.json > $out/pmc_summary.txt 2>&1
rm -rf $out/pmc/*/
grep -A 22 "k_ialm_pass_m<16, 2>" $out/pmc_summary.txt | head -30
echo "== fq trace"; date
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/fq -- python3 tools/bench_framequeue.py > $out/fq.log 2>&1; echo "fq rc=$?"
cp $(ls $out/fq/*/*kernel_stats.csv | head -1) $out/fq_kernel_stats.csv; rm -rf $out/fq
head -n 14 $out/fq_kernel_stats.csv | cut -c1-160
