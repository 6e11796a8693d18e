#!/bin/bash
# round profile of the bench command: kernel-trace stats, then PMC passes (separate runs), summaries -> gpurun_out/
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$TAG gpurun_out/pmcb_$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python bench.py --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1
echo "stats exit $?"; grep '^{' gpurun_out/prof_$TAG.log | tail -1 | cut -c1-400
mkdir -p gpurun_out/pmcb_$TAG
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcb_$TAG/$c -- python bench.py --no-cpu-baseline --steps 20 --warmup 5 --spmv-reps 10 > gpurun_out/pmcb_$TAG/$c.log 2>&1
  rc=$?; echo "pmc $c exit $rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
python scripts/pmc_traffic.py gpurun_out/pmcb_$TAG ${PMC_KERNEL:-spmv_rowblock_vcp_kernel} gpurun_out/${TAG}_pmc_traffic.json
cat gpurun_out/prof_$TAG/*/*_kernel_stats.csv | cut -c1-200
