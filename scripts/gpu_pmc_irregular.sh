#!/bin/bash
# rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE: separate runs, --kernel-trace only) over the irregular-pattern lines of
# scripts/bench_configs.py: usage gpu_pmc_irregular.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$1
mkdir -p "$OUT"
for grp in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$grp" -- python scripts/bench_configs.py irregular > "$OUT/$grp.log" 2>&1
  rc=$?; echo "pmc $grp exit $rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
python scripts/pmc_summary.py "$OUT" spmv_rowblock spmv_stream > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
