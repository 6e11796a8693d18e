#!/bin/bash
# rocprofv3 counter passes over an arbitrary python script: usage gpu_pmc_cmd.sh <tag> <script.py> [args...]
# (PMC_GROUPS="A B;C" overrides the counter groups; one pass per group, counters never mixed with tracing domains)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$1; shift
mkdir -p "$OUT"
GROUPS_DEFAULT=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE")
if [ -n "${PMC_GROUPS:-}" ]; then IFS=";" read -ra GROUPS_DEFAULT <<< "$PMC_GROUPS"; fi
for grp in "${GROUPS_DEFAULT[@]}"; do
  tag=$(echo "$grp" | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$tag" -- python "$@" > "$OUT/$tag.log" 2>&1
  rc=$?
  echo "pmc $grp exit $rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
python scripts/pmc_summary.py "$OUT" spmm spmv_rowblock
