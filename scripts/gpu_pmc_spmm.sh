#!/bin/bash
# rocprofv3 counter passes over scripts/spmm_ab.py (row-major matrix-core SpMM): usage gpu_pmc_spmm.sh <tag> [spmm_ab args...]
# one pass per counter group, counters never mixed with tracing domains other than --kernel-trace
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$1; shift
mkdir -p "$OUT"
GROUPS_DEFAULT=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE")
if [ -n "${PMC_GROUPS:-}" ]; then IFS=";" read -ra GROUPS_DEFAULT <<< "$PMC_GROUPS"; fi
for grp in "${GROUPS_DEFAULT[@]}"; do
  tag=$(echo "$grp" | tr ' ' '_' | cut -c1-60)
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$tag" -- python scripts/spmm_ab.py --rounds 1 --reps 5 "$@" > "$OUT/$tag.log" 2>&1
  rc=$?
  echo "pmc $grp exit $rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
python scripts/pmc_summary.py "$OUT" spmm_rm
