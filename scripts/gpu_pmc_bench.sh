#!/bin/bash
# rocprofv3 counter passes over bench.py for one kernel (default: the headline's SpMV): usage gpu_pmc_bench.sh <tag> [kernel substring]
# one pass per counter group, counters never mixed with tracing domains other than --kernel-trace
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcbench_$1
KERNEL=${2:-spmv_rowblock_vcp_kernel}
mkdir -p "$OUT"
GROUPS_DEFAULT=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE")
for grp in "${GROUPS_DEFAULT[@]}"; do
  tag=$(echo "$grp" | tr ' ' '_' | cut -c1-60)
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$tag" -- python bench.py --no-cpu-baseline --steps 20 --warmup 5 --spmv-reps 10 > "$OUT/$tag.log" 2>&1
  rc=$?
  echo "pmc $grp exit $rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
python scripts/pmc_summary.py "$OUT" "$KERNEL"
