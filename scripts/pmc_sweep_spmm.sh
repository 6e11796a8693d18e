#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "spmm_rowmajor=2" "spmm_rowmajor=2,dev.spmm_wgs=32" "spmm_rowmajor=2,dev.spmm_wgs=48" "spmm_rowmajor=2,dev.spmm_ynt=0" "spmm_rowmajor=2,dev.spmm_ynt=1" "spmm_rowmajor=2,dev.spmm_wgs=32,dev.spmm_ynt=0"; do
  tag=$(echo "$cfg" | tr ',=' '__')
  PMC_GROUPS="FETCH_SIZE" bash scripts/gpu_pmc_spmm.sh sweep_$tag --dtype f64 --nrhs 32 "$cfg" > gpurun_out/pmc_sweep_$tag.log 2>&1
  echo "$cfg: $(grep -A1 '== FETCH' gpurun_out/pmc_sweep_$tag.log | tail -1 | cut -c60-120)"
done
timeout -k 10 200 python scripts/spmm_ab.py --dtype f64 --nrhs 32 --rounds 3 "spmm_rowmajor=2" "spmm_rowmajor=2,dev.spmm_wgs=32" "spmm_rowmajor=2,dev.spmm_wgs=48" "spmm_rowmajor=2,dev.spmm_ynt=0" "spmm_rowmajor=2,dev.spmm_ynt=1" "spmm_rowmajor=2,dev.spmm_wgs=32,dev.spmm_ynt=0" 2>/dev/null | grep cfg | cut -c1-150
