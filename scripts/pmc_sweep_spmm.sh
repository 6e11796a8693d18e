#!/bin/bash
# FETCH_SIZE of the row-major SpMM for a list of tuning configurations (one rocprofv3 --pmc pass each), then their timings
# usage: pmc_sweep_spmm.sh "cfg1" "cfg2" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "$@"; do
  tag=$(echo "$cfg" | tr ',=' '__')
  PMC_GROUPS="FETCH_SIZE" bash scripts/gpu_pmc_spmm.sh sweep_$tag --dtype ${SPMM_DTYPE:-f64} --nrhs ${SPMM_NRHS:-32} "$cfg" > gpurun_out/pmc_sweep_$tag.log 2>&1 || exit 1
  echo "$cfg: $(grep -A1 '== FETCH' gpurun_out/pmc_sweep_$tag.log | tail -1 | cut -c60-120)"
done
timeout -k 10 200 python scripts/spmm_ab.py --dtype ${SPMM_DTYPE:-f64} --nrhs ${SPMM_NRHS:-32} --rounds 3 "$@" 2>/dev/null | grep cfg | cut -c1-150
