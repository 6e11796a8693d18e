#!/bin/bash
# per-kernel durations of the distributed loops on one rank's slab (rank as its own halo peer): rocprofv3 kernel trace per loop.
# usage (on the GPU box): scripts/prof_slab.sh <out tag> "<loop name>" ["<loop name>" ...]
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/$TAG
for m in "$@"; do
  d=gpurun_out/$TAG/prof_$(echo "$m" | tr ' .+' '___')
  rm -rf $d
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python scripts/dist_slab_probe.py --grid ${GRID:-250x200x25} --iters 400 --rounds 1 --only "$m" > $d.log 2>&1
  rc=$?; echo "== $m rc=$rc"; tail -2 $d.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  for f in $d/*/*_kernel_stats.csv; do [ -f "$f" ] && cut -c1-160 "$f" | head -9; done
done
