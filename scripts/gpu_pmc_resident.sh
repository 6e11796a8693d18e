#!/bin/bash
# rocprofv3 counter passes over scripts/bench_configs.py (resident loops): one counter per run, --kernel-trace only
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_resident
rm -rf "$OUT"; mkdir -p "$OUT"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -- python scripts/bench_configs.py asprec c2 c3c64 > "$OUT/$c.log" 2>&1
  rc=$?; echo "pmc $c exit $rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
python scripts/pmc_summary.py "$OUT" resident
