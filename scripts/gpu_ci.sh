#!/bin/bash
# Runs on the GPU box via gpurun: smoke, GPU tests, short bench.  Ordinary failures are logged and the
# next step still runs; a step that is killed or times out (exit 124/137) stops the sequence.
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
step() {
  name=$1; shift
  echo "=== $name: $*"
  timeout -k 10 "${STEP_TIMEOUT:-600}" "$@" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name exit $rc"
  tail -n "${TAIL:-15}" "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name killed/timed out: stopping"; exit $rc; fi
  return 0
}
for s in "$@"; do
  case $s in
    smoke) step smoke python -c "import __graft_entry__ as g; g.smoke()" ;;
    tests) step pytest python -m pytest tests -m gpu -x -q ;;
    tests_all) step pytest python -m pytest tests -m gpu -q ;;
    bench) step bench python bench.py --steps 100 --warmup 10 ;;
    bench_quick) step bench python bench.py --steps 50 --warmup 10 --no-cpu-baseline ;;
    *) echo "unknown step $s" ;;
  esac
done
