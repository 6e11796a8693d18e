#!/bin/bash
# rehearsal of the N>1 bench path on a ONE-GPU box: 2 ranks share device 0.  RCCL may refuse duplicate
# devices; the point is to learn whether it does, and to exercise the launch/bootstrap path.
export HSA_ENABLE_IPC_MODE_LEGACY=0 CG_DIST_BACKEND=gloo
timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${NP:-2} --master-addr 127.0.0.1 --master-port 29617 \
  bench.py --gpus ${NP:-2} --steps 20 --warmup 5 --grid ${GRID:-250x200x200} --no-cpu-baseline "$@"
echo "rehearsal exit $?"
