#!/bin/bash
# rehearsal of the N>1 bench path on a ONE-GPU box: 2 ranks share device 0.  RCCL may refuse duplicate
# devices; the point is to learn whether it does, and to exercise the launch/bootstrap path.
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 \
  bench.py --gpus 2 --steps 20 --warmup 5 --grid 64x64x64 --no-cpu-baseline "$@"
echo "rehearsal exit $?"
