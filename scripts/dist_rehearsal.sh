#!/bin/bash
# rehearsal of the N>1 bench path on a ONE-GPU box: 2 ranks share device 0.  RCCL may refuse duplicate
# devices; the point is to learn whether it does, and to exercise the launch/bootstrap path.
# vec_grid: in the four-launch loop EVERY work-group of the aypx launch waits for the other ranks' r.r; ranks that share
# one GPU must leave each other room to run (not an issue with one rank per GPU), so cap the vector grids here.
export HSA_ENABLE_IPC_MODE_LEGACY=0 CG_DIST_BACKEND=gloo CG_TUNE=${CG_TUNE:-vec_grid=$((1536 / ${NP:-2}))}
timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${NP:-2} --master-addr 127.0.0.1 --master-port 29617 \
  bench.py --gpus ${NP:-2} --steps 20 --warmup 5 --grid ${GRID:-250x200x200} --no-cpu-baseline "$@"
echo "rehearsal exit $?"
