#!/bin/bash
# On the 1-GPU box: run bench.py as 2 (or N <= 4) ranks sharing cuda:0 with a gloo gather, to exercise the N > 1 code path.
cd $GRAFT_REPO_ROOT
N=${1:-2}
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $N --steps 2 --warmup 1 --spp 32 --rehearse-gloo 2>&1 | tail -8 | cut -c1-900
