#!/bin/bash
# Rehearses bench.py's multi-rank path on a ONE-GPU box: N ranks share cuda:0, the library's exchange goes through its callback
# transport into gloo.  (RCCL refuses two ranks on one device; the driver runs the real N-GPU job over the library's RCCL communicator.)
N=${1:-2}; STEPS=${2:-300}; PRESET=${3:-syn3m5}
export ESIM_BENCH_SAME_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 \
     bench.py --gpus $N --steps $STEPS --warmup 8 --preset $PRESET --transport callback
