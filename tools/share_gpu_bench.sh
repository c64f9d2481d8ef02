#!/bin/bash
# bench.py as the driver launches it for N ranks, all ranks on the one GPU of a test box (RT_BENCH_SHARE_GPU=1):
# exercises rendezvous, sharding, barrier, max over ranks and the rank-0 print; the throughput is NOT a scaling figure.
# usage: tools/share_gpu_bench.sh <ranks> [bench flags]
n=$1; shift
RT_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 1000)) \
  bench.py --gpus $n "$@" 2>/dev/null | grep '^{'
