#!/bin/bash
# GPU box with ONE GPU: bench.py --gpus 2 with its defaults (log-text, 8 GiB a rank), both ranks on GPU 0 (ZAMD_BENCH_SHARE_GPU, gloo, the test double of RCCL).
# A rehearsal of the flow the driver launches on an 8-GPU node, no measurement.
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O2 -fPIC -shared -w -o /tmp/libfake_rccl.so tests/tools/fake_rccl.cpp -lrt
export ZAMD_RCCL_LIB=/tmp/libfake_rccl.so ZAMD_BENCH_SHARE_GPU=1 MASTER_ADDR=127.0.0.1
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 "$@"
