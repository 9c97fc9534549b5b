#!/bin/bash
# the driver's N > 1 invocation: bench.py under torch.distributed.run (two ranks sharing the one GPU of this box)
cd "$GRAFT_REPO_ROOT"
timeout 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --blocks 16384 2>&1 | grep -v "^\[W\|^W0\|warn" | tail -3 | cut -c1-700
