#!/bin/bash
# the driver's multi-GPU invocation on whatever devices exist: default per-GPU share (131072 blocks) on 2 ranks
( time python3 bench.py --gpus 2 --steps 5 --warmup 2 ) 2>&1 | tail -6 | cut -c1-1500
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 --blocks 131072 --no-cpu-baseline ) 2>&1 | tail -6 | cut -c1-900
